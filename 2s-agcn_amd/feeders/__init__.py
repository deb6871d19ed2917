"""Input side of the hot path: the reference's ``Feeder`` dataset and augmentations, plus what a GPU that
consumes ~700+ clips/s needs on top of it -- batched on-device augmentation and a pinned, double-buffered host-to-device
ring (``device.py``)."""
from .feeder import Feeder  # noqa: F401
from .device import DeviceAugment, DeviceLoader  # noqa: F401
