import agcn_amd  # noqa: F401
from agcn_amd.feeders.feeder import Feeder  # noqa: F401
