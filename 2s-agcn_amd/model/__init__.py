from . import agcn  # noqa: F401
