from . import aagcn, agcn  # noqa: F401
