import agcn_amd  # noqa: F401
from agcn_amd.feeders.tools import *  # noqa: F401,F403
