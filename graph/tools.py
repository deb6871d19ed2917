import agcn_amd  # noqa: F401
from agcn_amd.graph.tools import *  # noqa: F401,F403
