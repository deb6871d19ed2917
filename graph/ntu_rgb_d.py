import agcn_amd  # noqa: F401
from agcn_amd.graph.ntu_rgb_d import *  # noqa: F401,F403
from agcn_amd.graph.ntu_rgb_d import Graph  # noqa: F401
