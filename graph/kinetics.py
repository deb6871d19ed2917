import agcn_amd  # noqa: F401
from agcn_amd.graph.kinetics import *  # noqa: F401,F403
from agcn_amd.graph.kinetics import Graph  # noqa: F401
