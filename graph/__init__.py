"""Top-level ``graph`` package: keeps the reference's dotted class paths
(``graph.ntu_rgb_d.Graph``, ``graph.kinetics.Graph`` used by the yaml configs,
e.g. reference ``config/nturgbd-cross-view/train_joint.yaml:25``) resolvable."""
from . import kinetics, ntu_rgb_d, tools  # noqa: E402,F401
