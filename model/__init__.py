"""Top-level ``model`` package: keeps the reference's dotted class path ``model.agcn.Model``
(reference config/nturgbd-cross-view/train_joint.yaml:20) resolvable for ``import_class``."""
from . import aagcn, agcn  # noqa: E402,F401
