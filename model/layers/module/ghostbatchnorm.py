"""``model.layers.module.ghostbatchnorm`` -- the import path of the reference's GhostBatchNorm
(model/layers/module/ghostbatchnorm.py).  The implementation lives in ``2s-agcn_amd/model/ghostbatchnorm.py``."""
import agcn_amd  # noqa: F401
from agcn_amd.model.ghostbatchnorm import GhostBatchNorm1d, GhostBatchNorm2d  # noqa: F401
