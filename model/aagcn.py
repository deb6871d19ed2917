"""``model.aagcn`` -- the class path the reference AAGCN configs name (train_joint_aagcn.yaml:34).
The implementation lives in ``2s-agcn_amd/model/aagcn.py``."""
import agcn_amd  # noqa: F401
from agcn_amd.model.aagcn import (AdaptiveGCN, BaseModel, ChannelAttention, GCNUnit, Model,  # noqa: F401
                                  NonAdaptiveGCN, SpatialAttention, TCNGCNUnit, TCNUnit, TemporalAttention)
