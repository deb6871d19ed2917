"""``model.agcn`` -- the class path the reference yaml configs name (train_joint.yaml:20).
The implementation lives in ``2s-agcn_amd/model/agcn.py``."""
import agcn_amd  # noqa: F401
from agcn_amd.model.agcn import (Model, TCN_GCN_unit, bn_init, conv_branch_init, conv_init,  # noqa: F401
                                 import_class, unit_gcn, unit_tcn)
