"""How often a unit's f16x3 chain found the maximum of its input left behind by the BatchNorm pass that produced it
(ops._take_out_amax) in one training step of model.agcn.Model: expect 9 hits (l2..l10), 0 misses.
    python tools/amax_hits.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench  # noqa: F401  (puts the package on the path as agcn_amd)
import agcn_amd  # noqa: F401
from agcn_amd import ops
from agcn_amd.model.agcn import Model

torch.manual_seed(0)
m = Model(num_class=60, num_point=25, num_person=2, graph='graph.ntu_rgb_d.Graph', in_channels=3).cuda().train()
x = torch.randn(4, 3, 64, 25, 2, device='cuda')
loss = m(x).logsumexp(1).sum()
loss.backward()
torch.cuda.synchronize()
print('misses, hits =', ops._OUT_AMAX_STATS, 'chain:', ops._L().agcn_chain_mode().decode())
