import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import agcn_amd
from tests import golden_util as gu
from oracle import agcn_oracle as orc
from model.aagcn import Model
dev = torch.device('cuda:0')
gold = gu.load('am_ntu_b1_t64')
n, v, num_class, seed, t = [int(i) for i in gold['meta']]
model = Model(num_class=num_class, num_point=v, num_person=2, graph='graph.ntu_rgb_d.Graph', graph_args=dict(labeling_mode='spatial'))
shapes = orc.aagcn_model_param_shapes(num_class, v)
model.load_state_dict(orc.aagcn_randomized_state(shapes, seed, stress=float(gold['meta.stress'])))
model.to(dev)
xn, lab = gu.model_inputs(n, v, num_class, seed, t)
eps = float(os.environ.get('PERTURB', '0'))
if eps: xn = (xn * (1 + eps * np.random.default_rng(1).standard_normal(xn.shape))).astype(np.float32)
model.train()
logits, _ = model(torch.from_numpy(xn).to(dev))
loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(lab).to(dev))
loss.backward()
print('mode', os.environ.get('AGCN_GEMM'), 'perturb', eps, 'logits err', gu.rel_err(logits.detach().cpu().numpy(), gold['logits']))
bad = []
for k, p in model.named_parameters():
    if gu.is_zero_grad_bias(k): continue
    ok, e32, e64, noise = gu.grad_check(p.grad.cpu().numpy(), gold, k, 1e-3)
    if not ok: bad.append((k, e32, e64, noise))
print('bad', len(bad))
for b in bad: print('  %-40s e32 %.2e e64 %.2e noise %.2e' % b)
