import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import agcn_oracle as orc
from tests import golden_util as gu
import agcn_amd
from model.agcn import Model
from agcn_amd.model.agcn import TCN_GCN_unit
name = 'm_kin_b2_t64'
gold = gu.load(name)
n, v, nc, seed, t = [int(i) for i in gold['meta']]
sd0 = orc.randomized_state(orc.model_param_shapes(nc, v), seed, stress=float(gold['meta.stress']))
A = gu.graph_A(v)
xn, lab = gu.model_inputs(n, v, nc, seed, t)
dev = torch.device('cuda:0')
model = Model(num_class=nc, num_point=v, num_person=2, graph='graph.kinetics.Graph', graph_args=dict(labeling_mode='spatial'))
model.load_state_dict(sd0); model.to(dev).train()
cap = {}
def pre(mod, inp):
    inp[0].retain_grad(); cap['x'] = inp[0]
def post(mod, inp, out):
    out.retain_grad(); cap['y'] = out
model.l10.register_forward_pre_hook(pre); model.l10.register_forward_hook(post)
logits = model(torch.from_numpy(xn).to(dev))
torch.nn.functional.cross_entropy(logits, torch.from_numpy(lab).to(dev)).backward()
torch.cuda.synchronize()
x_in = cap['x'].detach().clone(); dout = cap['y'].grad.detach().clone(); dx_full = cap['x'].grad.detach().clone()
print('dout stats', float(dout.abs().max()), 'x_in max', float(x_in.abs().max()), 'zeros frac', float((x_in == 0).float().mean()))
# oracle fp64 on the captured tensors; l10 params BEFORE the step == sd0 (running stats were updated but unused in train mode)
sub = {k[len('l10.'):]: v_ for k, v_ in sd0.items() if k.startswith('l10.')}
sd64 = orc.with_grad({k: (v_.double() if v_.is_floating_point() else v_) for k, v_ in sub.items()})
xo = x_in.double().cpu().requires_grad_(True)
yo = orc.tcn_gcn_unit_forward(xo, sd64, '', A.double(), 1, True, training=True)
yo.backward(dout.double().cpu())
def rel(a, b): return float((a.double().cpu() - b.double()).abs().max() / max(1e-30, float(b.double().abs().max())))
print('y full vs oracle', rel(cap['y'].detach(), yo.detach()))
print('dx full vs oracle64', rel(dx_full, xo.grad))
unit = TCN_GCN_unit(256, 256, A.numpy(), stride=1, residual=True); unit.load_state_dict(sub); unit.to(dev).train()
xi = x_in.clone().requires_grad_(True)
yi = unit(xi); yi.backward(dout)
print('dx iso vs oracle64', rel(xi.grad, xo.grad), ' y iso', rel(yi.detach(), yo.detach()))
d = (dx_full.double().cpu() - xo.grad).abs(); thr = 0.01 * float(xo.grad.abs().max())
idx = torch.nonzero(d > thr); print('bad elements', idx.shape[0], 'of', d.numel()); print(idx[:40].tolist())
if idx.shape[0]:
    import collections
    for dim, nm in enumerate('nctv'):
        print(nm, sorted(collections.Counter(idx[:, dim].tolist()).items())[:40])
for k, p in unit.named_parameters():
    e = rel(p.grad, sd64[k].grad); ef = rel(dict(model.l10.named_parameters())[k].grad, sd64[k].grad)
    if (e > 2e-4 or ef > 2e-4) and not gu.is_zero_grad_bias(k): print('  ', k, 'iso %.1e full %.1e' % (e, ef))
