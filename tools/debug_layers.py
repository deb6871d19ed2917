"""Debug: per-layer activation / input-gradient comparison HIP vs CPU oracle (fp32 and fp64)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import agcn_oracle as orc
from tests import golden_util as gu
import agcn_amd
from model.agcn import Model

def run_oracle(xn, lab, sd0, A, dtype):
    sd = orc.with_grad({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd0.items()})
    acts, grads = {}, {}
    x = torch.from_numpy(xn).to(dtype)
    n, c, t, v, m = x.shape
    h = x.permute(0, 4, 3, 1, 2).reshape(n, m * v * c, t)
    h = orc._bn(h, sd, 'data_bn.', True)
    h = h.reshape(n, m, v, c, t).permute(0, 1, 3, 4, 2).reshape(n * m, c, t, v)
    for k, (_, _, stride, res) in enumerate(orc.LAYERS, start=1):
        h.retain_grad(); acts[k] = h
        h = orc.tcn_gcn_unit_forward(h, sd, f'l{k}.', A.to(dtype), stride, res, True)
    h.retain_grad(); acts[11] = h
    cn = h.shape[1]
    logits = torch.nn.functional.linear(h.reshape(n, m, cn, -1).mean(3).mean(1), sd['fc.weight'], sd['fc.bias'])
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(lab)); loss.backward()
    return acts, sd, logits

name = sys.argv[1] if len(sys.argv) > 1 else 'm_kin_b2_t64'
gold = gu.load(name)
n, v, nc, seed, t = [int(i) for i in gold['meta']]
sd0 = orc.randomized_state(orc.model_param_shapes(nc, v), seed, stress=float(gold['meta.stress']))
A = gu.graph_A(v)
xn, lab = gu.model_inputs(n, v, nc, seed, t)
a32, sd32, _ = run_oracle(xn, lab, sd0, A, torch.float32)
a64, sd64, _ = run_oracle(xn, lab, sd0, A, torch.float64)
dev = torch.device('cuda:0')
graph = {25: 'graph.ntu_rgb_d.Graph', 18: 'graph.kinetics.Graph'}[v]
model = Model(num_class=nc, num_point=v, num_person=2, graph=graph, graph_args=dict(labeling_mode='spatial'))
model.load_state_dict(sd0); model.to(dev).train()
hacts = {}
def mk(k):
    def hook(mod, inp):
        inp[0].retain_grad(); hacts[k] = inp[0]
    return hook
for k in range(1, 11):
    getattr(model, f'l{k}').register_forward_pre_hook(mk(k))
def post(mod, inp, out):
    out.retain_grad(); hacts[11] = out
model.l10.register_forward_hook(post)
logits = model(torch.from_numpy(xn).to(dev))
torch.nn.functional.cross_entropy(logits, torch.from_numpy(lab).to(dev)).backward()
def r(a, b): return float((a.double().cpu() - b.double()).abs().max() / max(1e-30, float(b.double().abs().max())))
for k in range(1, 12):
    print(f'layer {k}: act hip-vs-64 {r(hacts[k], a64[k]):.2e} (ref32-vs-64 {r(a32[k], a64[k]):.2e}) | '
          f'dact hip-vs-64 {r(hacts[k].grad, a64[k].grad):.2e} (ref32-vs-64 {r(a32[k].grad, a64[k].grad):.2e})')
rows = []
for k, p in model.named_parameters():
    e = r(p.grad, sd64[k].grad); nz = r(sd32[k].grad, sd64[k].grad)
    rows.append((e / max(nz, 2e-5), e, nz, k))
rows.sort(reverse=True)
for row in rows[:25]: print('%.1f  hip-vs-64 %.2e  ref32-vs-64 %.2e  %s' % row)
