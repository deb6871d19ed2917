"""Run in a process of its own with AGCN_GEMM=bf16 (the mode is fixed per process): BASELINE configs[3] arithmetic --
plain bf16 MFMA operands (one product per fp32 product), fp32 accumulation and storage -- against the fp32 fixtures
generated from the reference.  Tolerance (the reference has no bf16 path, so the build defines it against the
fp32 oracle): 2e-2 (max-normalised) on outputs / logits / loss, and on dx / parameter gradients with the ReLU patterns of
the bf16 run imposed on the fp64 oracle (see below); single-scalar parameters (one cancelling sum over the unit) 6e-2.
Prints one line per fixture; exit code 1 on a violation."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import agcn_amd  # noqa: E402,F401
from agcn_amd import lib  # noqa: E402
from oracle import agcn_oracle as orc  # noqa: E402
from tests import golden_util as gu  # noqa: E402

assert lib.load().agcn_gemm_mode() == b'bf16', 'run with AGCN_GEMM=bf16'
dev = torch.device('cuda:0')
TOL, STOL = 2e-2, 6e-2
bad = 0

# Units: outputs against the fp32 fixture; gradients against the fp64 oracle evaluated WITH THE bf16 RUN'S OWN ReLU
# PATTERNS.  A 3e-3 forward error flips the sign of ~0.3 % of the pre-activations, and a flipped element's gradient is
# wrong by its whole value -- in ANY bf16 implementation -- so an element-wise comparison with free masks measures mask
# flips (observed: 0.1-0.5 of max|g|), not the arithmetic.  With the patterns pinned it measures the arithmetic.
for name in ('au_64_64_s1_v25', 'au_64_128_s2_v25'):
    from agcn_amd.model.aagcn import AdaptiveGCN, TCNGCNUnit
    gold = gu.load(name)
    cin, cout, stride, residual, t, v, seed, adaptive, attention = [int(i) for i in gold['meta']]
    unit = TCNGCNUnit(cin, cout, gu.graph_A(v).numpy(), stride=stride, residual=bool(residual), adaptive=AdaptiveGCN,
                      attention=bool(attention))
    shapes = orc.aagcn_unit_param_shapes('', cin, cout, v, stride, bool(residual), True, bool(attention))
    sd0 = orc.aagcn_randomized_state(shapes, seed, stress=float(gold['meta.stress']))
    unit.load_state_dict(sd0)
    unit.to(dev).train()
    xn, rn = gu.unit_inputs(cin, cout, stride, t, v, seed)
    x = torch.from_numpy(xn).to(dev).requires_grad_(True)
    y = unit(x)
    (y * torch.from_numpy(rn).to(dev)).sum().backward()
    ey = gu.rel_err(y.detach().cpu().numpy(), gold['y'])
    with torch.no_grad():                       # ReLU pattern of the GCN core (before the attention gates)
        keep = (unit.gcn1.attn_s, unit.gcn1.attn_t, unit.gcn1.attn_c)
        unit.gcn1.attn_s = unit.gcn1.attn_t = unit.gcn1.attn_c = None
        g_k = unit.gcn1(x.detach())
        unit.gcn1.attn_s, unit.gcn1.attn_t, unit.gcn1.attn_c = keep
    masks = ((g_k > 0).double().cpu(), (y.detach() > 0).double().cpu())
    sd = orc.with_grad({k: (v_.double() if v_.is_floating_point() else v_) for k, v_ in sd0.items()})
    for k in list(sd):
        if gu.is_alias_key(k):
            sd[k] = sd[gu.canonical_key(k)]
    xo = torch.from_numpy(xn).double().requires_grad_(True)
    yo = orc.aagcn_unit_forward(xo, sd, '', None, stride, bool(residual), training=True, masks=masks)
    (yo * torch.from_numpy(rn).double()).sum().backward()
    edx = float((x.grad.double().cpu() - xo.grad).abs().max() / xo.grad.abs().max())
    worst, wname = 0.0, ''
    for k, p in unit.named_parameters():
        if gu.is_zero_grad_bias(k):
            continue
        ref = sd[k].grad
        den = float(ref.abs().max())
        if p.numel() == 1:
            # a single-scalar parameter (attention conv bias, alpha) is ONE sum over the whole unit that nearly cancels
            # (|g| << sum |terms|): its error is measured on the scale of its sibling tensor's gradient, which is built
            # from the same upstream terms (conv bias <-> conv weight, alpha <-> PA)
            sib = k[:-4] + 'weight' if k.endswith('bias') else k.replace('alpha', 'PA')
            den = max(den, float(sd[sib].grad.abs().max()))
        e = float((p.grad.double().cpu() - ref).abs().max() / max(1e-30, den))
        tol = STOL if p.numel() == 1 else TOL
        if e > worst:
            worst, wname = e, k
        if e > tol:
            bad += 1
            print(f'  FAIL {name} {k}: {e:.3e} > {tol}')
    print(f'bf16 {name}: y {ey:.2e} (vs fp32 fixture); with the ReLU patterns pinned, vs the fp64 oracle: dx {edx:.2e}, '
          f'worst parameter gradient {worst:.2e} [{wname}]')
    bad += int(ey > TOL) + int(edx > TOL)

from model.aagcn import Model  # noqa: E402
gold = gu.load('am_ntu_b1_t64')
n, v, num_class, seed, t = [int(i) for i in gold['meta']]
model = Model(num_class=num_class, num_point=v, num_person=2, graph='graph.ntu_rgb_d.Graph',
              graph_args=dict(labeling_mode='spatial'))
model.load_state_dict(orc.aagcn_randomized_state(orc.aagcn_model_param_shapes(num_class, v), seed,
                                                 stress=float(gold['meta.stress'])))
model.to(dev).train()
xn, lab = gu.model_inputs(n, v, num_class, seed, t)
logits, _ = model(torch.from_numpy(xn).to(dev))
loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(lab).to(dev))
loss.backward()
el = gu.rel_err(logits.detach().cpu().numpy(), gold['logits'])
eloss = abs(loss.item() - float(gold['loss'])) / max(1.0, abs(float(gold['loss'])))
print(f'bf16 am_ntu_b1_t64: logits {el:.2e} loss {eloss:.2e} (fp32 fixture; gradients of the full model are not '
      f'compared tensor by tensor: ReLU-kink conditioned, see test_gpu_parity.py)')
bad += int(el > TOL) + int(eloss > TOL)
sys.exit(1 if bad else 0)
