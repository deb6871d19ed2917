"""Pin the AAGCN part of the CPU oracle against fixtures produced by the REFERENCE aagcn.py
(tests/golden/make_golden.py aagcn).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import agcn_oracle as orc
from tests import golden_util as gu

TOL = 2e-5


@pytest.mark.parametrize('name', gu.AAGCN_UNIT_NAMES)
def test_aagcn_unit_matches_reference(name):
    gold = gu.load(name)
    cin, cout, stride, residual, t, v, seed, adaptive, attention = [int(i) for i in gold['meta']]
    gbn = gu.meta_int(gold, 'meta.gbn') or None
    shapes = orc.aagcn_unit_param_shapes('', cin, cout, v, stride, bool(residual), bool(adaptive), bool(attention), gbn)
    sd0 = orc.aagcn_randomized_state(shapes, seed, stress=float(gold['meta.stress']))
    A = gu.graph_A(v)
    xn, rn = gu.unit_inputs(cin, cout, stride, t, v, seed, n=gu.meta_int(gold, 'meta.n', 2))
    # the fixture was made as: load_state_dict, .eval() (GhostBatchNorm collates its running statistics there),
    # eval forward, .train(), training forward -- the same sequence on the functional state dict
    sd0 = orc.ghost_collate(sd0)
    sd = orc.with_grad(sd0)
    with torch.no_grad():
        ye = orc.aagcn_unit_forward(torch.from_numpy(xn), sd, '', A, stride, bool(residual), training=False)
    assert gu.rel_err(ye.numpy(), gold['y_eval']) < TOL
    sd = orc.with_grad(sd0)
    for k in list(sd):                       # conv_d aliases share one tensor, as in the reference module
        if gu.is_alias_key(k):
            sd[k] = sd[gu.canonical_key(k)]
    x = torch.from_numpy(xn).requires_grad_(True)
    y = orc.aagcn_unit_forward(x, sd, '', A, stride, bool(residual), training=True)
    (y * torch.from_numpy(rn)).sum().backward()
    assert gu.rel_err(y.detach().numpy(), gold['y']) < TOL
    assert gu.rel_err(x.grad.numpy(), gold['dx']) < TOL * max(1.0, np.abs(gold['dx']).max())
    for k, p in sd.items():
        if orc.is_buffer(k) or gu.is_alias_key(k) or gu.is_zero_grad_bias(k):
            continue
        ok, e32, e64, noise = gu.grad_check(p.grad.numpy(), gold, k, 1e-4)
        assert ok, (k, e32, e64, noise)
    for k, b in sd.items():
        if k.endswith(('running_mean', 'running_var')):
            assert gu.rel_err(b.numpy(), gold['buf.' + k]) < TOL, k


@pytest.mark.parametrize('name', gu.AAGCN_MODEL_NAMES)
def test_aagcn_model_matches_reference(name):
    """Full model, the 3/6/7-layer backbones of reference aagcn.py:407-428 and a GhostBatchNorm model."""
    gold = gu.load(name)
    n, v, num_class, seed, t = [int(i) for i in gold['meta']]
    layers, gbn = gu.meta_int(gold, 'meta.layers', 10), gu.meta_int(gold, 'meta.gbn') or None
    sd0 = orc.aagcn_randomized_state(orc.aagcn_model_param_shapes(num_class, v, model_layers=layers, gbn_split=gbn),
                                     seed, stress=float(gold['meta.stress']))
    xn, lab = gu.model_inputs(n, v, num_class, seed, t)
    sd0 = orc.ghost_collate(sd0)
    lyr = orc.AAGCN_LAYER_SUBSETS[layers]
    sd = orc.with_grad(sd0)
    with torch.no_grad():
        le = orc.aagcn_model_forward(torch.from_numpy(xn), sd, None, training=False, layers=lyr)
    assert gu.rel_err(le.numpy(), gold['logits_eval']) < TOL
    sd = orc.with_grad(sd0)
    for k in list(sd):
        if gu.is_alias_key(k):
            sd[k] = sd[gu.canonical_key(k)]
    logits = orc.aagcn_model_forward(torch.from_numpy(xn), sd, None, training=True, layers=lyr)
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(lab))
    loss.backward()
    assert gu.rel_err(logits.detach().numpy(), gold['logits']) < TOL
    assert abs(loss.item() - float(gold['loss'])) < TOL * max(1.0, abs(float(gold['loss'])))
    bad = []
    for k, p in sd.items():
        if orc.is_buffer(k) or gu.is_alias_key(k) or gu.is_zero_grad_bias(k):
            continue
        ok, e32, e64, noise = gu.grad_check(p.grad.numpy(), gold, k, 2e-4)
        if not ok:
            bad.append((k, e32, e64, noise))
    assert not bad, bad[:5]
