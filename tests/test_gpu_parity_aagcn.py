"""AAGCN (BASELINE config 4, fp32) on the HIP path vs fixtures generated from the REFERENCE aagcn.py.  GPU only."""
import numpy as np
import pytest
import torch

from oracle import agcn_oracle as orc
from tests import golden_util as gu

pytestmark = pytest.mark.gpu
TOL, GTOL = 1e-4, 2e-4


def _gpu():
    import agcn_amd  # noqa: F401
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device('cuda:0')


@pytest.mark.parametrize('name', gu.AAGCN_UNIT_NAMES)
def test_aagcn_unit_golden(name):
    dev = _gpu()
    from agcn_amd.model.aagcn import AdaptiveGCN, NonAdaptiveGCN, TCNGCNUnit
    gold = gu.load(name)
    cin, cout, stride, residual, t, v, seed, adaptive, attention = [int(i) for i in gold['meta']]
    A = gu.graph_A(v).numpy()
    unit = TCNGCNUnit(cin, cout, A, stride=stride, residual=bool(residual),
                      adaptive=AdaptiveGCN if adaptive else NonAdaptiveGCN, attention=bool(attention))
    shapes = orc.aagcn_unit_param_shapes('', cin, cout, v, stride, bool(residual), bool(adaptive), bool(attention))
    assert set(shapes) == set(unit.state_dict().keys())
    unit.load_state_dict(orc.aagcn_randomized_state(shapes, seed, stress=float(gold['meta.stress'])))
    unit.to(dev)
    xn, rn = gu.unit_inputs(cin, cout, stride, t, v, seed)
    unit.eval()
    with torch.no_grad():
        ye = unit(torch.from_numpy(xn).to(dev))
    assert gu.rel_err(ye.cpu().numpy(), gold['y_eval']) < TOL
    unit.train()
    x = torch.from_numpy(xn).to(dev).requires_grad_(True)
    y = unit(x)
    (y * torch.from_numpy(rn).to(dev)).sum().backward()
    assert gu.rel_err(y.detach().cpu().numpy(), gold['y']) < TOL
    assert float(np.abs(x.grad.cpu().numpy() - gold['dx']).max()) / max(1.0, float(np.abs(gold['dx']).max())) < GTOL
    for k, p in unit.named_parameters():
        if gu.is_zero_grad_bias(k):
            assert float(p.grad.abs().max()) < 1e-5, k
            continue
        # single-scalar parameters (attention conv biases, alpha): their gradient is ONE sum over every element of
        # the unit, so a single ReLU-kink flip (element within rounding of 0) moves it by ~1e-3 of its value
        tol = 5e-3 if p.numel() == 1 else GTOL
        ok, e32, e64, noise = gu.grad_check(p.grad.cpu().numpy(), gold, k, tol)
        assert ok, (k, e32, e64, noise)
    for k, b in unit.state_dict().items():
        if k.endswith(('running_mean', 'running_var')):
            assert gu.rel_err(b.cpu().numpy(), gold['buf.' + k]) < TOL, k


def test_aagcn_model_golden():
    dev = _gpu()
    from model.aagcn import Model
    gold = gu.load('am_ntu_b1_t64')
    n, v, num_class, seed, t = [int(i) for i in gold['meta']]
    model = Model(num_class=num_class, num_point=v, num_person=2, graph='graph.ntu_rgb_d.Graph',
                  graph_args=dict(labeling_mode='spatial'))
    shapes = orc.aagcn_model_param_shapes(num_class, v)
    assert set(shapes) == set(model.state_dict().keys())
    assert sum(p.numel() for p in model.parameters()) == 3781668
    model.load_state_dict(orc.aagcn_randomized_state(shapes, seed, stress=float(gold['meta.stress'])))
    model.to(dev)
    xn, lab = gu.model_inputs(n, v, num_class, seed, t)
    model.eval()
    with torch.no_grad():
        le, aux = model(torch.from_numpy(xn).to(dev))
    assert aux is None
    assert gu.rel_err(le.cpu().numpy(), gold['logits_eval']) < TOL
    model.train()
    logits, _ = model(torch.from_numpy(xn).to(dev))
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(lab).to(dev))
    loss.backward()
    assert gu.rel_err(logits.detach().cpu().numpy(), gold['logits']) < TOL
    assert abs(loss.item() - float(gold['loss'])) < TOL * max(1.0, abs(float(gold['loss'])))
    bad = []
    for k, p in model.named_parameters():
        if gu.is_zero_grad_bias(k):
            continue
        ok, e32, e64, noise = gu.grad_check(p.grad.cpu().numpy(), gold, k, GTOL)
        if not ok:
            bad.append((k, e32, e64, noise))
    assert not bad, bad[:8]
