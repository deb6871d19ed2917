"""Pin the CPU oracle (oracle/agcn_oracle.py) against the fixtures produced by the REFERENCE
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import agcn_oracle as orc
from tests import golden_util as gu

TOL = 2e-5   # oracle vs reference: both fp32 CPU; different op order only


def test_graph_matches_reference():
    import agcn_amd  # noqa: F401
    from agcn_amd.graph import kinetics, ntu_rgb_d
    g = np.load(gu.GOLDEN + '/graphs.npz')
    assert np.array_equal(ntu_rgb_d.Graph().A, g['A_v25'])
    assert np.array_equal(kinetics.Graph().A, g['A_v18'])
    # column-normalised: every non-empty column sums to one
    a = ntu_rgb_d.Graph().A
    assert a.shape == (3, 25, 25)
    assert int((a[0] != 0).sum()) == 25 and int((a[1] != 0).sum()) == 24 and int((a[2] != 0).sum()) == 24


@pytest.mark.parametrize('name', gu.UNIT_NAMES)
def test_unit_matches_reference(name):
    gold = gu.load(name)
    cin, cout, stride, residual, t, v, seed = [int(i) for i in gold['meta']]
    shapes = orc.unit_param_shapes('', cin, cout, v, stride, bool(residual))
    sd0 = orc.randomized_state(shapes, seed, stress=float(gold['meta.stress']))
    A = gu.graph_A(v)
    xn, rn = gu.unit_inputs(cin, cout, stride, t, v, seed)
    # eval mode
    sd = orc.with_grad(sd0)
    with torch.no_grad():
        ye = orc.tcn_gcn_unit_forward(torch.from_numpy(xn), sd, '', A, stride, bool(residual), training=False)
    assert gu.rel_err(ye.numpy(), gold['y_eval']) < TOL
    # train mode + backward
    sd = orc.with_grad(sd0)
    x = torch.from_numpy(xn).requires_grad_(True)
    y = orc.tcn_gcn_unit_forward(x, sd, '', A, stride, bool(residual), training=True)
    (y * torch.from_numpy(rn)).sum().backward()
    assert gu.rel_err(y.detach().numpy(), gold['y']) < TOL
    assert gu.rel_err(x.grad.numpy(), gold['dx']) < TOL * max(1.0, np.abs(gold['dx']).max())
    for k, p in sd.items():
        if orc.is_buffer(k):
            if k.endswith(('running_mean', 'running_var')):
                assert gu.rel_err(p.numpy(), gold['buf.' + k]) < TOL
            continue
        if gu.is_zero_grad_bias(k):
            assert float(p.grad.abs().max()) < 1e-4 * max(1.0, float(np.abs(gold['dx']).max()))
            continue
        assert gu.grad_err(p.grad.numpy(), gold, k) < 1e-4, k


@pytest.mark.parametrize('name', gu.MODEL_NAMES)
def test_model_matches_reference(name):
    gold = gu.load(name)
    n, v, num_class, seed, t = [int(i) for i in gold['meta']]
    sd0 = orc.randomized_state(orc.model_param_shapes(num_class, v), seed, stress=float(gold['meta.stress']))
    A = gu.graph_A(v)
    xn, lab = gu.model_inputs(n, v, num_class, seed, t)
    sd = orc.with_grad(sd0)
    with torch.no_grad():
        le = orc.model_forward(torch.from_numpy(xn), sd, A, training=False)
    assert gu.rel_err(le.numpy(), gold['logits_eval']) < TOL
    sd = orc.with_grad(sd0)
    logits = orc.model_forward(torch.from_numpy(xn), sd, A, training=True)
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(lab))
    loss.backward()
    assert gu.rel_err(logits.detach().numpy(), gold['logits']) < TOL
    assert abs(loss.item() - float(gold['loss'])) < TOL * max(1.0, abs(float(gold['loss'])))
    worst = 0.0
    for k, p in sd.items():
        if orc.is_buffer(k) or gu.is_zero_grad_bias(k):
            continue
        worst = max(worst, gu.grad_err(p.grad.numpy(), gold, k))
    assert worst < 2e-4, worst
