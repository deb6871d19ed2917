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
    gbn = gu.meta_int(gold, 'meta.gbn') or None          # GhostBatchNorm fixtures (reference ghostbatchnorm.py)
    A = gu.graph_A(v).numpy()
    unit = TCNGCNUnit(cin, cout, A, stride=stride, residual=bool(residual),
                      adaptive=AdaptiveGCN if adaptive else NonAdaptiveGCN, attention=bool(attention), gbn_split=gbn)
    shapes = orc.aagcn_unit_param_shapes('', cin, cout, v, stride, bool(residual), bool(adaptive), bool(attention), gbn)
    assert set(shapes) == set(unit.state_dict().keys())
    unit.load_state_dict(orc.aagcn_randomized_state(shapes, seed, stress=float(gold['meta.stress'])))
    unit.to(dev)
    xn, rn = gu.unit_inputs(cin, cout, stride, t, v, seed, n=gu.meta_int(gold, 'meta.n', 2))
    unit.eval()
    with torch.no_grad():
        ye = unit(torch.from_numpy(xn).to(dev))
    assert gu.rel_err(ye.cpu().numpy(), gold['y_eval']) < TOL
    unit.train()
    x = torch.from_numpy(xn).to(dev).requires_grad_(True)
    y = unit(x)
    (y * torch.from_numpy(rn).to(dev)).sum().backward()
    assert gu.audit_value(name, 'y', gu.rel_err(y.detach().cpu().numpy(), gold['y']), TOL)
    assert gu.audit_value(name, 'dx', float(np.abs(x.grad.cpu().numpy() - gold['dx']).max()) /
                          max(1.0, float(np.abs(gold['dx']).max())), GTOL)
    # single-scalar parameters (attention conv biases, alpha): their gradient is ONE sum over every element of
    # the unit, so a single ReLU-kink flip (element within rounding of 0) moves it by ~1e-3 of its value
    bad, rec = gu.audit_grads(name, [(k, p.grad.cpu().numpy()) for k, p in unit.named_parameters()], gold, GTOL,
                              scalar_tol=5e-3)
    assert not bad, bad[:8]
    assert rec['primary'] >= 0.9 * rec['tensors'], rec
    for k, b in unit.state_dict().items():
        if k.endswith(('running_mean', 'running_var')):
            assert gu.rel_err(b.cpu().numpy(), gold['buf.' + k]) < TOL, k


@pytest.mark.parametrize('name', gu.AAGCN_MODEL_NAMES)
def test_aagcn_model_golden(name):
    """Full AAGCN, the 3/6/7-layer backbones (reference aagcn.py:407-428) and a GhostBatchNorm model."""
    dev = _gpu()
    from model.aagcn import Model
    gold = gu.load(name)
    n, v, num_class, seed, t = [int(i) for i in gold['meta']]
    layers, gbn = gu.meta_int(gold, 'meta.layers', 10), gu.meta_int(gold, 'meta.gbn') or None
    model = Model(num_class=num_class, num_point=v, num_person=2, graph='graph.ntu_rgb_d.Graph',
                  graph_args=dict(labeling_mode='spatial'), model_layers=layers, gbn_split=gbn)
    shapes = orc.aagcn_model_param_shapes(num_class, v, model_layers=layers, gbn_split=gbn)
    assert set(shapes) == set(model.state_dict().keys())
    if layers == 10:
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
    fx = name
    assert gu.audit_value(fx, 'logits', gu.rel_err(logits.detach().cpu().numpy(), gold['logits']), TOL)
    assert gu.audit_value(fx, 'loss', abs(loss.item() - float(gold['loss'])) / max(1.0, abs(float(gold['loss']))), TOL)
    # full-model gradients are ReLU-kink conditioned (see test_gpu_parity.py::test_model_golden): every tensor must
    # pass grad_check; the audit prints by which route; the kink-free check is the layer-wise test below
    bad, rec = gu.audit_grads(fx, [(k, p.grad.cpu().numpy()) for k, p in model.named_parameters()], gold, GTOL)
    assert not bad, bad[:8]


def test_aagcn_model_layerwise_vs_oracle():
    """The kink-free check for AAGCN (mirror of test_gpu_parity.py::test_model_layerwise_vs_oracle_full_size): run the
    HIP model end to end at the NTU shape, capture every TCNGCNUnit's input x_k and output gradient dy_k, re-evaluate
    each layer with the fp64 CPU oracle on exactly those tensors with the HIP path's own ReLU patterns imposed, and
    compare y_k, dx_k and every parameter gradient of the layer at 2e-4."""
    dev = _gpu()
    from model.aagcn import Model
    gold = gu.load('am_ntu_b1_t64')
    n, v, num_class, seed, t = [int(i) for i in gold['meta']]
    shapes = orc.aagcn_model_param_shapes(num_class, v)
    sd0 = orc.aagcn_randomized_state(shapes, seed, stress=float(gold['meta.stress']))
    model = Model(num_class=num_class, num_point=v, num_person=2, graph='graph.ntu_rgb_d.Graph',
                  graph_args=dict(labeling_mode='spatial'))
    model.load_state_dict(sd0)
    model.to(dev).train()
    cap = {}

    def pre(k):
        def hook(mod, inp):
            inp[0].retain_grad()
            cap[('x', k)] = inp[0]
        return hook

    def post(k):
        def hook(mod, inp, out):
            out.retain_grad()
            cap[('y', k)] = out
        return hook

    def gcn_relu(k):        # the pre-attention activation of GCNUnit: captured from the HIP core's own output
        def hook(mod, inp, out):
            cap[('g', k)] = out.detach()
        return hook
    for k in range(1, 11):
        getattr(model, f'l{k}').register_forward_pre_hook(pre(k))
        getattr(model, f'l{k}').register_forward_hook(post(k))
    xn, lab = gu.model_inputs(n, v, num_class, seed, t)
    logits, _ = model(torch.from_numpy(xn).to(dev))
    torch.nn.functional.cross_entropy(logits, torch.from_numpy(lab).to(dev)).backward()
    fx = 'layerwise_am_ntu_b1_t64(masks imposed)'
    worst, worst_s = {}, {}
    for k, (cin, cout, stride, res) in enumerate(orc.LAYERS, start=1):
        unit = getattr(model, f'l{k}')
        x_k, y_k = cap[('x', k)], cap[('y', k)]
        # ReLU pattern of the GCN core (before the attention gates): re-run the HIP core on the same input
        with torch.no_grad():
            gu_attn = (unit.gcn1.attn_s, unit.gcn1.attn_t, unit.gcn1.attn_c)
            unit.gcn1.attn_s = unit.gcn1.attn_t = unit.gcn1.attn_c = None
            g_k = unit.gcn1(x_k.detach())
            unit.gcn1.attn_s, unit.gcn1.attn_t, unit.gcn1.attn_c = gu_attn
        masks = ((g_k > 0).double().cpu(), (y_k.detach() > 0).double().cpu())
        sub = {kk[len(f'l{k}.'):]: vv for kk, vv in sd0.items() if kk.startswith(f'l{k}.')}
        sd = orc.with_grad({kk: (vv.double() if vv.is_floating_point() else vv) for kk, vv in sub.items()})
        for kk in list(sd):                      # conv_d aliases share one tensor, as in the reference module
            if gu.is_alias_key(kk):
                sd[kk] = sd[gu.canonical_key(kk)]
        xo = x_k.detach().double().cpu().requires_grad_(True)
        yo = orc.aagcn_unit_forward(xo, sd, '', None, stride, res, training=True, masks=masks)
        yo.backward(y_k.grad.double().cpu())
        e_y = gu.rel_err(y_k.detach().cpu().numpy(), yo.detach().numpy())
        assert gu.audit_value(fx, f'l{k}.y', e_y, TOL), (k, e_y)
        e_dx = float((x_k.grad.double().cpu() - xo.grad).abs().max()) / max(1e-30, float(xo.grad.abs().max()))
        assert gu.audit_value(fx, f'l{k}.dx', e_dx, GTOL), (k, 'dx', e_dx)
        for kk, p in unit.named_parameters():
            if gu.is_zero_grad_bias(kk):
                continue
            ref = sd[kk].grad
            e = float((p.grad.double().cpu() - ref).abs().max()) / max(1e-30, float(ref.abs().max()))
            # single-scalar parameters (attention conv biases, alpha) are ONE cancelling sum over the whole unit:
            # 5e-3 as in test_aagcn_unit_golden; everything else 2e-4
            tol = 5e-3 if p.numel() == 1 else GTOL
            (worst_s if p.numel() == 1 else worst)[f'l{k}.{kk}'] = e
            assert e < tol, (k, kk, e)
    wk = max(worst, key=worst.get)
    gu.audit_value(fx, f'worst param grad [{wk}]', worst[wk], GTOL)
    ws_ = max(worst_s, key=worst_s.get)
    gu.audit_value(fx, f'worst single-scalar param grad [{ws_}]', worst_s[ws_], 5e-3)


@pytest.mark.parametrize('fixture', ['am_ntu_b1_t64', 'am_ntu_l3_t32'])
def test_aagcn_end_to_end_grads_with_pinned_relu_patterns(fixture):
    """END-TO-END gradient parity for AAGCN without the ReLU-kink lottery (mirror of
    test_gpu_parity.py::test_model_end_to_end_grads_with_pinned_relu_patterns): the HIP model runs forward + backward; the
    fp64 CPU oracle runs the WHOLE network (attention gates included) with every ReLU pattern of the HIP forward imposed,
    and every parameter gradient is compared at the primary 2e-4 of max|g| -- no band; single-scalar parameters (the gate
    convolutions' biases, alpha: ONE sum over the whole activation) at 5e-3 like the unit fixtures."""
    dev = _gpu()
    from model.aagcn import Model
    gold = gu.load(fixture)
    n, v, num_class, seed, t = [int(i) for i in gold['meta']]
    layers = gu.meta_int(gold, 'meta.layers', 10)
    shapes = orc.aagcn_model_param_shapes(num_class, v, model_layers=layers)
    sd0 = orc.aagcn_randomized_state(shapes, seed, stress=float(gold['meta.stress']))
    model = Model(num_class=num_class, num_point=v, num_person=2, graph='graph.ntu_rgb_d.Graph',
                  graph_args=dict(labeling_mode='spatial'), model_layers=layers)
    model.load_state_dict(sd0)
    model.to(dev).train()
    ks = list(orc.AAGCN_LAYER_SUBSETS[layers])
    cap = {}
    for k in ks:
        getattr(model, f'l{k}').register_forward_pre_hook(lambda mod, inp, k=k: cap.__setitem__(('x', k), inp[0].detach()))
        getattr(model, f'l{k}').register_forward_hook(lambda mod, inp, out, k=k: cap.__setitem__(('y', k), out.detach()))
    xn, lab = gu.model_inputs(n, v, num_class, seed, t)
    logits, _ = model(torch.from_numpy(xn).to(dev))
    torch.nn.functional.cross_entropy(logits, torch.from_numpy(lab).to(dev)).backward()
    masks = {}
    with torch.no_grad():
        for k in ks:
            unit = getattr(model, f'l{k}')
            saved = (unit.gcn1.attn_s, unit.gcn1.attn_t, unit.gcn1.attn_c)
            unit.gcn1.attn_s = unit.gcn1.attn_t = unit.gcn1.attn_c = None      # the GCN core's own ReLU pattern
            g_k = unit.gcn1(cap[('x', k)])
            unit.gcn1.attn_s, unit.gcn1.attn_t, unit.gcn1.attn_c = saved
            masks[k] = ((g_k > 0).double().cpu(), (cap[('y', k)] > 0).double().cpu())
    sd = orc.with_grad({kk: (vv.double() if vv.is_floating_point() else vv) for kk, vv in sd0.items()})
    lo = orc.aagcn_model_forward(torch.from_numpy(xn).double(), sd, gu.graph_A(v).double(), training=True,
                                 layers=ks, masks=masks)
    torch.nn.functional.cross_entropy(lo, torch.from_numpy(lab)).backward()
    fx = f'end_to_end_{fixture}(all ReLU patterns pinned, vs fp64 oracle)'
    assert gu.audit_value(fx, 'logits', gu.rel_err(logits.detach().cpu().numpy(), lo.detach().numpy()), TOL)
    errs, worst, wname = [], 0.0, ''
    seen = set()
    for k, p in model.named_parameters():
        if gu.is_zero_grad_bias(k) or id(p) in seen:
            continue
        seen.add(id(p))
        ref = sd[k].grad
        if ref is None:                           # (aliased conv_d entries: the gradient sits on the other name)
            ref = sd[k.replace('gcn1.conv_d.', 'gcn1.agcn.conv_d.')].grad
        e = float((p.grad.double().cpu() - ref).abs().max()) / max(1e-30, float(ref.abs().max()))
        tol = 5e-3 if p.numel() == 1 else GTOL
        errs.append(e / tol)
        if e / tol > worst:
            worst, wname = e / tol, k
    errs = np.array(errs)
    gu.audit_value(fx, f'worst param grad / its tolerance [{wname}]', worst, 1.0)
    gu.audit_value(fx, 'median param grad err / tolerance', float(np.median(errs)), 1.0)
    frac = float((errs <= 1.0).mean())
    gu.audit_value(fx, f'share of {len(errs)} tensors outside the primary criterion', 1.0 - frac, 0.1)
    assert frac >= 0.9, (frac, worst, wname)
