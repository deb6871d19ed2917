"""Generate the golden parity fixtures by running the REFERENCE implementation on CPU.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference module ``model/architecture/aagcn/agcn.py`` is loaded by file path
(``import model`` fails on missing optional deps) with the one shim
SURVEY.md F3 calls for: ``unit_gcn.forward`` does ``self.A.cuda(x.get_device())``
(agcn.py:94), which raises on CPU tensors, so ``Tensor.cuda`` is made an identity
in THIS process only.  Nothing from the reference is copied: the fixtures hold
inputs/outputs (data), and parameters are regenerated from a seeded recipe
(``oracle.agcn_oracle.randomized_state``).
"""
import importlib.util
import os
import sys

import numpy as np
import torch

# Model-level gradient conditioning band: an fp32 evaluation in another summation order reproduces the reference's
# activations to 1-3e-6 relative (measured), so the band is taken at 2e-6, over 8 draws (3 draws at 1e-6 under-sampled
# which ReLU-kink elements flip: a 3e-6 input perturbation of a correct fp32 run moved 13 tensors past that band).
MODEL_SENS_SAMPLES = 8
MODEL_SENS_EPS = 2e-6

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import agcn_oracle as orc  # noqa: E402


def load_reference():
    sys.path.insert(0, REF)          # for ``graph.*`` resolved by the reference's import_class
    # make sure the reference's own graph package wins inside this process
    for k in [k for k in sys.modules if k == 'graph' or k.startswith('graph.')]:
        del sys.modules[k]
    spec = importlib.util.spec_from_file_location(
        'ref_agcn', os.path.join(REF, 'model/architecture/aagcn/agcn.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.Tensor.cuda = lambda self, *a, **k: self      # F3 shim (oracle process only)
    return mod


def ref_graph(v):
    import importlib
    name = {25: 'graph.ntu_rgb_d', 18: 'graph.kinetics'}[v]
    return importlib.import_module(name).Graph(labeling_mode='spatial').A


def sample_idx(numel, k=64):
    return np.unique(np.linspace(0, numel - 1, num=min(k, numel)).astype(np.int64))


def pack_grads(out, named_params, prefix='g.', full_limit=40000):
    """prefix 'g.' = the reference in fp32 (the parity target); 'g64.' = the SAME reference code run in fp64,
    stored rounded to fp32: it measures the reference's own fp32 rounding noise per tensor."""
    for name, p in named_params:
        g = p.grad.detach().numpy().astype(np.float32)
        out[prefix + name + '.norm'] = np.float64(np.linalg.norm(g.astype(np.float64)))
        out[prefix + name + '.absmax'] = np.float32(np.abs(g).max())
        if g.size <= full_limit:
            out[prefix + name] = g
        else:
            idx = sample_idx(g.size, 256)
            out[prefix + name + '.idx'] = idx
            out[prefix + name + '.samples'] = g.reshape(-1)[idx]



def add_unit_sens(out, unit64, xn, rn, seed):
    """Per-tensor conditioning band for a unit fixture: how far the fp64 reference gradient moves when the input is
    perturbed by 1e-6 relative (ReLU-kink flips, cancelling sums); see make_model."""
    g64 = {k: p.grad.detach().clone() for k, p in unit64.named_parameters()}
    sens = {k: 0.0 for k in g64}
    prng = torch.Generator().manual_seed(seed)
    for _ in range(4):
        unit64.zero_grad()
        xp = torch.from_numpy(xn).double()
        xp = (xp * (1.0 + 1e-6 * torch.randn(xp.shape, generator=prng, dtype=torch.float64))).requires_grad_(True)
        (unit64(xp) * torch.from_numpy(rn).double()).sum().backward()
        for k, p in unit64.named_parameters():
            sens[k] = max(sens[k], float((p.grad - g64[k]).abs().max() / max(1e-300, float(g64[k].abs().max()))))
    for k, v_ in sens.items():
        out['sens.' + k] = np.float32(v_)


UNIT_CASES = [
    # name, cin, cout, stride, residual, T, V, seed, stress
    ('u_3_64_s1_v25', 3, 64, 1, False, 16, 25, 111, 1.0),    # seed 101 put a ReLU input within 1e-6 of the kink: its
    #                                                          whole perturbation band was 1e-2 instead of ~1e-6
    ('u_64_64_s1_v25', 64, 64, 1, True, 16, 25, 102, 1.0),
    ('u_64_64_s1_v25_stress', 64, 64, 1, True, 32, 25, 103, 6.0),
    ('u_64_128_s2_v25', 64, 128, 2, True, 16, 25, 104, 1.0),
    ('u_128_256_s2_v25', 128, 256, 2, True, 16, 25, 105, 4.0),
    ('u_64_64_s1_v18', 64, 64, 1, True, 16, 18, 106, 4.0),
    ('u_64_128_s2_v18_oddT', 64, 128, 2, True, 15, 18, 107, 1.0),
]


def unit_inputs(cin, cout, stride, t, v, seed, n=2):
    rng = np.random.default_rng(seed + 5000)
    x = rng.standard_normal((n, cin, t, v)).astype(np.float32)
    tout = (t + 2 * 4 - 9) // stride + 1
    r = rng.standard_normal((n, cout, tout, v)).astype(np.float32)
    return x, r


def make_unit(ref, name, cin, cout, stride, residual, t, v, seed, stress):
    A = ref_graph(v)
    unit = ref.TCN_GCN_unit(cin, cout, A, stride=stride, residual=residual)
    shapes = orc.unit_param_shapes('', cin, cout, v, stride, residual)
    assert set(shapes) == set(unit.state_dict().keys()), \
        (set(shapes) ^ set(unit.state_dict().keys()))
    sd = orc.randomized_state(shapes, seed, stress=stress)
    unit.load_state_dict(sd)
    xn, rn = unit_inputs(cin, cout, stride, t, v, seed)
    out = {}
    # eval-mode forward first (running stats untouched)
    unit.eval()
    with torch.no_grad():
        out['y_eval'] = unit(torch.from_numpy(xn)).numpy()
    # train-mode forward/backward
    unit.train()
    x = torch.from_numpy(xn).requires_grad_(True)
    y = unit(x)
    loss = (y * torch.from_numpy(rn)).sum()
    loss.backward()
    out['y'] = y.detach().numpy()
    out['dx'] = x.grad.numpy()
    out['loss'] = np.float64(loss.item())
    pack_grads(out, unit.named_parameters())
    # the same reference module in fp64 (noise-floor measurement)
    unit64 = ref.TCN_GCN_unit(cin, cout, A, stride=stride, residual=residual).double()
    unit64.load_state_dict({k: v_.double() if v_.is_floating_point() else v_ for k, v_ in sd.items()})
    for m_ in unit64.modules():
        if hasattr(m_, 'A') and isinstance(getattr(m_, 'A'), torch.Tensor) and not isinstance(m_.A, torch.nn.Parameter):
            m_.A = m_.A.double()
    unit64.train()
    x64 = torch.from_numpy(xn).double().requires_grad_(True)
    y64 = unit64(x64)
    (y64 * torch.from_numpy(rn).double()).sum().backward()
    out['y64'] = y64.detach().numpy().astype(np.float32)
    out['dx64'] = x64.grad.numpy().astype(np.float32)
    pack_grads(out, unit64.named_parameters(), prefix='g64.')
    add_unit_sens(out, unit64, xn, rn, seed)
    for k, b in unit.state_dict().items():
        if k.endswith(('running_mean', 'running_var')):
            out['buf.' + k] = b.numpy().copy()
    # diagnostic: spread of the adaptive-adjacency logits of subset 0 (recorded)
    with torch.no_grad():
        sdd = {k: v_.clone() for k, v_ in sd.items()}
        _, s, _ = orc.adaptive_adjacency(torch.from_numpy(xn), sdd, 'gcn1.', torch.from_numpy(A).float(), 0)
        out['meta.std_S'] = np.float32(s.std().item())
    out['meta'] = np.array([cin, cout, stride, int(residual), t, v, seed], dtype=np.int64)
    out['meta.stress'] = np.float32(stress)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(f'{name}: y {out["y"].shape} |y|max {np.abs(out["y"]).max():.3f} std(S) {out["meta.std_S"]:.3f}')


def model_inputs(n, v, num_class, seed, t=300):
    rng = np.random.default_rng(seed + 7000)
    x = rng.standard_normal((n, 3, t, v, 2)).astype(np.float32)
    lab = rng.integers(0, num_class, size=(n,)).astype(np.int64)
    return x, lab


MODEL_CASES = [
    # name, batch, V, num_class, graph, seed, stress, T
    ('m_ntu_b1', 1, 25, 60, 'graph.ntu_rgb_d.Graph', 201, 3.0, 300),
    ('m_ntu_b2', 2, 25, 60, 'graph.ntu_rgb_d.Graph', 202, 3.0, 300),
    ('m_kin_b2_t64', 2, 18, 400, 'graph.kinetics.Graph', 203, 3.0, 64),
    ('m_kin_b2', 2, 18, 400, 'graph.kinetics.Graph', 204, 3.0, 300),      # BASELINE configs[4] at its full T
]


def make_model(ref, name, n, v, num_class, graph, seed, stress, t):
    model = ref.Model(num_class=num_class, num_point=v, num_person=2, graph=graph,
                      graph_args=dict(labeling_mode='spatial'))
    shapes = orc.model_param_shapes(num_class, v)
    assert set(shapes) == set(model.state_dict().keys())
    sd = orc.randomized_state(shapes, seed, stress=stress)
    model.load_state_dict(sd)
    xn, lab = model_inputs(n, v, num_class, seed, t)
    out = {}
    model.eval()
    with torch.no_grad():
        out['logits_eval'] = model(torch.from_numpy(xn)).numpy()
    model.train()
    logits = model(torch.from_numpy(xn))
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(lab))
    loss.backward()
    out['logits'] = logits.detach().numpy()
    out['loss'] = np.float64(loss.item())
    pack_grads(out, model.named_parameters(), full_limit=2000)
    model64 = ref.Model(num_class=num_class, num_point=v, num_person=2, graph=graph,
                        graph_args=dict(labeling_mode='spatial')).double()
    model64.load_state_dict({k: v_.double() if v_.is_floating_point() else v_ for k, v_ in sd.items()})
    for m_ in model64.modules():
        if hasattr(m_, 'A') and isinstance(getattr(m_, 'A'), torch.Tensor) and not isinstance(m_.A, torch.nn.Parameter):
            m_.A = m_.A.double()
    model64.train()
    logits64 = model64(torch.from_numpy(xn).double())
    torch.nn.functional.cross_entropy(logits64, torch.from_numpy(lab)).backward()
    out['logits64'] = logits64.detach().numpy().astype(np.float32)
    pack_grads(out, model64.named_parameters(), prefix='g64.', full_limit=2000)
    # Conditioning of the gradients: the network has ~1e6 ReLU inputs per layer, a few of them within 1e-6 of
    # the kink; ANY perturbation of that size (fp32 rounding, another summation order) flips their masks and
    # moves some parameter gradients by ~1e-2 of max|g|.  Measure it with the reference itself (fp64, inputs
    # perturbed by MODEL_SENS_EPS relative, MODEL_SENS_SAMPLES draws) and store the per-tensor band next to the gradients.
    g64 = {k: p.grad.detach().clone() for k, p in model64.named_parameters()}
    sens = {k: 0.0 for k in g64}
    prng = torch.Generator().manual_seed(seed)
    for _ in range(MODEL_SENS_SAMPLES):
        model64.zero_grad()
        xp = torch.from_numpy(xn).double()
        xp = xp * (1.0 + MODEL_SENS_EPS * torch.randn(xp.shape, generator=prng, dtype=torch.float64))
        torch.nn.functional.cross_entropy(model64(xp), torch.from_numpy(lab)).backward()
        for k, p in model64.named_parameters():
            d = float((p.grad - g64[k]).abs().max() / max(1e-300, float(g64[k].abs().max())))
            sens[k] = max(sens[k], d)
    for k, v_ in sens.items():
        out['sens.' + k] = np.float32(v_)
    out['meta'] = np.array([n, v, num_class, seed, t], dtype=np.int64)
    out['meta.stress'] = np.float32(stress)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(f'{name}: loss {out["loss"]:.6f} logits absmax {np.abs(out["logits"]).max():.3f}')


def make_train_trace(ref):
    """3 SGD steps as reference utils/processor.py:696-703 does them: zero_grad, backward,
    clip_grad_norm_(1.0), SGD(momentum .9, nesterov, wd 1e-4) step; lr 0.1 (train_joint.yaml:29-39)."""
    n, v, num_class, seed = 2, 25, 60, 301
    model = ref.Model(num_class=num_class, num_point=v, num_person=2, graph='graph.ntu_rgb_d.Graph',
                      graph_args=dict(labeling_mode='spatial'))
    sd = orc.randomized_state(orc.model_param_shapes(num_class, v), seed, stress=3.0)
    model.load_state_dict(sd)
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4)
    losses, gnorms = [], []
    for step in range(3):
        xn, lab = model_inputs(n, v, num_class, seed + step, 300)
        logits = model(torch.from_numpy(xn))
        loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(lab))
        opt.zero_grad()
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        losses.append(loss.item())
        gnorms.append(float(gn))
    out = {'losses': np.array(losses, dtype=np.float64), 'grad_norms': np.array(gnorms, dtype=np.float64)}
    for k in ('fc.weight', 'l1.gcn1.PA', 'l5.tcn1.conv.weight', 'l10.gcn1.conv_d.2.weight', 'l1.gcn1.bn.running_var'):
        t = model.state_dict()[k].numpy().astype(np.float64)
        out['final.' + k + '.sum'] = np.float64(t.sum())
        out['final.' + k + '.norm'] = np.float64(np.linalg.norm(t))
        idx = sample_idx(t.size, 32)
        out['final.' + k + '.idx'] = idx
        out['final.' + k + '.samples'] = t.reshape(-1)[idx].astype(np.float32)
    out['meta'] = np.array([n, v, num_class, seed], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'train_trace_ntu_b2.npz'), **out)
    print('train trace losses', losses, 'gnorms', gnorms)


def make_graphs():
    out = {}
    for v in (25, 18):
        out[f'A_v{v}'] = np.asarray(ref_graph(v), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, 'graphs.npz'), **out)


if __name__ == '__main__' and len(sys.argv) == 1:
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = load_reference()
    make_graphs()
    for case in UNIT_CASES:
        make_unit(ref, *case)
    for case in MODEL_CASES:
        make_model(ref, *case)
    make_train_trace(ref)


if __name__ == '__main__' and len(sys.argv) > 2 and sys.argv[1] == 'only':     # regenerate the named fixtures only
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = load_reference()
    for case in UNIT_CASES:
        if case[0] in sys.argv[2:]:
            make_unit(ref, *case)
    for case in MODEL_CASES:
        if case[0] in sys.argv[2:]:
            make_model(ref, *case)


# ------------------------------------------------------------------------------------------------
# AAGCN (reference model/architecture/aagcn/aagcn.py).  Run: python tests/golden/make_golden.py aagcn
# ------------------------------------------------------------------------------------------------
def load_reference_aagcn():
    """aagcn.py imports torchinfo (absent) and model.layers.module.ghostbatchnorm: register an empty `torchinfo`
    stub and load ghostbatchnorm.py by path under its dotted name."""
    import types
    sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k == 'graph' or k.startswith('graph.') or k == 'model' or k.startswith('model.')]:
        del sys.modules[k]
    ti = types.ModuleType('torchinfo')
    ti.summary = lambda *a, **k: None
    sys.modules['torchinfo'] = ti
    for name in ('model', 'model.layers', 'model.layers.module'):
        sys.modules[name] = types.ModuleType(name)
    spec = importlib.util.spec_from_file_location(
        'model.layers.module.ghostbatchnorm', os.path.join(REF, 'model/layers/module/ghostbatchnorm.py'))
    gb = importlib.util.module_from_spec(spec)
    sys.modules['model.layers.module.ghostbatchnorm'] = gb
    spec.loader.exec_module(gb)
    spec = importlib.util.spec_from_file_location('ref_aagcn', os.path.join(REF, 'model/architecture/aagcn/aagcn.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


AAGCN_UNIT_CASES = [
    # name, cin, cout, stride, residual, T, V, seed, stress, adaptive, attention
    ('au_64_64_s1_v25', 64, 64, 1, True, 16, 25, 401, 4.0, True, True),
    ('au_64_128_s2_v25', 64, 128, 2, True, 16, 25, 402, 3.0, True, True),
    ('au_3_64_s1_v18', 3, 64, 1, False, 16, 18, 403, 1.0, True, True),
    ('au_64_64_s1_v25_plain', 64, 64, 1, True, 16, 25, 404, 1.0, False, False),
    # GhostBatchNorm (gbn_split = 2, batch of 4 person-samples: two per virtual sub-batch); down + residual BNs included
    ('au_64_128_s2_v25_gbn2', 64, 128, 2, True, 16, 25, 405, 3.0, True, True, 2, 4),
    ('au_64_64_s1_v25_gbn2', 64, 64, 1, True, 16, 25, 406, 3.0, True, True, 2, 4),
]


def make_aagcn_unit(ref, name, cin, cout, stride, residual, t, v, seed, stress, adaptive, attention, gbn=None, n=2):
    A = ref_graph(v)
    fn = ref.AdaptiveGCN if adaptive else ref.NonAdaptiveGCN
    kw = dict(stride=stride, residual=residual, adaptive=fn, attention=attention, gbn_split=gbn)
    unit = ref.TCNGCNUnit(cin, cout, A, **kw)
    shapes = orc.aagcn_unit_param_shapes('', cin, cout, v, stride, residual, adaptive, attention, gbn)
    assert set(shapes) == set(unit.state_dict().keys()), set(shapes) ^ set(unit.state_dict().keys())
    sd = orc.aagcn_randomized_state(shapes, seed, stress=stress)
    unit.load_state_dict(sd)
    xn, rn = unit_inputs(cin, cout, stride, t, v, seed, n=n)
    out = {}
    unit.eval()
    with torch.no_grad():
        out['y_eval'] = unit(torch.from_numpy(xn)).numpy()
    unit.train()
    x = torch.from_numpy(xn).requires_grad_(True)
    y = unit(x)
    (y * torch.from_numpy(rn)).sum().backward()
    out['y'] = y.detach().numpy()
    out['dx'] = x.grad.numpy()
    pack_grads(out, unit.named_parameters())
    unit64 = ref.TCNGCNUnit(cin, cout, A, **kw).double()
    unit64.load_state_dict({k: v_.double() if v_.is_floating_point() else v_ for k, v_ in sd.items()})
    for m_ in unit64.modules():
        if isinstance(m_, ref.NonAdaptiveGCN):
            m_.A = m_.A.double()
    unit64.train()
    x64 = torch.from_numpy(xn).double().requires_grad_(True)
    (unit64(x64) * torch.from_numpy(rn).double()).sum().backward()
    out['dx64'] = x64.grad.numpy().astype(np.float32)
    pack_grads(out, unit64.named_parameters(), prefix='g64.')
    add_unit_sens(out, unit64, xn, rn, seed)
    for k, b in unit.state_dict().items():
        if k.endswith(('running_mean', 'running_var')):
            out['buf.' + k] = b.numpy().copy()
    out['meta'] = np.array([cin, cout, stride, int(residual), t, v, seed, int(adaptive), int(attention)], dtype=np.int64)
    out['meta.stress'] = np.float32(stress)
    out['meta.gbn'] = np.int64(gbn or 0)
    out['meta.n'] = np.int64(n)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(f'{name}: y {out["y"].shape} |y|max {np.abs(out["y"]).max():.3f}')


# backbones of reference aagcn.py:407-474 reachable through Model(model_layers=...) (101-103 end in 64 channels, which
# Model's 256-wide fc cannot take: they only serve the aagcn_vNN research heads), and a GhostBatchNorm model
AAGCN_MODEL_CASES = [
    # name, batch, V, classes, seed, stress, T, model_layers, gbn_split, grad samples per tensor
    ('am_ntu_b1_t64', 1, 25, 60, 501, 3.0, 64, 10, None, 256),
    ('am_ntu_l3_t32', 1, 25, 60, 502, 3.0, 32, 3, None, 32),
    ('am_ntu_l6_t32', 1, 25, 60, 503, 3.0, 32, 6, None, 32),
    ('am_ntu_l7_t32', 1, 25, 60, 504, 3.0, 32, 7, None, 32),
    ('am_ntu_l3_gbn2_t32', 2, 25, 60, 505, 3.0, 32, 3, 2, 32),
]


def make_aagcn_model(ref, name='am_ntu_b1_t64', n=1, v=25, num_class=60, seed=501, stress=3.0, t=64, model_layers=10,
                     gbn=None, nsamp=256):
    global sample_idx
    _orig_idx = sample_idx
    sample_idx = lambda numel, k=64: _orig_idx(numel, min(k, nsamp))     # noqa: E731  (smaller fixtures for variants)
    mk = dict(num_class=num_class, num_point=v, num_person=2, graph='graph.ntu_rgb_d.Graph',
              graph_args=dict(labeling_mode='spatial'), model_layers=model_layers, gbn_split=gbn)
    model = ref.Model(**mk)
    shapes = orc.aagcn_model_param_shapes(num_class, v, model_layers=model_layers, gbn_split=gbn)
    assert set(shapes) == set(model.state_dict().keys()), set(shapes) ^ set(model.state_dict().keys())
    sd = orc.aagcn_randomized_state(shapes, seed, stress=stress)
    model.load_state_dict(sd)
    xn, lab = model_inputs(n, v, num_class, seed, t)
    out = {}
    model.eval()
    with torch.no_grad():
        le, aux = model(torch.from_numpy(xn))
        assert aux is None
        out['logits_eval'] = le.numpy()
    model.train()
    logits, _ = model(torch.from_numpy(xn))
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(lab))
    loss.backward()
    out['logits'] = logits.detach().numpy()
    out['loss'] = np.float64(loss.item())
    pack_grads(out, model.named_parameters(), full_limit=2000)
    model64 = ref.Model(**mk).double()
    model64.load_state_dict({k: v_.double() if v_.is_floating_point() else v_ for k, v_ in sd.items()})
    model64.train()
    l64, _ = model64(torch.from_numpy(xn).double())
    torch.nn.functional.cross_entropy(l64, torch.from_numpy(lab)).backward()
    pack_grads(out, model64.named_parameters(), prefix='g64.', full_limit=2000)
    g64 = {k: p.grad.detach().clone() for k, p in model64.named_parameters()}
    sens = {k: 0.0 for k in g64}
    prng = torch.Generator().manual_seed(seed)
    for _ in range(MODEL_SENS_SAMPLES):
        model64.zero_grad()
        xp = torch.from_numpy(xn).double()
        xp = xp * (1.0 + MODEL_SENS_EPS * torch.randn(xp.shape, generator=prng, dtype=torch.float64))
        torch.nn.functional.cross_entropy(model64(xp)[0], torch.from_numpy(lab)).backward()
        for k, p in model64.named_parameters():
            sens[k] = max(sens[k], float((p.grad - g64[k]).abs().max() / max(1e-300, float(g64[k].abs().max()))))
    for k, v_ in sens.items():
        out['sens.' + k] = np.float32(v_)
    out['meta'] = np.array([n, v, num_class, seed, t], dtype=np.int64)
    out['meta.stress'] = np.float32(stress)
    out['meta.layers'] = np.int64(model_layers)
    out['meta.gbn'] = np.int64(gbn or 0)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    sample_idx = _orig_idx
    print(f'{name}: loss {out["loss"]:.6f}')


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'aagcn':
    torch.manual_seed(0)
    ref_a = load_reference_aagcn()
    only = sys.argv[2:]
    for case in AAGCN_UNIT_CASES:
        if not only or case[0] in only:
            make_aagcn_unit(ref_a, *case)
    for case in AAGCN_MODEL_CASES:
        if not only or case[0] in only:
            make_aagcn_model(ref_a, *case)
