"""Shared helpers for the golden-fixture tests (oracle pinning on CPU, HIP parity on GPU)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

UNIT_NAMES = ['u_3_64_s1_v25', 'u_64_64_s1_v25', 'u_64_64_s1_v25_stress', 'u_64_128_s2_v25',
              'u_128_256_s2_v25', 'u_64_64_s1_v18', 'u_64_128_s2_v18_oddT']
MODEL_NAMES = ['m_ntu_b1', 'm_ntu_b2', 'm_kin_b2_t64', 'm_kin_b2']


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz')))


def unit_inputs(cin, cout, stride, t, v, seed, n=2):
    """Same recipe as tests/golden/make_golden.py::unit_inputs."""
    rng = np.random.default_rng(seed + 5000)
    x = rng.standard_normal((n, cin, t, v)).astype(np.float32)
    tout = (t + 2 * 4 - 9) // stride + 1
    r = rng.standard_normal((n, cout, tout, v)).astype(np.float32)
    return x, r


def model_inputs(n, v, num_class, seed, t=300):
    rng = np.random.default_rng(seed + 7000)
    x = rng.standard_normal((n, 3, t, v, 2)).astype(np.float32)
    lab = rng.integers(0, num_class, size=(n,)).astype(np.int64)
    return x, lab


def graph_A(v):
    g = np.load(os.path.join(GOLDEN, 'graphs.npz'))
    return torch.from_numpy(g[f'A_v{v}'].astype(np.float32))


def rel_err(a, ref):
    """SURVEY 8c parity metric: max|a-ref| / max(1, max|ref|)."""
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(a - ref).max() / max(1.0, np.abs(ref).max()))


def grad_err(g, gold, name):
    """Gradient parity: normalised by the per-tensor max|g_ref| (floor 1e-6: structurally-zero
    gradients, SURVEY F9, are compared absolutely)."""
    g = np.asarray(g, dtype=np.float64)
    scale = max(float(gold['g.' + name + '.absmax']), 1e-6)
    if ('g.' + name) in gold:
        return float(np.abs(g - gold['g.' + name]).max() / scale) if scale > 1e-6 else float(np.abs(g).max())
    idx = gold['g.' + name + '.idx']
    return float(np.abs(g.reshape(-1)[idx] - gold['g.' + name + '.samples']).max() / scale)


ZERO_GRAD_BIAS = ('conv_d.0.bias', 'conv_d.1.bias', 'conv_d.2.bias', 'down.0.bias', 'tcn1.conv.bias',
                  'residual.conv.bias', 'conv_a.0.bias', 'conv_a.1.bias', 'conv_a.2.bias')


def is_zero_grad_bias(name):
    """Structurally-zero gradients, compared with an ABSOLUTE tolerance only:
    * conv biases feeding a train-mode BN;
    * conv_a biases: they add a term to the logits S[u,v] that is constant in u, and the softmax over u
      (agcn.py:80,101) is invariant to it.  The reference's fp32 values there are rounding noise (~1e-7)."""
    return name.endswith(ZERO_GRAD_BIAS)


def _pick(gold, prefix, name, g):
    if (prefix + name) in gold:
        return np.asarray(g, dtype=np.float64), gold[prefix + name].astype(np.float64)
    idx = gold['g.' + name + '.idx']
    return np.asarray(g, dtype=np.float64).reshape(-1)[idx], gold[prefix + name + '.samples'].astype(np.float64)


def sens_floor(gold):
    vals = [float(v) for k, v in gold.items() if k.startswith('sens.') and not is_zero_grad_bias(k)]
    return float(np.median(vals)) if vals else 0.0


BAND_CAP = 5e-2


def grad_check(g, gold, name, tol):
    """Noise-aware gradient parity.  err32 = distance to the reference run in fp32 (the parity target);
    the fixtures also hold the SAME reference code run in fp64, which gives the reference's own fp32 rounding
    noise per tensor (up to ~1e-2 of max|g| for the adjacency parameters of the full model, whose gradients are
    sums of ~1e5 cancelling terms).  Pass if err32 <= tol, or if the result is as close to the fp64 answer as
    the fp32 reference itself is (within 3x): err64 <= tol + 3*noise (8x for the perturbation band of the model
    fixtures, whose 8 draws sample the tail of the kink-flip distribution only coarsely: a 3e-6 input perturbation
    of a correct fp32 run was measured to move individual tensors to 10x a 3-draw band).
    Returns (ok, err32, err64, noise)."""
    scale = max(float(gold['g.' + name + '.absmax']), 1e-12)
    a, r32 = _pick(gold, 'g.', name, g)
    err32 = float(np.abs(a - r32).max() / scale)
    if ('g64.' + name + '.absmax') not in gold:
        return err32 <= tol, err32, None, None
    _, r64 = _pick(gold, 'g64.', name, g)
    noise = float(np.abs(r32 - r64).max() / scale)
    # 'sens.*' (model fixtures): how far the REFERENCE's own fp64 gradient moves when its input is perturbed by
    # 1e-6 relative -- ReLU masks of elements sitting on the kink flip (see make_golden.py).  It bounds what any
    # two correct fp32 evaluations can be expected to agree to.
    factor = 3.0
    if ('sens.' + name) in gold:
        factor = 8.0
        # three perturbation samples under-sample which kink elements flip, so the band of a tensor is at
        # least the model-wide median band (a flip in layer L moves the gradients of every layer below it)
        noise = max(noise, float(gold['sens.' + name]), sens_floor(gold))
    err64 = float(np.abs(a - r64).max() / scale)
    # the band is CAPPED: whatever the perturbation draws say, a tensor further than BAND_CAP of max|g| from the fp64
    # reference fails (round 2's audit showed bands of 0.1-0.2 on single tensors, i.e. a criterion that could not fail;
    # the largest deviation any correct run has shown on these fixtures is 2.2e-2).  The kink-free gradient evidence is
    # the pinned-ReLU end-to-end tests, which assert the primary criterion.
    band = min(factor * noise, BAND_CAP)
    return (err32 <= tol) or (err64 <= tol + band), err32, err64, noise

# ---- parity audit: every golden gradient comparison leaves a record; tests/conftest.py prints one summary line per
# fixture at the end of the run (so the test log shows HOW each tensor passed, not just that it did) ----
AUDIT = {}


def audit_grads(fixture, named_grads, gold, tol, scalar_tol=None):
    """Check every (name, gradient) against the fixture with ``grad_check`` and record, per fixture: tensors compared,
    how many passed the PRIMARY criterion (err32 <= tol, distance to the reference's fp32 run), how many only through
    the fp64/perturbation band, the worst err32 / err64 / band and their tensor names.
    Structurally-zero bias gradients are checked absolutely (<= 1e-5) and counted separately.
    Returns (bad, summary): bad = [(name, err32, err64, band)] of failing tensors."""
    rec = AUDIT.setdefault(fixture, {'tensors': 0, 'primary': 0, 'band': 0, 'failed': 0, 'zero_bias': 0,
                                     'worst_err32': (0.0, ''), 'worst_err64': (0.0, ''), 'max_band': (0.0, ''),
                                     'tol': tol})
    bad = []
    for name, g in named_grads:
        g = np.asarray(g)
        if is_zero_grad_bias(name):
            rec['zero_bias'] += 1
            if float(np.abs(g).max()) >= 1e-5:
                bad.append((name, float(np.abs(g).max()), None, None))
                rec['failed'] += 1
            continue
        t = scalar_tol if (scalar_tol is not None and g.size == 1) else tol
        ok, e32, e64, noise = grad_check(g, gold, name, t)
        rec['tensors'] += 1
        rec.setdefault('err32_all', []).append(e32)
        # the reference's OWN fp32-vs-fp64 distance for this tensor (no perturbation band mixed in)
        n32 = ref_noise32(gold, name)
        if n32 is not None:
            rec.setdefault('ref_noise32_all', []).append(n32)
            rec['ref32_within_tol'] = rec.get('ref32_within_tol', 0) + int(n32 <= t)
        if e32 <= t:
            rec['primary'] += 1
        elif e64 is not None and n32 is not None and e64 <= t + 3.0 * n32:
            rec['as_ref'] = rec.get('as_ref', 0) + 1      # as close to fp64 as the reference's fp32 run (x3)
        elif ok:
            rec['band'] += 1
        else:
            rec['failed'] += 1
            bad.append((name, e32, e64, noise))
        if e32 > rec['worst_err32'][0]:
            rec['worst_err32'] = (e32, name)
        if e64 is not None and e64 > rec['worst_err64'][0]:
            rec['worst_err64'] = (e64, name)
        if noise is not None and noise > rec['max_band'][0]:
            rec['max_band'] = (noise, name)
    return bad, rec


def ref_noise32(gold, name):
    """|reference fp32 gradient - reference fp64 gradient| / max|g|: what the reference itself achieves in fp32."""
    if ('g64.' + name + '.absmax') not in gold:
        return None
    scale = max(float(gold['g.' + name + '.absmax']), 1e-12)
    key = 'g.' + name if ('g.' + name) in gold else 'g.' + name + '.samples'
    key64 = 'g64.' + name if ('g64.' + name) in gold else 'g64.' + name + '.samples'
    return float(np.abs(gold[key].astype(np.float64) - gold[key64].astype(np.float64)).max() / scale)


def audit_value(fixture, what, err, tol):
    """Record a scalar comparison (outputs, logits, dx, layer-wise errors) next to the gradient audit."""
    rec = AUDIT.setdefault(fixture, {'tensors': 0, 'primary': 0, 'band': 0, 'failed': 0, 'zero_bias': 0,
                                     'worst_err32': (0.0, ''), 'worst_err64': (0.0, ''), 'max_band': (0.0, ''),
                                     'tol': tol})
    rec.setdefault('values', {})[what] = (float(err), float(tol))
    return err <= tol


def audit_lines():
    out = []
    for fx, r in AUDIT.items():
        n = max(1, r['tensors'])
        if r['tensors'] == 0:
            out.append(f"parity-audit {fx}: " + ', '.join(f"{k} {e:.2e}(tol {t:g})" for k, (e, t) in
                                                             r.get('values', {}).items()))
            continue
        line = (f"parity-audit {fx}: grads {r['tensors']} tensors, primary(err32<={r['tol']:g}) {r['primary']} "
                f"({100.0 * r['primary'] / n:.1f}%), as close to fp64 as 3x the reference's own fp32 run "
                f"{r.get('as_ref', 0)}, via ReLU-kink perturbation band {r['band']}, failed {r['failed']}, "
                f"zero-bias(abs) {r['zero_bias']}; worst err32 {r['worst_err32'][0]:.2e} [{r['worst_err32'][1]}], "
                f"worst err64 {r['worst_err64'][0]:.2e} [{r['worst_err64'][1]}], "
                f"max band {r['max_band'][0]:.2e} [{r['max_band'][1]}]")
        if r.get('err32_all'):
            line += f"; median err32 {float(np.median(r['err32_all'])):.2e}"
        if r.get('ref_noise32_all'):
            line += (f"; the REFERENCE's own fp32-vs-fp64 distance: median {float(np.median(r['ref_noise32_all'])):.2e}, "
                     f"within {r['tol']:g} for {r.get('ref32_within_tol', 0)}/{len(r['ref_noise32_all'])} tensors")
        if r.get('values'):
            line += '; ' + ', '.join(f"{k} {e:.2e}(tol {t:g})" for k, (e, t) in r['values'].items())
        out.append(line)
    return out


AAGCN_UNIT_NAMES = ['au_64_64_s1_v25', 'au_64_128_s2_v25', 'au_3_64_s1_v18', 'au_64_64_s1_v25_plain',
                    'au_64_128_s2_v25_gbn2', 'au_64_64_s1_v25_gbn2']          # *_gbn2: GhostBatchNorm, 2 splits
AAGCN_MODEL_NAMES = ['am_ntu_b1_t64', 'am_ntu_l3_t32', 'am_ntu_l6_t32', 'am_ntu_l7_t32', 'am_ntu_l3_gbn2_t32']


def meta_int(gold, key, default=0):
    return int(gold[key]) if key in gold else default


def is_alias_key(name):
    """``gcn1.agcn.conv_d.*`` aliases ``gcn1.conv_d.*`` in the reference AAGCN state_dict (aagcn.py:228-233);
    ``named_parameters()`` reports the tensor under the first name only."""
    return 'agcn.conv_d.' in name


def canonical_key(name):
    return name.replace('agcn.conv_d.', 'conv_d.')
