"""Shared helpers for the golden-fixture tests (oracle pinning on CPU, HIP parity on GPU)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

UNIT_NAMES = ['u_3_64_s1_v25', 'u_64_64_s1_v25', 'u_64_64_s1_v25_stress', 'u_64_128_s2_v25',
              'u_128_256_s2_v25', 'u_64_64_s1_v18', 'u_64_128_s2_v18_oddT']
MODEL_NAMES = ['m_ntu_b1', 'm_ntu_b2', 'm_kin_b2_t64']


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
                  'residual.conv.bias')


def is_zero_grad_bias(name):
    """Conv biases feeding a train-mode BN have mathematically zero gradient."""
    return name.endswith(ZERO_GRAD_BIAS)
