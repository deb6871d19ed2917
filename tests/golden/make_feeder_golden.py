"""Golden vectors for the input side: clips produced by the REFERENCE augmentation functions (feeders/tools.py) with
fixed seeds.  Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_feeder_golden.py

Only data is stored (input clip, seeds, outputs); tests/test_feeders.py re-runs ``agcn_amd.feeders.tools`` with the same
seeds and compares."""
import importlib.util
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
spec = importlib.util.spec_from_file_location('ref_tools', '/root/reference/feeders/tools.py')
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def clip(seed, T=24, valid=(3, 17), V=5, M=2):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((3, T, V, M))
    x[:, :valid[0]] = 0
    x[:, valid[1]:] = 0
    return x


CASES = [  # name, function, kwargs, input seed, rng seed
    ('auto_pading_front', 'auto_pading', dict(size=32), 1, 11),
    ('auto_pading_random', 'auto_pading', dict(size=32, random_pad=True), 1, 12),
    ('random_choose_crop', 'random_choose', dict(size=16), 2, 13),
    ('random_choose_pad', 'random_choose', dict(size=40), 2, 14),
    ('random_shift', 'random_shift', dict(), 3, 15),
    ('random_move', 'random_move', dict(), 4, 16),
    ('random_xaxis_scale', 'random_xaxis_scale', dict(), 5, 17),
    ('random_yaxis_scale', 'random_yaxis_scale', dict(), 5, 18),
    ('random_zaxis_flip_a', 'random_zaxis_flip', dict(), 6, 19),
    ('random_zaxis_flip_b', 'random_zaxis_flip', dict(), 6, 20),
    ('random_subsample', 'random_subsample', dict(freq=6), 7, 21),
    # random_rotation is NOT pinned: the reference's numpy helper `_rot` (tools.py:164-185) is shadowed by a second,
    # torch-only `_rot` defined later in the same file (:278-302), so tools.random_rotation raises AttributeError on
    # the feeder's numpy clips.  agcn_amd.feeders.tools follows the maths of the numpy `_rot`; its test is a property
    # test (orthonormal matrix, norms preserved): parity unpinned for this one transform.
    ('stretch', 'stretch_to_maximum_length', dict(), 9, 23),
    ('downsample', 'downsample', dict(step=3), 10, 24),
]

out = {}
for name, fn, kw, iseed, rseed in CASES:
    x = clip(iseed, valid=(0, 17) if fn == 'stretch_to_maximum_length' else (3, 17))
    random.seed(rseed)
    np.random.seed(rseed)
    y = getattr(ref, fn)(x.copy(), **kw)
    out[name + '.y'] = np.asarray(y, dtype=np.float64)
    out[name + '.seeds'] = np.array([iseed, rseed])
    print(name, np.asarray(y).shape)
np.savez_compressed(os.path.join(HERE, 'feeder_tools.npz'), **out)
