"""Input side: ``agcn_amd.feeders`` against clips the REFERENCE augmentation functions produced with fixed
seeds (tests/golden/feeder_tools.npz, made by tests/golden/make_feeder_golden.py), the Feeder dataset on the reference's
on-disk format, and the batched device transforms / pinned loader against the per-clip host versions."""
import os
import pickle
import random

import numpy as np
import pytest
import torch

import agcn_amd  # noqa: F401
from agcn_amd.feeders import DeviceAugment, DeviceLoader, Feeder, tools

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'feeder_tools.npz')


def clip(seed, T=24, valid=(3, 17), V=5, M=2):          # same recipe as make_feeder_golden.py
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((3, T, V, M))
    x[:, :valid[0]] = 0
    x[:, valid[1]:] = 0
    return x


CASES = [('auto_pading_front', 'auto_pading', dict(size=32)), ('auto_pading_random', 'auto_pading', dict(size=32, random_pad=True)),
         ('random_choose_crop', 'random_choose', dict(size=16)), ('random_choose_pad', 'random_choose', dict(size=40)),
         ('random_shift', 'random_shift', dict()), ('random_move', 'random_move', dict()),
         ('random_xaxis_scale', 'random_xaxis_scale', dict()), ('random_yaxis_scale', 'random_yaxis_scale', dict()),
         ('random_zaxis_flip_a', 'random_zaxis_flip', dict()), ('random_zaxis_flip_b', 'random_zaxis_flip', dict()),
         ('random_subsample', 'random_subsample', dict(freq=6)), ('stretch', 'stretch_to_maximum_length', dict()),
         ('downsample', 'downsample', dict(step=3))]


@pytest.mark.parametrize('name,fn,kw', CASES)
def test_tools_reproduce_the_reference_clips(name, fn, kw):
    gold = np.load(GOLD)
    iseed, rseed = [int(i) for i in gold[name + '.seeds']]
    x = clip(iseed, valid=(0, 17) if fn == 'stretch_to_maximum_length' else (3, 17))
    random.seed(rseed)
    np.random.seed(rseed)
    y = getattr(tools, fn)(x.copy(), **kw)
    ref = gold[name + '.y']
    assert y.shape == ref.shape
    assert np.abs(np.asarray(y, dtype=np.float64) - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


def test_random_rotation_is_a_rotation():
    """parity unpinned (the reference's tools.random_rotation raises on numpy clips, see make_feeder_golden.py):
    property test of the documented maths -- R orthonormal with det +1, joint norms preserved, angles within theta."""
    np.random.seed(3)
    x = clip(5)
    y = tools.random_rotation(x.copy(), theta=0.5)
    assert np.allclose(np.linalg.norm(y, axis=0), np.linalg.norm(x, axis=0), atol=1e-12)
    R = tools.rotation_matrix(np.array([0.1, -0.2, 0.3]))
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
    assert np.allclose(tools.rotation_matrix(np.array([0.3, 0, 0])), [[1, 0, 0], [0, np.cos(.3), np.sin(.3)],
                                                                       [0, -np.sin(.3), np.cos(.3)]])


def _dataset(tmp_path, n=12, T=20, V=5, M=2):
    rng = np.random.default_rng(0)
    data = rng.standard_normal((n, 3, T, V, M)).astype(np.float32)
    data[:, :, 14:] = 0
    labels = [int(i) for i in rng.integers(0, 4, n)]
    names = [f's{i:03d}.skeleton' for i in range(n)]
    np.save(tmp_path / 'data.npy', data)
    with open(tmp_path / 'label.pkl', 'wb') as f:
        pickle.dump((names, labels), f)
    return data, labels, names


def test_feeder_reads_the_reference_format(tmp_path):
    data, labels, names = _dataset(tmp_path)
    f = Feeder(str(tmp_path / 'data.npy'), str(tmp_path / 'label.pkl'), window_size=24)
    assert len(f) == 12 and list(f.sample_name) == names
    x, lab, idx = f[3]
    assert x.shape == (3, 24, 5, 2) and lab == labels[3] and idx == 3
    assert np.array_equal(x[:, :20], data[3]) and not x[:, 20:].any()          # auto_pading at the front
    score = np.zeros((12, 4))
    score[np.arange(12), labels] = 1.0
    assert f.top_k(score, 1) == 1.0
    random.seed(0)
    np.random.seed(0)
    g = Feeder(str(tmp_path / 'data.npy'), str(tmp_path / 'label.pkl'), window_size=16, random_choose=True,
               random_shift=True, random_move=True, normalization=True, random_zaxis_flip=True)
    y, _, _ = g[5]
    assert y.shape == (3, 16, 5, 2) and np.isfinite(y).all()


def test_device_augment_matches_host_transforms_on_cpu_tensors():
    """The batched transforms with PINNED parameters equal the per-clip numpy versions (torch CPU tensors here; the
    same code runs on the GPU in test_gpu_feeders)."""
    B, T = 3, 24
    xs = np.stack([clip(20 + b) for b in range(B)])
    x = torch.from_numpy(xs)
    # random_shift: same draws as tools.random_shift per clip
    random.seed(7)
    ref = np.stack([tools.random_shift(xs[b].copy()) for b in range(B)])
    random.seed(7)
    got = DeviceAugment(random_shift=True)(x)
    assert np.allclose(got.numpy(), ref, atol=1e-12)
    # random_choose (crop) and padding at a random offset
    for W in (16, 40):
        random.seed(8)
        ref = np.stack([tools.random_choose(xs[b].copy(), W) for b in range(B)])
        random.seed(8)
        got = DeviceAugment(window_size=W, random_choose=True)(x)
        assert np.allclose(got.numpy(), ref, atol=1e-12)
    assert np.allclose(DeviceAugment(window_size=32)(x).numpy(), np.stack([tools.auto_pading(xs[b], 32) for b in range(B)]))
    # random_move, axis scale, flip, rotation: parameters drawn identically
    random.seed(9)
    np.random.seed(9)
    ref = np.stack([tools.random_move(xs[b].copy()) for b in range(B)])
    random.seed(9)
    np.random.seed(9)
    got = DeviceAugment(random_move=True)(x)
    assert np.allclose(got.numpy(), ref, atol=1e-10)
    np.random.seed(10)
    S = np.random.choice(tools.SCALE_CANDIDATES, B)
    ref = xs.copy()
    for b in range(B):
        d = ref[b, 0, :, :, 1] - ref[b, 0, :, :, 0]
        ref[b, 0, :, :, 1] = ref[b, 0, :, :, 0] + d * S[b]
    np.random.seed(10)
    assert np.allclose(DeviceAugment(random_xaxis_scale=True)(x).numpy(), ref, atol=1e-12)
    np.random.seed(11)
    Rs = [tools.rotation_matrix(np.random.uniform(-0.3, 0.3, (1, 3))[0]) for _ in range(B)]
    np.random.seed(11)
    got = DeviceAugment(random_rotation=True, rotation_theta=0.3)(x)
    assert np.allclose(got.numpy(), np.stack([np.einsum('ij,jtvm->itvm', Rs[b], xs[b]) for b in range(B)]), atol=1e-12)


def test_device_loader_yields_every_batch_in_order_on_cpu():
    batches = [(torch.full((2, 3), float(i)), torch.tensor([i, i + 1]), torch.tensor([i])) for i in range(5)]
    got = list(DeviceLoader(batches, 'cpu', depth=2))
    assert len(got) == 5
    for i, (d, lab, idx) in enumerate(got):
        assert torch.equal(d, batches[i][0]) and torch.equal(lab, batches[i][1]) and int(idx) == i


@pytest.mark.gpu
def test_gpu_feeders_device_loader_and_augment(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    dev = torch.device('cuda:0')
    data, labels, names = _dataset(tmp_path, n=16, T=40)
    f = Feeder(str(tmp_path / 'data.npy'), str(tmp_path / 'label.pkl'), window_size=32, random_choose=True,
               random_shift=True, random_move=True, device_augment=True)
    loader = torch.utils.data.DataLoader(f, batch_size=4, shuffle=False)
    aug = DeviceAugment.from_feeder(f)
    seen = 0
    for x, lab, idx in DeviceLoader(loader, dev, depth=2):
        assert x.is_cuda and x.shape == (4, 3, 40, 5, 2)
        assert torch.equal(x.cpu(), torch.from_numpy(data[idx.cpu().numpy()]))       # H2D ring delivers the right clips
        random.seed(int(idx[0]))
        np.random.seed(int(idx[0]))
        y = aug(x)
        random.seed(int(idx[0]))
        np.random.seed(int(idx[0]))
        y_cpu = aug(x.cpu().double())                 # same code path on the host: identical draws
        assert y.shape == (4, 3, 32, 5, 2)
        assert torch.allclose(y.cpu().double(), y_cpu, atol=1e-5)
        seen += 4
    assert seen == 16


def test_eval_input_equals_host_feeder_output_with_normalization_and_window(tmp_path):
    """With device_augment the feeder returns RAW clips; what Processor.train AND Processor.eval feed the model
    (processor.prepare_batch) must equal what the host feeder's __getitem__ returns -- the deterministic transforms
    (normalization, window_size padding) apply in both phases (reference feeder.py:182-221)."""
    from agcn_amd.processor import prepare_batch, use_device_augment
    _dataset(tmp_path)
    kw = dict(data_path=str(tmp_path / 'data.npy'), label_path=str(tmp_path / 'label.pkl'), window_size=24,
              normalization=True)
    host = Feeder(**kw)
    assert use_device_augment(Feeder, kw, True)
    dev = Feeder(device_augment=True, **kw)
    aug = DeviceAugment.from_feeder(dev)
    idx = [0, 3, 7]
    raw = torch.from_numpy(np.stack([dev[i][0] for i in idx]))
    want = np.stack([host[i][0] for i in idx])
    assert raw.shape[2] == 20 and want.shape[2] == 24                  # raw clips are neither padded nor normalised
    got = prepare_batch(raw, aug).numpy()
    assert got.shape == want.shape and np.allclose(got, want, atol=1e-5)
    assert prepare_batch(raw, None) is not None and prepare_batch(raw.double(), None).dtype == torch.float32


def test_transforms_missing_on_the_device_keep_the_host_path():
    """stretch / random_subsample exist only as host transforms: such a config must not lose them silently."""
    from agcn_amd.processor import use_device_augment
    assert use_device_augment(Feeder, dict(random_move=True, window_size=150, random_choose=True), True)
    assert not use_device_augment(Feeder, dict(random_subsample=60), True)
    assert not use_device_augment(Feeder, dict(stretch=True), True)
    assert use_device_augment(Feeder, dict(stretch=False, random_subsample=None), True)
    assert not use_device_augment(Feeder, dict(), False)
    assert not use_device_augment(object, dict(), True)               # feeders without the device protocol
