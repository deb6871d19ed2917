"""Host logic of the Processor counterpart that needs no GPU: config precedence, feeders, LR rule."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_precedence_defaults_yaml_cli():
    import agcn_amd  # noqa: F401
    from agcn_amd.processor import load_args
    cfg = os.path.join(ROOT, 'config', 'nturgbd-cross-view', 'train_joint.yaml')
    a = load_args(['--config', cfg])
    assert a.model == 'model.agcn.Model' and a.batch_size == 64 and a.step == [30, 40] and a.nesterov is True
    assert a.model_args['graph'] == 'graph.ntu_rgb_d.Graph' and a.num_epoch == 50
    b = load_args(['--config', cfg, '--batch-size', '8', '--base-lr', '0.05'])
    assert b.batch_size == 8 and b.base_lr == 0.05 and b.weight_decay == 0.0001
    k = load_args(['--config', os.path.join(ROOT, 'config', 'kinetics-skeleton', 'train_joint.yaml')])
    assert k.model_args['num_class'] == 400 and k.model_args['num_point'] == 18 and k.batch_size == 128


def test_synthetic_feeder_shapes():
    import agcn_amd  # noqa: F401
    from agcn_amd.processor import SyntheticFeeder
    f = SyntheticFeeder(num_samples=5, num_point=18, num_class=400, window_size=32)
    x, y, i = f[3]
    assert x.shape == (3, 32, 18, 2) and x.dtype == np.float32 and 0 <= y < 400 and i == 3 and len(f) == 5


def test_npy_feeder_roundtrip(tmp_path):
    import agcn_amd  # noqa: F401
    from agcn_amd.processor import NpyFeeder
    data = np.random.default_rng(0).standard_normal((4, 3, 8, 25, 2)).astype(np.float32)
    np.save(tmp_path / 'd.npy', data)
    np.save(tmp_path / 'l.npy', np.array([1, 2, 3, 4]))
    f = NpyFeeder(str(tmp_path / 'd.npy'), str(tmp_path / 'l.npy'))
    x, y, i = f[2]
    assert np.array_equal(x, data[2]) and y == 3 and len(f) == 4


def test_import_class_paths():
    import agcn_amd  # noqa: F401
    from agcn_amd.processor import import_class
    assert import_class('model.agcn.Model').__name__ == 'Model'
    assert import_class('graph.kinetics.Graph')().A.shape == (3, 18, 18)


def test_ensemble_fuses_by_sample_name(tmp_path):
    """ensemble.py (reference ensemble.py:13-33): argmax(joint + alpha*bone) against the label pickle."""
    import pickle
    import numpy as np
    import ensemble
    names = [f's{i}' for i in range(6)]
    labels = [0, 1, 2, 3, 0, 1]
    rng = np.random.default_rng(0)
    joint = {n: rng.standard_normal(6) * 0.1 for n in names}
    bone = {n: rng.standard_normal(6) * 0.1 for n in names}
    for n, lab in zip(names[:4], labels[:4]):
        bone[n][lab] += 5.0                         # the bone stream knows the first four
    with open(tmp_path / 'j.pkl', 'wb') as f:
        pickle.dump(dict(reversed(list(joint.items()))), f)      # other order: fusion must match by name
    with open(tmp_path / 'b.pkl', 'wb') as f:
        pickle.dump(bone, f)
    with open(tmp_path / 'l.pkl', 'wb') as f:
        pickle.dump((names, labels), f)
    acc = ensemble.main(['--joint-score', str(tmp_path / 'j.pkl'), '--bone-score', str(tmp_path / 'b.pkl'),
                         '--label', str(tmp_path / 'l.pkl'), '--alpha', '1.0'])
    ref = np.mean([int(np.argmax(joint[n] + bone[n])) == lab for n, lab in zip(names, labels)])
    assert abs(acc[1] - ref) < 1e-12 and acc[1] >= 4 / 6


def test_eval_shards_deinterleave_like_the_reference():
    """rank r evaluates samples r, r+W, ... of the wrap-around padded order; the gathered scores go back with
    score[r::W] = part_r and the padding is cut (reference processor.py:862-868)."""
    import numpy as np
    n, W = 7, 3
    per = (n + W - 1) // W
    order = list(range(n)) + list(range(per * W - n))
    truth = np.arange(n, dtype=np.float32)[:, None] * np.ones((1, 4), dtype=np.float32)
    parts = [truth[[i for i in order[r::W]]] for r in range(W)]
    full = np.zeros((per * W, 4), dtype=np.float32)
    for r, val in enumerate(parts):
        full[r::W] = val
    assert np.array_equal(full[:n], truth)
