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
