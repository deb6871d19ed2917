"""Multi-process data-parallel path on CPU (gloo, world_size 2): flat-gradient all-reduce + sharding.

Keeps the reference's only reusable self-check (model/architecture/aagcn/aagcn.py:592-616): with per-replica
BatchNorm, the gradient of the DDP job equals the AVERAGE of the gradients of its shards."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, (9, 1), padding=(4, 0)), torch.nn.BatchNorm2d(8),
                               torch.nn.ReLU(), torch.nn.Conv2d(8, 5, 1), torch.nn.AdaptiveAvgPool2d(1),
                               torch.nn.Flatten())


def _batch():
    g = torch.Generator().manual_seed(1)
    return torch.randn(6, 3, 12, 5, generator=g), torch.randint(0, 5, (6,), generator=g)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import agcn_amd  # noqa: F401
    from agcn_amd import dp
    from agcn_amd.trainer import FlatParams
    r, w = dp.init_distributed(backend='gloo')
    assert (r, w) == (rank, world)
    net = _net()
    if rank == 1:                                   # ranks start different; broadcast must fix that
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    fp = FlatParams(net)
    dp.broadcast_parameters(fp.flat, world)
    x, y = _batch()
    idx = dp.shard_indices(x.shape[0], rank, world)
    fp.zero_grad()
    loss = torch.nn.functional.cross_entropy(net(x[idx]), y[idx])
    loss.backward()
    for p, o in zip(fp.params, fp.offsets):        # autograd accumulated into the flat buffer's views
        assert p.grad.data_ptr() == fp.grad.data_ptr() + 4 * o
    dp.allreduce_gradients(fp.grad, world)
    total = dp.allreduce_scalar(loss.item(), world, 'cpu')
    torch.save({'grad': fp.grad.clone() / world, 'flat': fp.flat.clone(), 'idx': idx, 'loss_sum': total},
               os.path.join(out_dir, f'rank{rank}.pt'))
    torch.distributed.destroy_process_group()


def test_flat_allreduce_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f'rank{r}.pt')) for r in range(world)]
    assert torch.equal(res[0]['grad'], res[1]['grad'])          # every rank holds the same averaged gradient
    assert torch.equal(res[0]['flat'], res[1]['flat'])          # and the same (broadcast) parameters
    assert sorted(res[0]['idx'] + res[1]['idx']) == list(range(6))
    # single-process restatement: average of the per-shard gradients (per-replica BN)
    sys.path.insert(0, ROOT)
    import agcn_amd  # noqa: F401
    from agcn_amd.trainer import FlatParams
    x, y = _batch()
    acc, losses = None, 0.0
    for r in range(world):
        net = _net()
        fp = FlatParams(net)
        fp.zero_grad()
        loss = torch.nn.functional.cross_entropy(net(x[res[r]['idx']]), y[res[r]['idx']])
        loss.backward()
        losses += loss.item()
        acc = fp.grad.clone() if acc is None else acc + fp.grad
    assert torch.allclose(res[0]['grad'], acc / world, rtol=1e-5, atol=1e-7)
    assert abs(res[0]['loss_sum'] - losses) < 1e-5


def _engine_net():
    """Several top-level children so that the engine finds a cut for its early (tail) bucket."""
    torch.manual_seed(0)
    net = torch.nn.Sequential()
    net.add_module('l1', torch.nn.Sequential(torch.nn.Conv2d(3, 4, (9, 1), padding=(4, 0)), torch.nn.BatchNorm2d(4),
                                             torch.nn.ReLU()))
    net.add_module('l2', torch.nn.Sequential(torch.nn.Conv2d(4, 16, 1), torch.nn.BatchNorm2d(16), torch.nn.ReLU()))
    net.add_module('l3', torch.nn.Sequential(torch.nn.Conv2d(16, 16, 1), torch.nn.ReLU()))
    net.add_module('pool', torch.nn.Sequential(torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten()))
    net.add_module('fc', torch.nn.Linear(16, 5))
    return net


def _engine_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import agcn_amd  # noqa: F401
    from agcn_amd import dp
    from agcn_amd.trainer import TrainEngine
    dp.init_distributed(backend='gloo')
    net = _engine_net()
    eng = TrainEngine(net, world_size=world)
    assert eng._tail_lo is not None and 0 < eng._tail_lo < len(eng.fp.params)      # an early bucket exists
    x, y = _batch()
    idx = dp.shard_indices(x.shape[0], rank, world)
    for _ in range(3):                               # several times: the views/hooks must survive a step
        loss = torch.nn.functional.cross_entropy(net(x[idx]), y[idx])
        eng.backward_and_reduce(loss)
    # step 1 only probes and the ranks agree on the (static) schedule; steps 2 and 3 reduce their tail bucket from
    # inside the backward
    assert eng._tail_static is True and eng.early_buckets == 2
    for p, o in zip(eng.fp.params, eng.fp.offsets):
        assert p.grad.data_ptr() == eng.fp.grad.data_ptr() + 4 * o
    torch.save({'grad': eng.fp.grad.clone() / world, 'idx': idx, 'tail_lo': eng._tail_lo},
               os.path.join(out_dir, f'eng{rank}.pt'))
    torch.distributed.destroy_process_group()


def test_engine_bucketed_allreduce_world2(tmp_path):
    """TrainEngine.backward_and_reduce: tail bucket reduced from the backward hook, head at the end; the result is the
    average of the shard gradients (per-replica BN), identical on both ranks."""
    world = 2
    mp.spawn(_engine_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f'eng{r}.pt')) for r in range(world)]
    assert torch.equal(res[0]['grad'], res[1]['grad'])
    sys.path.insert(0, ROOT)
    import agcn_amd  # noqa: F401
    from agcn_amd.trainer import TrainEngine
    x, y = _batch()
    acc = None
    for r in range(world):
        net = _engine_net()
        eng = TrainEngine(net, world_size=1)
        for _ in range(3):
            loss = torch.nn.functional.cross_entropy(net(x[res[r]['idx']]), y[res[r]['idx']])
            eng.backward_and_reduce(loss)
        acc = eng.fp.grad.clone() if acc is None else acc + eng.fp.grad
    assert torch.allclose(res[0]['grad'], acc / world, rtol=1e-5, atol=1e-7)


def test_shard_indices_cover_and_pad():
    import agcn_amd  # noqa: F401
    from agcn_amd import dp
    assert dp.shard_indices(7, 0, 2) == [0, 2, 4, 6] and dp.shard_indices(7, 1, 2) == [1, 3, 5, 0]
    assert dp.shard_indices(8, 3, 4) == [3, 7]


def test_learning_rate_rule():
    import agcn_amd  # noqa: F401
    from agcn_amd.trainer import learning_rate
    # reference utils/processor.py:349-360 with train_joint.yaml: base_lr 0.1, step [30, 40]
    assert learning_rate(0, 0.1, [30, 40]) == 0.1
    assert abs(learning_rate(30, 0.1, [30, 40]) - 0.01) < 1e-12
    assert abs(learning_rate(45, 0.1, [30, 40]) - 0.001) < 1e-12
    assert abs(learning_rate(1, 0.1, [30, 40], warm_up_epoch=5) - 0.04) < 1e-12
