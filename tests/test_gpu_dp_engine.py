"""The data-parallel training step of the HIP model through ``TrainEngine`` with two ranks (gloo carries the
collectives, both ranks share the one GPU of the test box; RCCL needs one GPU per rank):

* per-replica BatchNorm: the reference's own DP property (model/architecture/aagcn/aagcn.py:592-616) -- the flat
  gradient after the all-reduce, averaged, equals the average of the two shard gradients computed by ONE rank, and the
  parameters after clip + SGD equal that rank's update with the averaged gradient;
* synchronised BatchNorm (reference DDP path, utils/processor.py:295): two ranks with half a batch each reproduce one
  rank with the whole batch, tensor by tensor (BN weight/bias gradients, running_mean AND running_var included).
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(dev):
    import torch
    from model.agcn import Model
    torch.manual_seed(3)
    m = Model(num_class=60, num_point=25, num_person=2, graph='graph.ntu_rgb_d.Graph',
              graph_args=dict(labeling_mode='spatial'))
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith('gcn1.bn.weight'):
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            elif name.endswith('gcn1.PA'):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return m.to(dev).train()


def _per_tensor_err(engine_a, flat_a, flat_b):
    """max over parameter tensors of max|a-b| / max|b| (per tensor), and the name of the worst tensor."""
    import torch
    worst, name = 0.0, ''
    names = [n for n, p in engine_a.model.named_parameters() if p.requires_grad]
    gmax = float(flat_b.abs().max())
    for n, p, o in zip(names, engine_a.fp.params, engine_a.fp.offsets):
        a, b = flat_a[o:o + p.numel()], flat_b[o:o + p.numel()]
        den = float(b.abs().max())
        if den < 1e-5 * gmax:        # structurally-zero gradients (conv biases in front of a BatchNorm, the conv_b bias under
            e = float((a - b).abs().max()) / gmax      # the column softmax): rounding residue, measured on the gradient's scale
        else:
            e = float((a - b).abs().max()) / den
        if e > worst:
            worst, name = e, n
    return worst, name


def worker(mode):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import agcn_amd  # noqa: F401
    from agcn_amd import dp
    from agcn_amd.trainer import TrainEngine
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dev = torch.device('cuda:0')
    dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 3, 32, 25, 2, generator=g).to(dev)
    y = torch.randint(0, 60, (4,), generator=g).to(dev)
    kw = dict(base_lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4, max_grad_norm=1.0)

    # ---- the two-rank job through TrainEngine.train_step ----
    m = _build(dev)
    if mode == 'sync':
        m = dp.enable_sync_bn(m, world)
    eng = TrainEngine(m, world_size=world, tail_bucket=True, **kw)     # static two-bucket schedule from step 1
    dp.broadcast_parameters(eng.fp.flat, world)
    p0 = eng.fp.flat.clone()
    eng.train_step(x[rank::world], y[rank::world])
    assert eng.early_buckets == 1, 'the tail bucket was not reduced from inside the backward'
    g_job = eng.fp.grad.clone() / world            # train_step leaves the SUM in the buffer
    p_job = eng.fp.flat.clone()

    # ---- single-rank restatement ----
    ref = _build(dev)
    e1 = TrainEngine(ref, world_size=1, **kw)
    assert torch.equal(e1.fp.flat, p0)
    if mode == 'replica':
        acc = torch.zeros_like(e1.fp.grad)
        for r in range(world):
            loss = torch.nn.functional.cross_entropy(ref(x[r::world]), y[r::world])
            e1.backward_and_reduce(loss)
            acc += e1.fp.grad
        e1.fp.grad.copy_(acc / world)
    else:
        loss = torch.nn.functional.cross_entropy(ref(x), y)
        e1.backward_and_reduce(loss)
    g_ref = e1.fp.grad.clone()
    e1.apply_update()
    p_ref = e1.fp.flat.clone()

    eg, ng = _per_tensor_err(eng, g_job, g_ref)
    ep = float((p_job - p_ref).abs().max() / p_ref.abs().max())
    msg = f'rank {rank} mode {mode}: worst per-tensor grad err {eg:.2e} ({ng}), params {ep:.2e}'
    if mode == 'sync':
        bufs, rbufs = dict(m.named_buffers()), dict(ref.named_buffers())
        erm = max(float((bufs[k] - rbufs[k]).abs().max()) for k in bufs if k.endswith('running_mean'))
        erv = max(float((bufs[k] - rbufs[k]).abs().max() / rbufs[k].abs().max()) for k in bufs
                  if k.endswith('running_var'))
        msg += f' running_mean {erm:.2e} running_var {erv:.2e}'
        assert erm < 1e-5 and erv < 1e-5, msg
    print(msg, flush=True)
    # replica: same maths in the same order on both sides -> tight.  sync: a 4-clip batch evaluated in two halves:
    # ReLU-kink flips move single tensors by ~1e-3..1e-2 of their maximum; a wrong reduction or a wrong
    # 1/world scale shows as O(0.5-1) on that tensor.
    tol_g = 2e-4 if mode == 'replica' else 3e-2
    assert eg < tol_g and ep < 1e-4, msg
    dist.destroy_process_group()


def _run(mode, port):
    import torch
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), 'worker', mode], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=400)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    print('\n'.join(o.strip().splitlines()[-1] for o in outs))


@pytest.mark.gpu
def test_engine_two_ranks_replica_bn_matches_shard_average():
    _run('replica', 29541)


@pytest.mark.gpu
def test_engine_two_ranks_sync_bn_matches_full_batch():
    _run('sync', 29543)


if __name__ == '__main__' and len(sys.argv) > 2 and sys.argv[1] == 'worker':
    worker(sys.argv[2])
