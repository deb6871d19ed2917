"""Synchronised BatchNorm (reference DDP path, utils/processor.py:295): two ranks with half the batch each and the
per-channel sums all-reduced must reproduce one rank with the whole batch -- logits of the own half and the averaged
gradient.  Two processes share the one GPU of the test box (gloo carries the collectives)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker():
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import agcn_amd  # noqa: F401
    from agcn_amd import dp
    from model.agcn import Model
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dev = torch.device('cuda:0')
    dist.init_process_group(backend='gloo', rank=rank, world_size=world)

    def build():
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

    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 3, 32, 25, 2, generator=g).to(dev)
    y = torch.randint(0, 60, (4,), generator=g).to(dev)
    # reference: the whole batch on one rank, per-replica statistics
    ref = build()
    out_ref = ref(x)
    torch.nn.functional.cross_entropy(out_ref, y).backward()
    gref = torch.cat([p.grad.flatten() for p in ref.parameters()])
    # this rank's half with synchronised statistics
    # the reference way: convert the BatchNorms; the HIP stages follow (model/agcn.py::follow_sync_batchnorm)
    m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(build())
    xs, ys = x[rank::world], y[rank::world]
    out = m(xs)
    torch.nn.functional.cross_entropy(out, ys).backward()
    gs = torch.cat([p.grad.flatten() for p in m.parameters()])
    dist.all_reduce(gs)
    gs /= world
    e_out = float((out - out_ref[rank::world]).abs().max() / out_ref.abs().max())
    e_g = float((gs - gref).abs().max() / gref.abs().max())
    rm = float((dict(m.named_buffers())['l5.tcn1.bn.running_mean'] -
                dict(ref.named_buffers())['l5.tcn1.bn.running_mean']).abs().max())
    print(f'rank {rank} logits {e_out:.2e} grads {e_g:.2e} running_mean {rm:.2e}', flush=True)
    # gradients of a 4-clip batch: ReLU-kink flips between the two evaluation orders move single tensors by ~1e-3 of the
    # global maximum; a wrong reduction would show as O(0.1-1)
    assert e_out < 1e-4 and e_g < 1e-2 and rm < 1e-5, (e_out, e_g, rm)
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sync_bn_two_ranks_match_full_batch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), 'worker'], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'worker':
    worker()
