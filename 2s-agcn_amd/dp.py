"""Single-node data parallelism: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests).

Counterpart of the reference's DDP path (main.py:20-40, utils/processor.py:294-296) built MI355X-first: the model's
gradients live in ONE flat fp32 buffer (trainer.FlatParams), so a step needs exactly one SUM all-reduce of ~14 MB
(3.47 M parameters); the division by world_size is folded into the fused clip+SGD kernel (``grad_scale``).  The clip
(processor.py:698) therefore sees the averaged gradient, as in the reference.  Batches are sharded like the
reference's ``DistributedSampler`` (feeders/loader.py:378-383): rank r takes samples r, r+W, r+2W, ...
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None, device=None):
    """Initialise the default process group from the torchrun environment (RANK/WORLD_SIZE/MASTER_*).
    Returns (rank, world_size).  No-op for world_size 1."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '8020')       # the reference's fixed port, main.py:22
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kwargs = {}
        if backend == 'nccl' and device is not None:
            kwargs['device_id'] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world


def enable_sync_bn(model, world_size, group=None):
    """Reference DDP semantics (utils/processor.py:295 converts every BatchNorm to SyncBatchNorm): the model's
    BatchNorm modules are converted by torch exactly as the reference does; the HIP units see the converted class of
    the module whose parameters they borrow (``ops.sync_of``) and all-reduce their per-channel sums over its process
    group, the stock ``data_bn`` runs torch's SyncBatchNorm.  The policy lives on the modules: nothing process-global.
    Returns the (possibly converted) model.  With world_size 1 nothing changes."""
    if world_size > 1:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model, process_group=group)
    return model


def allreduce_gradients(flat_grad, world_size, async_op=False, force=False):
    """SUM all-reduce of (a slice of) the flat gradient buffer.  The caller applies 1/world_size.
    async_op: returns the work handle (None for world_size 1); with RCCL the collective runs on the communicator's own
    stream and ``handle.wait()`` makes the compute stream wait for it.  force: issue the collective even in a group of
    one rank (the one-GPU rehearsal of the multi-GPU schedule, bench.py --rehearse-dp)."""
    if world_size > 1 or (force and dist.is_initialized()):
        from . import ops as _ops
        _ops.COLLECTIVES['grad'] += 1
        w = dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, async_op=async_op)
        return w if async_op else flat_grad
    return None if async_op else flat_grad


def broadcast_parameters(flat_param, world_size, src=0):
    """Make every rank start from rank 0's parameters (DDP does this at wrap time)."""
    if world_size > 1:
        dist.broadcast(flat_param, src=src)
    return flat_param


def shard_indices(num_samples, rank, world_size):
    """DistributedSampler striding without shuffle: rank r owns r, r+W, ... (padded by wrap-around so every rank
    gets ceil(n/W) samples, as torch's sampler does)."""
    per = (num_samples + world_size - 1) // world_size
    idx = list(range(num_samples)) + list(range(per * world_size - num_samples))
    return idx[rank::world_size]


def allreduce_scalar(value, world_size, device):
    """Sum of a python scalar over ranks (loss / accuracy logging, reference processor.py:178-183)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if world_size > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def allreduce_min_flag(flag, world_size, device):
    """True only if ``flag`` holds on EVERY rank (one tiny MIN all-reduce; used once, to agree on a static schedule)."""
    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float32, device=device)
    if world_size > 1 and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)
