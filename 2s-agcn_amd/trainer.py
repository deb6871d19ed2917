"""Training-step engine: the MI355X counterpart of the hot loop of the reference ``Processor.train``
(utils/processor.py:658-756): forward, mean cross-entropy, zero_grad, backward, ``clip_grad_norm_(params, 1.0)``
(:698), SGD(momentum 0.9, nesterov, weight decay) step (:395-401, :703), with the learning-rate rule of
``adjust_learning_rate`` (:349-360).

MI355X-first differences from the reference's DDP path:
  * all parameters live in ONE flat fp32 buffer and all gradients in another (autograd accumulates straight
    into views of it), so data parallelism is a single RCCL all-reduce of ~14 MB per step over xGMI instead of
    DDP's bucket machinery, and the clip + SGD update is one fused HIP launch chain over the flat buffers
    (``agcn_sgd_step``), not 274 per-tensor updates;
  * one process per GPU (``torch.distributed``, backend "nccl" = RCCL); per-rank batch stays fixed (weak scaling,
    like the reference's DistributedSampler + per-rank batch_size, processor.py:500);
  * the gradient all-reduce is two buckets: the tail of the model (l8..fc, 75 % of the parameters) goes out
    asynchronously while l7..l1 still run their backward, the head follows at the end;
  * BatchNorm statistics: synchronised when the model's BatchNorms were converted (``dp.enable_sync_bn``, the
    reference's DDP semantics, processor.py:295), else per replica (its ``nn.DataParallel`` path, :336-343).
"""
import numpy as np
import torch
from . import dp as _dp
from . import lib as _lib


class FlatParams:
    """Re-homes every parameter of ``model`` into one contiguous fp32 buffer (and its gradient into another)."""

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        # pad every tensor to 4 floats so views stay 16-byte aligned
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.total = off
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.momentum = torch.zeros(self.total, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, offs):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
        self.offsets = offs

    def views(self):
        return [self.grad[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, self.offsets)]

    def detach_grads(self):
        """Let autograd ASSIGN the gradients (p.grad = None beforehand) instead of accumulating into the flat views:
        an accumulation is one tiny add kernel per parameter tensor (274 launches, 1.2 ms per step)."""
        for p in self.params:
            p.grad = None

    def gather_grads(self, lo=0, hi=None, require_all=False):
        """Bring the gradients autograd produced for parameters [lo, hi) into the flat buffer with one multi-tensor
        copy and re-attach the views (parameters that received no gradient count as zero, like zero_grad + accumulate
        would give).  Parameters whose .grad already IS the flat view are skipped (gathered earlier in this step).
        require_all: return False and do nothing unless every parameter of the range has its gradient."""
        hi = len(self.params) if hi is None else hi
        ps = self.params[lo:hi]
        views = [self.grad[o:o + p.numel()].view(p.shape) for p, o in zip(ps, self.offsets[lo:hi])]
        if require_all and any(p.grad is None for p in ps):
            return False
        todo = [(v, p) for v, p in zip(views, ps) if p.grad is None or p.grad.data_ptr() != v.data_ptr()]
        got = [(v, p.grad) for v, p in todo if p.grad is not None]
        missing = [v for v, p in todo if p.grad is None]
        if missing:
            torch._foreach_zero_(missing)
        if got:
            torch._foreach_copy_([v for v, _ in got], [g for _, g in got])
        for v, p in todo:
            p.grad = v
        return True

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):      # re-attach if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


def learning_rate(epoch, base_lr, steps, warm_up_epoch=0):
    """reference utils/processor.py:349-360 (optimizer 'SGD')."""
    if epoch < warm_up_epoch:
        return base_lr * (epoch + 1) / warm_up_epoch
    return base_lr * (0.1 ** int(np.sum(epoch >= np.array(steps))))


class TrainEngine:
    def __init__(self, model, base_lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4, max_grad_norm=1.0,
                 world_size=1, tail_bucket='auto'):
        """tail_bucket: 'auto' = the first step probes whether the early (tail) all-reduce bucket is possible and the
        ranks agree on it; True / False = fixed from the start (must be the same on every rank)."""
        self.model = model
        self.fp = FlatParams(model)
        self.lr = base_lr
        self.momentum, self.nesterov, self.weight_decay = momentum, nesterov, weight_decay
        self.max_grad_norm = max_grad_norm
        self.world_size = world_size
        self.steps_done = 0
        L = _lib.load()
        nbytes = L.agcn_sgd_step_workspace(self.fp.total)
        dev = self.fp.flat.device
        self.ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
        self.norm = torch.zeros(2, dtype=torch.float32, device=dev)
        self.loss_fn = torch.nn.CrossEntropyLoss()
        self._tail_lo = None          # first parameter index of the early all-reduce bucket (world_size > 1)
        self._tail_static = None      # None: undecided (first step only probes); True / False: agreed by ALL ranks
        self._tail_probe = False      # what this rank's hook saw during the probing step
        self._pending = []            # async all-reduce handles of this step
        self.rehearse = False         # True: take the multi-rank schedule (buckets, collectives) with a group of ONE rank
        self.early_buckets = 0        # how many steps sent their tail bucket from inside the backward (diagnostic)
        if world_size > 1:
            self._setup_overlap()
            if tail_bucket != 'auto':
                self._tail_static = bool(tail_bucket) and self._tail_lo is not None

    # ---- gradient all-reduce overlapped with the backward -----------------------------------------------------------
    def _setup_overlap(self):
        """Two buckets instead of DDP's ~25: the parameters of the LAST top-level modules (l8..l10 + fc hold 75 % of
        the 3.47 M parameters and finish their backward first) are all-reduced asynchronously (RCCL runs on its own
        stream) as soon as the backward has passed them, under the backward of l7..l1; the head bucket follows at the
        end.  The cut is the latest top-level child with at least half of the parameter mass behind it."""
        children = [(n, m) for n, m in self.model.named_children() if any(True for _ in m.parameters())]
        index = {id(p): i for i, p in enumerate(self.fp.params)}
        total = float(self.fp.numel)
        cut = None
        for n, m in children:
            ids = [index[id(p)] for p in m.parameters() if id(p) in index]
            if not ids:
                continue
            lo = min(ids)
            behind = sum(p.numel() for p in self.fp.params[lo:])
            if behind >= 0.5 * total:
                cut = (m, lo)
        if cut is None or cut[1] == 0:
            return
        self._tail_lo = cut[1]
        cut[0].register_full_backward_hook(self._tail_ready)

    def _tail_ready(self, module, grad_input, grad_output):
        """Backward has passed the cut module: every parameter from it to the end of the model has its gradient
        (AccumulateGrad nodes run at top priority).  If one is missing anyway, the bucket simply waits for the end."""
        if self._tail_lo is None or self._pending:
            return None
        # The bucket schedule must be IDENTICAL on every rank (ranks issuing tail+head against ranks issuing one full
        # all-reduce would hang or mix buffers), so it is decided once: the first step only probes whether the tail is
        # complete when the hook fires, the ranks agree on the answer (MIN all-reduce in _finish_allreduce), and from
        # then on the schedule is static -- a rank that cannot honour it raises instead of silently deviating.
        if self._tail_static is None:
            self._tail_probe = all(p.grad is not None for p in self.fp.params[self._tail_lo:])
            return None
        if not self._tail_static:
            return None
        if not self.fp.gather_grads(self._tail_lo, None, require_all=True):
            raise RuntimeError("agcn_amd.TrainEngine: a parameter behind the all-reduce cut had no gradient when the "
                               "backward passed the cut module, although all ranks agreed on the two-bucket schedule "
                               "at the first step; the schedule is static (every rank must issue the same collectives)")
        off = self.fp.offsets[self._tail_lo]
        self._pending.append(_dp.allreduce_gradients(self.fp.grad[off:], self.world_size, async_op=True, force=self.rehearse))
        self.early_buckets += 1
        return None

    def _finish_allreduce(self):
        if self._tail_lo is not None and self._tail_static is None:
            # first step: agree on the schedule (every rank must see its tail complete at the cut)
            self._tail_static = bool(_dp.allreduce_min_flag(self._tail_probe, max(self.world_size, 2 if self.rehearse else 1),
                                                            self.fp.grad.device))
        if self._pending:             # tail bucket already in flight: only the head remains
            off = self.fp.offsets[self._tail_lo]
            self.fp.gather_grads(0, self._tail_lo)
            self._pending.append(_dp.allreduce_gradients(self.fp.grad[:off], self.world_size, async_op=True, force=self.rehearse))
        else:
            self.fp.gather_grads()
            self._pending.append(_dp.allreduce_gradients(self.fp.grad, self.world_size, async_op=True, force=self.rehearse))
        for w in self._pending:
            if w is not None:
                w.wait()
        self._pending = []

    def backward_and_reduce(self, loss):
        """loss.backward() with autograd ASSIGNING the gradients, gathered into the flat buffer and (world_size > 1)
        SUM-all-reduced over the ranks in two overlapped buckets.  Device-agnostic (the gloo CPU test drives it)."""
        self.fp.detach_grads()
        self._pending = []
        loss.backward()
        if self.world_size > 1 or self.rehearse:
            self._finish_allreduce()      # RCCL over xGMI (SUM); the 1/world average is folded into the update
        else:
            self.fp.gather_grads()

    def train_step(self, data, label, before_step=None):
        """One optimisation step on a device-resident batch.  Returns the (device) loss tensor.
        ``before_step`` (optional) runs after the gradient all-reduce and before the fused clip+SGD update."""
        self.model.train()
        output = self.model(data)
        if isinstance(output, tuple):
            output = output[0]
        loss = self.loss_fn(output, label)
        self.backward_and_reduce(loss)
        if before_step is not None:
            before_step()
        self.apply_update()
        return loss

    def apply_update(self):
        """Fused global-norm clip + SGD(momentum, nesterov, weight decay) over the flat buffers (``agcn_sgd_step``);
        the flat gradient holds the SUM over ranks, 1/world_size is applied inside."""
        L = _lib.load()
        _lib.check(L.agcn_sgd_step(self.fp.flat.data_ptr(), self.fp.grad.data_ptr(), self.fp.momentum.data_ptr(),
                                   self.fp.total, float(self.lr), float(self.momentum), float(self.weight_decay),
                                   int(self.nesterov), float(self.max_grad_norm), 1.0 / self.world_size,
                                   int(self.steps_done == 0), self.ws.data_ptr(), self.ws.numel() * 4,
                                   self.norm.data_ptr(), _lib.stream()), "agcn_sgd_step")
        self.steps_done += 1
        from . import ops as _ops
        _ops.note_params_changed()        # raw-pointer update: invalidates caches derived from the parameters

    def grad_norm(self):
        return float(self.norm[0])


def synthetic_batch(batch, num_point=25, num_class=60, T=300, seed=1234, device='cuda'):
    """x ~ N(0,1) of the NTU shape (N,3,300,V,2), labels uniform."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(batch, 3, T, num_point, 2, generator=g, dtype=torch.float32)
    y = torch.randint(0, num_class, (batch,), generator=g, dtype=torch.int64)
    return x.to(device), y.to(device)
