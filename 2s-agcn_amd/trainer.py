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
  * BatchNorm statistics are per replica (what the reference's ``nn.DataParallel`` path does, processor.py:336-343).
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

    def gather_grads(self):
        """Bring the gradients autograd produced into the flat buffer with one multi-tensor copy and re-attach the
        views (parameters that received no gradient count as zero, like zero_grad + accumulate would give)."""
        views = self.views()
        got = [(v, p.grad) for v, p in zip(views, self.params) if p.grad is not None]
        missing = [v for v, p in zip(views, self.params) if p.grad is None]
        if missing:
            torch._foreach_zero_(missing)
        if got:
            torch._foreach_copy_([v for v, _ in got], [g for _, g in got])
        for p, v in zip(self.params, views):
            p.grad = v

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
                 world_size=1):
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

    def train_step(self, data, label, before_step=None):
        """One optimisation step on a device-resident batch.  Returns the (device) loss tensor.
        ``before_step`` (optional) runs after the gradient all-reduce and before the fused clip+SGD update."""
        self.model.train()
        output = self.model(data)
        if isinstance(output, tuple):
            output = output[0]
        loss = self.loss_fn(output, label)
        self.fp.detach_grads()
        loss.backward()
        self.fp.gather_grads()
        _dp.allreduce_gradients(self.fp.grad, self.world_size)     # RCCL over xGMI; averaged inside the update
        if before_step is not None:
            before_step()
        L = _lib.load()
        _lib.check(L.agcn_sgd_step(self.fp.flat.data_ptr(), self.fp.grad.data_ptr(), self.fp.momentum.data_ptr(),
                                   self.fp.total, float(self.lr), float(self.momentum), float(self.weight_decay),
                                   int(self.nesterov), float(self.max_grad_norm), 1.0 / self.world_size,
                                   int(self.steps_done == 0), self.ws.data_ptr(), self.ws.numel() * 4,
                                   self.norm.data_ptr(), _lib.stream()), "agcn_sgd_step")
        self.steps_done += 1
        return loss

    def grad_norm(self):
        return float(self.norm[0])


def synthetic_batch(batch, num_point=25, num_class=60, T=300, seed=1234, device='cuda'):
    """x ~ N(0,1) of the NTU shape (N,3,300,V,2), labels uniform."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(batch, 3, T, num_point, 2, generator=g, dtype=torch.float32)
    y = torch.randint(0, num_class, (batch,), generator=g, dtype=torch.int64)
    return x.to(device), y.to(device)
