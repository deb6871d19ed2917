"""GhostBatchNorm (reference ``model/layers/module/ghostbatchnorm.py:17-61, 77-120``): BatchNorm whose training
statistics are taken over ``num_splits`` virtual sub-batches -- sample n belongs to sub-batch ``n % num_splits`` (the
reference views the (N, C, ...) input as (N / S, S * C, ...)) -- with ONE shared weight/bias and S*C running statistics
that are averaged over the sub-batches when the module is switched to eval.

Same class names, constructor arguments, parameters and buffer shapes as the reference, so its checkpoints load.  The
2-D variant is only a parameter holder for the HIP units (``ops._bn_coeffs`` recognises it by its S*C running
statistics and calls the BatchNorm kernels on the regrouped shape); ``forward`` exists for stand-alone use and for the
1-D ``data_bn`` of the model prologue, and uses stock PyTorch operators exactly as the reference does."""
import torch
import torch.nn.functional as F


class _GhostMixin:
    def _init_ghost(self, num_features, num_splits):
        self.num_splits = int(num_splits)
        self.register_buffer('running_mean', torch.zeros(num_features * self.num_splits))
        self.register_buffer('running_var', torch.ones(num_features * self.num_splits))

    def train(self, mode=True):
        if self.training and not mode:      # leaving training: collate the per-split running statistics
            s, c = self.num_splits, self.num_features
            self.running_mean = self.running_mean.view(s, c).mean(0).repeat(s)
            self.running_var = self.running_var.view(s, c).mean(0).repeat(s)
        return super().train(mode)

    def _ghost_forward(self, x):
        s, c = self.num_splits, self.num_features
        if self.training or not self.track_running_stats:
            shape = x.shape
            y = F.batch_norm(x.reshape(-1, c * s, *shape[2:]), self.running_mean, self.running_var,
                             self.weight.repeat(s), self.bias.repeat(s), True, self.momentum, self.eps)
            return y.view(shape)
        return F.batch_norm(x, self.running_mean[:c], self.running_var[:c], self.weight, self.bias, False,
                            self.momentum, self.eps)


class GhostBatchNorm1d(_GhostMixin, torch.nn.BatchNorm1d):
    def __init__(self, num_features, num_splits=16, **kw):
        super().__init__(num_features, **kw)
        self._init_ghost(num_features, num_splits)

    def forward(self, x):
        return self._ghost_forward(x)


class GhostBatchNorm2d(_GhostMixin, torch.nn.BatchNorm2d):
    def __init__(self, num_features, num_splits=16, **kw):
        super().__init__(num_features, **kw)
        self._init_ghost(num_features, num_splits)

    def forward(self, x):
        return self._ghost_forward(x)
