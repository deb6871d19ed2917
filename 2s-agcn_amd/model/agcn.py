"""2s-AGCN network on the MI355X-native hot path.

Same public surface as the reference ``model/architecture/aagcn/agcn.py`` (``unit_tcn`` :36-50,
``unit_gcn`` :53-109, ``TCN_GCN_unit`` :112-129, ``Model`` :132-183): class names, constructor
arguments, ``forward(x)`` with x of shape (N, C, T, V, M), parameter/buffer names and shapes
 -- so reference checkpoints load with ``load_state_dict`` and
``'PA' in name`` style parameter matching (reference utils/processor.py:621-629) keeps working.

Underneath, every operator of unit_gcn / unit_tcn runs as a hand-written gfx950 HIP kernel through the
C-ABI in ``libagcn_hip.so`` (see ``ops.py``).  There is no CPU / stock-PyTorch fallback for them: a CPU tensor
or a missing extension raises.  The model prologue/epilogue (data_bn on (N, M*V*C, T), the global average pool and
the 256->num_class Linear; reference agcn.py:163-165,179-183) run on deterministic HIP entry points as well
(``agcn_data_bn_*``, ``agcn_pool_*``, ``agcn_linear_*``: csrc/small_ops.hip) -- the stock operators were the one source of
run-to-run differences between processes.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import ops


def import_class(name):
    """Resolve a dotted class path such as ``graph.ntu_rgb_d.Graph`` (reference agcn.py:9-14)."""
    import importlib
    module_name, _, cls = name.rpartition('.')
    return getattr(importlib.import_module(module_name), cls)


def conv_branch_init(conv, branches):
    # reference agcn.py:17-23: N(0, sqrt(2 / (out * in * k * branches))), zero bias
    out_c, in_c, k = conv.weight.size(0), conv.weight.size(1), conv.weight.size(2)
    nn.init.normal_(conv.weight, 0, math.sqrt(2.0 / (out_c * in_c * k * branches)))
    nn.init.constant_(conv.bias, 0)


def conv_init(conv):
    nn.init.kaiming_normal_(conv.weight, mode='fan_out')   # reference agcn.py:26-28
    nn.init.constant_(conv.bias, 0)


def bn_init(bn, scale):
    nn.init.constant_(bn.weight, scale)
    nn.init.constant_(bn.bias, 0)


def _require_gpu(x, who):
    if not x.is_cuda:
        raise RuntimeError(f"agcn_amd.{who}: input is on {x.device}; the hot path exists only as gfx950 HIP kernels "
                           f"(no CPU fallback). Move the model and data to a GPU.")
    if x.dtype != torch.float32:
        raise RuntimeError(f"agcn_amd.{who}: expected float32 input, got {x.dtype}")


def _bn_args(bn):
    return bn.weight, bn.bias, bn.running_mean, bn.running_var


def _bn_tick(bn, training):
    if training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)


def data_bn_forward(bn, x):
    """Model prologue: (N, C, T, V, M) -> BatchNorm1d over channels (m, v, c) -> (N*M, C, T, V) (reference
    agcn.py:163-165) on the deterministic HIP kernels; modules other than plain / synchronised BatchNorm1d (GhostBatchNorm1d)
    keep their own forward."""
    N, C, T, V, M = x.size()
    if ops.data_bn_supported(bn):
        y = ops.DataBNFunction.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training,
                                     ops.sync_of(bn))
        _bn_tick(bn, bn.training)
        return y
    x = x.permute(0, 4, 3, 1, 2).contiguous().view(N, M * V * C, T)
    x = bn(x)
    return x.view(N, M, V, C, T).permute(0, 1, 3, 4, 2).contiguous().view(N * M, C, T, V)


class unit_tcn(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=9, stride=1):
        super().__init__()
        if kernel_size not in (1, 9):
            raise ValueError("agcn_amd.unit_tcn: kernel_size must be 1 or 9")
        pad = int((kernel_size - 1) / 2)
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=(kernel_size, 1), padding=(pad, 0),
                              stride=(stride, 1))
        self.bn = nn.BatchNorm2d(out_channels)
        self.relu = nn.ReLU()
        self.stride = stride
        conv_init(self.conv)
        bn_init(self.bn, 1)

    def forward(self, x):
        _require_gpu(x, 'unit_tcn')
        y = ops.UnitTCNFunction.apply(x, self.conv.weight, self.conv.bias, *_bn_args(self.bn), self.stride,
                                      self.training, ops.sync_of(self.bn))
        _bn_tick(self.bn, self.training)
        return y


class unit_gcn(nn.Module):
    def __init__(self, in_channels, out_channels, A, coff_embedding=4, num_subset=3):
        super().__init__()
        if num_subset != 3:
            raise ValueError("agcn_amd.unit_gcn: num_subset must be 3")
        inter_channels = out_channels // coff_embedding
        self.inter_c = inter_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        A = np.asarray(A, dtype=np.float32)
        self.PA = nn.Parameter(torch.full(A.shape, 1e-6, dtype=torch.float32))
        # constant graph: follows .to()/.cuda() but stays out of the state_dict (reference keeps a plain attribute)
        self.register_buffer('A', torch.from_numpy(A.copy()), persistent=False)
        self.num_subset = num_subset
        self.conv_a = nn.ModuleList()
        self.conv_b = nn.ModuleList()
        self.conv_d = nn.ModuleList()
        for _ in range(num_subset):
            self.conv_a.append(nn.Conv2d(in_channels, inter_channels, 1))
            self.conv_b.append(nn.Conv2d(in_channels, inter_channels, 1))
            self.conv_d.append(nn.Conv2d(in_channels, out_channels, 1))
        if in_channels != out_channels:
            self.down = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1), nn.BatchNorm2d(out_channels))
        else:
            self.down = lambda x: x
        self.bn = nn.BatchNorm2d(out_channels)
        self.soft = nn.Softmax(-2)
        self.relu = nn.ReLU()
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                conv_init(m)
            elif isinstance(m, nn.BatchNorm2d):
                bn_init(m, 1)
        bn_init(self.bn, 1e-6)
        for i in range(num_subset):
            conv_branch_init(self.conv_d[i], num_subset)

    def packed_args(self):
        """Parameters in the layout the kernels take: theta/phi weights stacked row-wise
        [a0|b0|a1|b1|a2|b2], projection weights side by side [Wd0|Wd1|Wd2], conv_d biases summed."""
        ws, bs = [], []
        for i in range(3):
            ws += [self.conv_a[i].weight, self.conv_b[i].weight]
            bs += [self.conv_a[i].bias, self.conv_b[i].bias]
        wab = torch.cat(ws, dim=0)
        bab = torch.cat(bs, dim=0)
        co, ci = self.out_channels, self.in_channels
        wd = torch.cat([self.conv_d[i].weight.view(co, ci) for i in range(3)], dim=1)
        bd = self.conv_d[0].bias + self.conv_d[1].bias + self.conv_d[2].bias
        if isinstance(self.down, nn.Sequential):
            dn = (self.down[0].weight, self.down[0].bias) + _bn_args(self.down[1])
        else:
            dn = (None,) * 6
        return (self.A, self.PA, wab, bab, wd, bd) + _bn_args(self.bn) + dn

    def tick(self):
        _bn_tick(self.bn, self.training)
        if isinstance(self.down, nn.Sequential):
            _bn_tick(self.down[1], self.training)

    def forward(self, x):
        _require_gpu(x, 'unit_gcn')
        y = ops.UnitGCNFunction.apply(x, *self.packed_args(), self.training, None, True, ops.sync_of(self.bn))
        self.tick()
        return y


class TCN_GCN_unit(nn.Module):
    def __init__(self, in_channels, out_channels, A, stride=1, residual=True):
        super().__init__()
        self.gcn1 = unit_gcn(in_channels, out_channels, A)
        self.tcn1 = unit_tcn(out_channels, out_channels, stride=stride)
        self.relu = nn.ReLU()
        self.stride = stride
        if not residual:
            self.residual = lambda x: 0
            self.res_mode = 0
        elif (in_channels == out_channels) and (stride == 1):
            self.residual = lambda x: x
            self.res_mode = 1
        else:
            self.residual = unit_tcn(in_channels, out_channels, kernel_size=1, stride=stride)
            self.res_mode = 2

    def train(self, mode=True):
        # folded inference weights are derived from parameters the training path rewrites in place
        self.__dict__.pop('_infer_cache', None)
        return super().train(mode)

    def forward(self, x):
        _require_gpu(x, 'TCN_GCN_unit')
        t = self.tcn1
        if self.res_mode == 2:
            r = self.residual
            rargs = (r.conv.weight, r.conv.bias) + _bn_args(r.bn)
        else:
            rargs = (None,) * 6
        if not self.training and not torch.is_grad_enabled() and ops.infer_fold_enabled():
            # inference: BatchNorms folded into the contractions, residual + ReLU in their epilogues
            ga = self.gcn1.packed_args()
            down = None if ga[10] is None else tuple(ga[10:16])
            y = ops.unit_infer(x, *ga[:6], tuple(ga[6:10]), down, t.conv.weight, t.conv.bias, _bn_args(t.bn),
                               self.res_mode, rargs if self.res_mode == 2 else None, self.stride,
                               cache=self.__dict__.setdefault('_infer_cache', {}))
            if y is not None:
                return y
        y = ops.TCNGCNUnitFunction.apply(x, *self.gcn1.packed_args(), t.conv.weight, t.conv.bias, *_bn_args(t.bn),
                                         self.res_mode, *rargs, self.stride, self.training, ops.sync_of(t.bn))
        self.gcn1.tick()
        _bn_tick(t.bn, self.training)
        if self.res_mode == 2:
            _bn_tick(self.residual.bn, self.training)
        return y


class Model(nn.Module):
    def __init__(self, num_class=60, num_point=25, num_person=2, graph=None, graph_args=dict(), in_channels=3):
        super().__init__()
        if graph is None:
            raise ValueError()
        Graph = import_class(graph) if isinstance(graph, str) else graph
        self.graph = Graph(**graph_args)
        A = self.graph.A
        self.data_bn = nn.BatchNorm1d(num_person * in_channels * num_point)
        self.l1 = TCN_GCN_unit(3, 64, A, residual=False)      # in-channels hard-coded 3, reference agcn.py:145
        self.l2 = TCN_GCN_unit(64, 64, A)
        self.l3 = TCN_GCN_unit(64, 64, A)
        self.l4 = TCN_GCN_unit(64, 64, A)
        self.l5 = TCN_GCN_unit(64, 128, A, stride=2)
        self.l6 = TCN_GCN_unit(128, 128, A)
        self.l7 = TCN_GCN_unit(128, 128, A)
        self.l8 = TCN_GCN_unit(128, 256, A, stride=2)
        self.l9 = TCN_GCN_unit(256, 256, A)
        self.l10 = TCN_GCN_unit(256, 256, A)
        self.fc = nn.Linear(256, num_class)
        nn.init.normal_(self.fc.weight, 0, math.sqrt(2. / num_class))
        bn_init(self.data_bn, 1)

    def forward(self, x):
        N, C, T, V, M = x.size()
        _require_gpu(x, 'Model')
        x = data_bn_forward(self.data_bn, x)            # (N*M, C, T, V), reference agcn.py:163-165
        for k in range(1, 11):
            x = getattr(self, f'l{k}')(x)
        # global average pool over (T, V), then persons, and the classifier (agcn.py:179-183)
        return ops.PoolFCFunction.apply(x, self.fc.weight, self.fc.bias, M)
