"""AAGCN (attention-enhanced adaptive GCN, BASELINE config 4) on the MI355X-native hot path.

Same public surface as the reference ``model/architecture/aagcn/aagcn.py``: ``SpatialAttention`` / ``TemporalAttention`` /
``ChannelAttention`` (:59-116), ``NonAdaptiveGCN`` / ``AdaptiveGCN`` (:119-177), ``TCNUnit`` (:184-207), ``GCNUnit``
(:210-271), ``TCNGCNUnit`` (:274-322), ``BaseModel`` (:328-533), ``Model`` (:536-577): constructor arguments,
``forward(x) -> (logits, None)``, and the identical state_dict, including the reference's duplicate registration of the
shared ``conv_d`` ModuleList under ``gcn1.conv_d.*`` and ``gcn1.agcn.conv_d.*`` (:228-233).

What runs where: the GCN core (theta/phi contraction, adaptive adjacency ``PA + alpha*softmax``, fused
aggregate+project, BN + down + ReLU) and the temporal unit (9x1 conv, BN, residual, ReLU) with their backward are the
same gfx950 HIP kernels as AGCN (``ops.UnitGCNFunction`` / ``ops.TCNResidualFunction``).  The three STC attention
gates between them (bandwidth-bound mean -> tiny conv/fc -> sigmoid -> ``y*s + y`` passes) are one autograd
node (``ops.STCAttentionFunction``): every pass over the (N,C,T,V) activation is a HIP kernel (``csrc/attention.hip``:
3 reads + 1 write forward, 4 reads + 1 write backward); only the few-KB gate networks (Conv1d C->1, two Linears on
(N,C,V)/(N,C,T)/(N,C) tensors) are tensor code.
GhostBatchNorm (``gbn_split >= 2``) runs on the same HIP BatchNorm stages (``ghostbatchnorm.py``).  Not supported:
``data_norm='ln'``.
"""
import math
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .agcn import (_bn_args, _bn_tick, _require_gpu, bn_init, conv_branch_init, conv_init, data_bn_forward,
                   import_class)
from .ghostbatchnorm import GhostBatchNorm1d, GhostBatchNorm2d


def batch_norm_1d(num_channels, gbn_split=None):
    """reference aagcn.py:45-49"""
    if gbn_split is None or gbn_split < 2:
        return nn.BatchNorm1d(num_channels)
    return GhostBatchNorm1d(num_channels, gbn_split)


def batch_norm_2d(num_channels, gbn_split=None):
    """reference aagcn.py:52-56"""
    if gbn_split is None or gbn_split < 2:
        return nn.BatchNorm2d(num_channels)
    return GhostBatchNorm2d(num_channels, gbn_split)


class _Gate(torch.autograd.Function):
    """y = x * se + x with a broadcast gate se (reference aagcn.py:264-271), as one pass forward and three backward
    (stock autograd spends 2 + 5 full-tensor passes on the same expression)."""

    @staticmethod
    def forward(ctx, x, se):
        ctx.save_for_backward(x, se)
        return torch.addcmul(x, x, se)

    @staticmethod
    def backward(ctx, g):
        x, se = ctx.saved_tensors
        g = g.contiguous()
        return torch.addcmul(g, g, se), (g * x).sum_to_size(se.shape)


class SpatialAttention(nn.Module):
    def __init__(self, in_channels: int, out_channels: int = 1, kernel_size: int = 9):
        super().__init__()
        pad = (kernel_size - 1) // 2
        self.conv_sa = nn.Conv1d(in_channels, out_channels, kernel_size, padding=pad)
        nn.init.xavier_normal_(self.conv_sa.weight)
        nn.init.constant_(self.conv_sa.bias, 0)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        se = self.sigmoid(self.conv_sa(x.mean(-2)))          # N 1 V
        return _Gate.apply(x, se.unsqueeze(-2))


class TemporalAttention(nn.Module):
    def __init__(self, in_channels: int, out_channels: int = 1, kernel_size: int = 9):
        super().__init__()
        pad = (kernel_size - 1) // 2
        self.conv_ta = nn.Conv1d(in_channels, out_channels, kernel_size, padding=pad)
        nn.init.constant_(self.conv_ta.weight, 0)
        nn.init.constant_(self.conv_ta.bias, 0)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        se = self.sigmoid(self.conv_ta(x.mean(-1)))          # N 1 T
        return _Gate.apply(x, se.unsqueeze(-1))


class ChannelAttention(nn.Module):
    def __init__(self, in_channels: int, rr: int = 2):
        super().__init__()
        self.fc1c = nn.Linear(in_channels, in_channels // rr)
        self.fc2c = nn.Linear(in_channels // rr, in_channels)
        nn.init.kaiming_normal_(self.fc1c.weight)
        nn.init.constant_(self.fc1c.bias, 0)
        nn.init.constant_(self.fc2c.weight, 0)
        nn.init.constant_(self.fc2c.bias, 0)
        self.sigmoid = nn.Sigmoid()
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        se = self.relu(self.fc1c(x.mean(-1).mean(-1)))
        se = self.sigmoid(self.fc2c(se))
        return _Gate.apply(x, se.unsqueeze(-1).unsqueeze(-1))


class NonAdaptiveGCN(nn.Module):
    """Fixed-graph aggregation; holds the graph and the shared conv_d list (compute happens in GCNUnit)."""

    def __init__(self, in_channels, out_channels, A, conv_d, num_subset=3):
        super().__init__()
        self.num_subset = num_subset
        self.register_buffer('A', torch.from_numpy(np.asarray(A, dtype=np.float32).copy()), persistent=False)
        self.conv_d = conv_d


class AdaptiveGCN(nn.Module):
    """Parameters of the adaptive branch: PA (init = A), alpha (init 0), conv_a/conv_b (C -> Cout/4)."""

    def __init__(self, in_channels, out_channels, A, conv_d, num_subset=3):
        super().__init__()
        self.num_subset = num_subset
        self.PA = nn.Parameter(torch.from_numpy(np.asarray(A, dtype=np.float32).copy()))
        self.alpha = nn.Parameter(torch.zeros(1))
        self.conv_a = nn.ModuleList()
        self.conv_b = nn.ModuleList()
        for _ in range(num_subset):
            self.conv_a.append(nn.Conv2d(in_channels, out_channels, 1))
            self.conv_b.append(nn.Conv2d(in_channels, out_channels, 1))
        self.soft = nn.Softmax(-2)
        self.conv_d = conv_d


class TCNUnit(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=9, stride=1, pad=True, gbn_split=None):
        super().__init__()
        if kernel_size not in (1, 9) or not pad:
            raise NotImplementedError("agcn_amd.aagcn.TCNUnit: kernel_size in {1, 9} with padding only")
        padding = (kernel_size - 1) // 2
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=(kernel_size, 1), padding=(padding, 0),
                              stride=(stride, 1))
        self.bn = batch_norm_2d(out_channels, gbn_split)
        self.stride = stride
        conv_init(self.conv)
        bn_init(self.bn, 1)

    def forward(self, x):
        _require_gpu(x, 'aagcn.TCNUnit')
        y = ops.UnitTCNFunction.apply(x, self.conv.weight, self.conv.bias, *_bn_args(self.bn), self.stride,
                                      self.training, ops.sync_of(self.bn))
        _bn_tick(self.bn, self.training)
        return y


class GCNUnit(nn.Module):
    def __init__(self, in_channels, out_channels, A, coff_embedding=4, num_subset=3, adaptive=AdaptiveGCN,
                 attention=True, gbn_split=None):
        super().__init__()
        if num_subset != 3:
            raise ValueError("agcn_amd.aagcn.GCNUnit: num_subset must be 3")
        inter_channels = out_channels // coff_embedding
        self.inter_c, self.out_c, self.in_c = inter_channels, out_channels, in_channels
        self.num_subset = num_subset
        num_jpts = A.shape[-1]
        self.conv_d = nn.ModuleList()
        for _ in range(num_subset):
            self.conv_d.append(nn.Conv2d(in_channels, out_channels, 1))
        self.agcn = adaptive(in_channels, inter_channels, A, self.conv_d, num_subset)
        if attention:
            ker_jpt = num_jpts - 1 if not num_jpts % 2 else num_jpts
            self.attn_s = SpatialAttention(out_channels, kernel_size=ker_jpt)
            self.attn_t = TemporalAttention(out_channels)
            self.attn_c = ChannelAttention(out_channels)
        else:
            self.attn_s, self.attn_t, self.attn_c = None, None, None
        if in_channels != out_channels:
            self.down = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1), batch_norm_2d(out_channels, gbn_split))
        else:
            self.down = lambda x: x
        self.bn = batch_norm_2d(out_channels, gbn_split)
        self.relu = nn.ReLU(inplace=True)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                conv_init(m)
            elif isinstance(m, nn.BatchNorm2d):
                bn_init(m, 1)
        bn_init(self.bn, 1e-6)
        for i in range(num_subset):
            conv_branch_init(self.conv_d[i], num_subset)

    def forward(self, x):
        _require_gpu(x, 'aagcn.GCNUnit')
        co, ci = self.out_c, self.in_c
        wd = torch.cat([self.conv_d[i].weight.view(co, ci) for i in range(3)], dim=1)
        bd = self.conv_d[0].bias + self.conv_d[1].bias + self.conv_d[2].bias
        if isinstance(self.down, nn.Sequential):
            dn = (self.down[0].weight, self.down[0].bias) + _bn_args(self.down[1])
        else:
            dn = (None,) * 6
        if isinstance(self.agcn, AdaptiveGCN):
            ws, bs = [], []
            for i in range(3):
                ws += [self.agcn.conv_a[i].weight, self.agcn.conv_b[i].weight]
                bs += [self.agcn.conv_a[i].bias, self.agcn.conv_b[i].bias]
            y = ops.UnitGCNFunction.apply(x, None, self.agcn.PA, torch.cat(ws, 0), torch.cat(bs, 0), wd, bd,
                                          *_bn_args(self.bn), *dn, self.training, self.agcn.alpha, True,
                                          ops.sync_of(self.bn))
        else:
            y = ops.UnitGCNFunction.apply(x, self.agcn.A, None, None, None, wd, bd, *_bn_args(self.bn), *dn,
                                          self.training, None, False, ops.sync_of(self.bn))
        _bn_tick(self.bn, self.training)
        if isinstance(self.down, nn.Sequential):
            _bn_tick(self.down[1], self.training)
        if self.attn_s is not None and self.attn_t is not None and self.attn_c is not None:
            # the three gates as one autograd node on the HIP reduction / apply passes (ops.STCAttentionFunction); the
            # attention modules only hold the parameters (their own forward is the stand-alone tensor-op version)
            s_, t_, c_ = self.attn_s.conv_sa, self.attn_t.conv_ta, self.attn_c
            return ops.STCAttentionFunction.apply(y, s_.weight, s_.bias, t_.weight, t_.bias, c_.fc1c.weight,
                                                  c_.fc1c.bias, c_.fc2c.weight, c_.fc2c.bias)
        y = y if self.attn_s is None else self.attn_s(y)
        y = y if self.attn_t is None else self.attn_t(y)
        y = y if self.attn_c is None else self.attn_c(y)
        return y


class TCNGCNUnit(nn.Module):
    def __init__(self, in_channels, out_channels, A, num_subset=3, kernel_size=9, stride=1, pad=True, residual=True,
                 adaptive=AdaptiveGCN, attention=True, gbn_split=None):
        super().__init__()
        self.gcn1 = GCNUnit(in_channels, out_channels, A, num_subset=num_subset, adaptive=adaptive,
                            attention=attention, gbn_split=gbn_split)
        self.tcn1 = TCNUnit(out_channels, out_channels, kernel_size=kernel_size, stride=stride, pad=pad,
                            gbn_split=gbn_split)
        self.stride = stride
        if not residual:
            self.residual = lambda x: 0
            self.res_mode = 0
        elif (in_channels == out_channels) and (stride == 1):
            self.residual = lambda x: x
            self.res_mode = 1
        else:
            self.residual = TCNUnit(in_channels, out_channels, kernel_size=1, stride=stride, gbn_split=gbn_split)
            self.res_mode = 2
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        _require_gpu(x, 'aagcn.TCNGCNUnit')
        y = self.gcn1(x)
        t = self.tcn1
        if self.res_mode == 2:
            r = self.residual
            rargs = (r.conv.weight, r.conv.bias) + _bn_args(r.bn)
        else:
            rargs = (None,) * 6
        out = ops.TCNResidualFunction.apply(y, x if self.res_mode else None, t.conv.weight, t.conv.bias,
                                            *_bn_args(t.bn), self.res_mode, *rargs, self.stride, self.training,
                                            ops.sync_of(t.bn))
        _bn_tick(t.bn, self.training)
        if self.res_mode == 2:
            _bn_tick(self.residual.bn, self.training)
        return out


class BaseModel(nn.Module):
    def __init__(self, num_class=60, num_point=25, num_person=2, in_channels=3, drop_out=0, adaptive=True,
                 gbn_split: Optional[int] = None, fc_cv=False, data_norm='bn'):
        super().__init__()
        if data_norm != 'bn':
            raise NotImplementedError("agcn_amd.aagcn: data_norm must be 'bn'")
        self.num_class, self.num_person, self.num_point = num_class, num_person, num_point
        self.graph = None
        self.adaptive_fn = AdaptiveGCN if adaptive else NonAdaptiveGCN
        self.data_norm = data_norm
        self.data_bn = batch_norm_1d(num_person * in_channels * num_point, gbn_split)
        bn_init(self.data_bn, 1)
        for k in range(1, 11):
            setattr(self, f'l{k}', None)
        self.fc = None
        self.fc_cv = fc_cv
        self.drop_out = nn.Dropout(drop_out) if drop_out else lambda x: x

    def init_graph(self, graph, graph_args):
        if graph is None:
            raise ValueError()
        self.graph = import_class(graph)(**graph_args)

    def init_empty_model_backbone(self):
        for k in range(1, 11):
            setattr(self, f'l{k}', lambda x: x)

    def init_model_backbone(self, model_layers, tcngcn_unit, output_channel=None):
        """Layer subsets of reference aagcn.py:403-474 (3, 6, 7, 10 and the 101-103 one-width stacks)."""
        self.init_empty_model_backbone()
        full = {1: (3, 64, 1, False), 2: (64, 64, 1, True), 3: (64, 64, 1, True), 4: (64, 64, 1, True),
                5: (64, 128, 2, True), 6: (128, 128, 1, True), 7: (128, 128, 1, True), 8: (128, 256, 2, True),
                9: (256, 256, 1, True), 10: (256, 256, 1, True)}
        subsets = {0: [], 3: [1, 5, 8], 6: [1, 4, 5, 7, 8, 10], 7: [1, 3, 4, 5, 7, 8, 10], 10: list(range(1, 11))}
        if model_layers in subsets:
            for k in subsets[model_layers]:
                ci, co, st, res = full[k]
                setattr(self, f'l{k}', tcngcn_unit(ci, co, stride=st, residual=res))
        elif model_layers in (101, 102, 103):
            c = output_channel if output_channel is not None else 64
            self.l1 = tcngcn_unit(3, c, residual=False)
            for k in range(2, model_layers - 100 + 1):
                setattr(self, f'l{k}', tcngcn_unit(c, c))
        else:
            raise ValueError(f"Model with {model_layers} layers is not supported.")

    def init_fc(self, in_channels, out_channels):
        self.fc = nn.Linear(in_channels, out_channels)
        nn.init.normal_(self.fc.weight, 0, math.sqrt(2. / out_channels))

    def forward_preprocess(self, x, size):
        _require_gpu(x, 'aagcn.Model')
        return data_bn_forward(self.data_bn, x)

    def forward_model_backbone(self, x, size):
        for k in range(1, 11):
            x = getattr(self, f'l{k}')(x)
        return x

    def forward_postprocess(self, x, size):
        N, C, T, V, M = size
        c_new = x.size(1)
        if self._fused_head():
            return x, None                    # pooled inside forward_classifier (one autograd node with the Linear)
        if self.fc_cv:
            x = x.view(N, M, c_new, -1, V).mean(3).mean(1).view(N, -1)
        else:
            x = x.view(N, M, c_new, -1).mean(3).mean(1)
        return x, None

    def _fused_head(self):
        """global average pool + Linear on the deterministic HIP kernels (ops.PoolFCFunction): the default head (no
        fc_cv, no dropout) on a backbone that ends in a unit."""
        return (not self.fc_cv) and not isinstance(self.drop_out, nn.Dropout) and isinstance(self.fc, nn.Linear)

    def forward_classifier(self, x, size):
        if self._fused_head() and x.dim() == 4:
            return ops.PoolFCFunction.apply(x, self.fc.weight, self.fc.bias, size[4])
        return self.fc(self.drop_out(x))

    def forward(self, x):
        size = x.size()
        x = self.forward_preprocess(x, size)
        x = self.forward_model_backbone(x, size)
        x, attn = self.forward_postprocess(x, size)
        return self.forward_classifier(x, size), attn


class Model(BaseModel):
    def __init__(self, num_class=60, num_point=25, num_person=2, num_subset=3, graph=None, graph_args=dict(),
                 in_channels=3, drop_out=0, adaptive=True, attention=True, gbn_split=None, fc_cv=False,
                 model_layers=10):
        super().__init__(num_class, num_point, num_person, in_channels, drop_out, adaptive, gbn_split, fc_cv)
        if graph is None:
            raise ValueError()
        self.graph = (import_class(graph) if isinstance(graph, str) else graph)(**graph_args)

        def _unit(in_c, out_c, stride=1, residual=True):
            return TCNGCNUnit(in_channels=in_c, out_channels=out_c, A=self.graph.A, num_subset=num_subset,
                              stride=stride, residual=residual, adaptive=self.adaptive_fn, attention=attention,
                              gbn_split=gbn_split)

        self.init_model_backbone(model_layers=model_layers, tcngcn_unit=_unit)
        self.init_fc(256 * num_point if fc_cv else 256, num_class)
