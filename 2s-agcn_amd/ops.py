"""Host-side operators of the AGCN hot path: thin wrappers over the C-ABI (``lib.py``) and the
``torch.autograd.Function``s that chain them into unit_gcn / unit_tcn / TCN_GCN_unit forward+backward.

Tensors are (N', C, T, V) contiguous fp32 on the GPU, allocated by PyTorch's caching allocator; the
extension never allocates or keeps device memory.  Everything here launches on the current stream.

Maths: SURVEY.md Appendix A (checked against the reference ``agcn.py:92-109`` by the oracle tests).
"""
import os
import weakref

import torch

from . import lib as _lib

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _L():
    return _lib.load()


def _empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _ws(nbytes, like):
    """Byte workspace (packed weight images etc.) from the caching allocator."""
    return torch.empty((int(nbytes) + 3) // 4, dtype=torch.float32, device=like.device)


def _conv_ws(Cin, Cout, T, V, taps, stride, like):
    n = _L().agcn_conv_workspace(Cin, Cout, T, V, taps, stride)
    return _ws(n, like), n


def _gcn_ws(C, Cout, T, V, like):
    n = _L().agcn_gcn_workspace(C, Cout, T, V)
    return _ws(n, like), n


def _scratch(width, like):
    """Scratch for the two-stage column reductions (agcn_colsum_scratch_bytes)."""
    nbytes = _L().agcn_colsum_scratch_bytes(int(width))
    return torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=like.device)


# ------------------------------------------------------------------------------------------------
# thin wrappers (one C entry point each)
# ------------------------------------------------------------------------------------------------

# ---- side stream for the weight gradients (AGCN_SIDE_STREAM=0 disables): they do not feed the backward's critical path
# (dx), so they run beside the backward-data kernels and the two fill the tails of each other's grids (+1 % on the
# training step, same-box A/B; results are bitwise the same: no kernel changes, only their placement) ----
_SIDE = {}


def side_stream_enabled():
    return os.environ.get('AGCN_SIDE_STREAM', '1') != '0'


_SIDE_SCOPE = [0]     # > 0 while a backward that joins once at its end is running (TCNGCNUnitFunction)


def _side_run(fn, inputs):
    """Run fn() on the side stream after everything enqueued so far on the current stream; returns its result.  The
    caller must _side_join() before the results are consumed on the current stream.  Only inside the fused unit's
    backward: the stand-alone unit_gcn / unit_tcn nodes (AAGCN) join right after their one weight-gradient kernel, which
    measured 1.2 % SLOWER than no side stream at all."""
    if not side_stream_enabled() or _SIDE_SCOPE[0] <= 0:
        return fn()
    main = torch.cuda.current_stream()
    dev = main.device_index
    side = _SIDE.get(dev)
    if side is None:
        side = _SIDE[dev] = torch.cuda.Stream(device=dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        out = fn()
    for t in inputs:
        if t is not None:
            t.record_stream(side)
    outs = out if isinstance(out, (tuple, list)) else (out,)
    for t in outs:
        if torch.is_tensor(t):
            t.record_stream(main)
    return out


def _side_join():
    if side_stream_enabled():
        main = torch.cuda.current_stream()
        side = _SIDE.get(main.device_index)
        if side is not None:
            main.wait_stream(side)


def fused_amax_enabled():
    """AGCN_FUSED_AMAX=0: the split-fp16 convolutions compute their operand's maximum in a pass of their own (A/B)."""
    return os.environ.get('AGCN_FUSED_AMAX', '1') != '0'


# max |out| of the last unit output a BatchNorm pass produced, for the f16x3 chain of the unit that consumes it:
# (weak reference to the tensor, its version counter, the device scalar).  Taken only by the very tensor object it
# describes, unmodified since; anything else makes the chain take the maximum with a pass of its own.
_OUT_AMAX = [None]
_OUT_AMAX_STATS = [0, 0]      # [misses, hits] of _take_out_amax (diagnostic)


def _note_out_amax(t, amax):
    _OUT_AMAX[0] = (weakref.ref(t), t._version, amax) if amax is not None else None


def _take_out_amax(t):
    e = _OUT_AMAX[0]
    hit = e is not None and e[0]() is t and t._version == e[1]
    _OUT_AMAX_STATS[int(hit)] += 1
    return e[2] if hit else None


def conv_out_frames(T, taps, stride):
    pad = (taps - 1) // 2
    return (T + 2 * pad - taps) // stride + 1


def conv_fwd(x, w, b, stride=1, want_stats=False, x_amax=None):
    """y = conv2d(x, w(k,1), b, stride=(s,1), padding=((k-1)/2,0)); optional per-channel (sum,sumsq) partials."""
    N, Cin, T, V = x.shape
    Cout, Cin2, taps, one = w.shape
    assert Cin2 == Cin and one == 1
    To = conv_out_frames(T, taps, stride)
    y = _empty((N, Cout, To, V), x)
    stats = None
    if want_stats:
        nt = _L().agcn_conv_stats_tiles(Cin, Cout, To, V, taps, stride)
        stats = _empty((N * nt, 2, Cout), x)
    ws, nb = _conv_ws(Cin, Cout, T, V, taps, stride, x)
    # x_amax: 1-element tensor with max |x| left by bn_act_fwd(..., want_amax=True): the split-fp16 temporal convolution
    # then skips its own pass over x
    _lib.check(_L().agcn_conv_fwd_ex(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), _lib.ptr(stats), ws.data_ptr(),
                                     nb, N, Cin, Cout, T, V, taps, stride, _lib.ptr(x_amax), _lib.stream()),
               "agcn_conv_fwd")
    return y, stats


def conv_bwd_data(dy, w, x_shape, stride=1, out=None, accumulate=False, add1=None, mask1=None, add2=None,
                  mask2=None, dy_amax=None):
    N, Cin, T, V = x_shape
    Cout, _, taps, _ = w.shape
    dx = out if out is not None else _empty(x_shape, dy)
    ws, nb = _conv_ws(Cin, Cout, T, V, taps, stride, dy)
    _lib.check(_L().agcn_conv_bwd_data_ex(_lib.ptr(dy), _lib.ptr(w), _lib.ptr(dx), int(accumulate), _lib.ptr(add1),
                                          _lib.ptr(mask1), _lib.ptr(add2), _lib.ptr(mask2), ws.data_ptr(), nb, N, Cin,
                                          Cout, T, V, taps, stride, _lib.ptr(dy_amax), _lib.stream()),
               "agcn_conv_bwd_data")
    return dx


def conv_bwd_weight(dy, x, w_shape, stride=1, dy_amax=None, x_amax=None):
    """dy_amax / x_amax: device scalars max |dy| / max |x| left behind by their producers; with both, the tap-free
    gradients run on f16x3 (agcn_conv_bwd_weight_ex)."""
    N, Cin, T, V = x.shape
    Cout, _, taps, _ = w_shape
    nbytes = _L().agcn_conv_bwd_weight_workspace(N, Cin, Cout, T, V, taps, stride)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    dw = _empty(tuple(w_shape), x)
    _lib.check(_L().agcn_conv_bwd_weight_ex(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw), _lib.ptr(ws), nbytes, N, Cin, Cout,
                                            T, V, taps, stride, _lib.ptr(dy_amax), _lib.ptr(x_amax), _lib.stream()),
               "agcn_conv_bwd_weight")
    return dw


def adjacency_fwd(tp, A, PA, alpha=None):
    """tp: (N, 6Ci, T, V) theta/phi; returns P (softmax) and adj = [alpha*]P + A + PA, both (N,3,V,V)."""
    N, C6, T, V = tp.shape
    Ci = C6 // 6
    nt = _L().agcn_scores_num_tiles(V, T)
    spart = _empty((N, 3, nt, V, V), tp)
    P = _empty((N, 3, V, V), tp)
    adj = _empty((N, 3, V, V), tp)
    _lib.check(_L().agcn_adjacency_fwd(_lib.ptr(tp), _lib.ptr(A), _lib.ptr(PA), _lib.ptr(alpha), _lib.ptr(spart),
                                       _lib.ptr(P), _lib.ptr(adj), N, Ci, T, V, _lib.stream()), "agcn_adjacency_fwd")
    return P, adj


def adjacency_fused_supported(C, Ci, T, V):
    return bool(_L().agcn_adjacency_fused_supported(int(C), int(Ci), int(T), int(V)))


def adjacency_recompute():
    """Backward policy of the adaptive branch: recompute theta/phi on chip from x (AGCN_ADJ_RECOMPUTE=1: nothing of
    size 6*Ci*T*V is kept per layer) or re-read the copy the fused forward leaves behind as a by-product (default:
    measured faster on MI355X)."""
    import os
    return os.environ.get('AGCN_ADJ_RECOMPUTE', '0') == '1'


def adjacency_fused_fwd(x, wab, bab, A, PA, alpha=None, keep_tp=False, x_amax_out=None, x_amax=None):
    """P, adj straight from x: [theta;phi] = wab.x + bab is formed and reduced on chip (no tp round trip); wab
    (6Ci, C[,1,1]).  keep_tp: also return theta/phi, written once as a by-product (never re-read by the forward).
    x_amax_out: optional 1-element tensor that receives max |x| (the pass reads all of x anyway); x_amax: the same
    scalar where the producer of x already took it."""
    N, C, T, V = x.shape
    Ci = wab.shape[0] // 6
    tp = _empty((N, 6 * Ci, T, V), x) if keep_tp else None
    nt = _L().agcn_scores_num_tiles(V, T)
    spart = _empty((N, 3, nt, V, V), x)
    P = _empty((N, 3, V, V), x)
    adj = _empty((N, 3, V, V), x)
    nb = _L().agcn_adjacency_fused_workspace(C, Ci)
    ws = _ws(nb, x)
    _lib.check(_L().agcn_adjacency_fused_fwd_ex(_lib.ptr(x), _lib.ptr(wab.reshape(6 * Ci, C)), _lib.ptr(bab), _lib.ptr(A),
                                                _lib.ptr(PA), _lib.ptr(alpha), _lib.ptr(tp), _lib.ptr(spart), _lib.ptr(P),
                                                _lib.ptr(adj), _lib.ptr(x_amax_out), _lib.ptr(x_amax), ws.data_ptr(), nb, N,
                                                C, Ci, T, V, _lib.stream()), "agcn_adjacency_fused_fwd")
    return (P, adj, tp) if keep_tp else (P, adj)


def first_layer_enabled():
    """AGCN_FIRST_LAYER=0 keeps the 3-channel layer on the generic kernels (A/B)."""
    return os.environ.get('AGCN_FIRST_LAYER', '1') != '0'


def gcn_first_fwd(x, adj, wcat, bias, wdown, bdown, want_stats=False):
    """First-layer unit_gcn forward (C <= 4): ypre = bias + sum_i Wd_i (x . adj_i) and dpre = bdown + Wdown x with
    their BatchNorm partials, one pass over x."""
    N, C, T, V = x.shape
    Cout = wcat.shape[0]
    ypre, dpre = _empty((N, Cout, T, V), x), _empty((N, Cout, T, V), x)
    st = st2 = None
    if want_stats:
        nt = _L().agcn_gcn_first_tiles(T, V)
        st, st2 = _empty((N * nt, 2, Cout), x), _empty((N * nt, 2, Cout), x)
    wdown2 = wdown.reshape(Cout, C).contiguous()
    _lib.check(_L().agcn_gcn_first_fwd(_lib.ptr(x), _lib.ptr(adj), _lib.ptr(wcat), _lib.ptr(bias), _lib.ptr(wdown2),
                                       _lib.ptr(bdown), _lib.ptr(ypre), _lib.ptr(st), _lib.ptr(dpre), _lib.ptr(st2),
                                       N, C, Cout, T, V, _lib.stream()), "agcn_gcn_first_fwd")
    return ypre, st, dpre, st2


def aggregate_project_fwd(x, adj, wcat, bias, want_stats=False, x_amax=None):
    """y = sum_i Wd_i (x . adj_i) + bias ; wcat: (Cout, 3C) = [Wd_0 | Wd_1 | Wd_2]."""
    N, C, T, V = x.shape
    Cout = wcat.shape[0]
    y = _empty((N, Cout, T, V), x)
    stats = None
    if want_stats:
        stats = _empty((_L().agcn_gcn_stats_slots(N, C, Cout, T, V), 2, Cout), x)
    ws, nb = _gcn_ws(C, Cout, T, V, x)
    _lib.check(_L().agcn_gcn_aggregate_project_fwd_ex(_lib.ptr(x), _lib.ptr(adj), _lib.ptr(wcat), _lib.ptr(bias),
                                                      _lib.ptr(y), _lib.ptr(stats), ws.data_ptr(), nb, N, C, Cout, T, V,
                                                      _lib.ptr(x_amax), _lib.stream()),
               "agcn_gcn_aggregate_project_fwd")
    return y, stats


def aggregate_project_bwd_data(dy, adj, wcat, x_shape, out=None, accumulate=False, add1=None, mask1=None,
                               add2=None, mask2=None, dtp=None, wab=None, dy_amax=None, dtp_amax=None):
    """dx (+)= sum_i Wd_i^T (dy . adj_i^T) + add1*[mask1] + add2*[mask2] [+ wab^T dtp].  The masks are fp32 tensors
    (> 0 passes) or, both of them, int32 sign bit masks (bn_act_fwd(..., want_bits=True)).  dtp/wab: the 1x1 term of
    the adaptive branch fused into the same pass (only where fused_bwd_data_supported says so)."""
    N, C, T, V = x_shape
    Cout = wcat.shape[0]
    dx = out if out is not None else _empty(x_shape, dy)
    ws, nb = _gcn_ws(C, Cout, T, V, dy)
    kinds = {m.dtype for m in (mask1, mask2) if m is not None}
    if len(kinds) > 1:
        raise RuntimeError("agcn_amd: mask1 and mask2 must be of one kind (fp32 tensors or int32 sign bit masks)")
    mbits = int(torch.int32 in kinds)
    mp = _lib.ptr_bits if mbits else _lib.ptr
    if dtp is not None:
        K2 = dtp.shape[1]
        _lib.check(_L().agcn_gcn_aggregate_project_bwd_data_ex(
            _lib.ptr(dy), _lib.ptr(adj), _lib.ptr(wcat), _lib.ptr(dtp), _lib.ptr(wab.reshape(K2, C)), K2, _lib.ptr(dx),
            int(accumulate), _lib.ptr(add1), mp(mask1), _lib.ptr(add2), mp(mask2), mbits, ws.data_ptr(), nb, N, C, Cout,
            T, V, _lib.ptr(dy_amax), _lib.ptr(dtp_amax), _lib.stream()), "agcn_gcn_aggregate_project_bwd_data_fused")
        return dx
    _lib.check(_L().agcn_gcn_aggregate_project_bwd_data_ex(
        _lib.ptr(dy), _lib.ptr(adj), _lib.ptr(wcat), None, None, 0, _lib.ptr(dx), int(accumulate), _lib.ptr(add1),
        mp(mask1), _lib.ptr(add2), mp(mask2), mbits, ws.data_ptr(), nb, N, C, Cout, T, V, _lib.ptr(dy_amax), None,
        _lib.stream()), "agcn_gcn_aggregate_project_bwd_data")
    return dx


def fused_bwd_data_supported(C, Cout, V):
    return bool(_L().agcn_gcn_bwd_data_fused_supported(int(C), int(Cout), int(V)))


def project_bwd_weight(dy, x, adj, Cout, dy_amax=None, x_amax=None):
    N, C, T, V = x.shape
    nbytes = _L().agcn_gcn_project_bwd_weight_workspace(N, C, Cout, T, V)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    dw = _empty((Cout, 3 * C), x)
    _lib.check(_L().agcn_gcn_project_bwd_weight_ex(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(adj), _lib.ptr(dw), _lib.ptr(ws),
                                                   nbytes, N, C, Cout, T, V, _lib.ptr(dy_amax), _lib.ptr(x_amax),
                                                   _lib.stream()),
               "agcn_gcn_project_bwd_weight")
    return dw


def adjacency_bwd(dy, wcat, x, tp, P, alpha=None, wab=None, bab=None, dy_amax=None, x_amax=None):
    """Gradient of the adaptive adjacency branch.  Returns dPA (3,V,V), dtp (N,6Ci,T,V), dbab (6Ci), dalpha, dadj and
    the device scalar max |dtp| its producer left behind (None where it does not: the consumers take it themselves).
    tp = None: theta/phi were never stored (adjacency_fused_fwd); they are recomputed from x, wab, bab on chip."""
    N, C, T, V = x.shape
    Cout = wcat.shape[0]
    Ci = (tp.shape[1] if tp is not None else wab.shape[0]) // 6
    nslots = _L().agcn_dadj_num_slots(C, V, T)
    dpart = _empty((N, 3, nslots, V, V), x)
    ws, nb = _gcn_ws(C, Cout, T, V, x)
    _lib.check(_L().agcn_gcn_dadj_ex(_lib.ptr(dy), _lib.ptr(wcat), _lib.ptr(x), _lib.ptr(dpart), ws.data_ptr(), nb, N, C,
                                     Cout, T, V, _lib.ptr(dy_amax), _lib.ptr(x_amax), _lib.stream()), "agcn_gcn_dadj")
    dadj = _empty((N, 3, V, V), x)
    dS = _empty((N, 3, V, V), x)
    dPA = _empty((3, V, V), x)
    dal_part = _empty((N * 3,), x) if alpha is not None else None
    _lib.check(_L().agcn_adjacency_bwd_softmax(_lib.ptr(dpart), _lib.ptr(P), _lib.ptr(alpha), _lib.ptr(dadj),
                                               _lib.ptr(dS), _lib.ptr(dPA), _lib.ptr(dal_part), N, Ci, T, V, nslots,
                                               _lib.stream()), "agcn_adjacency_bwd_softmax")
    nt = _L().agcn_scores_num_tiles(V, T)
    dtp = _empty((N, 6 * Ci, T, V), x)
    dbpart = _empty((N * nt, 6 * Ci), x)
    dbab = _empty((6 * Ci,), x)
    scratch = _scratch(6 * Ci, x)
    dtp_amax = None
    if tp is None:
        nb = _L().agcn_adjacency_fused_workspace(C, Ci)
        ws = _ws(nb, x)
        _lib.check(_L().agcn_adjacency_fused_bwd_scores(
            _lib.ptr(x), _lib.ptr(wab.reshape(6 * Ci, C)), _lib.ptr(bab), _lib.ptr(dS), _lib.ptr(dtp), _lib.ptr(dbpart),
            scratch.data_ptr(), _lib.ptr(dbab), ws.data_ptr(), nb, N, C, Ci, T, V, _lib.stream()),
            "agcn_adjacency_fused_bwd_scores")
    else:
        dtp_amax = _empty((1,), x) if fused_amax_enabled() else None
        _lib.check(_L().agcn_adjacency_bwd_scores_ex(_lib.ptr(tp), _lib.ptr(dS), _lib.ptr(dtp), _lib.ptr(dbpart),
                                                     scratch.data_ptr(), _lib.ptr(dbab), _lib.ptr(dtp_amax), N, Ci, T, V,
                                                     _lib.stream()),
                   "agcn_adjacency_bwd_scores")
    dalpha = dal_part.sum() if dal_part is not None else None
    return dPA, dtp, dbab, dalpha, dadj, dtp_amax


def stc_row_reduce(y, g=None, wv=None, wt=None, want_t=False, want_v=False, scale_t=1.0, scale_v=1.0):
    """One pass over y (N,C,T,V) [times g elementwise]: out_t (N,C,T) = scale_t * sum_v wv[n,v]*y*g and / or
    out_v (N,C,V) = scale_v * sum_t wt[...,t]*y*g ; wv (N,V), wt (N,T) or (N,C,T); None weights = ones."""
    N, C, T, V = y.shape
    out_t = _empty((N, C, T), y) if want_t else None
    out_v = _empty((N, C, V), y) if want_v else None
    per_row = int(wt is not None and wt.dim() == 3)
    _lib.check(_L().agcn_stc_row_reduce(_lib.ptr(y), _lib.ptr(g), _lib.ptr(wv), _lib.ptr(wt), per_row, _lib.ptr(out_t),
                                        _lib.ptr(out_v), float(scale_t), float(scale_v), N, C, T, V, _lib.stream()),
               "agcn_stc_row_reduce")
    return out_t, out_v


# ---- the small operators around the unit stack (csrc/small_ops.hip): deterministic, no vendor library --------------
def linear_fwd(x, w, b, act=0):
    """out = act(x @ w.T + b); act 0 identity, 1 ReLU, 2 "1 + sigmoid".  x (N, K), w (O, K)."""
    N, K = x.shape
    O = w.shape[0]
    out = _empty((N, O), x)
    _lib.check(_L().agcn_linear_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), N, K, O, int(act),
                                    _lib.stream()), "agcn_linear_fwd")
    return out


def linear_bwd(dout, out, x, w, act=0, need_din=True):
    """Returns din (or None), dw, db of linear_fwd."""
    N, K = x.shape
    O = w.shape[0]
    dpre = _empty((N, O), x)
    din = _empty((N, K), x) if need_din else None
    dw, db = _empty((O, K), x), _empty((O,), x)
    _lib.check(_L().agcn_linear_bwd(_lib.ptr(dout), _lib.ptr(out), _lib.ptr(x), _lib.ptr(w), _lib.ptr(dpre),
                                    _lib.ptr(din), _lib.ptr(dw), _lib.ptr(db), N, K, O, int(act), _lib.stream()),
               "agcn_linear_bwd")
    return din, dw, db


def gate_conv_fwd(x, w, b):
    """a (N, L) = 1 + sigmoid(Conv1d(C -> 1, Ks, padding (Ks-1)/2)(x)); x (N, C, L), w (1, C, Ks), b (1)."""
    N, C, L = x.shape
    Ks = w.shape[-1]
    a = _empty((N, L), x)
    _lib.check(_L().agcn_gate_conv_fwd(_lib.ptr(x), _lib.ptr(w.reshape(C, Ks)), _lib.ptr(b), _lib.ptr(a), N, C, L, Ks,
                                       _lib.stream()), "agcn_gate_conv_fwd")
    return a


def gate_conv_bwd(da, a, x, w):
    """Returns dx (N, C, L), dw (1, C, Ks), db (1) of gate_conv_fwd given da = dL/da."""
    N, C, L = x.shape
    Ks = w.shape[-1]
    dpre = _empty((N, L), x)
    dx, dw, db = _empty((N, C, L), x), _empty((1, C, Ks), x), _empty((1,), x)
    _lib.check(_L().agcn_gate_conv_bwd(_lib.ptr(da), _lib.ptr(a), _lib.ptr(x), _lib.ptr(w.reshape(C, Ks)),
                                       _lib.ptr(dpre), _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), N, C, L, Ks,
                                       _lib.stream()), "agcn_gate_conv_bwd")
    return dx, dw, db


class DataBNFunction(torch.autograd.Function):
    """The model prologue (reference agcn.py:163-165): x (N, C, T, V, M) -> permute/view (N, M*V*C, T) -> BatchNorm1d ->
    view/permute -> (N*M, C, T, V), as three small deterministic kernels (statistics, finalize, apply).
    args: x, weight, bias, running_mean, running_var, training, sync"""

    @staticmethod
    def forward(ctx, x, w, b, rm, rv, training, sync=None):
        x = x.contiguous()
        N, C, T, V, M = x.shape
        CH = C * V * M
        L = _L()
        if training:
            part = _empty((N, 2, CH), x)
            _lib.check(L.agcn_data_bn_stats(_lib.ptr(x), _lib.ptr(part), N, C, T, V, M, _lib.stream()),
                       "agcn_data_bn_stats")
            (st,), gcount = _bn_coeffs(True, [part], N * T, [(w, b, rm, rv)], sync, N)
        else:
            (st,), gcount = _bn_coeffs(False, [None], N * T, [(w, b, rm, rv)], None, N)
        out = _empty((N * M, C, T, V), x)
        _lib.check(L.agcn_data_bn_apply(_lib.ptr(x), _lib.ptr(st.scale), _lib.ptr(st.shift), _lib.ptr(out), N, C, T, V, M,
                                        _lib.stream()), "agcn_data_bn_apply")
        ctx.st, ctx.sync, ctx.gcount, ctx.training = st, sync, gcount, training
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, dy):
        _need_train(ctx.training)
        x, w = ctx.saved_tensors
        st, sync = ctx.st, ctx.sync
        dy = dy.contiguous()
        N, C, T, V, M = x.shape
        CH = C * V * M
        L = _L()
        part = _empty((N, 2, CH), x)
        _lib.check(L.agcn_data_bn_bwd_reduce(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(st.mean), _lib.ptr(st.invstd),
                                             _lib.ptr(part), N, C, T, V, M, _lib.stream()), "agcn_data_bn_bwd_reduce")
        sums = _colsum(part, N, 2 * CH)
        scale = 1.0
        if sync is not None:
            sums = _allreduce_sum(sums, sync)
            scale = 1.0 / sync.world       # global sums, restored by the gradient average of the data-parallel step
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(L.agcn_data_bn_bwd_apply(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(w), _lib.ptr(st.mean),
                                                _lib.ptr(st.invstd), _lib.ptr(sums), float(ctx.gcount), _lib.ptr(dx), N, C,
                                                T, V, M, _lib.stream()), "agcn_data_bn_bwd_apply")
        dbeta, dgamma = sums[:CH], sums[CH:]
        if scale != 1.0:
            dbeta, dgamma = dbeta * scale, dgamma * scale
        return dx, dgamma, dbeta, None, None, None, None


def data_bn_supported(bn):
    """Plain / synchronised BatchNorm1d only (GhostBatchNorm1d keeps its own module path)."""
    return type(bn) in (torch.nn.BatchNorm1d, torch.nn.SyncBatchNorm)


class PoolFCFunction(torch.autograd.Function):
    """The model epilogue (reference agcn.py:179-183): mean over (T, V) per person, mean over persons, Linear.
    args: x (N*M, C, T, V), fc weight (K, C), fc bias (K), M"""

    @staticmethod
    def forward(ctx, x, w, b, M):
        x = x.contiguous()
        NM, C, T, V = x.shape
        N = NM // M
        rowmean, pooled = _empty((NM, C), x), _empty((N, C), x)
        _lib.check(_L().agcn_pool_fwd(_lib.ptr(x), _lib.ptr(rowmean), _lib.ptr(pooled), N, M, C, T * V, _lib.stream()),
                   "agcn_pool_fwd")
        logits = linear_fwd(pooled, w, b, 0)
        ctx.save_for_backward(pooled, logits, w)
        ctx.shape, ctx.M = (NM, C, T, V), M
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        pooled, logits, w = ctx.saved_tensors
        NM, C, T, V = ctx.shape
        M = ctx.M
        dpooled, dw, db = linear_bwd(dlogits.contiguous(), logits, pooled, w, 0, need_din=True)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _empty((NM, C, T, V), pooled)
            _lib.check(_L().agcn_pool_bwd(_lib.ptr(dpooled), _lib.ptr(dx), NM // M, M, C, T * V, _lib.stream()),
                       "agcn_pool_bwd")
        return dx, dw, db, None


class STCAttentionFunction(torch.autograd.Function):
    """AAGCN's three attention gates (reference aagcn.py:59-116 applied at :268-270) as ONE autograd node:
        y3 = y * (1+se_s[n,v]) * (1+se_t[n,t]) * (1+se_c[n,c])
    Full-tensor work = HIP passes (forward: two reductions + one apply = 3 reads, 1 write of the activation; backward:
    one pass over (dout, y), one over y, one apply = 4 reads, 1 write); the gate networks (Conv1d C->1 over joints /
    frames, two Linears) act on (N,C,V) / (N,C,T) / (N,C) tensors and run as ordinary tensor code, differentiated by
    explicit small kernels too (gate_conv_* / linear_* above).
    args: y, sa_w (1,C,Ks), sa_b (1), ta_w (1,C,9), ta_b (1), fc1_w, fc1_b, fc2_w, fc2_b"""

    @staticmethod
    def _gates(m_s, mv1, sa_w, sa_b, ta_w, ta_b, f1w, f1b, f2w, f2b, a_s=None):
        """a_s (N,V) from mean_t y; a_t (N,T) from mean_v y1; a_c (N,C) from mean_t(a_t * mean_v y1).  The gate
        networks are the deterministic kernels of csrc/small_ops.hip (gate convolution, small Linear).
        Returns (a_s, a_t, a_c, m_c, h) with m_c / h the channel gate's input and hidden activations."""
        if a_s is None:
            a_s = gate_conv_fwd(m_s, sa_w, sa_b)
        if mv1 is None:
            return a_s, None, None, None, None
        a_t = gate_conv_fwd(mv1, ta_w, ta_b)
        m_c = (mv1 * a_t.unsqueeze(1)).mean(-1)
        h = linear_fwd(m_c, f1w, f1b, 1)
        a_c = linear_fwd(h, f2w, f2b, 2)
        return a_s, a_t, a_c, m_c, h

    @staticmethod
    def forward(ctx, y, sa_w, sa_b, ta_w, ta_b, f1w, f1b, f2w, f2b):
        y = y.contiguous()
        N, C, T, V = y.shape
        par = (sa_w, sa_b, ta_w, ta_b, f1w, f1b, f2w, f2b)
        _, m_s = stc_row_reduce(y, want_v=True, scale_v=1.0 / T)                         # mean_t y
        a_s = STCAttentionFunction._gates(m_s, None, *par)[0]
        mv1, _ = stc_row_reduce(y, wv=a_s, want_t=True, scale_t=1.0 / V)                 # mean_v y*(1+se_s)
        _, a_t, a_c, m_c, h = STCAttentionFunction._gates(m_s, mv1, *par, a_s=a_s)
        out = torch.empty_like(y)
        # the gated tensor feeds the f16x3 temporal convolution (and its weight gradient): leave its maximum behind
        o_amax = _empty((1,), y) if fused_amax_enabled() else None
        _lib.check(_L().agcn_stc_apply_ex(_lib.ptr(y), _lib.ptr(a_s), _lib.ptr(a_t), _lib.ptr(a_c), _lib.ptr(out),
                                          _lib.ptr(o_amax), N, C, T, V, _lib.stream()), "agcn_stc_apply")
        _note_out_amax(out, o_amax)
        ctx.save_for_backward(y, m_s, mv1, a_s, a_t, a_c, m_c, h, *par)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, m_s, mv1, a_s, a_t, a_c, m_c, h, *par = ctx.saved_tensors
        sa_w, sa_b, ta_w, ta_b, f1w, f1b, f2w, f2b = par
        dout = dout.contiguous()
        N, C, T, V = y.shape
        # one pass over (dout, y): P1[n,c,t] = sum_v a_s dout*y ; P2[n,c,v] = sum_t a_t dout*y
        P1, P2 = stc_row_reduce(y, g=dout, wv=a_s, wt=a_t, want_t=True, want_v=True)
        da_c = (P1 * a_t.unsqueeze(1)).sum(-1).contiguous()
        da_t = (P1 * a_c.unsqueeze(-1)).sum(1)
        da_s = (P2 * a_c.unsqueeze(-1)).sum(1)
        # channel gate: a_c = 1 + sigmoid(fc2(relu(fc1(m_c)))), m_c = mean_t(a_t * mv1)
        dh, df2w, df2b = linear_bwd(da_c, a_c, h, f2w, 2)
        dm_c, df1w, df1b = linear_bwd(dh, h, m_c, f1w, 1)
        da_t = (da_t + (dm_c.unsqueeze(-1) * mv1).sum(1) / T).contiguous()
        dmv1 = dm_c.unsqueeze(-1) * a_t.unsqueeze(1) / T
        # temporal gate: a_t = 1 + sigmoid(conv_9(mv1))
        dmv1_t, dta_w, dta_b = gate_conv_bwd(da_t, a_t, mv1, ta_w)
        dmv1 = ((dmv1 + dmv1_t) / V).contiguous()       # mv1 = (1/V) sum_v y*a_s
        # its two other consumers: y (folded into the apply pass below) and a_s
        _, R = stc_row_reduce(y, wt=dmv1, want_v=True)
        da_s = (da_s + R.sum(1)).contiguous()
        # spatial gate: a_s = 1 + sigmoid(conv_V(m_s)), m_s = (1/T) sum_t y
        dm_s, dsa_w, dsa_b = gate_conv_bwd(da_s, a_s, m_s, sa_w)
        dms = (dm_s / T).contiguous()
        dy = torch.empty_like(y)
        _lib.check(_L().agcn_stc_bwd_apply(_lib.ptr(dout), _lib.ptr(a_s), _lib.ptr(a_t), _lib.ptr(a_c), _lib.ptr(dmv1),
                                           _lib.ptr(dms), _lib.ptr(dy), N, C, T, V, _lib.stream()),
                   "agcn_stc_bwd_apply")
        return dy, dsa_w, dsa_b, dta_w, dta_b, df1w, df1b, df2w, df2b


class BNState:
    """Per-BatchNorm forward products kept for the backward.  ``S`` > 1: GhostBatchNorm (reference
    model/layers/module/ghostbatchnorm.py:77-120) -- the statistics are per virtual sub-batch s = n % S, which is an
    ordinary BatchNorm over (N/S, S*C, T, V): the HIP stages are simply called with that shape (channel index of a row
    = (n*C + c) % (S*C) = (n % S)*C + c), the coefficient vectors hold S*C entries and the shared weight/bias are
    repeated S times; nothing in the kernels changes."""
    __slots__ = ("mean", "invstd", "scale", "shift", "S")

    def __init__(self):
        self.S = 1


class SyncBN:
    """BatchNorm statistics policy of ONE unit's HIP BatchNorm stages, carried by the calling module and stored on the
    autograd context (no process-global state: two models with different policies can live in one process).
    ``None`` = per replica (reference nn.DataParallel semantics, utils/processor.py:336-343).  ``SyncBN(world, group)``
    = synchronised over the process group like the reference's DDP path (SyncBatchNorm.convert_sync_batchnorm,
    processor.py:295): the per-channel sums are all-reduced between the two stages of the forward statistics (ONE
    collective for the main and the down/residual BatchNorm together) and of the backward (one collective)."""
    __slots__ = ("world", "group")

    def __init__(self, world, group=None):
        self.world, self.group = int(world), group


def sync_of(bn_module):
    """SyncBN policy implied by a BatchNorm module: the reference converts the model with
    ``SyncBatchNorm.convert_sync_batchnorm`` before DDP (processor.py:295); the HIP units only borrow the BN modules'
    parameters, so the conversion is honoured by looking at the module's class."""
    import torch.distributed as dist
    if isinstance(bn_module, torch.nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized():
        world = dist.get_world_size(bn_module.process_group)
        if world > 1:
            return SyncBN(world, bn_module.process_group)
    return None


def _colsum(slab, nslots, width):
    out = _empty((width,), slab)
    scratch = _scratch(width, slab)
    _lib.check(_L().agcn_colsum(_lib.ptr(slab), int(nslots), int(width), scratch.data_ptr(), _lib.ptr(out),
                                _lib.stream()), "agcn_colsum")
    return out


COLLECTIVES = {'bn': 0, 'grad': 0}      # diagnostic: collectives issued so far (bench.py reports them per step)


def _allreduce_sum(t, sync):
    import torch.distributed as dist
    COLLECTIVES['bn'] += 1
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=sync.group)
    return t


def sync_stats(stats_parts, count, sync):
    """Global (sum, sumsq) of several BatchNorm stages of one unit with ONE all-reduce: every slab is reduced over its
    slots locally, the per-channel sums travel in one buffer, and each stage gets back a 1-slot slab.  Every rank must
    hold the same number of elements per channel (equal per-rank batches: the trainer's sampler pads like torch's
    DistributedSampler), so the global count is count * world.  Returns [slab (1,2,C) ...]."""
    sums = [_colsum(sp, sp.shape[0], sp.shape[1] * sp.shape[2]) for sp in stats_parts]
    buf = torch.cat(sums) if len(sums) > 1 else sums[0]
    _allreduce_sum(buf, sync)
    out, o = [], 0
    for sp in stats_parts:
        w = sp.shape[1] * sp.shape[2]
        out.append(buf[o:o + w].view(1, 2, sp.shape[2]))
        o += w
    return out


def bn_train_coeffs(stats_part, count, gamma, beta, running_mean, running_var, momentum=BN_MOMENTUM, eps=BN_EPS):
    """stats_part: (slots, 2, C) partial (sum, sumsq); count: elements per channel behind them (python number)."""
    C = gamma.numel()
    st = BNState()
    st.mean, st.invstd = _empty((C,), gamma), _empty((C,), gamma)
    st.scale, st.shift = _empty((C,), gamma), _empty((C,), gamma)
    nslots = stats_part.shape[0]
    scratch = _scratch(2 * C, gamma)
    _lib.check(_L().agcn_bn_stats_finalize(_lib.ptr(stats_part), nslots, C, float(count), _lib.ptr(gamma),
                                           _lib.ptr(beta), _lib.ptr(running_mean), _lib.ptr(running_var),
                                           float(momentum), float(eps), scratch.data_ptr(), _lib.ptr(st.mean),
                                           _lib.ptr(st.invstd),
                                           _lib.ptr(st.scale), _lib.ptr(st.shift), _lib.stream()),
               "agcn_bn_stats_finalize")
    return st


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps=BN_EPS):
    C = gamma.numel()
    st = BNState()
    st.mean = st.invstd = None
    st.scale, st.shift = _empty((C,), gamma), _empty((C,), gamma)
    _lib.check(_L().agcn_bn_eval_coeff(_lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean),
                                       _lib.ptr(running_var), float(eps), C, _lib.ptr(st.scale), _lib.ptr(st.shift),
                                       _lib.stream()), "agcn_bn_eval_coeff")
    return st


def bn_act_fwd(y1, st1, r=None, st2=None, relu=True, want_bits=False, amax_out=None):
    """out = act(scale1*y1 + shift1 + res); res = 0 (r None) | r (st2 None) | scale2*r + shift2.
    want_bits: also return the sign bit mask of ``out`` (int32 words, 32 elements each) for the BatchNorm backward."""
    N, C, T, V = y1.shape
    if st1.S > 1:                      # GhostBatchNorm: (N, C) rows regrouped as (N/S, S*C)
        N, C = N // st1.S, C * st1.S
    out = torch.empty_like(y1)
    bits = torch.empty((y1.numel() + 31) // 32, dtype=torch.int32, device=y1.device) if want_bits else None
    mode = 0 if r is None else (1 if st2 is None else 2)
    # amax_out: optional 1-element tensor that receives max |out| (for the split-fp16 kernels that read `out` next)
    _lib.check(_L().agcn_bn_act_fwd_ex(_lib.ptr(y1), _lib.ptr(st1.scale), _lib.ptr(st1.shift), _lib.ptr(r),
                                       _lib.ptr(st2.scale) if st2 else None, _lib.ptr(st2.shift) if st2 else None,
                                       _lib.ptr(out), _lib.ptr_bits(bits), _lib.ptr(amax_out), N, C, T * V, mode,
                                       int(relu), _lib.stream()),
               "agcn_bn_act_fwd")
    return (out, bits) if want_bits else out


def bn_bwd(dout, mask, y1, gamma1, st1, y2=None, gamma2=None, st2=None, sync=None, gcount=None, amax_out=None):
    """Backward through out = relu(bn1(y1) [+ bn2(y2)] [+ identity]) in train mode.  ``mask``: the fp32 output tensor
    (positive elements pass) or the int32 sign bit mask of ``bn_act_fwd(..., want_bits=True)``; None = no ReLU.
    Returns dy1, dgamma1, dbeta1, dy2, dgamma2, dbeta2 (branch-2 entries None without y2)."""
    N, C, T, V = y1.shape
    S = st1.S
    if S > 1:                          # GhostBatchNorm: see BNState; gamma repeated S times, dgamma/dbeta folded below
        N, C = N // S, C * S
        gamma1 = gamma1.repeat(S)
        gamma2 = gamma2.repeat(S) if gamma2 is not None else None
    part = _empty((N * C * 3,), y1)
    coef = _empty((6 * C,), y1)
    dy1 = torch.empty_like(y1)
    dg1, db1 = _empty((C,), y1), _empty((C,), y1)
    dy2 = dg2 = db2 = None
    if y2 is not None:
        dy2 = torch.empty_like(y2)
        dg2, db2 = _empty((C,), y1), _empty((C,), y1)
    mbits = int(mask is not None and mask.dtype == torch.int32)
    mptr = _lib.ptr_bits(mask) if mbits else _lib.ptr(mask)
    if sync is None and amax_out is not None:
        # (amax_out: 1-element tensor that receives max |dy1| for the split-fp16 kernels that read dy1 next)
        _lib.check(_L().agcn_bn_bwd_reduce(_lib.ptr(dout), mptr, mbits, _lib.ptr(y1), _lib.ptr(y2),
                                           _lib.ptr(part), N, C, T * V, _lib.stream()), "agcn_bn_bwd_reduce")
        _lib.check(_L().agcn_bn_bwd_apply_ex(
            _lib.ptr(part), N, float(N) * float(T * V), 1.0, _lib.ptr(dout), mptr, mbits, _lib.ptr(y1),
            _lib.ptr(gamma1), _lib.ptr(st1.mean), _lib.ptr(st1.invstd), _lib.ptr(y2), _lib.ptr(gamma2),
            _lib.ptr(st2.mean) if st2 else None, _lib.ptr(st2.invstd) if st2 else None, _lib.ptr(coef), _lib.ptr(dy1),
            _lib.ptr(dg1), _lib.ptr(db1), _lib.ptr(dy2), _lib.ptr(dg2), _lib.ptr(db2), _lib.ptr(amax_out), N, C, T * V,
            _lib.stream()), "agcn_bn_bwd_apply")
    elif sync is None:
        _lib.check(_L().agcn_bn_bwd(
            _lib.ptr(dout), mptr, mbits, _lib.ptr(y1), _lib.ptr(gamma1), _lib.ptr(st1.mean), _lib.ptr(st1.invstd),
            _lib.ptr(y2), _lib.ptr(gamma2), _lib.ptr(st2.mean) if st2 else None, _lib.ptr(st2.invstd) if st2 else None,
            _lib.ptr(part), _lib.ptr(coef), _lib.ptr(dy1), _lib.ptr(dg1), _lib.ptr(db1), _lib.ptr(dy2), _lib.ptr(dg2),
            _lib.ptr(db2), N, C, T * V, _lib.stream()), "agcn_bn_bwd")
    else:
        _lib.check(_L().agcn_bn_bwd_reduce(_lib.ptr(dout), mptr, mbits, _lib.ptr(y1), _lib.ptr(y2),
                                           _lib.ptr(part), N, C, T * V, _lib.stream()), "agcn_bn_bwd_reduce")
        # one collective for both branches; dgamma/dbeta come out as GLOBAL sums, scaled by 1/world so that the
        # gradient average of the data-parallel step restores them (every rank holds the same value)
        sums = _allreduce_sum(_colsum(part, N, 3 * C), sync)
        total = float(gcount) if gcount is not None else float(N * T * V * sync.world)
        _lib.check(_L().agcn_bn_bwd_apply_ex(
            _lib.ptr(sums), 1, total, 1.0 / sync.world, _lib.ptr(dout), mptr, mbits, _lib.ptr(y1),
            _lib.ptr(gamma1), _lib.ptr(st1.mean), _lib.ptr(st1.invstd), _lib.ptr(y2), _lib.ptr(gamma2),
            _lib.ptr(st2.mean) if st2 else None, _lib.ptr(st2.invstd) if st2 else None, _lib.ptr(coef), _lib.ptr(dy1),
            _lib.ptr(dg1), _lib.ptr(db1), _lib.ptr(dy2), _lib.ptr(dg2), _lib.ptr(db2), _lib.ptr(amax_out), N, C, T * V,
            _lib.stream()), "agcn_bn_bwd_apply")
    if S > 1:                          # the weight / bias are shared by the S virtual sub-batches
        dg1, db1 = dg1.view(S, -1).sum(0), db1.view(S, -1).sum(0)
        if dg2 is not None:
            dg2, db2 = dg2.view(S, -1).sum(0), db2.view(S, -1).sum(0)
    return dy1, dg1, db1, dy2, dg2, db2


# ------------------------------------------------------------------------------------------------
# unit_gcn / unit_tcn forward and backward as plain functions over a context object
# ------------------------------------------------------------------------------------------------

class _Ctx:
    pass


def _bn_coeffs(training, stats, count, bns, sync=None, nsamples=None):
    """Coefficients of the BatchNorm stages of one unit (main [+ down/residual]): ``stats`` / ``bns`` are parallel
    lists of partial-sum slabs and (weight, bias, running_mean, running_var).  Returns ([BNState ...], global count)."""
    S = bns[0][2].numel() // bns[0][0].numel()        # GhostBatchNorm keeps S*C running statistics
    if not training:                                  # eval: plain BN on the first C running entries (reference
        out = []                                      # ghostbatchnorm.py:108-117; .eval() has averaged them over S)
        for w, b, rm, rv in bns:
            C = w.numel()
            out.append(bn_eval_coeffs(w, b, rm[:C].contiguous(), rv[:C].contiguous()))
        return out, count
    if S > 1:
        if nsamples is None or nsamples % S:
            raise RuntimeError(f"agcn_amd: GhostBatchNorm with {S} splits needs a batch (x persons) divisible by {S}")
        stats = [_ghost_regroup(sp, S, nsamples) for sp in stats]
        bns = [(w.repeat(S), b.repeat(S), rm, rv) for w, b, rm, rv in bns]
        count = count // S
    if sync is not None:
        stats = sync_stats(stats, count, sync)
        count = count * sync.world
    note_params_changed()                             # running statistics are about to be rewritten in place
    res = [bn_train_coeffs(sp, count, *bn) for sp, bn in zip(stats, bns)]
    for st in res:
        st.S = S
    return res, count


def _ghost_regroup(stats_part, S, nsamples):
    """(N*nt, 2, C) per-(sample, tile) partial sums -> (N/S*nt, 2, S*C): sample n = n'*S + s feeds virtual channel
    s*C + c of row n' (GhostBatchNorm views (N, C, ...) as (N/S, S*C, ...), ghostbatchnorm.py:98-99)."""
    slots, two, C = stats_part.shape
    nt = slots // nsamples
    return stats_part.view(nsamples // S, S, nt, two, C).permute(0, 2, 3, 1, 4).reshape(nsamples // S * nt, two, S * C)


def gcn_forward(c, x, A, PA, wab, bab, wd, bd, bn, down, training, alpha=None, adaptive=True, sync=None,
                need_bwd=True):
    """unit_gcn.forward (reference agcn.py:92-109) and AAGCN's GCNUnit core (aagcn.py:164-177, 264-266).
    wab: (6Ci, C, 1, 1) rows [a0|b0|a1|b1|a2|b2]; wd: (Cout, 3C); bd: summed conv_d biases;
    bn = (weight, bias, running_mean, running_var); down = None | (w, b, bn_w, bn_b, bn_rm, bn_rv).
    AGCN: adj = P + A + PA.  AAGCN: A = None, adj = PA + alpha*P.  adaptive=False (NonAdaptiveGCN): adj = A."""
    N, C, T, V = x.shape
    count = N * T * V
    first = down is not None and first_layer_enabled() and bool(_L().agcn_gcn_first_supported(C, wd.shape[0], V))
    # max |x| left behind by the pass that produced x (None: nobody did); the 3-channel first layer does not use it
    x_amax = None if first else _take_out_amax(x)
    if adaptive and adjacency_fused_supported(C, wab.shape[0] // 6, T, V):
        # theta/phi are formed and reduced on chip.  A training forward either keeps nothing (the backward recomputes
        # them, adjacency_bwd with tp = None) or lets the kernel drop a copy for the backward to re-read.
        keep = training and need_bwd and not adjacency_recompute()
        amax_here = None
        if x_amax is None and not first and fused_amax_enabled():
            amax_here = _empty((1,), x)    # this pass reads all of x: it takes the maximum along the way
        if keep:
            P, adj, tp = adjacency_fused_fwd(x, wab, bab, A, PA, alpha, keep_tp=True, x_amax_out=amax_here, x_amax=x_amax)
        else:
            tp = None
            P, adj = adjacency_fused_fwd(x, wab, bab, A, PA, alpha, x_amax_out=amax_here, x_amax=x_amax)
        if amax_here is not None:
            x_amax = amax_here
    elif adaptive:
        tp, _ = conv_fwd(x, wab, bab)
        P, adj = adjacency_fwd(tp, A, PA, alpha)
    else:
        tp = P = None
        adj = A.unsqueeze(0).expand(N, 3, V, V).contiguous()
    if first:
        # 3-channel first layer: aggregate+project and the `down` convolution in one pass over x (csrc/gcn_first.hip)
        ypre, st, dpre, st2 = gcn_first_fwd(x, adj, wd, bd, down[0], down[1], want_stats=training)
    else:
        c.g_x_amax = x_amax                # kept for the weight gradients of the backward
        ypre, st = aggregate_project_fwd(x, adj, wd, bd, want_stats=training, x_amax=c.g_x_amax)
    bn2 = None
    if not first:
        dpre = None
    if down is not None:
        if not first:
            dpre, st2 = conv_fwd(x, down[0], down[1], want_stats=training)
        (bn1, bn2), gcount = _bn_coeffs(training, [st, st2], count, [bn, down[2:]], sync, N)
        c.g_amax = _empty((1,), x) if fused_amax_enabled() else None   # max |g| for the temporal convolution that reads g
        out, bits = bn_act_fwd(ypre, bn1, dpre, bn2, relu=True, want_bits=True, amax_out=c.g_amax)
    else:
        (bn1,), gcount = _bn_coeffs(training, [st], count, [bn], sync, N)
        c.g_amax = _empty((1,), x) if fused_amax_enabled() else None
        out, bits = bn_act_fwd(ypre, bn1, x, None, relu=True, want_bits=True, amax_out=c.g_amax)
    c.g_sync, c.g_count = sync, gcount
    c.g_bits = bits          # sign bit mask of `out` for the BatchNorm backward (32x less traffic than `out`)
    c.g_x, c.g_tp, c.g_P, c.g_adj, c.g_ypre, c.g_dpre, c.g_out = x, tp, P, adj, ypre, dpre, out
    c.g_bn1, c.g_bn2 = bn1, bn2
    c.g_params = (wab, wd, bn[0], down[0] if down is not None else None, down[2] if down is not None else None)
    c.g_bab = bab
    c.g_alpha, c.g_adaptive = alpha, adaptive
    return out


def gcn_backward(c, dout, extra_add=None, extra_mask=None):
    """Backward of gcn_forward.  ``extra_add`` (masked by ``extra_mask``) is an additional dx contribution folded
    into the epilogue of the first dx kernel (the TCN_GCN_unit identity residual)."""
    x, tp, P, adj, ypre, dpre, out = c.g_x, c.g_tp, c.g_P, c.g_adj, c.g_ypre, c.g_dpre, c.g_out
    wab, wd, gamma1, wdown, gamma2 = c.g_params
    Cout = wd.shape[0]
    dy_amax = _empty((1,), dout) if fused_amax_enabled() else None   # max |dypre| for the f16x3 backward-data chain
    dypre, dg1, db1, ddpre, dg2, db2 = bn_bwd(dout, c.g_bits, ypre, gamma1, c.g_bn1, dpre, gamma2, c.g_bn2,
                                              sync=c.g_sync, gcount=c.g_count, amax_out=dy_amax)
    x_amax = getattr(c, 'g_x_amax', None)
    dwd = _side_run(lambda: project_bwd_weight(dypre, x, adj, Cout, dy_amax, x_amax), (dypre, x, adj, dy_amax, x_amax))
    dPA = dwab = dbab = dalpha = dtp = dtp_amax = None
    if c.g_adaptive:      # adjacency branch first: its dtp rides along in the dx kernel where that is supported
        dPA, dtp, dbab, dalpha, _, dtp_amax = adjacency_bwd(dypre, wd, x, tp, P, c.g_alpha, wab, c.g_bab,
                                                            dy_amax=dy_amax, x_amax=x_amax)
        dwab = _side_run(lambda: conv_bwd_weight(dtp, x, wab.shape, 1, dtp_amax, x_amax), (dtp, x, dtp_amax, x_amax))
    fuse = dtp is not None and fused_bwd_data_supported(x.shape[1], Cout, x.shape[3])
    ftp = dict(dtp=dtp, wab=wab) if fuse else {}
    ftp['dy_amax'] = dy_amax
    if fuse:
        ftp['dtp_amax'] = dtp_amax
    if dpre is None:      # identity `down`: dx += dout * (out > 0)
        dx = aggregate_project_bwd_data(dypre, adj, wd, x.shape, add1=dout, mask1=c.g_bits, add2=extra_add,
                                        mask2=extra_mask, **ftp)
    else:
        dx = aggregate_project_bwd_data(dypre, adj, wd, x.shape, add1=extra_add, mask1=extra_mask, **ftp)
    if dtp is not None and not fuse:
        conv_bwd_data(dtp, wab, x.shape, out=dx, accumulate=True)
    c.g_dalpha = dalpha
    dwdown = None
    if dpre is not None:
        dwdown = _side_run(lambda: conv_bwd_weight(ddpre, x, wdown.shape), (ddpre, x))
        conv_bwd_data(ddpre, wdown, x.shape, out=dx, accumulate=True)
    _side_join()
    return dx, dPA, dwab, dbab, dwd, dg1, db1, dwdown, dg2, db2


def tcn_forward(c, g, w, b, bn, stride, res_x, res, relu, training, sync=None):
    """unit_tcn.forward (reference agcn.py:48-50) optionally fused with the TCN_GCN_unit tail
    relu(tcn(g) + residual(x)) (agcn.py:127-129).  res = None (no residual) | 'identity' |
    (w, b, bn_w, bn_b, bn_rm, bn_rv) for the unit_tcn(kernel_size=1, stride) residual."""
    N, C, T, V = g.shape
    g_amax = getattr(c, 'g_amax', None) if getattr(c, 'g_out', None) is g else None   # only for the tensor it describes
    if g_amax is None and getattr(c, 'g_out', None) is None:
        g_amax = _take_out_amax(g)       # stand-alone unit_tcn (AAGCN): the producer of g (attention gates / unit_gcn) noted it
    zpre, st = conv_fwd(g, w, b, stride, want_stats=training, x_amax=g_amax)
    To = zpre.shape[2]
    count = N * To * V
    rpre = bn2 = None
    if res is None or isinstance(res, str):
        (bn1,), gcount = _bn_coeffs(training, [st], count, [bn], sync, N)
        o_amax = _empty((1,), g) if fused_amax_enabled() else None   # max |out| for the next unit's f16x3 chain
        out, bits = bn_act_fwd(zpre, bn1, None if res is None else res_x, None, relu=relu, want_bits=True,
                               amax_out=o_amax)
    else:
        rpre, st2 = conv_fwd(res_x, res[0], res[1], stride, want_stats=training)
        (bn1, bn2), gcount = _bn_coeffs(training, [st, st2], count, [bn, res[2:]], sync, N)
        o_amax = _empty((1,), g) if fused_amax_enabled() else None
        out, bits = bn_act_fwd(zpre, bn1, rpre, bn2, relu=relu, want_bits=True, amax_out=o_amax)
    _note_out_amax(out, o_amax)
    c.t_sync, c.t_count = sync, gcount
    c.t_bits = bits
    c.t_g, c.t_zpre, c.t_rpre, c.t_out, c.t_bn1, c.t_bn2 = g, zpre, rpre, out, bn1, bn2
    c.t_g_amax = g_amax
    c.t_resx, c.t_res_identity = res_x, isinstance(res, str)
    c.t_params = (w, bn[0], res[0] if isinstance(res, tuple) else None, res[2] if isinstance(res, tuple) else None)
    c.t_stride, c.t_relu = stride, relu
    return out


def tcn_backward(c, dout, join=True):
    """Returns dg, dw, dgamma, dbeta, (drpre, dw_res, dgamma_res, dbeta_res)."""
    w, gamma1, wres, gamma2 = c.t_params
    mask = c.t_bits if c.t_relu else None
    dz_amax = _empty((1,), dout) if fused_amax_enabled() else None   # max |dzpre| for the backward-data convolution
    dzpre, dg1, db1, drpre, dg2, db2 = bn_bwd(dout, mask, c.t_zpre, gamma1, c.t_bn1, c.t_rpre, gamma2, c.t_bn2,
                                              sync=c.t_sync, gcount=c.t_count, amax_out=dz_amax)
    t_g, t_stride = c.t_g, c.t_stride
    t_g_amax = getattr(c, 't_g_amax', None)
    # (the device scalars are inputs of the side-stream kernels too: dz_amax dies with this frame, possibly before the join)
    dw = _side_run(lambda: conv_bwd_weight(dzpre, t_g, w.shape, t_stride, dz_amax, t_g_amax),
                   (dzpre, t_g, dz_amax, t_g_amax))
    dg = conv_bwd_data(dzpre, w, c.t_g.shape, c.t_stride, dy_amax=dz_amax)
    dwres = None
    if drpre is not None:
        t_resx = c.t_resx
        dwres = _side_run(lambda: conv_bwd_weight(drpre, t_resx, wres.shape, t_stride), (drpre, t_resx))
    if join:
        _side_join()
    return dg, dw, dg1, db1, drpre, dwres, dg2, db2


# ---- BN-folded inference (eval mode under no_grad): adjacency + two kernels per TCN_GCN_unit ----------------------
ERR_UNSUPPORTED = -3

# Parameters and BatchNorm running statistics are written through RAW POINTERS by the training path (agcn_sgd_step on the
# flat buffer, agcn_bn_stats_finalize), which moves neither data_ptr nor the tensors' version counters.  Everything that
# caches something derived from them (the folded inference weights) keys on this counter too; it is bumped by every
# optimiser step and every training-mode BatchNorm stage.
_PARAM_EPOCH = [0]


def note_params_changed():
    _PARAM_EPOCH[0] += 1


def infer_fold_enabled():
    """AGCN_INFER_FOLD=0 keeps the eval forward on the unfused passes (A/B and debugging)."""
    return os.environ.get('AGCN_INFER_FOLD', '1') != '0'


def _fold(bn):
    """(scale, shift) of an eval-mode BatchNorm: y = scale * x + shift  (reference nn.BatchNorm2d in eval mode)."""
    w, b, rm, rv = bn
    s = w / torch.sqrt(rv + BN_EPS)
    return s, b - rm * s


def gcn_unit_infer(x, adj, wcat, bias, res=None, x2=None, w2=None, relu=True):
    """y = act(bias + sum_i W_i (x . adj_i) [+ res] [+ W2 . x2]); None when only the exact-f32 kernels apply."""
    N, C, T, V = x.shape
    Cout = wcat.shape[0]
    K2 = 0 if x2 is None else x2.shape[1]
    nbytes = _L().agcn_gcn_unit_infer_workspace(C, Cout, K2, T, V)
    ws = _ws(nbytes, x)
    y = _empty((N, Cout, T, V), x)
    rc = _L().agcn_gcn_unit_infer(_lib.ptr(x), _lib.ptr(adj), _lib.ptr(wcat), _lib.ptr(bias), _lib.ptr(res),
                                  _lib.ptr(x2), _lib.ptr(w2), K2, 1 if relu else 0, _lib.ptr(y), ws.data_ptr(), nbytes,
                                  N, C, Cout, T, V, _lib.stream())
    if rc == ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "agcn_gcn_unit_infer")
    return y


def conv9_infer(x, w, b, res=None, relu=True, stride=1):
    """y = act(b + conv9x1(x; w, stride) [+ res]); None when only the exact-f32 kernels apply."""
    N, Cin, T, V = x.shape
    Cout = w.shape[0]
    To = conv_out_frames(T, 9, stride)
    ws, nbytes = _conv_ws(Cin, Cout, T, V, 9, stride, x)
    y = _empty((N, Cout, To, V), x)
    rc = _L().agcn_conv9_infer(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(res), 1 if relu else 0, _lib.ptr(y),
                               ws.data_ptr(), nbytes, N, Cin, Cout, T, V, stride, _lib.stream())
    if rc == ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "agcn_conv9_infer")
    return y


def unit_infer(x, A, PA, wab, bab, wd, bd, gbn, down, tw, tb, tbn, res_mode, res, stride, alpha=None, adaptive=True,
               cache=None):
    """Eval-mode TCN_GCN_unit (reference agcn.py:92-109, 48-50, 127-129) with every BatchNorm folded into the contraction
    in front of it: adjacency, one aggregate+project kernel (conv `down` as extra plain stages, residual add and ReLU
    in its epilogue), [the 1x1 stride residual conv], one temporal-conv kernel (residual add + ReLU in its epilogue).
    gbn / tbn = (weight, bias, running_mean, running_var); down / res = None | (w, b, bn_w, bn_b, bn_rm, bn_rv).
    Returns None where the fused kernels do not apply (3-channel first layer, AGCN_GEMM=f32): the caller then runs the
    unfused eval passes.  ``cache`` (a dict owned by the module) keeps the folded weights between calls."""
    x = x.contiguous()
    N, C, T, V = x.shape
    Cout = wd.shape[0]
    if C < 32:
        return None
    srcs = [wd, bd, *gbn, tw, tb, *tbn] + (list(down) if down is not None else []) + \
           (list(res) if isinstance(res, tuple) else [])
    key = (_PARAM_EPOCH[0],) + tuple((t.data_ptr(), t._version) for t in srcs)
    f = cache.get('folded') if cache is not None else None
    if f is None or f[0] != key:
        s1, sh1 = _fold(gbn)
        wdf = (wd * s1[:, None]).contiguous()
        bias = bd * s1 + sh1
        w2 = None
        if down is not None:
            s2, sh2 = _fold(down[2:])
            w2 = (down[0].reshape(Cout, C) * s2[:, None]).contiguous()
            bias = bias + down[1] * s2 + sh2
        s3, sh3 = _fold(tbn)
        twf = (tw * s3[:, None, None, None]).contiguous()
        tbf = tb * s3 + sh3
        rwf = rbf = None
        if isinstance(res, tuple):
            s4, sh4 = _fold(res[2:])
            rwf = (res[0] * s4[:, None, None, None]).contiguous()
            rbf = res[1] * s4 + sh4
        f = (key, wdf, bias.contiguous(), w2, twf, tbf.contiguous(), rwf, rbf)
        if cache is not None:
            cache['folded'] = f
    _, wdf, bias, w2, twf, tbf, rwf, rbf = f
    if adaptive and adjacency_fused_supported(C, wab.shape[0] // 6, T, V):
        _, adj = adjacency_fused_fwd(x, wab, bab, A, PA, alpha)
    elif adaptive:
        tp, _ = conv_fwd(x, wab, bab)
        _, adj = adjacency_fwd(tp, A, PA, alpha)
    else:
        adj = A.unsqueeze(0).expand(N, 3, V, V).contiguous()
    if down is None:
        g = gcn_unit_infer(x, adj, wdf, bias, res=x)
    else:
        g = gcn_unit_infer(x, adj, wdf, bias, x2=x, w2=w2)
    if g is None:
        return None
    if res_mode == 0:
        r = None
    elif res_mode == 1:
        r = x
    else:
        r, _ = conv_fwd(x, rwf, rbf, stride)
    return conv9_infer(g, twf, tbf, r, relu=True, stride=stride)


def _need_train(training):
    if not training:
        raise NotImplementedError("agcn_amd: backward through eval-mode BatchNorm is not implemented; call "
                                  "model.train() for training or torch.no_grad() for inference")


class UnitGCNFunction(torch.autograd.Function):
    """unit_gcn: args (x, A, PA, wab, bab, wd, bd, bn_w, bn_b, bn_rm, bn_rv,
                       down_w, down_b, dbn_w, dbn_b, dbn_rm, dbn_rv, training, alpha=None, adaptive=True)"""

    @staticmethod
    def forward(ctx, x, A, PA, wab, bab, wd, bd, bn_w, bn_b, bn_rm, bn_rv, down_w, down_b, dbn_w, dbn_b, dbn_rm,
                dbn_rv, training, alpha=None, adaptive=True, sync=None):
        c = _Ctx()
        down = None if down_w is None else (down_w, down_b, dbn_w, dbn_b, dbn_rm, dbn_rv)
        out = gcn_forward(c, x.contiguous(), A, PA, wab, bab, wd, bd, (bn_w, bn_b, bn_rm, bn_rv), down, training,
                          alpha, adaptive, sync, need_bwd=any(ctx.needs_input_grad))
        ctx.c, ctx.training = c, training
        ctx.has_down = down is not None
        _note_out_amax(out, getattr(c, 'g_amax', None))     # stand-alone node: a unit_tcn may read this tensor next
        return out

    @staticmethod
    def backward(ctx, dout):
        _need_train(ctx.training)
        dx, dPA, dwab, dbab, dwd, dg1, db1, dwdown, dg2, db2 = gcn_backward(ctx.c, dout.contiguous())
        zb = lambda t: torch.zeros(t.shape[0], dtype=torch.float32, device=dout.device)  # noqa: E731
        dbd = zb(dwd)
        dbdown = zb(dwdown) if ctx.has_down else None
        dalpha = ctx.c.g_dalpha
        if dalpha is not None:
            dalpha = dalpha.reshape(1)
        ctx.c = None
        return (dx, None, dPA, dwab, dbab, dwd, dbd, dg1, db1, None, None, dwdown, dbdown, dg2, db2, None, None, None,
                dalpha, None, None)


class UnitTCNFunction(torch.autograd.Function):
    """unit_tcn (no residual, no ReLU): args (x, w, b, bn_w, bn_b, bn_rm, bn_rv, stride, training)"""

    @staticmethod
    def forward(ctx, x, w, b, bn_w, bn_b, bn_rm, bn_rv, stride, training, sync=None):
        c = _Ctx()
        out = tcn_forward(c, x.contiguous(), w, b, (bn_w, bn_b, bn_rm, bn_rv), stride, None, None, False, training,
                          sync)
        ctx.c, ctx.training = c, training
        return out

    @staticmethod
    def backward(ctx, dout):
        _need_train(ctx.training)
        dg, dw, dg1, db1, _, _, _, _ = tcn_backward(ctx.c, dout.contiguous())
        dbias = torch.zeros(dw.shape[0], dtype=torch.float32, device=dout.device)
        ctx.c = None
        return dg, dw, dbias, dg1, db1, None, None, None, None, None


class TCNResidualFunction(torch.autograd.Function):
    """relu(bn(conv9(g)) + residual(x)) as its own autograd node (AAGCN: the attention ops sit between the GCN core and
    this).  args: g, x, w, b, bn_w, bn_b, bn_rm, bn_rv, res_mode (0 none, 1 identity, 2 conv), rw, rb, rbn_w, rbn_b,
    rbn_rm, rbn_rv, stride, training"""

    @staticmethod
    def forward(ctx, g, x, w, b, bn_w, bn_b, bn_rm, bn_rv, res_mode, rw, rb, rbn_w, rbn_b, rbn_rm, rbn_rv, stride,
                training, sync=None):
        c = _Ctx()
        res = None if res_mode == 0 else ('identity' if res_mode == 1 else (rw, rb, rbn_w, rbn_b, rbn_rm, rbn_rv))
        x = x.contiguous() if x is not None else None
        out = tcn_forward(c, g.contiguous(), w, b, (bn_w, bn_b, bn_rm, bn_rv), stride, x, res, True, training, sync)
        ctx.c, ctx.training, ctx.res_mode, ctx.stride = c, training, res_mode, stride
        ctx.x_shape = tuple(x.shape) if x is not None else None
        return out

    @staticmethod
    def backward(ctx, dout):
        _need_train(ctx.training)
        c = ctx.c
        dout = dout.contiguous()
        dg, dw, dg1, db1, drpre, dwres, dg2, db2 = tcn_backward(c, dout)
        dev = dout.device
        zb = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)  # noqa: E731
        dx = None
        if ctx.res_mode == 1:
            dx = torch.where(c.t_out > 0, dout, torch.zeros_like(dout))
        elif ctx.res_mode == 2:
            dx = conv_bwd_data(drpre, c.t_params[2], ctx.x_shape, ctx.stride)
        ctx.c = None
        return (dg, dx, dw, zb(dw.shape[0]), dg1, db1, None, None, None,
                dwres, zb(dwres.shape[0]) if ctx.res_mode == 2 else None, dg2, db2, None, None, None, None, None)


class TCNGCNUnitFunction(torch.autograd.Function):
    """TCN_GCN_unit = relu(tcn1(gcn1(x)) + residual(x)) as ONE autograd node, so that every dx contribution is
    accumulated in a contraction epilogue instead of separate elementwise passes.

    args: x, A, PA, wab, bab, wd, bd, gbn_w, gbn_b, gbn_rm, gbn_rv, down_w, down_b, dbn_w, dbn_b, dbn_rm, dbn_rv,
          tw, tb, tbn_w, tbn_b, tbn_rm, tbn_rv, res_mode(0 none,1 identity,2 conv), rw, rb, rbn_w, rbn_b, rbn_rm,
          rbn_rv, stride, training, sync (SyncBN policy or None)"""

    @staticmethod
    def forward(ctx, x, A, PA, wab, bab, wd, bd, gbn_w, gbn_b, gbn_rm, gbn_rv, down_w, down_b, dbn_w, dbn_b, dbn_rm,
                dbn_rv, tw, tb, tbn_w, tbn_b, tbn_rm, tbn_rv, res_mode, rw, rb, rbn_w, rbn_b, rbn_rm, rbn_rv, stride,
                training, sync=None):
        c = _Ctx()
        x = x.contiguous()
        down = None if down_w is None else (down_w, down_b, dbn_w, dbn_b, dbn_rm, dbn_rv)
        g = gcn_forward(c, x, A, PA, wab, bab, wd, bd, (gbn_w, gbn_b, gbn_rm, gbn_rv), down, training, sync=sync,
                        need_bwd=any(ctx.needs_input_grad))
        res = None if res_mode == 0 else ('identity' if res_mode == 1 else (rw, rb, rbn_w, rbn_b, rbn_rm, rbn_rv))
        out = tcn_forward(c, g, tw, tb, (tbn_w, tbn_b, tbn_rm, tbn_rv), stride, x, res, True, training, sync)
        ctx.c, ctx.training, ctx.has_down, ctx.res_mode, ctx.stride = c, training, down is not None, res_mode, stride
        return out

    @staticmethod
    def backward(ctx, dout):
        _need_train(ctx.training)
        c = ctx.c
        dout = dout.contiguous()
        _SIDE_SCOPE[0] += 1
        try:
            dg, dtw, dtg, dtb, drpre, drw, drg, drb = tcn_backward(c, dout, join=False)   # (gcn_backward joins)
            if ctx.res_mode == 1:
                gres = gcn_backward(c, dg, extra_add=dout, extra_mask=c.t_bits)
            else:
                gres = gcn_backward(c, dg)
        finally:
            _SIDE_SCOPE[0] -= 1
        dx, dPA, dwab, dbab, dwd, dg1, db1, dwdown, dg2, db2 = gres
        if ctx.res_mode == 2:
            conv_bwd_data(drpre, c.t_params[2], c.g_x.shape, ctx.stride, out=dx, accumulate=True)
        dev = dout.device
        zb = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)  # noqa: E731
        ctx.c = None
        return (dx, None, dPA, dwab, dbab, dwd, zb(dwd.shape[0]), dg1, db1, None, None,
                dwdown, zb(dwdown.shape[0]) if ctx.has_down else None, dg2, db2, None, None,
                dtw, zb(dtw.shape[0]), dtg, dtb, None, None, None,
                drw, zb(drw.shape[0]) if ctx.res_mode == 2 else None, drg, drb, None, None, None, None, None)
