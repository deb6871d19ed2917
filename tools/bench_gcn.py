"""Time the unit_gcn contraction kernels per layer shape with HIP events (N'=128, V=25).
    python tools/bench_gcn.py [fwd,bwd,dadj,wgrad] [layer names...]     e.g.  python tools/bench_gcn.py fwd,bwd l9"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import agcn_amd
from agcn_amd import ops, lib
dev = torch.device('cuda:0')
SHAPES = {'l1': (3, 64, 300), 'l2': (64, 64, 300), 'l5': (64, 128, 300), 'l6': (128, 128, 150), 'l8': (128, 256, 150),
          'l9': (256, 256, 75)}
which = sys.argv[1].split(',') if len(sys.argv) > 1 else ['fwd', 'bwd', 'dadj', 'wgrad']
layers = sys.argv[2:] or list(SHAPES)
reps = int(os.environ.get('REPS', '5'))
N, V = 128, 25
g = torch.Generator().manual_seed(0)


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for name in layers:
    C, Cout, T = SHAPES[name]
    x = torch.randn(N, C, T, V, generator=g).to(dev)
    dy = torch.randn(N, Cout, T, V, generator=g).to(dev)
    adj = (0.2 * torch.randn(N, 3, V, V, generator=g)).to(dev)
    w = (torch.randn(Cout, 3 * C, generator=g) / (3 * C) ** 0.5).to(dev)
    b = torch.zeros(Cout, device=dev)
    proj = 3 * 2 * C * Cout * T * V * N
    agg = 3 * 2 * C * T * V * V * N
    out = [f'{name:4s} C{C:3d}->{Cout:3d} T{T:3d}']
    if 'fwd' in which:
        us = timed(lambda: ops.aggregate_project_fwd(x, adj, w, b, want_stats=True))
        out.append(f'fwd {us:7.0f} us {(proj + agg) / us / 1e6:6.1f} TF')
    if 'fwdn' in which:
        us = timed(lambda: ops.aggregate_project_fwd(x, adj, w, b, want_stats=False))
        out.append(f'fwdn {us:7.0f} us {(proj + agg) / us / 1e6:6.1f} TF')
    if 'bwd' in which:
        aggb = 3 * 2 * Cout * T * V * V * N
        us = timed(lambda: ops.aggregate_project_bwd_data(dy, adj, w, tuple(x.shape)))
        out.append(f'bwd {us:7.0f} us {(proj + aggb) / us / 1e6:6.1f} TF')
    if 'bwdx' in which:   # as in training: identity-residual gradient folded in (add1 masked by the forward output)
        aggb = 3 * 2 * Cout * T * V * V * N
        us = timed(lambda: ops.aggregate_project_bwd_data(dy, adj, w, tuple(x.shape), add1=x, mask1=x))
        out.append(f'bwdx {us:7.0f} us {(proj + aggb) / us / 1e6:6.1f} TF')
    if 'bwdf' in which and ops.fused_bwd_data_supported(C, Cout, V):   # as in training: fused 1x1 term + two masked residuals
        K2 = 6 * (Cout // 4)
        dtp = torch.randn(N, K2, T, V, generator=g).to(dev)
        wab = (torch.randn(K2, C, 1, 1, generator=g) / K2 ** 0.5).to(dev)
        bits = (torch.randint(0, 2 ** 31 - 1, ((N * C * T * V + 31) // 32,), generator=g, dtype=torch.int64).to(torch.int32)).to(dev)
        aggb = 3 * 2 * Cout * T * V * V * N
        amax = torch.full((1,), 4.0, device=dev)
        us = timed(lambda: ops.aggregate_project_bwd_data(dy, adj, w, tuple(x.shape), add1=x, mask1=bits, add2=x, mask2=bits,
                                                          dtp=dtp, wab=wab, dy_amax=amax, dtp_amax=amax))
        out.append(f'bwdf {us:7.0f} us {(proj + aggb + 2.0 * K2 * C * T * V * N) / us / 1e6:6.1f} TF')
    if 'adj' in which and C >= 16:     # adaptive adjacency forward (theta/phi kept for the backward, max |x| known)
        Ci = Cout // 4
        wab = (torch.randn(6 * Ci, C, generator=g) / C ** 0.5).to(dev)
        bab = torch.zeros(6 * Ci, device=dev)
        A = torch.zeros(3, V, V, device=dev)
        amax = x.abs().max().reshape(1)
        us = timed(lambda: ops.adjacency_fused_fwd(x, wab, bab, A, A, keep_tp=True, x_amax=amax))
        us2 = timed(lambda: ops.adjacency_fused_fwd(x, wab, bab, A, A, x_amax=amax))
        fl = 2.0 * 6 * Ci * C * T * V * N + 3 * 2.0 * Ci * T * V * V * N
        out.append(f'adj {us:7.0f} us (no tp copy {us2:5.0f}) {fl / us / 1e6:6.1f} TF {ops._L().agcn_last_kernel().decode()}')
    if 'sbwd' in which and C >= 16:    # backward of the scores: dtp from tp and dS
        Ci = Cout // 4
        tp = torch.randn(N, 6 * Ci, T, V, generator=g).to(dev)
        dS = torch.randn(N, 3, V, V, generator=g).to(dev)
        L = ops._L(); nt = L.agcn_scores_num_tiles(V, T)
        dtp = torch.empty_like(tp); dbp = torch.empty((N * nt, 6 * Ci), device=dev); db = torch.empty(6 * Ci, device=dev)
        scr = ops._scratch(6 * Ci, tp); am = torch.empty(1, device=dev)
        us = timed(lambda: lib.check(L.agcn_adjacency_bwd_scores_ex(lib.ptr(tp), lib.ptr(dS), lib.ptr(dtp), lib.ptr(dbp),
                                                                    scr.data_ptr(), lib.ptr(db), lib.ptr(am), N, Ci, T, V,
                                                                    lib.stream()), 'sbwd'))
        out.append(f'sbwd {us:7.0f} us {2 * tp.numel() * 4 / us / 1e6:5.2f} TB/s {L.agcn_last_kernel().decode()}')
    if 'dadj' in which:
        L = ops._L(); ns = L.agcn_dadj_num_slots(C, V, T)
        dpart = torch.empty((N, 3, ns, V, V), device=dev)
        ws, nb = ops._gcn_ws(C, Cout, T, V, x)
        am_dy, am_x = dy.abs().max().reshape(1), x.abs().max().reshape(1)     # (left behind by the producers in the step)
        us = timed(lambda: lib.check(L.agcn_gcn_dadj_ex(lib.ptr(dy), lib.ptr(w), lib.ptr(x), lib.ptr(dpart), ws.data_ptr(),
                                                        nb, N, C, Cout, T, V, lib.ptr(am_dy), lib.ptr(am_x), lib.stream()),
                                     'dadj'))
        out.append(f'dadj {us:7.0f} us {(proj + agg) / us / 1e6:6.1f} TF')
    if 'wgrad' in which:
        am_dy2, am_x2 = dy.abs().max().reshape(1), x.abs().max().reshape(1)   # (left behind by the producers in the step)
        us = timed(lambda: ops.project_bwd_weight(dy, x, adj, Cout, am_dy2, am_x2))
        out.append(f'wgrad {us:7.0f} us {(proj + agg) / us / 1e6:6.1f} TF {ops._L().agcn_last_kernel().decode()}')
    if 'w1' in which:     # 1x1 weight gradient at the shape of the fused conv_a/conv_b projection (M = 6*Cout/4 rows)
        M1 = 6 * (Cout // 4)
        dy1 = torch.randn(N, M1, T, V, generator=g).to(dev)
        us = timed(lambda: ops.conv_bwd_weight(dy1, x, (M1, C, 1, 1), 1))
        out.append(f'w1 {us:7.0f} us {2.0 * M1 * C * T * V * N / us / 1e6:6.1f} TF')
    print('  '.join(out), flush=True)
