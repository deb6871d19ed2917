"""Time the adaptive-adjacency forward (fused vs theta/phi materialised) and the recomputing backward per layer shape.
    python tools/bench_adj.py [layers...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import agcn_amd
from agcn_amd import ops, lib
dev = torch.device('cuda:0')
SHAPES = {'l1': (3, 64, 300), 'l2': (64, 64, 300), 'l5': (64, 128, 300), 'l6': (128, 128, 150), 'l8': (128, 256, 150),
          'l9': (256, 256, 75)}
layers = sys.argv[1:] or list(SHAPES)
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
    Ci = Cout // 4
    x = torch.randn(N, C, T, V, generator=g).to(dev)
    wab = (torch.randn(6 * Ci, C, 1, 1, generator=g) / C ** 0.5).to(dev)
    bab = torch.zeros(6 * Ci, device=dev)
    A = (0.2 * torch.randn(3, V, V, generator=g)).to(dev)
    PA = (0.05 * torch.randn(3, V, V, generator=g)).to(dev)
    dS = (1e-3 * torch.randn(N, 3, V, V, generator=g)).to(dev)
    out = [f'{name:4s} C{C:3d} Ci{Ci:3d} T{T:3d}']

    def old():
        tp, _ = ops.conv_fwd(x, wab, bab)
        return ops.adjacency_fwd(tp, A, PA)
    out.append(f'two-kernel fwd {timed(old):7.0f} us')
    if ops.adjacency_fused_supported(C, Ci, T, V):
        out.append(f'fused fwd {timed(lambda: ops.adjacency_fused_fwd(x, wab, bab, A, PA)):7.0f} us')
        out.append(f'fused+tp {timed(lambda: ops.adjacency_fused_fwd(x, wab, bab, A, PA, keep_tp=True)):7.0f} us')
        L = ops._L()
        nt = L.agcn_scores_num_tiles(V, T)
        dtp = torch.empty((N, 6 * Ci, T, V), device=dev)
        dbpart = torch.empty((N * nt, 6 * Ci), device=dev)
        db = torch.empty((6 * Ci,), device=dev)
        scratch = ops._scratch(6 * Ci, x)
        nb = L.agcn_adjacency_fused_workspace(C, Ci)
        ws = ops._ws(nb, x)
        tp, _ = ops.conv_fwd(x, wab, bab)
        us_old = timed(lambda: lib.check(L.agcn_adjacency_bwd_scores(
            lib.ptr(tp), lib.ptr(dS), lib.ptr(dtp), lib.ptr(dbpart), scratch.data_ptr(), lib.ptr(db), N, Ci, T, V,
            lib.stream()), 'bs'))
        us_new = timed(lambda: lib.check(L.agcn_adjacency_fused_bwd_scores(
            lib.ptr(x), lib.ptr(wab.view(6 * Ci, C)), lib.ptr(bab), lib.ptr(dS), lib.ptr(dtp), lib.ptr(dbpart),
            scratch.data_ptr(), lib.ptr(db), ws.data_ptr(), nb, N, C, Ci, T, V, lib.stream()), 'fbs'))
        out.append(f'bwd scores stored-tp {us_old:7.0f} us  recompute {us_new:7.0f} us')
    print('  '.join(out), flush=True)
