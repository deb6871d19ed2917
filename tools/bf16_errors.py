"""Per-operator error of the current AGCN_GEMM mode vs fp64 (diagnostic): python tools/bf16_errors.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import agcn_amd
from agcn_amd import ops, lib
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
rnd = lambda *s, scale=1.0: torch.randn(*s, generator=g, dtype=torch.float64) * scale
rel = lambda a, b: float((a.detach().double().cpu() - b.detach()).abs().max() / b.detach().abs().max())
print('mode', lib.load().agcn_gemm_mode().decode())
for (N, C, Co, T, V, taps, st) in [(2, 64, 64, 23, 25, 9, 1), (2, 64, 128, 22, 25, 9, 2), (2, 128, 128, 21, 25, 9, 1), (2, 128, 128, 22, 25, 9, 2),
                                   (2, 64, 128, 22, 25, 1, 2), (2, 64, 192, 23, 25, 1, 1)]:
    x = rnd(N, C, T, V).requires_grad_(True); w = rnd(Co, C, taps, 1, scale=1 / np.sqrt(C * taps)).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=(st, 1), padding=((taps - 1) // 2, 0)); dy = rnd(*y.shape); y.backward(dy)
    xg, wg, dyg = x.detach().float().to(dev), w.detach().float().to(dev), dy.float().to(dev)
    yo, _ = ops.conv_fwd(xg, wg, None, st)
    dx = ops.conv_bwd_data(dyg, wg, tuple(x.shape), st)
    dw = ops.conv_bwd_weight(dyg, xg, tuple(w.shape), st)
    print(f'conv C{C}->{Co} k{taps} s{st}: fwd {rel(yo, y):.2e} dgrad {rel(dx, x.grad):.2e} wgrad {rel(dw, w.grad):.2e}')
for (N, C, Co, T, V) in [(2, 64, 64, 23, 25), (2, 64, 128, 22, 25), (2, 128, 256, 12, 25)]:
    x = rnd(N, C, T, V).requires_grad_(True); adj = rnd(N, 3, V, V, scale=0.3).requires_grad_(True)
    w = rnd(Co, 3 * C, scale=1 / np.sqrt(3 * C)).requires_grad_(True)
    y = 0
    for i in range(3):
        y = y + torch.einsum('oc,nctv->notv', w[:, i * C:(i + 1) * C], torch.einsum('nctu,nuv->nctv', x, adj[:, i]))
    dy = rnd(*y.shape); y.backward(dy)
    xg, ag, wg, dyg = [t.detach().float().to(dev) for t in (x, adj, w, dy)]
    yo, _ = ops.aggregate_project_fwd(xg, ag, wg, torch.zeros(Co, device=dev))
    dx = ops.aggregate_project_bwd_data(dyg, ag, wg, tuple(x.shape))
    dw = ops.project_bwd_weight(dyg, xg, ag, Co)
    L = ops._L(); ns = L.agcn_dadj_num_slots(C, V, T); dpart = torch.empty((N, 3, ns, V, V), device=dev)
    ws, nb = ops._gcn_ws(C, Co, T, V, xg)
    lib.check(L.agcn_gcn_dadj(lib.ptr(dyg), lib.ptr(wg), lib.ptr(xg), lib.ptr(dpart), ws.data_ptr(), nb, N, C, Co, T, V, lib.stream()), 'dadj')
    print(f'gcn C{C}->{Co}: fwd {rel(yo, y):.2e} dgrad {rel(dx, x.grad):.2e} wgrad {rel(dw, w.grad):.2e} dadj {rel(dpart.sum(2), adj.grad):.2e}')
