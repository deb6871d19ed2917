import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import agcn_amd
from agcn_amd import ops, lib
from oracle import agcn_oracle as orc
from tests import golden_util as gu
dev = torch.device('cuda:0')
def rel(a, ref):
    a = a.detach().double().cpu(); ref = ref.detach().double().cpu()
    return float((a - ref).abs().max() / max(1e-30, float(ref.abs().max())))
def gcn_ref(x, adj, wcat, bias):
    N, C, T, V = x.shape; y = 0
    for i in range(3):
        gi = torch.einsum('nctu,nuv->nctv', x, adj[:, i])
        y = y + torch.einsum('oc,nctv->notv', wcat[:, i * C:(i + 1) * C], gi)
    return y + bias.view(1, -1, 1, 1)
g = torch.Generator().manual_seed(0)
print('== kernel sweep (aggregate/project)')
for (C, Cout) in [(64, 64), (128, 128), (256, 256), (64, 128), (128, 256), (3, 64)]:
    for V in (25, 18):
        for T in (9, 16, 31):
            N = 4
            x = torch.randn(N, C, T, V, generator=g, dtype=torch.float64, requires_grad=True)
            adj = (0.3 * torch.randn(N, 3, V, V, generator=g, dtype=torch.float64)).requires_grad_(True)
            wcat = (torch.randn(Cout, 3 * C, generator=g, dtype=torch.float64) / np.sqrt(3 * C)).requires_grad_(True)
            bias = 0.1 * torch.randn(Cout, generator=g, dtype=torch.float64)
            y = gcn_ref(x, adj, wcat, bias); dy = torch.randn(*y.shape, generator=g, dtype=torch.float64); y.backward(dy)
            xg, ag, wg, bg, dyg = [t.detach().float().to(dev) for t in (x, adj, wcat, bias, dy)]
            yo, st = ops.aggregate_project_fwd(xg, ag, wg, bg, want_stats=True)
            dx = ops.aggregate_project_bwd_data(dyg, ag, wg, tuple(x.shape))
            dw = ops.project_bwd_weight(dyg, xg, ag, Cout)
            L = ops._L(); ns = L.agcn_dadj_num_slots(C, V, T)
            dpart = torch.empty((N, 3, ns, V, V), device=dev)
            ws, nb = ops._gcn_ws(C, Cout, T, V, xg); lib.check(L.agcn_gcn_dadj(lib.ptr(dyg), lib.ptr(wg), lib.ptr(xg), lib.ptr(dpart), ws.data_ptr(), nb, N, C, Cout, T, V, lib.stream()), 'dadj')
            e = (rel(yo, y), rel(dx, x.grad), rel(dw, wcat.grad), rel(dpart.sum(2), adj.grad))
            print((N, C, Cout, T, V), 'fwd %.1e dx %.1e dw %.1e dadj %.1e' % e, '' if max(e) < 1e-4 else '  <<<<<< FAIL')
print('== unit isolation with GAP-like loss')
for (cin, cout, stride, v, t, n) in [(256, 256, 1, 18, 16, 4), (256, 256, 1, 25, 75, 2), (64, 64, 1, 25, 300, 2), (128, 128, 1, 25, 150, 2)]:
    from agcn_amd.model.agcn import TCN_GCN_unit
    A = gu.graph_A(v)
    sd0 = orc.randomized_state(orc.unit_param_shapes('', cin, cout, v, stride, True), 11, stress=3.0)
    rng = np.random.default_rng(5)
    xn = np.maximum(rng.standard_normal((n, cin, t, v)), 0).astype(np.float32)
    rn = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32))
    res = {}
    for dt in (torch.float32, torch.float64):
        sd = orc.with_grad({k: (v_.to(dt) if v_.is_floating_point() else v_) for k, v_ in sd0.items()})
        xo = torch.from_numpy(xn).to(dt).requires_grad_(True)
        yo = orc.tcn_gcn_unit_forward(xo, sd, '', A.to(dt), stride, True, training=True)
        (yo.mean((2, 3)) * rn.to(dt)).sum().backward()
        res[dt] = (yo, xo, sd)
    unit = TCN_GCN_unit(cin, cout, A.numpy(), stride=stride, residual=True); unit.load_state_dict(sd0); unit.to(dev).train()
    x = torch.from_numpy(xn).to(dev).requires_grad_(True)
    y = unit(x); (y.mean((2, 3)) * rn.to(dev)).sum().backward()
    y64, x64, sd64 = res[torch.float64]; y32, x32, sd32 = res[torch.float32]
    print((cin, cout, stride, v, t, n), 'y %.1e dx hip %.1e (ref32 %.1e)' % (rel(y, y64), rel(x.grad, x64.grad), rel(x32.grad, x64.grad)))
    d = (x.grad.double().cpu() - x64.grad).abs(); idx = torch.nonzero(d > 0.01 * x64.grad.abs().max())
    print('   bad dx elements:', idx.shape[0], 'of', d.numel(), ' first:', idx[:12].tolist())
    for k, p in unit.named_parameters():
        e, nz = rel(p.grad, sd64[k].grad), rel(sd32[k].grad, sd64[k].grad)
        if e > 2e-4 and e > 3 * nz and not gu.is_zero_grad_bias(k): print('   ', k, 'hip %.1e ref32 %.1e' % (e, nz))
