"""Kernel-level parity of every C-ABI entry point against plain CPU fp64 PyTorch maths of the same operator.
GPU only (``-m gpu``); calls go through ``agcn_amd.ops`` -> ctypes -> ``libagcn_hip.so``.

Tolerance (fp32, north_star): max|a-ref| <= 1e-4 * max(1, max|ref|).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _gpu():
    import agcn_amd  # noqa: F401
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device('cuda:0')


def rel(a, ref):
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return float((a - ref).abs().max() / max(1.0, float(ref.abs().max())))


def rnd(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen, dtype=torch.float64) * scale)


CONV_CASES = [
    # N, Cin, Cout, T, V, taps, stride
    (2, 64, 64, 23, 25, 9, 1),
    (3, 64, 128, 21, 25, 9, 2),
    (2, 128, 128, 12, 25, 9, 1),
    (2, 128, 256, 14, 18, 9, 2),
    (2, 3, 64, 23, 25, 1, 1),
    (2, 3, 96, 11, 25, 1, 1),
    (2, 64, 192, 23, 25, 1, 1),
    (2, 64, 128, 21, 25, 1, 2),
    (2, 128, 256, 16, 18, 1, 2),
    (2, 256, 256, 9, 25, 9, 1),
    (1, 64, 64, 300, 25, 9, 1),
]


def test_mfma_layout_probe():
    """A = I (via a 1x1 conv with identity weights) and asymmetric inputs: catches any row/col swap."""
    from agcn_amd import ops
    dev = _gpu()
    g = torch.Generator().manual_seed(1)
    x = rnd(g, 1, 64, 7, 25)
    w = torch.eye(64, dtype=torch.float64).view(64, 64, 1, 1)
    y, _ = ops.conv_fwd(x.float().to(dev), w.float().to(dev), None)
    assert rel(y, x) < 1e-6


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_bwd(case):
    from agcn_amd import ops
    dev = _gpu()
    N, Cin, Cout, T, V, taps, stride = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = rnd(g, N, Cin, T, V).requires_grad_(True)
    w = rnd(g, Cout, Cin, taps, 1, scale=1.0 / np.sqrt(Cin * taps)).requires_grad_(True)
    b = rnd(g, Cout, scale=0.1)
    pad = (taps - 1) // 2
    y_ref = F.conv2d(x, w, b, stride=(stride, 1), padding=(pad, 0))
    dy = rnd(g, *y_ref.shape)
    y_ref.backward(dy)
    xg, wg, bg, dyg = x.detach().float().to(dev), w.detach().float().to(dev), b.float().to(dev), dy.float().to(dev)
    y, stats = ops.conv_fwd(xg, wg, bg, stride, want_stats=True)
    assert rel(y, y_ref) < TOL
    s = stats.double().sum(0).cpu()
    assert rel(s[0], y_ref.detach().sum((0, 2, 3))) < TOL * 10
    assert rel(s[1], (y_ref.detach() ** 2).sum((0, 2, 3))) < TOL * 10
    dx = ops.conv_bwd_data(dyg, wg, tuple(x.shape), stride)
    assert rel(dx, x.grad) < TOL
    # accumulate + masked addends
    base = rnd(g, *x.shape).float().to(dev)
    add = rnd(g, *x.shape).float().to(dev)
    mask = rnd(g, *x.shape).float().to(dev)
    out = base.clone()
    ops.conv_bwd_data(dyg, wg, tuple(x.shape), stride, out=out, accumulate=True, add1=add, mask1=mask, add2=add)
    ref2 = x.grad + base.double().cpu() + (add.double().cpu() * (mask.cpu() > 0)) + add.double().cpu()
    assert rel(out, ref2) < TOL
    dw = ops.conv_bwd_weight(dyg, xg, tuple(w.shape), stride)
    assert rel(dw, w.grad) < TOL


GCN_CASES = [
    # N, C, Cout, T, V
    (2, 64, 64, 23, 25),
    (2, 3, 64, 12, 25),
    (3, 64, 128, 11, 25),
    (2, 128, 256, 9, 25),
    (2, 256, 256, 7, 25),
    (2, 64, 64, 17, 18),
    (2, 128, 128, 16, 25),
    (1, 256, 128, 33, 25),
    (2, 64, 80, 10, 20),        # rows/K not multiples of the 32-channel stage, generic V
]


def _gcn_ref(x, adj, wcat, bias):
    N, C, T, V = x.shape
    y = 0
    for i in range(3):
        gi = torch.einsum('nctu,nuv->nctv', x, adj[:, i])
        y = y + torch.einsum('oc,nctv->notv', wcat[:, i * C:(i + 1) * C], gi)
    return y + bias.view(1, -1, 1, 1)


@pytest.mark.parametrize('case', GCN_CASES)
def test_aggregate_project(case):
    from agcn_amd import ops
    dev = _gpu()
    N, C, Cout, T, V = case
    g = torch.Generator().manual_seed(7 + C + T)
    x = rnd(g, N, C, T, V).requires_grad_(True)
    adj = rnd(g, N, 3, V, V, scale=0.3).requires_grad_(True)
    wcat = rnd(g, Cout, 3 * C, scale=1.0 / np.sqrt(3 * C)).requires_grad_(True)
    bias = rnd(g, Cout, scale=0.1)
    y_ref = _gcn_ref(x, adj, wcat, bias)
    dy = rnd(g, *y_ref.shape)
    y_ref.backward(dy)
    xg, ag, wg, bg, dyg = [t.detach().float().to(dev) for t in (x, adj, wcat, bias, dy)]
    y, stats = ops.aggregate_project_fwd(xg, ag, wg, bg, want_stats=True)
    assert rel(y, y_ref) < TOL
    s = stats.double().sum(0).cpu()
    assert rel(s[0], y_ref.detach().sum((0, 2, 3))) < TOL * 10
    dx = ops.aggregate_project_bwd_data(dyg, ag, wg, tuple(x.shape))
    assert rel(dx, x.grad) < TOL
    # fused epilogue operands: dx = prior + grad + add1*[mask1>0] + add2*[mask2>0]  (unit residual / identity `down`)
    a1, m1, a2, m2 = [rnd(g, N, C, T, V).float().to(dev) for _ in range(4)]
    prior = rnd(g, N, C, T, V).float().to(dev)
    ref2 = x.grad.float().to(dev) + prior + a1 * (m1 > 0) + a2 * (m2 > 0)
    dx2 = ops.aggregate_project_bwd_data(dyg, ag, wg, tuple(x.shape), out=prior.clone(), accumulate=True, add1=a1,
                                         mask1=m1, add2=a2, mask2=m2)
    assert rel(dx2, ref2) < TOL
    dx3 = ops.aggregate_project_bwd_data(dyg, ag, wg, tuple(x.shape), add1=a1)
    assert rel(dx3, x.grad.float().to(dev) + a1) < TOL
    # the same masks as sign bit masks (what the product path passes): bit-identical result

    def pack(m):
        b = np.packbits((m.flatten() > 0).cpu().numpy(), bitorder='little')
        b = np.concatenate([b, np.zeros((-len(b)) % 4, np.uint8)])
        return torch.from_numpy(b.view(np.int32).copy()).to(dev)
    dx4 = ops.aggregate_project_bwd_data(dyg, ag, wg, tuple(x.shape), out=prior.clone(), accumulate=True, add1=a1,
                                         mask1=pack(m1), add2=a2, mask2=pack(m2))
    assert torch.equal(dx4, dx2)
    # the adaptive branch's 1x1 term fused into the same pass: dx += wab^T dtp
    if ops.fused_bwd_data_supported(C, Cout, V):
        K2 = 6 * (Cout // 4)
        dtp = rnd(g, N, K2, T, V)
        wab = rnd(g, K2, C, scale=1.0 / np.sqrt(K2))
        ref5 = x.grad + torch.einsum('kc,nktv->nctv', wab, dtp)
        dx5 = ops.aggregate_project_bwd_data(dyg, ag, wg, tuple(x.shape), dtp=dtp.float().to(dev),
                                             wab=wab.float().to(dev).view(K2, C, 1, 1))
        assert rel(dx5, ref5) < TOL
    dw = ops.project_bwd_weight(dyg, xg, ag, Cout)
    assert rel(dw, wcat.grad) < TOL
    # adjacency gradient via the slot partials
    L = ops._L()
    from agcn_amd import lib
    nslots = L.agcn_dadj_num_slots(C, V, T)
    dpart = torch.empty((N, 3, nslots, V, V), device=dev)
    ws, nb = ops._gcn_ws(C, Cout, T, V, xg)
    lib.check(L.agcn_gcn_dadj(lib.ptr(dyg), lib.ptr(wg), lib.ptr(xg), lib.ptr(dpart), ws.data_ptr(), nb, N, C, Cout, T,
                              V, lib.stream()), 'dadj')
    assert rel(dpart.sum(2), adj.grad) < TOL


ADJ_CASES = [(2, 16, 23, 25, 1.0), (2, 32, 12, 25, 8.0), (3, 64, 9, 18, 4.0), (2, 16, 300, 25, 30.0)]


@pytest.mark.parametrize('case', ADJ_CASES)
def test_adjacency_fwd_bwd(case):
    from agcn_amd import ops
    dev = _gpu()
    N, Ci, T, V, scale = case
    g = torch.Generator().manual_seed(11 + Ci + T)
    tp = rnd(g, N, 6 * Ci, T, V, scale=scale).requires_grad_(True)
    A = rnd(g, 3, V, V, scale=0.2)
    PA = rnd(g, 3, V, V, scale=0.05).requires_grad_(True)
    adjs, Ps = [], []
    for i in range(3):
        th = tp[:, (2 * i) * Ci:(2 * i + 1) * Ci].permute(0, 3, 1, 2).reshape(N, V, Ci * T)
        ph = tp[:, (2 * i + 1) * Ci:(2 * i + 2) * Ci].reshape(N, Ci * T, V)
        p = torch.softmax(torch.matmul(th, ph) / (Ci * T), dim=-2)
        Ps.append(p)
        adjs.append(p + A[i] + PA[i])
    adj_ref = torch.stack(adjs, 1)
    tpg, Ag, PAg = tp.detach().float().to(dev), A.float().to(dev), PA.detach().float().to(dev)
    P, adj = ops.adjacency_fwd(tpg, Ag, PAg)
    assert rel(P, torch.stack(Ps, 1)) < TOL
    assert rel(adj, adj_ref) < TOL
    # backward: feed a synthetic dadj through a one-slot partial buffer
    dadj = rnd(g, N, 3, V, V)
    adj_ref.backward(dadj)
    from agcn_amd import lib
    L = ops._L()
    dpart = dadj.float().to(dev).view(N, 3, 1, V, V).contiguous()
    dadj_o, dS, dPA = (torch.empty((N, 3, V, V), device=dev), torch.empty((N, 3, V, V), device=dev),
                       torch.empty((3, V, V), device=dev))
    lib.check(L.agcn_adjacency_bwd_softmax(lib.ptr(dpart), lib.ptr(P), None, lib.ptr(dadj_o), lib.ptr(dS),
                                           lib.ptr(dPA), None, N, Ci, T, V, 1, lib.stream()), 'bwd_softmax')
    assert rel(dPA, PA.grad) < TOL
    nt = L.agcn_scores_num_tiles(V, T)
    dtp = torch.empty_like(tpg)
    dbpart = torch.empty((N * nt, 6 * Ci), device=dev)
    db = torch.empty((6 * Ci,), device=dev)
    scratch = ops._scratch(6 * Ci, tpg)
    lib.check(L.agcn_adjacency_bwd_scores(lib.ptr(tpg), lib.ptr(dS), lib.ptr(dtp), lib.ptr(dbpart),
                                          scratch.data_ptr(), lib.ptr(db), N, Ci, T, V, lib.stream()), 'bwd_scores')
    gmax = max(1e-30, float(tp.grad.abs().max()))
    assert float((dtp.double().cpu() - tp.grad).abs().max()) / gmax < 2e-4
    db_ref = tp.grad.sum((0, 2, 3))
    assert float((db.double().cpu() - db_ref).abs().max()) / max(1e-30, float(db_ref.abs().max())) < 2e-3


# (N, C, Ci, T, V, weight scale): every (TM, NSUB) instantiation, partial last tiles, V = 18, K not a multiple of 16
ADJ_FUSED_CASES = [(2, 64, 16, 23, 25, 4.0), (2, 64, 32, 12, 25, 6.0), (2, 128, 64, 9, 18, 6.0), (1, 64, 16, 300, 25, 8.0),
                   (2, 128, 32, 31, 18, 6.0), (2, 256, 64, 75, 25, 8.0), (2, 3, 16, 17, 25, 1.0), (2, 40, 16, 10, 25, 3.0)]


@pytest.mark.parametrize('case', ADJ_FUSED_CASES)
def test_adjacency_fused_fwd_bwd(case):
    """theta/phi never stored: P/adj straight from x (reference agcn.py:99-102) and the recomputing backward."""
    from agcn_amd import ops, lib
    dev = _gpu()
    N, C, Ci, T, V, scale = case
    if not ops.adjacency_fused_supported(C, Ci, T, V):
        pytest.skip("fused adjacency not built for this shape / AGCN_GEMM mode")
    g = torch.Generator().manual_seed(5 + C + Ci + T)
    x = rnd(g, N, C, T, V).requires_grad_(True)
    wab = rnd(g, 6 * Ci, C, scale=scale / np.sqrt(C)).requires_grad_(True)
    bab = rnd(g, 6 * Ci, scale=0.3).requires_grad_(True)
    A = rnd(g, 3, V, V, scale=0.2)
    PA = rnd(g, 3, V, V, scale=0.05)
    tp = torch.einsum('oc,nctv->notv', wab, x) + bab.view(1, -1, 1, 1)
    tp.retain_grad()
    adjs, Ps = [], []
    for i in range(3):
        th = tp[:, (2 * i) * Ci:(2 * i + 1) * Ci].permute(0, 3, 1, 2).reshape(N, V, Ci * T)
        ph = tp[:, (2 * i + 1) * Ci:(2 * i + 2) * Ci].reshape(N, Ci * T, V)
        p = torch.softmax(torch.matmul(th, ph) / (Ci * T), dim=-2)
        Ps.append(p)
        adjs.append(p + A[i] + PA[i])
    adj_ref = torch.stack(adjs, 1)
    xg, wg, bg = x.detach().float().to(dev), wab.detach().float().to(dev), bab.detach().float().to(dev)
    P, adj = ops.adjacency_fused_fwd(xg, wg, bg, A.float().to(dev), PA.float().to(dev))
    assert rel(P, torch.stack(Ps, 1)) < TOL
    assert rel(adj, adj_ref) < TOL
    P2, adj2, tpk = ops.adjacency_fused_fwd(xg, wg, bg, A.float().to(dev), PA.float().to(dev), keep_tp=True)
    assert torch.equal(P2, P) and torch.equal(adj2, adj)          # the by-product copy does not change the result
    assert rel(tpk, tp) < TOL
    # backward of the scores: dS from the reference softmax backward, then the recomputing kernel
    dadj = rnd(g, N, 3, V, V)
    adj_ref.backward(dadj)
    Pd = torch.stack(Ps, 1).detach()
    dS = Pd * (dadj - (Pd * dadj).sum(2, keepdim=True)) / (Ci * T)
    L = ops._L()
    nt = L.agcn_scores_num_tiles(V, T)
    dtp = torch.empty((N, 6 * Ci, T, V), device=dev)
    dbpart = torch.empty((N * nt, 6 * Ci), device=dev)
    db = torch.empty((6 * Ci,), device=dev)
    scratch = ops._scratch(6 * Ci, xg)
    nb = L.agcn_adjacency_fused_workspace(C, Ci)
    ws = ops._ws(nb, xg)
    lib.check(L.agcn_adjacency_fused_bwd_scores(lib.ptr(xg), lib.ptr(wg), lib.ptr(bg), lib.ptr(dS.float().to(dev)),
                                                lib.ptr(dtp), lib.ptr(dbpart), scratch.data_ptr(), lib.ptr(db),
                                                ws.data_ptr(), nb, N, C, Ci, T, V, lib.stream()), 'fused_bwd_scores')
    gmax = max(1e-30, float(tp.grad.abs().max()))
    assert float((dtp.double().cpu() - tp.grad).abs().max()) / gmax < 2e-4
    db_ref = tp.grad.sum((0, 2, 3))
    assert float((db.double().cpu() - db_ref).abs().max()) / max(1e-30, float(db_ref.abs().max())) < 2e-3


ADJ_WS_CASES = [(2, 64, 16, 47, 25, 1), (2, 64, 16, 47, 25, 2), (2, 64, 32, 35, 25, 1), (2, 128, 32, 61, 18, 2),
                (3, 64, 16, 300, 25, 4), (2, 128, 32, 75, 25, 1), (2, 64, 16, 5, 25, 1)]


@pytest.mark.parametrize('case', ADJ_WS_CASES)
def test_adj_ws_multi_tile(case, monkeypatch):
    """The persistent adjacency forward (csrc/adj_ws.hip) with several tiles per workgroup (the partial scores stay in
    registers across tiles), ragged last tiles and frame splits; with and without the producer's max |x|; large and
    tiny activations (the f16x3 range scaling)."""
    from agcn_amd import ops, lib
    dev = _gpu()
    N, C, Ci, T, V, nsplit = case
    L = ops._L()
    if L.agcn_gemm_mode().decode() != 'bf16x6' or L.agcn_chain_mode().decode() != 'f16x3':
        pytest.skip("persistent adjacency kernel runs in the default arithmetic mode only")
    monkeypatch.setenv('AGCN_AW_SPLIT', str(nsplit))
    for mag in (1.0, 3e4, 1e-5):
        g = torch.Generator().manual_seed(11 + C + Ci + T)
        x = rnd(g, N, C, T, V) * mag
        wab = rnd(g, 6 * Ci, C, scale=5.0 / np.sqrt(C) * min(1.0 / mag, 16.0))   # (weights stay below 256: F16_W_SCALE)
        bab = rnd(g, 6 * Ci, scale=0.3)
        A, PA = rnd(g, 3, V, V, scale=0.2), rnd(g, 3, V, V, scale=0.05)
        tp = torch.einsum('oc,nctv->notv', wab, x) + bab.view(1, -1, 1, 1)
        Ps = []
        for i in range(3):
            th = tp[:, (2 * i) * Ci:(2 * i + 1) * Ci].permute(0, 3, 1, 2).reshape(N, V, Ci * T)
            ph = tp[:, (2 * i + 1) * Ci:(2 * i + 2) * Ci].reshape(N, Ci * T, V)
            Ps.append(torch.softmax(torch.matmul(th, ph) / (Ci * T), dim=-2))
        P_ref = torch.stack(Ps, 1)
        xg, wg, bg = x.float().to(dev), wab.float().to(dev), bab.float().to(dev)
        Ag, PAg = A.float().to(dev), PA.float().to(dev)
        P, adj, tpk = ops.adjacency_fused_fwd(xg, wg, bg, Ag, PAg, keep_tp=True)
        assert L.agcn_last_kernel().decode().startswith('adj_ws_kernel'), L.agcn_last_kernel()
        assert rel(P, P_ref) < TOL, (mag, rel(P, P_ref))
        assert rel(adj, P_ref + A + PA) < TOL
        assert rel(tpk, tp) < TOL
        amax = xg.abs().max().reshape(1)
        out = torch.empty(1, device=dev)
        P2, adj2 = ops.adjacency_fused_fwd(xg, wg, bg, Ag, PAg, x_amax=amax, x_amax_out=out)
        assert torch.equal(P2, P) and torch.equal(adj2, adj) and torch.equal(out, amax)
        out.zero_()
        P3, _ = ops.adjacency_fused_fwd(xg, wg, bg, Ag, PAg, x_amax_out=out)
        assert torch.equal(P3, P) and torch.equal(out, amax)
        for _ in range(5):                                      # bit-for-bit on repeats
            P4, adj4, tp4 = ops.adjacency_fused_fwd(xg, wg, bg, Ag, PAg, keep_tp=True, x_amax=amax)
            assert torch.equal(P4, P) and torch.equal(adj4, adj) and torch.equal(tp4, tpk)


BN_CASES = [(2, 64, 23, 25, 0), (2, 64, 23, 25, 1), (3, 128, 11, 25, 2), (2, 256, 7, 25, 2), (2, 64, 17, 18, 1)]


@pytest.mark.parametrize('case', BN_CASES)
def test_bn_act_fwd_bwd(case):
    from agcn_amd import ops
    dev = _gpu()
    N, C, T, V, mode = case
    g = torch.Generator().manual_seed(3 + C + mode)
    y1 = (rnd(g, N, C, T, V) * 1.5 + 0.3).requires_grad_(True)
    r = rnd(g, N, C, T, V).requires_grad_(True)
    g1, b1 = (rnd(g, C, scale=0.3) + 1).requires_grad_(True), rnd(g, C, scale=0.1).requires_grad_(True)
    g2, b2 = (rnd(g, C, scale=0.3) + 1).requires_grad_(True), rnd(g, C, scale=0.1).requires_grad_(True)
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    rm2, rv2 = rm.clone(), rv.clone()
    z = F.batch_norm(y1, rm, rv, g1, b1, True, 0.1, 1e-5)
    if mode == 1:
        z = z + r
    elif mode == 2:
        z = z + F.batch_norm(r, rm2, rv2, g2, b2, True, 0.1, 1e-5)
    out_ref = F.relu(z)
    dout = rnd(g, N, C, T, V)
    out_ref.backward(dout)
    f = lambda t: t.detach().float().to(dev).contiguous()  # noqa: E731
    y1g, rg = f(y1), f(r)
    # statistics partials computed here with torch on the GPU (the contraction epilogues are tested elsewhere)
    def part(t):
        return torch.stack([t.sum((0, 2, 3)), (t * t).sum((0, 2, 3))]).view(1, 2, C).contiguous()
    rmg, rvg = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    st1 = ops.bn_train_coeffs(part(y1g), N * T * V, f(g1), f(b1), rmg, rvg)
    assert rel(rmg, rm) < TOL and rel(rvg, rv) < TOL
    st2 = None
    if mode == 2:
        rm2g, rv2g = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        st2 = ops.bn_train_coeffs(part(rg), N * T * V, f(g2), f(b2), rm2g, rv2g)
    out = ops.bn_act_fwd(y1g, st1, None if mode == 0 else rg, st2, relu=True)
    assert rel(out, out_ref) < TOL
    dy1, dg1, db1, dy2, dg2, db2 = ops.bn_bwd(f(dout), out, y1g, f(g1), st1, rg if mode == 2 else None,
                                              f(g2) if mode == 2 else None, st2)
    assert rel(dy1, y1.grad) < TOL
    assert rel(dg1, g1.grad) < TOL * 5 and rel(db1, b1.grad) < TOL * 5
    if mode == 2:
        assert rel(dy2, r.grad) < TOL
        assert rel(dg2, g2.grad) < TOL * 5 and rel(db2, b2.grad) < TOL * 5
    # the sign bit mask of `out` (what the product path keeps for the backward) gives bit-identical results
    out2, bits = ops.bn_act_fwd(y1g, st1, None if mode == 0 else rg, st2, relu=True, want_bits=True)
    assert torch.equal(out2, out)
    flat = (out.flatten() > 0)
    words = bits.cpu().numpy().view(np.uint32)
    unpacked = np.unpackbits(words.view(np.uint8), bitorder='little')[:flat.numel()].astype(bool)
    assert np.array_equal(unpacked, flat.cpu().numpy())
    res_b = ops.bn_bwd(f(dout), bits, y1g, f(g1), st1, rg if mode == 2 else None, f(g2) if mode == 2 else None, st2)
    for a_, b_ in zip(res_b, (dy1, dg1, db1, dy2, dg2, db2)):
        assert (a_ is None and b_ is None) or torch.equal(a_, b_)


def test_sgd_step_matches_torch():
    from agcn_amd import lib, ops
    dev = _gpu()
    L = ops._L()
    g = torch.Generator().manual_seed(5)
    n = 100003
    p0 = rnd(g, n).float()
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([p_ref], lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4)
    p = p0.to(dev)
    buf = torch.zeros(n, device=dev)
    ws = torch.empty(L.agcn_sgd_step_workspace(n) // 4 + 1, device=dev)
    norm = torch.zeros(2, device=dev)
    for step in range(3):
        grad = rnd(g, n, scale=0.05).float()
        p_ref.grad = grad.clone()
        gn = torch.nn.utils.clip_grad_norm_([p_ref], 1.0)
        opt.step()
        gg = grad.to(dev)
        lib.check(L.agcn_sgd_step(lib.ptr(p), lib.ptr(gg), lib.ptr(buf), n, 0.1, 0.9, 1e-4, 1, 1.0, 1.0,
                                  int(step == 0), lib.ptr(ws), ws.numel() * 4, lib.ptr(norm), lib.stream()), 'sgd')
        assert abs(float(norm[0]) - float(gn)) < 1e-4 * float(gn)
        assert rel(p, p_ref) < 1e-5


# N, C(=M), T, V, stride: 9-tap weight gradient on the split-precision kernel (wgrad9_bf16.hip; f16x3 in the default
# mode): 64/128-row tiles, ragged last chunk (T % 80 != 0), T not a multiple of 4 (zero padded rows), V = 18, stride 2
WGRAD9_CASES = [(2, 64, 23, 25, 1), (2, 128, 85, 25, 1), (3, 64, 300, 18, 1), (2, 256, 75, 25, 1), (1, 128, 150, 25, 1),
                (2, 128, 85, 25, 2), (2, 128, 300, 18, 2)]


@pytest.mark.parametrize('case', WGRAD9_CASES)
def test_conv9_weight_gradient_split(case):
    from agcn_amd import ops
    dev = _gpu()
    N, C, T, V, stride = case
    g = torch.Generator().manual_seed(17 + C + T)
    L = ops._L()
    for dy_mag, x_mag in ((1.0, 1.0), (1e-6, 50.0), (3e3, 1e-3)):       # gradients of 1e-6, large activations: range scales
        x = rnd(g, N, C, T, V) * x_mag
        w = rnd(g, C, C, 9, 1, scale=1.0 / np.sqrt(9 * C)).requires_grad_(True)
        y = F.conv2d(x, w, None, stride=(stride, 1), padding=(4, 0))
        dy = rnd(g, *y.shape) * dy_mag
        y.backward(dy)
        dyg, xg = dy.float().to(dev), x.float().to(dev)
        dw = ops.conv_bwd_weight(dyg, xg, tuple(w.shape), stride)
        name = L.agcn_last_kernel().decode()
        assert name.startswith('wgrad9_') or L.agcn_gemm_mode().decode() != 'bf16x6', name
        assert rel(dw, w.grad) < TOL, (dy_mag, x_mag, rel(dw, w.grad))
        # the producers' maxima handed in: the same bits as with the kernel's own reduction pass; and on repeats
        dw2 = ops.conv_bwd_weight(dyg, xg, tuple(w.shape), stride, dyg.abs().max().reshape(1), xg.abs().max().reshape(1))
        assert torch.equal(dw2, dw)
        assert torch.equal(ops.conv_bwd_weight(dyg, xg, tuple(w.shape), stride), dw)


@pytest.mark.parametrize('case', [(2, 64, 23, 25), (3, 128, 30, 18), (2, 64, 300, 25), (2, 32, 7, 25)])
def test_stc_attention_gates_fwd_bwd(case):
    """The fused AAGCN attention node (HIP reductions + apply) vs the reference's three gates in fp64 tensor code
    (aagcn.py:59-116, 268-270): output, dy and all eight parameter gradients."""
    from agcn_amd import ops
    dev = _gpu()
    N, C, T, V = case
    g = torch.Generator().manual_seed(23 + C + T)
    ks = V if V % 2 else V - 1
    y = rnd(g, N, C, T, V).requires_grad_(True)
    par = [rnd(g, 1, C, ks, scale=0.3 / np.sqrt(C)), rnd(g, 1, scale=0.1), rnd(g, 1, C, 9, scale=0.3 / np.sqrt(C)),
           rnd(g, 1, scale=0.1), rnd(g, C // 2, C, scale=1 / np.sqrt(C)), rnd(g, C // 2, scale=0.1),
           rnd(g, C, C // 2, scale=1 / np.sqrt(C)), rnd(g, C, scale=0.1)]
    par = [p.requires_grad_(True) for p in par]
    sa_w, sa_b, ta_w, ta_b, f1w, f1b, f2w, f2b = par
    se = torch.sigmoid(F.conv1d(y.mean(-2), sa_w, sa_b, padding=(ks - 1) // 2))
    y1 = y * se.unsqueeze(-2) + y
    se = torch.sigmoid(F.conv1d(y1.mean(-1), ta_w, ta_b, padding=4))
    y2 = y1 * se.unsqueeze(-1) + y1
    se = torch.sigmoid(F.linear(F.relu(F.linear(y2.mean(-1).mean(-1), f1w, f1b)), f2w, f2b))
    ref = y2 * se.unsqueeze(-1).unsqueeze(-1) + y2
    dout = rnd(g, N, C, T, V)
    ref.backward(dout)
    yg = y.detach().float().to(dev).requires_grad_(True)
    pg = [p.detach().float().to(dev).requires_grad_(True) for p in par]
    out = ops.STCAttentionFunction.apply(yg, *pg)
    out.backward(dout.float().to(dev))
    assert rel(out, ref) < TOL
    assert rel(yg.grad, y.grad) < TOL
    for a, b in zip(pg, par):
        den = max(1e-30, float(b.grad.abs().max()))
        assert float((a.grad.double().cpu() - b.grad).abs().max()) / den < 2e-4, (tuple(b.shape),)


INFER_GCN_CASES = [
    # N, C, Cout, T, V, conv `down`
    (2, 64, 64, 23, 25, False),
    (2, 64, 128, 21, 25, True),
    (2, 128, 128, 12, 18, False),
    (2, 128, 256, 9, 25, True),
]


@pytest.mark.parametrize('case', INFER_GCN_CASES)
def test_gcn_unit_infer(case):
    """y = relu(bias + sum_i W_i (x . adj_i) + residual), residual = x (identity `down`) or a folded 1x1 conv on x that
    runs as extra plain stages of the same kernel (BN-folded inference, reference agcn.py:103-109 in eval mode)."""
    from agcn_amd import ops
    dev = _gpu()
    N, C, Cout, T, V, has_down = case
    g = torch.Generator().manual_seed(23 + C + T)
    x = rnd(g, N, C, T, V)
    adj = rnd(g, N, 3, V, V, scale=0.3)
    wcat = rnd(g, Cout, 3 * C, scale=1.0 / np.sqrt(3 * C))
    bias = rnd(g, Cout, scale=0.1)
    y_ref = _gcn_ref(x, adj, wcat, bias)
    if has_down:
        w2 = rnd(g, Cout, C, scale=1.0 / np.sqrt(C))
        y_ref = y_ref + torch.einsum('oc,nctv->notv', w2, x)
    else:
        y_ref = y_ref + x
    y_ref = torch.relu(y_ref)
    xg, ag, wg, bg = [t.float().to(dev) for t in (x, adj, wcat, bias)]
    if has_down:
        y = ops.gcn_unit_infer(xg, ag, wg, bg, x2=xg, w2=w2.float().to(dev).contiguous())
    else:
        y = ops.gcn_unit_infer(xg, ag, wg, bg, res=xg)
    if y is None:
        pytest.skip("fused inference kernels not available in this AGCN_GEMM mode")
    assert rel(y, y_ref) < TOL


INFER_CONV_CASES = [
    # N, Cin, Cout, T, V, stride, residual, relu
    (2, 64, 64, 23, 25, 1, True, True),
    (3, 64, 128, 21, 25, 2, True, True),
    (2, 128, 128, 12, 25, 1, False, False),
    (2, 256, 256, 9, 18, 1, True, True),
]


@pytest.mark.parametrize('case', INFER_CONV_CASES)
def test_conv9_infer(case):
    """y = act(b + conv9x1(x) [+ res]): unit_tcn + the TCN_GCN_unit tail with the BatchNorm folded (agcn.py:48-50,127-129)."""
    from agcn_amd import ops
    dev = _gpu()
    N, Cin, Cout, T, V, stride, has_res, relu = case
    g = torch.Generator().manual_seed(29 + Cin + T)
    x = rnd(g, N, Cin, T, V)
    w = rnd(g, Cout, Cin, 9, 1, scale=1.0 / np.sqrt(9 * Cin))
    b = rnd(g, Cout, scale=0.1)
    y_ref = F.conv2d(x, w, b, stride=(stride, 1), padding=(4, 0))
    res = rnd(g, *y_ref.shape) if has_res else None
    if has_res:
        y_ref = y_ref + res
    if relu:
        y_ref = torch.relu(y_ref)
    y = ops.conv9_infer(x.float().to(dev), w.float().to(dev), b.float().to(dev),
                        None if res is None else res.float().to(dev), relu=relu, stride=stride)
    if y is None:
        pytest.skip("fused inference kernels not available in this AGCN_GEMM mode")
    assert rel(y, y_ref) < TOL


FIRST_CASES = [(8, 3, 64, 300, 25), (3, 3, 64, 23, 25), (2, 3, 64, 37, 18), (2, 2, 48, 11, 20)]


@pytest.mark.parametrize('case', FIRST_CASES)
def test_gcn_first_layer_forward(case):
    """First-layer unit_gcn forward (3 input channels): aggregate+project and the `down` convolution in one pass, with the
    (sum, sumsq) partials of both; bitwise reproducible (reference agcn.py:103-108 with in_channels = 3)."""
    from agcn_amd import ops
    dev = _gpu()
    N, C, Cout, T, V = case
    g = torch.Generator().manual_seed(31 + T + V)
    x = rnd(g, N, C, T, V)
    adj = rnd(g, N, 3, V, V, scale=0.3)
    wcat = rnd(g, Cout, 3 * C, scale=1.0 / np.sqrt(3 * C))
    bias = rnd(g, Cout, scale=0.1)
    wdown = rnd(g, Cout, C, 1, 1, scale=1.0 / np.sqrt(C))
    bdown = rnd(g, Cout, scale=0.1)
    y_ref = _gcn_ref(x, adj, wcat, bias)
    d_ref = torch.einsum('oc,nctv->notv', wdown.view(Cout, C), x) + bdown.view(1, -1, 1, 1)
    args = [t.float().to(dev) for t in (x, adj, wcat, bias, wdown, bdown)]
    y, st, d, st2 = ops.gcn_first_fwd(*args, want_stats=True)
    y2, st_b, d2, st2_b = ops.gcn_first_fwd(*args, want_stats=True)
    assert torch.equal(y, y2) and torch.equal(st, st_b) and torch.equal(d, d2) and torch.equal(st2, st2_b)
    assert rel(y, y_ref) < TOL and rel(d, d_ref) < TOL
    for slab, ref in ((st, y_ref), (st2, d_ref)):
        s = slab.double().sum(0).cpu()
        assert rel(s[0], ref.detach().sum((0, 2, 3))) < TOL * 10
        assert rel(s[1], (ref.detach() ** 2).sum((0, 2, 3))) < TOL * 10


def test_producer_consumer_kernels_are_bitwise_reproducible():
    """The producer / consumer kernels hand tiles over through LDS rings and counted waits: a missing wait shows up as a
    run-to-run difference, not as a large error.  Repeated launches on the same operands must agree bit for bit (temporal
    conv forward / backward-data at a 128-row shape, projection and 1x1 weight gradients)."""
    from agcn_amd import ops
    dev = _gpu()
    g = torch.Generator().manual_seed(41)
    N, C, T, V = 16, 256, 75, 25
    x = rnd(g, N, C, T, V).float().to(dev)
    w9 = rnd(g, C, C, 9, 1, scale=1.0 / np.sqrt(9 * C)).float().to(dev)
    b = rnd(g, C, scale=0.1).float().to(dev)
    dy = rnd(g, N, C, T, V).float().to(dev)
    adj = rnd(g, N, 3, V, V, scale=0.3).float().to(dev)
    w1 = (C // 4 * 6, C, 1, 1)
    dy1 = rnd(g, N, w1[0], T, V).float().to(dev)
    junk = torch.empty(64 << 20, device=dev)             # churn the allocator / caches between repeats
    ref = None
    for rep in range(6):
        junk.normal_()
        cur = (ops.conv_fwd(x, w9, b, 1, want_stats=True)[0],
               ops.conv_bwd_data(dy, w9, (N, C, T, V), 1),
               ops.project_bwd_weight(dy, x, adj, C),
               ops.conv_bwd_weight(dy1, x, w1, 1))
        torch.cuda.synchronize()
        if ref is None:
            ref = [t.clone() for t in cur]
        else:
            for k, (a_, b_) in enumerate(zip(ref, cur)):
                assert torch.equal(a_, b_), f'output {k} differs in repeat {rep}'


# ---- f16x3 range scaling: the split-fp16 kernels multiply their streamed operand by a power of two taken from
# the tensor's maximum, so operands far outside fp16's range (activations of 1e4, gradients of 1e-6) or spanning many
# decades must come out as accurately as O(1) ones.  Errors here are measured against the fp64 result's own scale (not
# max(1, .) as elsewhere in this file).  Maxima supplied by the caller (what the BatchNorm passes leave behind in the
# product path) and maxima taken by the kernels' own pre-pass must give bit-identical results.
def _rel_strict(a, ref):
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    return float((a - ref).abs().max() / ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(1e4, 1e-6, False), (3e-5, 2e3, False), (1.0, 1.0, True), (7e4, 1e-7, True)])
def test_f16x3_range_scaling(case):
    from agcn_amd import ops, lib
    dev = _gpu()
    sx, sdy, wide = case
    N, C, Cout, T, V = 2, 128, 128, 12, 25
    g = torch.Generator().manual_seed(11)
    x = rnd(g, N, C, T, V) * sx
    dy = rnd(g, N, Cout, T, V) * sdy
    if wide:     # seven decades inside one tensor: every other channel 1e-4 / 1e-7 of the rest
        x[:, ::2] *= 1e-4
        dy[:, 1::2] *= 1e-7
    x.requires_grad_(True)
    adj = rnd(g, N, 3, V, V, scale=0.3).requires_grad_(True)
    wcat = rnd(g, Cout, 3 * C, scale=1.0 / np.sqrt(3 * C)).requires_grad_(True)
    bias = rnd(g, Cout, scale=0.1 * sx)
    y_ref = _gcn_ref(x, adj, wcat, bias)
    y_ref.backward(dy)
    xg, ag, wg, bg, dyg = [t.detach().float().to(dev) for t in (x, adj, wcat, bias, dy)]
    x_amax, dy_amax = xg.abs().max().reshape(1), dyg.abs().max().reshape(1)
    tol = 2e-5
    y, _ = ops.aggregate_project_fwd(xg, ag, wg, bg)
    assert _rel_strict(y, y_ref) < tol
    y2, _ = ops.aggregate_project_fwd(xg, ag, wg, bg, x_amax=x_amax)
    assert torch.equal(y, y2)
    dx = ops.aggregate_project_bwd_data(dyg, ag, wg, tuple(x.shape))
    assert _rel_strict(dx, x.grad) < tol
    assert torch.equal(dx, ops.aggregate_project_bwd_data(dyg, ag, wg, tuple(x.shape), dy_amax=dy_amax))
    # the fused 1x1 term shares the scale with dy: a dtp four decades above and below it
    K2 = 6 * (Cout // 4)
    wab = rnd(g, K2, C, scale=1.0 / np.sqrt(K2))
    for sd in (1e4, 1e-4) if ops.fused_bwd_data_supported(C, Cout, V) else ():
        dtp = rnd(g, N, K2, T, V) * sdy * sd
        ref = x.grad + torch.einsum('kc,nktv->nctv', wab, dtp)
        dtpg = dtp.float().to(dev)
        dx5 = ops.aggregate_project_bwd_data(dyg, ag, wg, tuple(x.shape), dtp=dtpg, wab=wab.float().to(dev).view(K2, C, 1, 1),
                                             dy_amax=dy_amax, dtp_amax=dtpg.abs().max().reshape(1))
        assert _rel_strict(dx5, ref) < tol
    # weight gradients: bf16x6 without the maxima, f16x3 with both
    for kw in ({}, dict(dy_amax=dy_amax, x_amax=x_amax)):
        dw = ops.project_bwd_weight(dyg, xg, ag, Cout, **kw)
        assert _rel_strict(dw, wcat.grad) < tol
        dw1 = ops.conv_bwd_weight(dyg, xg, (Cout, C, 1, 1), **kw)
        assert _rel_strict(dw1.view(Cout, C), torch.einsum('notv,nctv->oc', dy, x.detach())) < tol
    # adjacency gradient
    L = ops._L()
    nslots = L.agcn_dadj_num_slots(C, V, T)
    dpart = torch.empty((N, 3, nslots, V, V), device=dev)
    ws, nb = ops._gcn_ws(C, Cout, T, V, xg)
    for xm in (None, x_amax):           # max |x| taken by a pass inside / handed in: the same bits
        lib.check(L.agcn_gcn_dadj_ex(lib.ptr(dyg), lib.ptr(wg), lib.ptr(xg), lib.ptr(dpart), ws.data_ptr(), nb, N, C, Cout,
                                     T, V, lib.ptr(dy_amax), lib.ptr(xm), lib.stream()), 'dadj')
        assert _rel_strict(dpart.sum(2), adj.grad) < tol, (xm is not None, _rel_strict(dpart.sum(2), adj.grad))
        if xm is None:
            first = dpart.clone()
        else:
            assert torch.equal(dpart, first)
    # temporal convolution (forward and backward-data run on f16x3 too)
    w9 = rnd(g, Cout, C, 9, 1, scale=1.0 / np.sqrt(9 * C))
    z_ref = torch.nn.functional.conv2d(x.detach(), w9, None, padding=(4, 0))
    z, _ = ops.conv_fwd(xg, w9.float().to(dev), torch.zeros(Cout, device=dev), 1, x_amax=x_amax)
    assert _rel_strict(z, z_ref) < tol
    dz_ref = torch.nn.functional.conv_transpose2d(dy, w9, None, padding=(4, 0))
    dz = ops.conv_bwd_data(dyg, w9.float().to(dev), tuple(x.shape), 1, dy_amax=dy_amax)
    assert _rel_strict(dz, dz_ref) < tol


@pytest.mark.gpu
def test_absmax_pass_is_exact():
    """agcn_absmax (the pass the f16x3 kernels run when no producer supplied the maximum) returns exactly max |x| for any
    length and any 4-byte alignment (16-byte loads from the first aligned element, scalar head and tail)."""
    from agcn_amd import ops, lib
    dev = _gpu()
    L = ops._L()
    g = torch.Generator().manual_seed(3)
    for n in (1, 3, 5, 1000, 76800, 76801, (1 << 20) + 3):
        for off in (0, 1, 2, 3):
            base = (torch.randn(n + 4, generator=g) * 1e-6).to(dev)
            x = base[off:off + n]
            out = torch.zeros(1, device=dev)
            lib.check(L.agcn_absmax(x.data_ptr(), n, out.data_ptr(), lib.stream()), 'absmax')
            assert float(out) == float(x.abs().max()), (n, off)


@pytest.mark.parametrize('case', [(4, 3, 20, 25, 2), (3, 3, 17, 18, 2), (2, 3, 9, 25, 1)])
def test_data_bn_prologue_fwd_bwd(case):
    """Model prologue (reference agcn.py:163-165): permute/view -> BatchNorm1d(M*V*C) over (N, T) -> view/permute, as the
    deterministic HIP kernels (ops.DataBNFunction) vs the same tensor code in fp64: output, dx, dweight, dbias and the
    running statistics, train and eval."""
    from agcn_amd import ops
    dev = _gpu()
    N, C, T, V, M = case
    g = torch.Generator().manual_seed(5 + T)
    x = (rnd(g, N, C, T, V, M) * 2 + 0.5).requires_grad_(True)
    w = (0.5 + torch.rand(M * V * C, generator=g, dtype=torch.float64)).requires_grad_(True)
    b = rnd(g, M * V * C, scale=0.3).requires_grad_(True)
    rm0, rv0 = rnd(g, M * V * C, scale=0.2), 0.5 + torch.rand(M * V * C, generator=g, dtype=torch.float64)
    r = rnd(g, N * M, C, T, V)

    def ref(training):
        rm, rv = rm0.clone(), rv0.clone()
        h = x.permute(0, 4, 3, 1, 2).contiguous().view(N, M * V * C, T)
        h = F.batch_norm(h, rm, rv, w, b, training, 0.1, 1e-5)
        return h.view(N, M, V, C, T).permute(0, 1, 3, 4, 2).contiguous().view(N * M, C, T, V), rm, rv
    for training in (True, False):
        for t in (x, w, b):
            t.grad = None
        yr, rmr, rvr = ref(training)
        xg = x.detach().float().to(dev).requires_grad_(True)
        wg = w.detach().float().to(dev).requires_grad_(True)
        bg = b.detach().float().to(dev).requires_grad_(True)
        rm, rv = rm0.float().to(dev), rv0.float().to(dev)
        y = ops.DataBNFunction.apply(xg, wg, bg, rm, rv, training, None)
        assert rel(y, yr) < TOL
        assert rel(rm, rmr) < TOL and rel(rv, rvr) < TOL
        if training:
            (yr * r).sum().backward()
            (y * r.float().to(dev)).sum().backward()
            assert rel(xg.grad, x.grad) < TOL
            assert rel(wg.grad, w.grad) < TOL and rel(bg.grad, b.grad) < TOL


@pytest.mark.parametrize('case', [(3, 2, 64, 7, 25, 60), (2, 2, 256, 5, 18, 400), (4, 1, 32, 3, 25, 7)])
def test_pool_fc_head_fwd_bwd(case):
    """Model epilogue (agcn.py:179-183): x.view(N, M, C, -1).mean(3).mean(1) -> Linear, ops.PoolFCFunction vs fp64."""
    from agcn_amd import ops
    dev = _gpu()
    N, M, C, T, V, K = case
    g = torch.Generator().manual_seed(77 + C)
    x = rnd(g, N * M, C, T, V).requires_grad_(True)
    w = rnd(g, K, C, scale=0.1).requires_grad_(True)
    b = rnd(g, K, scale=0.1).requires_grad_(True)
    r = rnd(g, N, K)
    ref = F.linear(x.view(N, M, C, -1).mean(3).mean(1), w, b)
    (ref * r).sum().backward()
    xg, wg, bg = [t.detach().float().to(dev).requires_grad_(True) for t in (x, w, b)]
    out = ops.PoolFCFunction.apply(xg, wg, bg, M)
    (out * r.float().to(dev)).sum().backward()
    assert rel(out, ref) < TOL
    for a, b_ in ((xg, x), (wg, w), (bg, b)):
        den = max(1e-30, float(b_.grad.abs().max()))
        assert float((a.grad.double().cpu() - b_.grad).abs().max()) / den < 2e-4


@pytest.mark.parametrize('case', [(3, 64, 25, 25), (2, 128, 40, 9), (2, 256, 75, 9), (2, 32, 18, 17), (5, 16, 300, 9)])
def test_gate_conv_and_linear_kernels(case):
    """agcn_gate_conv_* (Conv1d(C -> 1, Ks, 'same') + 1 + sigmoid, aagcn.py:72-76 / 92-96) and agcn_linear_* (act 0/1/2)
    vs fp64 tensor code, forward and every gradient; and twice in a row bit for bit."""
    from agcn_amd import ops
    dev = _gpu()
    N, C, L, Ks = case
    g = torch.Generator().manual_seed(31 + C + L)
    x = rnd(g, N, C, L).requires_grad_(True)
    w = rnd(g, 1, C, Ks, scale=0.5 / np.sqrt(C)).requires_grad_(True)
    b = rnd(g, 1, scale=0.1).requires_grad_(True)
    da = rnd(g, N, L)
    ref = 1 + torch.sigmoid(F.conv1d(x, w, b, padding=(Ks - 1) // 2)).squeeze(1)
    (ref * da).sum().backward()
    xg, wg, bg = [t.detach().float().to(dev) for t in (x, w, b)]
    a = ops.gate_conv_fwd(xg, wg, bg)
    assert rel(a, ref) < TOL
    dx, dw, db = ops.gate_conv_bwd(da.float().to(dev), a, xg, wg)
    dx2, dw2, db2 = ops.gate_conv_bwd(da.float().to(dev), a, xg, wg)
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(db, db2)
    for got, want in ((dx, x.grad), (dw, w.grad), (db, b.grad)):
        den = max(1e-30, float(want.abs().max()))
        assert float((got.double().cpu() - want).abs().max()) / den < 2e-4
    for act in (0, 1, 2):
        xi = rnd(g, N, C).requires_grad_(True)
        wl = rnd(g, C // 2, C, scale=1 / np.sqrt(C)).requires_grad_(True)
        bl = rnd(g, C // 2, scale=0.1).requires_grad_(True)
        pre = F.linear(xi, wl, bl)
        ref = pre if act == 0 else (F.relu(pre) if act == 1 else 1 + torch.sigmoid(pre))
        do = rnd(g, N, C // 2)
        (ref * do).sum().backward()
        xi_g, wl_g, bl_g = [t.detach().float().to(dev) for t in (xi, wl, bl)]
        out = ops.linear_fwd(xi_g, wl_g, bl_g, act)
        assert rel(out, ref) < TOL
        din, dwl, dbl = ops.linear_bwd(do.float().to(dev), out, xi_g, wl_g, act)
        for got, want in ((din, xi.grad), (dwl, wl.grad), (dbl, bl.grad)):
            den = max(1e-30, float(want.abs().max()))
            assert float((got.double().cpu() - want).abs().max()) / den < 2e-4


@pytest.mark.parametrize('case', [(2, 64, 64, 44, 25, 1), (2, 64, 64, 44, 25, 2), (3, 64, 128, 24, 25, 1),
                                  (2, 64, 64, 20, 18, 1), (2, 64, 64, 37, 25, 1), (2, 3, 64, 40, 25, 1),
                                  (2, 128, 128, 28, 25, 1), (2, 128, 256, 21, 25, 2), (2, 128, 128, 22, 18, 1),
                                  (2, 96, 64, 16, 25, 1)])
def test_ws_chain_multi_tile(case, monkeypatch):
    """The persistent weight-stationary chain kernel (gcn_ws_kernel: 64 streamed channels) with SEVERAL 8-frame tiles per
    workgroup (AGCN_WS_SPLIT pins the frame splits per sample; small test batches would otherwise get one tile per
    workgroup): forward + BatchNorm partials, backward-data with every fused epilogue operand and the fused 1x1 term, on
    16-byte-aligned rows (T*V % 4 == 0) and unaligned ones, a partial last tile, V = 18, and a 3-row block (first layer's
    backward).  Against fp64 tensor code."""
    from agcn_amd import ops
    dev = _gpu()
    N, C, Cout, T, V, nsplit = case
    monkeypatch.setenv('AGCN_WS_SPLIT', str(nsplit))
    g = torch.Generator().manual_seed(100 + C + T + V)
    x = rnd(g, N, C, T, V).requires_grad_(True)
    adj = rnd(g, N, 3, V, V, scale=0.3)
    wcat = rnd(g, Cout, 3 * C, scale=1.0 / np.sqrt(3 * C))
    bias = rnd(g, Cout, scale=0.1)
    y_ref = _gcn_ref(x, adj, wcat, bias)
    dy = rnd(g, *y_ref.shape)
    y_ref.backward(dy)
    xg, ag, wg, bg, dyg = [t.detach().float().to(dev) for t in (x, adj, wcat, bias, dy)]
    L = ops._L()
    # (AGCN_GEMM=f32 / bf16 runs take the generic kernels)
    ws_mode = L.agcn_chain_mode().decode() == 'f16x3' and L.agcn_gemm_mode().decode() == 'bf16x6'
    if C >= 32:
        y, stats = ops.aggregate_project_fwd(xg, ag, wg, bg, want_stats=True)
        assert (not ws_mode) or 'gcn_ws_kernel' in L.agcn_last_kernel().decode(), L.agcn_last_kernel()
        assert rel(y, y_ref) < TOL
        s = stats.double().sum(0).cpu()
        assert rel(s[0], y_ref.detach().sum((0, 2, 3))) < TOL * 10
        assert rel(s[1], (y_ref.detach() ** 2).sum((0, 2, 3))) < TOL * 10
        y2, _ = ops.aggregate_project_fwd(xg, ag, wg, bg, want_stats=False)
        assert torch.equal(y, y2)
    if Cout != 64:
        return
    # backward-data: streamed source = dy (64 channels)
    a1, m1, a2, m2 = [rnd(g, N, C, T, V).float().to(dev) for _ in range(4)]
    prior = rnd(g, N, C, T, V).float().to(dev)

    def pack(m):
        b = np.packbits((m.flatten() > 0).cpu().numpy(), bitorder='little')
        b = np.concatenate([b, np.zeros((-len(b)) % 4, np.uint8)])
        return torch.from_numpy(b.view(np.int32).copy()).to(dev)
    dx = ops.aggregate_project_bwd_data(dyg, ag, wg, tuple(x.shape))
    assert (not ws_mode) or 'gcn_ws_kernel' in L.agcn_last_kernel().decode(), L.agcn_last_kernel()
    assert rel(dx, x.grad) < TOL
    ref2 = x.grad.float().to(dev) + prior + a1 * (m1 > 0) + a2 * (m2 > 0)
    dx2 = ops.aggregate_project_bwd_data(dyg, ag, wg, tuple(x.shape), out=prior.clone(), accumulate=True, add1=a1,
                                         mask1=m1, add2=a2, mask2=m2)
    assert rel(dx2, ref2) < TOL
    dx4 = ops.aggregate_project_bwd_data(dyg, ag, wg, tuple(x.shape), out=prior.clone(), accumulate=True, add1=a1,
                                         mask1=pack(m1), add2=a2, mask2=pack(m2))
    assert torch.equal(dx4, dx2)
    if C >= 32 and ops.fused_bwd_data_supported(C, Cout, V):
        K2 = 6 * (Cout // 4)
        dtp = rnd(g, N, K2, T, V)
        wab = rnd(g, K2, C, scale=1.0 / np.sqrt(K2))
        ref5 = x.grad + torch.einsum('kc,nktv->nctv', wab, dtp) + (a1 * (m1 > 0)).double().cpu()
        dx5 = ops.aggregate_project_bwd_data(dyg, ag, wg, tuple(x.shape), add1=a1, mask1=pack(m1),
                                             dtp=dtp.float().to(dev), wab=wab.float().to(dev).view(K2, C, 1, 1))
        assert (not ws_mode) or 'gcn_ws_kernel<2, 2, 3>' in L.agcn_last_kernel().decode(), L.agcn_last_kernel()
        assert rel(dx5, ref5) < TOL


@pytest.mark.parametrize('case', [(8, 64, 64, 64, 25), (8, 128, 128, 48, 25), (6, 128, 256, 32, 25), (8, 64, 128, 56, 18)])
def test_ws_chain_repeats_bit_for_bit(case, monkeypatch):
    """The persistent chain kernel synchronises its matrix and store waves with one or two workgroup barriers per tile and
    keeps its operands in flight across them: a missing wait would show as run-to-run differences.  Forty launches of the
    forward (BatchNorm partials included) and of the backward-data on the same operands must agree bit for bit, with
    several tiles per workgroup."""
    from agcn_amd import ops
    dev = _gpu()
    N, C, Cout, T, V = case
    monkeypatch.setenv('AGCN_WS_SPLIT', '2')
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, C, T, V, generator=g).to(dev)
    adj = (0.3 * torch.randn(N, 3, V, V, generator=g)).to(dev)
    w = (torch.randn(Cout, 3 * C, generator=g) / np.sqrt(3 * C)).to(dev)
    b = (0.1 * torch.randn(Cout, generator=g)).to(dev)
    dy = torch.randn(N, Cout, T, V, generator=g).to(dev)
    y0, s0 = ops.aggregate_project_fwd(x, adj, w, b, want_stats=True)
    dx0 = ops.aggregate_project_bwd_data(dy, adj, w, tuple(x.shape), add1=x, mask1=x)
    torch.cuda.synchronize()
    for _ in range(40):
        y, s = ops.aggregate_project_fwd(x, adj, w, b, want_stats=True)
        dx = ops.aggregate_project_bwd_data(dy, adj, w, tuple(x.shape), add1=x, mask1=x)
        assert torch.equal(y, y0) and torch.equal(s, s0)
        assert torch.equal(dx, dx0)


@pytest.mark.parametrize('case', [(2, 64, 64, 24, 25), (2, 256, 256, 9, 25), (2, 128, 128, 16, 18)])
def test_f16x3_adjacency_headroom(case):
    """The f16x3 graph chain splits G = x . A^ (not x), so its range scale must leave room for the adjacency: PA is an
    unconstrained trained parameter.  With column / row sums of |A^| around 1500 (round 2's scale overflowed fp16 beyond
    128) the forward, the backward-data and the projection's weight gradient still match fp64 (persistent 64- and
    128-channel kernels and the tile-per-workgroup one at 256 channels)."""
    from agcn_amd import ops
    dev = _gpu()
    N, C, Cout, T, V = case
    g = torch.Generator().manual_seed(41 + C)
    x = rnd(g, N, C, T, V).requires_grad_(True)
    adj = rnd(g, N, 3, V, V, scale=75.0)
    assert float(adj.abs().sum(-2).max()) > 1000 and float(adj.abs().sum(-1).max()) > 1000
    wcat = rnd(g, Cout, 3 * C, scale=1.0 / np.sqrt(3 * C)).requires_grad_(True)
    bias = rnd(g, Cout, scale=0.1)
    y_ref = _gcn_ref(x, adj, wcat, bias)
    dy = rnd(g, *y_ref.shape)
    y_ref.backward(dy)
    xg, ag, wg, bg, dyg = [t.detach().float().to(dev) for t in (x, adj, wcat, bias, dy)]
    y, _ = ops.aggregate_project_fwd(xg, ag, wg, bg, want_stats=True)
    assert torch.isfinite(y).all()
    assert rel(y, y_ref) < TOL
    dx = ops.aggregate_project_bwd_data(dyg, ag, wg, tuple(x.shape))
    assert torch.isfinite(dx).all()
    assert rel(dx, x.grad) < TOL
    amax_x = xg.abs().max().reshape(1)
    amax_dy = dyg.abs().max().reshape(1)
    dw = ops.project_bwd_weight(dyg, xg, ag, Cout, amax_dy, amax_x)       # (both maxima: the f16x3 weight-gradient kernel)
    assert torch.isfinite(dw).all()
    assert rel(dw, wcat.grad) < TOL
