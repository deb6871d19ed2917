import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import agcn_amd
from agcn_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
for (N, T, V) in [(8, 300, 25), (3, 23, 25), (2, 37, 18)]:
    C, Cout = 3, 64
    x = torch.randn(N, C, T, V, generator=g).to(dev)
    adj = (0.2 * torch.randn(N, 3, V, V, generator=g)).to(dev)
    w = (torch.randn(Cout, 3 * C, generator=g) / 3).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    wd = torch.randn(Cout, C, 1, 1, generator=g).to(dev)
    bd = torch.randn(Cout, generator=g).to(dev)
    outs = [ops.gcn_first_fwd(x, adj, w, b, wd, bd, want_stats=True) for _ in range(3)]
    torch.cuda.synchronize()
    for k in range(4):
        same = all(torch.equal(outs[0][k], o[k]) for o in outs[1:])
        print((N, T, V), 'output', k, 'bitwise equal across runs:', same)
    xd, ad = x.double(), adj.double()
    y = sum(torch.einsum('oc,nctv->notv', w.double()[:, i * C:(i + 1) * C], torch.einsum('nctu,nuv->nctv', xd, ad[:, i])) for i in range(3)) + b.double().view(1, -1, 1, 1)
    d = torch.einsum('oc,nctv->notv', wd.double().view(Cout, C), xd) + bd.double().view(1, -1, 1, 1)
    print('  err y %.2e d %.2e  stats y %.2e %.2e' % (float((outs[0][0].double() - y).abs().max() / y.abs().max()),
          float((outs[0][2].double() - d).abs().max() / d.abs().max()),
          float((outs[0][1].double().sum(0)[0] - y.sum((0, 2, 3))).abs().max() / y.sum((0, 2, 3)).abs().max()),
          float((outs[0][1].double().sum(0)[1] - (y * y).sum((0, 2, 3))).abs().max() / (y * y).sum((0, 2, 3)).abs().max())))

import time
N, T, V, C, Cout = 128, 300, 25, 3, 64
x = torch.randn(N, C, T, V, device=dev); adj = 0.2 * torch.randn(N, 3, V, V, device=dev)
w = torch.randn(Cout, 9, device=dev); b = torch.randn(Cout, device=dev)
wd = torch.randn(Cout, C, 1, 1, device=dev); bd = torch.randn(Cout, device=dev)
def timed(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print('gcn_first_fwd (ypre + dpre + stats): %.0f us' % timed(lambda: ops.gcn_first_fwd(x, adj, w, b, wd, bd, want_stats=True)))
print('generic: aggregate_project_fwd %.0f us + conv_fwd(down) %.0f us' % (
    timed(lambda: ops.aggregate_project_fwd(x, adj, w, b, want_stats=True)),
    timed(lambda: ops.conv_fwd(x, wd, bd, want_stats=True))))
