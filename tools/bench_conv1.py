"""Time the 1x1 convolutions (theta/phi, down, residual) forward / backward-data per layer shape (N'=128, V=25)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import agcn_amd
from agcn_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
def timed(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
N, V = 128, 25
for name, C, M, T, s in [('l2 thetaphi', 64, 96, 300, 1), ('l6 thetaphi', 128, 192, 150, 1), ('l9 thetaphi', 256, 384, 75, 1),
                         ('l5 down', 64, 128, 300, 1), ('l8 down', 128, 256, 150, 1), ('l8 res s2', 128, 256, 150, 2)]:
    x = torch.randn(N, C, T, V, generator=g).to(dev)
    w = (torch.randn(M, C, 1, 1, generator=g) / C ** 0.5).to(dev)
    b = torch.randn(M, generator=g).to(dev)
    To = (T - 1) // s + 1
    dy = torch.randn(N, M, To, V, generator=g).to(dev)
    fl = 2.0 * C * M * To * V * N
    y, _ = ops.conv_fwd(x, w, b, s, want_stats=True)
    ref = torch.nn.functional.conv2d(x[:8].double(), w.double(), b.double(), stride=(s, 1))
    e1 = rel(y[:8], ref)
    us = timed(lambda: ops.conv_fwd(x, w, b, s, want_stats=True))
    out = f'{name:12s} C{C}->{M} T{T} s{s}: fwd {us:6.0f} us {fl/us/1e6:6.1f} TF err {e1:.1e}'
    if s == 1:
        dx = ops.conv_bwd_data(dy, w, tuple(x.shape), s)
        refx = torch.nn.functional.conv_transpose2d(dy[:8].double(), w.double())
        e2 = rel(dx[:8], refx)
        us = timed(lambda: ops.conv_bwd_data(dy, w, tuple(x.shape), s, accumulate=False))
        out += f'  bwd {us:6.0f} us {fl/us/1e6:6.1f} TF err {e2:.1e}'
    print(out, flush=True)
