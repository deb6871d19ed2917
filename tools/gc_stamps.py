"""In-kernel cycle stamps of gcn_chain (AGCN_GC_DBG=8): cycles wave 0 of one mid-grid workgroup spends per section.
    AGCN_GC_DBG=8 python tools/gc_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import agcn_amd
from agcn_amd import ops
dev = torch.device('cuda:0')
N, V = 128, 25
for name, (C, Cout, T) in {'l2': (64, 64, 300), 'l6': (128, 128, 150), 'l9': (256, 256, 75)}.items():
    x = torch.randn(N, C, T, V, device=dev)
    adj = 0.2 * torch.randn(N, 3, V, V, device=dev)
    w = torch.randn(Cout, 3 * C, device=dev) / (3 * C) ** 0.5
    b = torch.zeros(Cout, device=dev)
    for _ in range(3):
        y, _ = ops.aggregate_project_fwd(x, adj, w, b, want_stats=True)
    torch.cuda.synchronize()
    flat = y.flatten()
    idx = (flat == -12345.0).nonzero()
    if idx.numel() == 0:
        print(name, 'no stamps found'); continue
    i = int(idx[0]) - 6
    v = flat[i:i + 6].tolist()
    S = int(v[5])
    print('%s: loop %8.0f cyc (%d stages: %.0f per stage) = staging %.0f + matrix %.0f + barrier %.0f per stage; epilogue %.0f'
          % (name, v[0], S, v[0] / S, v[1] / S, v[2] / S, v[3] / S, v[4]), flush=True)
