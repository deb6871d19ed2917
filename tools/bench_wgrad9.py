"""Time unit_tcn's 9-tap weight gradient per layer shape (N'=128, V=25).  AGCN_WGRAD9_BF16=0 selects the f32 kernel.
    python tools/bench_wgrad9.py [layers...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import agcn_amd
from agcn_amd import ops
dev = torch.device('cuda:0')
SHAPES = {'l2': (64, 300, 1), 'l5': (128, 300, 2), 'l6': (128, 150, 1), 'l8': (256, 150, 2), 'l9': (256, 75, 1)}
layers = sys.argv[1:] or list(SHAPES)
reps = int(os.environ.get('REPS', '5'))
N, V = 128, 25
g = torch.Generator().manual_seed(0)
for name in layers:
    C, T, stride = SHAPES[name]
    To = (T - 1) // stride + 1
    x = torch.randn(N, C, T, V, generator=g).to(dev)
    dy = torch.randn(N, C, To, V, generator=g).to(dev)
    am_dy, am_x = dy.abs().max().reshape(1), x.abs().max().reshape(1)      # (left behind by the producers in the step)
    fn = lambda: ops.conv_bwd_weight(dy, x, (C, C, 9, 1), stride, am_dy, am_x)
    for _ in range(2):
        fn()
    kern = ops._L().agcn_last_kernel().decode()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / reps * 1e3
    fl = 2.0 * C * C * 9 * N * To * V
    print(f'{name} C{C} T{T} s{stride}: {us:8.0f} us  {fl / us / 1e6:6.1f} TF  [{kern}]', flush=True)
