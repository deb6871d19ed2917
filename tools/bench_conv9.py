"""Time unit_tcn's 9x1 convolution forward / backward-data alone at the model's layer shapes (HIP events).
    python tools/bench_conv9.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import agcn_amd
from agcn_amd import ops
dev = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shapes = [('l2-4', 64, 64, 300, 1), ('l5', 64, 128, 300, 2), ('l6-7', 128, 128, 150, 1), ('l8', 128, 256, 150, 2),
          ('l9-10', 256, 256, 75, 1)]
N, V = 128, 25
print('AGCN_CB_DBG=%s' % os.environ.get('AGCN_CB_DBG', '0'))
for name, cin, cout, T, stride in shapes:
    x = torch.randn(N, cin, T, V, device=dev)
    w = torch.randn(cout, cin, 9, 1, device=dev) / (cin * 9) ** 0.5
    b = torch.zeros(cout, device=dev)
    To = (T - 1) // stride + 1
    dy = torch.randn(N, cout, To, V, device=dev)
    def timeit(f):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    tf = timeit(lambda: ops.conv_fwd(x, w, b, stride))
    kf = agcn_amd.lib.load().agcn_last_kernel().decode()
    tb = timeit(lambda: ops.conv_bwd_data(dy, w, (N, cin, T, V), stride))
    fl = 2.0 * cin * cout * 9 * N * To * V
    print('%-6s fwd %.3f ms (%5.1f TF)  bwd-data %.3f ms (%5.1f TF)  %s' % (name, tf, fl / tf / 1e9, tb, fl / tb / 1e9, kf))
