import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import agcn_amd
from agcn_amd import ops
dev = torch.device('cuda:0')
def rel(a, ref):
    a = a.detach().double().cpu(); ref = ref.detach().double().cpu()
    return float((a - ref).abs().max() / max(1.0, float(ref.abs().max())))
cases = []
for (cin, cout) in [(64, 64), (128, 128), (256, 256), (64, 128), (128, 256)]:
    for V in (25, 18):
        for T in (9, 16, 31):
            for taps in (9, 1):
                for stride in (1, 2):
                    cases.append((4, cin, cout, T, V, taps, stride))
g = torch.Generator().manual_seed(0)
for case in cases:
    N, Cin, Cout, T, V, taps, stride = case
    x = torch.randn(N, Cin, T, V, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, taps, 1, generator=g, dtype=torch.float64) / np.sqrt(Cin * taps)).requires_grad_(True)
    pad = (taps - 1) // 2
    y = F.conv2d(x, w, None, stride=(stride, 1), padding=(pad, 0))
    dy = torch.randn(*y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    xg, wg, dyg = x.detach().float().to(dev), w.detach().float().to(dev), dy.float().to(dev)
    yo, _ = ops.conv_fwd(xg, wg, None, stride)
    dx = ops.conv_bwd_data(dyg, wg, tuple(x.shape), stride)
    dw = ops.conv_bwd_weight(dyg, xg, tuple(w.shape), stride)
    e = (rel(yo, y), rel(dx, x.grad), rel(dw, w.grad))
    flag = '' if max(e) < 1e-4 else '   <<<<<< FAIL'
    print(case, 'fwd %.1e dgrad %.1e wgrad %.1e' % e, flag)
