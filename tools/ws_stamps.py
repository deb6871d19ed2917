"""In-kernel wall-clock stamps of gcn_ws_kernel (AGCN_WS_DBG bit 8; the stamps overwrite a few outputs): per role, the time
from kernel entry to the end of the prologue, the tile loop, and the tail, averaged over the workgroups (units: 10 ns)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import agcn_amd
from agcn_amd import ops
dev = torch.device('cuda:0')
N, C, Cout, T, V = 128, 64, 64, 300, 25
g = torch.Generator().manual_seed(0)
x = torch.randn(N, C, T, V, generator=g).to(dev)
adj = (0.2 * torch.randn(N, 3, V, V, generator=g)).to(dev)
w = (torch.randn(Cout, 3 * C, generator=g) / (3 * C) ** 0.5).to(dev)
b = torch.zeros(Cout, device=dev)
for dbg in sys.argv[1:] or ['8']:
    os.environ['AGCN_WS_DBG'] = dbg
    for _ in range(3):
        y, _ = ops.aggregate_project_fwd(x, adj, w, b, want_stats=False)
    torch.cuda.synchronize()
    flat = y.flatten()
    im = (flat == -12345.0).nonzero().flatten()
    is_ = (flat == -54321.0).nonzero().flatten()
    m = torch.stack([flat[im - 3], flat[im - 2]], 1).cpu() / 100.0
    cyc = flat[im - 1].cpu()
    brk = torch.stack([flat[im + 1], flat[im + 2], flat[im + 3], flat[im + 4]], 1).cpu()
    s = torch.stack([flat[is_ - 3], flat[is_ - 2], flat[is_ - 1]], 1).cpu() / 100.0
    print(f'dbg={dbg}: {len(im)} workgroups; matrix role: prologue {m[:,0].mean():.1f} us (max {m[:,0].max():.1f}), loop '
          f'{m[:,1].mean():.1f} us (max {m[:,1].max():.1f}); store role: prologue {s[:,0].mean():.1f}, loop {s[:,1].mean():.1f} '
          f'(max {s[:,1].max():.1f}), tail {s[:,2].mean():.1f} (max {s[:,2].max():.1f}); loop cycles {cyc.mean():.0f} -> '
          f'{float(cyc.mean()) / float(m[:,1].mean()) / 1e3:.2f} GHz; wave 0 cycles: stages {brk[:,0].mean():.0f}, barrier C '
          f'{brk[:,1].mean():.0f}, O write {brk[:,2].mean():.0f}, barrier E {brk[:,3].mean():.0f}')
