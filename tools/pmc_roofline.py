"""Launch the kernels whose HBM traffic we want from PMC counters, plus two calibration kernels with known byte counts
(MI355X_MICROARCH.md: FETCH_SIZE under-reports wide streaming reads by 2x on gfx950; other widths are uncalibrated).
Run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- python tools/pmc_roofline.py
and again with --pmc WRITE_SIZE (separate passes: the TCC block has 4 counter slots)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import agcn_amd
from agcn_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
N, C, T, V = 128, 64, 300, 25
x = torch.randn(N, C, T, V, generator=g).to(dev)
dy = torch.randn(N, C, T, V, generator=g).to(dev)
w9 = (torch.randn(C, C, 9, 1, generator=g) / 24).to(dev)
b = torch.zeros(C, device=dev)
gam = torch.ones(C, device=dev)
for rep in range(2):
    # calibration A: 4-byte-per-lane loads, 3 x 245.76 MB read, ~0 written  (bn_bwd_reduce_kernel<false>)
    # calibration B: 16-byte-per-lane, 2 x 245.76 MB read + 245.76 MB written (bn_act_fwd_kernel<1,true>)
    st = ops.BNState(); st.mean = torch.zeros(C, device=dev); st.invstd = torch.ones(C, device=dev)
    st.scale = torch.ones(C, device=dev); st.shift = torch.zeros(C, device=dev)
    out = ops.bn_act_fwd(x, st, dy, None, relu=True)
    ops.bn_bwd(dy, out, x, gam, st)
    # kernels of interest at the l2-l4 shape (N'=128, C=64, T=300, V=25)
    y, stats = ops.conv_fwd(x, w9, b, 1, want_stats=True)          # conv_gemm_kernel<9,...> fwd: 70.8 GFLOP
    ops.conv_bwd_weight(dy, x, tuple(w9.shape), 1)                 # conv_wgrad_kernel<9,...>: 70.8 GFLOP
    ops.conv_bwd_data(dy, w9, tuple(x.shape), 1)                   # conv_gemm_kernel<9,...> bwd-data
# dominant instantiation of the step: unit_tcn forward at the l9/l10 shape (N'=128, C=256, T=75, V=25): 283.1 GFLOP
x9 = torch.randn(128, 256, 75, 25, generator=g).to(dev)
w99 = (torch.randn(256, 256, 9, 1, generator=g) / 48).to(dev)
b9 = torch.zeros(256, device=dev)
for rep in range(2):
    ops.conv_fwd(x9, w99, b9, 1, want_stats=True)
# unit_gcn aggregate+project (gcn_chain_kernel) forward / backward-data and the adjacency gradient at the same shape
adj9 = (0.2 * torch.randn(128, 3, 25, 25, generator=g)).to(dev)
wd9 = (torch.randn(256, 768, generator=g) / 27.7).to(dev)
for rep in range(2):
    ops.aggregate_project_fwd(x9, adj9, wd9, b9, want_stats=True)
    ops.aggregate_project_bwd_data(x9, adj9, wd9, tuple(x9.shape))
    ops.adjacency_bwd  # (dadj is launched through the C entry point below)
    from agcn_amd import lib
    L = ops._L(); ns = L.agcn_dadj_num_slots(256, 25, 75)
    dpart = torch.empty((128, 3, ns, 25, 25), device=dev)
    ws, nb = ops._gcn_ws(256, 256, 75, 25, x9)
    lib.check(L.agcn_gcn_dadj(lib.ptr(x9), lib.ptr(wd9), lib.ptr(x9), lib.ptr(dpart), ws.data_ptr(), nb, 128, 256, 256,
                              75, 25, lib.stream()), 'dadj')
torch.cuda.synchronize()
print('done')
