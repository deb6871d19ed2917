"""Eval-mode (inference) throughput of the AGCN joint model, NTU shape, batch 64, synthetic input resident in HBM:
the BN-folded chain (adjacency + two kernels per unit; AGCN_INFER_FOLD=1, default) against the unfused eval passes.
    python tools/infer_bench.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device('cuda:0')
model = bench.build_model('ntu_agcn').to(dev).eval()
x = torch.randn(64, 3, 300, 25, 2, device=dev)
for fold in ('1', '0'):
    os.environ['AGCN_INFER_FOLD'] = fold
    with torch.no_grad():
        for _ in range(3):
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            model(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    print('AGCN_INFER_FOLD=%s: %.2f ms per batch of 64, %.0f clips/s' % (fold, dt * 1e3, 64 / dt), flush=True)
