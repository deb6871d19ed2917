"""Host-side enqueue time of one training step (no device sync inside the timed region) vs the GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import agcn_amd
from agcn_amd.trainer import TrainEngine, synthetic_batch
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = bench.build_model(); bench.randomize_like_training(model, 0); model.to(dev)
eng = TrainEngine(model)
x, y = synthetic_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 64, device=dev)
for _ in range(3): eng.train_step(x, y)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); eng.train_step(x, y); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'enqueue {1e3*(t1-t0):.1f} ms   until done {1e3*(t2-t0):.1f} ms')
