"""Soak: N training steps of the headline configuration; loss must stay finite and the allocator's footprint flat
(the side stream's record_stream bookkeeping must not leak blocks).
    python tools/soak.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from agcn_amd.trainer import TrainEngine, synthetic_batch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = bench.build_model('ntu_agcn')
bench.randomize_like_training(model, seed=0)
model.to(dev)
eng = TrainEngine(model, base_lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-4, max_grad_norm=1.0)
marks = []
for i in range(steps):
    data, label = synthetic_batch(64, num_point=25, num_class=60, seed=i % 7, device=dev)
    loss = eng.train_step(data, label)
    if i % 50 == 49 or i == steps - 1:
        torch.cuda.synchronize()
        marks.append((i + 1, float(loss.detach()), torch.cuda.memory_reserved() >> 20, torch.cuda.max_memory_allocated() >> 20))
        print('step %4d loss %.4f reserved %d MiB peak allocated %d MiB' % marks[-1], flush=True)
assert all(m[1] == m[1] and abs(m[1]) < 1e4 for m in marks)
assert marks[-1][2] <= marks[0][2] * 1.05 + 64, 'allocator footprint keeps growing'
print('soak ok')
