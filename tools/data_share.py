"""Input-side share of a training step at full size: synthetic NTU-shaped dataset in the reference's on-disk
format (.npy (N,3,300,25,2) + label pickle), read by feeders.feeder.Feeder with the reference's training transforms
(random_choose window 300 of 300... here: random_shift + random_move + random_choose), delivered through the pinned
double-buffered H2D ring and augmented on the GPU; prints the three-bucket timer of Processor.train
(dataloader / model / statistics, reference utils/processor.py:759-775) and clips/s.
    python tools/data_share.py [--samples 512] [--workers 4] [--host-augment]"""
import argparse, os, pickle, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import agcn_amd  # noqa: F401
from agcn_amd.processor import Processor, load_args

ap = argparse.ArgumentParser()
ap.add_argument('--samples', type=int, default=1536)
ap.add_argument('--workers', type=int, default=8)
ap.add_argument('--host-augment', action='store_true', help='per-sample numpy transforms on the host (the reference way)')
a = ap.parse_args()
tmp = tempfile.mkdtemp(prefix='agcn_data_')
rng = np.random.default_rng(0)
data = rng.standard_normal((a.samples, 3, 300, 25, 2)).astype(np.float32)
L = rng.integers(50, 300, a.samples)
for i, l in enumerate(L):
    data[i, :, l:] = 0                                  # real NTU clips are zero padded (data_gen/ntu_gendata.py:161-172)
np.save(os.path.join(tmp, 'train_data_joint.npy'), data)
with open(os.path.join(tmp, 'train_label.pkl'), 'wb') as f:
    pickle.dump(([f'S{i:06d}' for i in range(a.samples)], [int(x) for x in rng.integers(0, 60, a.samples)]), f)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
argv = ['--config', os.path.join(root, 'config', 'nturgbd-cross-view', 'train_joint.yaml'), '--work-dir', tmp,
        '--model-saved-name', '', '--feeder', 'feeders.feeder.Feeder', '--batch-size', '64', '--num-epoch', '1',
        '--log-interval', '1000', '--print-log', 'False', '--num-worker', str(a.workers),
        '--device-augment', 'False' if a.host_augment else 'True']
arg = load_args(argv)
arg.train_feeder_args = dict(data_path=os.path.join(tmp, 'train_data_joint.npy'), label_path=os.path.join(tmp, 'train_label.pkl'),
                             window_size=300, random_choose=True, random_shift=True, random_move=True)
arg.test_feeder_args = dict(arg.train_feeder_args, random_choose=False, random_shift=False, random_move=False)
p = Processor(arg)
p.train(0)                                             # warm-up epoch (kernel first-launch costs, worker start-up)
t0 = time.perf_counter()
p.train(1)
dt = time.perf_counter() - t0
t = p.last_timer
tot = sum(t.values())
steps = a.samples // 64
print(f"data_share: {steps} steps of 64 clips, {steps * 64 / dt:.1f} clips/s; dataloader {100 * t['dataloader'] / tot:.1f}% "
      f"model {100 * t['model'] / tot:.1f}% statistics {100 * t['statistics'] / tot:.1f}%  "
      f"({'host numpy' if a.host_augment else 'device'} augmentation, {a.workers} loader workers)")
