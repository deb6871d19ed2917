"""``Feeder``: the reference's skeleton dataset (``feeders/feeder.py:35-227``, the AGCN/AAGCN branch: ``.npy`` clips
``(N, C, T, V, M)`` memory-mapped + a ``(sample_name, label)`` pickle, ``data_gen/ntu_gendata.py:158-173``), same
constructor flags, same order of the per-sample transforms in ``__getitem__`` (:182-221), ``top_k`` (:224-227).
The SGN-specific branches (pickled SGN arrays, ``joint_15`` remapping) are out of this path's scope.

With ``device_augment=True`` the random transforms are NOT applied per sample on the host: ``__getitem__`` returns the
raw clip and ``DeviceAugment.from_feeder(feeder)`` applies the same transforms to whole batches on the GPU."""
import pickle

import numpy as np
from torch.utils.data import Dataset

from . import tools


class Feeder(Dataset):
    def __init__(self, data_path, label_path, dataset='NTU60-CV', random_choose=False, random_shift=False,
                 random_move=False, window_size=-1, normalization=False, random_zaxis_flip=False,
                 random_xaxis_scale=False, random_yaxis_scale=False, random_subsample=None, random_rotation=False,
                 stretch=False, debug=False, use_mmap=True, device_augment=False, **_):
        self.data_path, self.label_path, self.dataset = data_path, label_path, dataset
        self.random_choose, self.random_shift, self.random_move = random_choose, random_shift, random_move
        self.window_size, self.normalization = window_size, normalization
        self.random_zaxis_flip = random_zaxis_flip
        self.random_xaxis_scale, self.random_yaxis_scale = random_xaxis_scale, random_yaxis_scale
        self.random_subsample, self.random_rotation, self.stretch = random_subsample, random_rotation, stretch
        self.debug, self.use_mmap, self.device_augment = debug, use_mmap, device_augment
        self.load_data()
        if normalization:
            self.get_mean_map()

    def load_data(self):
        if self.label_path.endswith('.npy'):
            self.label = np.load(self.label_path).astype(np.int64)
            self.sample_name = np.arange(len(self.label))
        else:
            with open(self.label_path, 'rb') as f:       # the user's own dataset file (names, labels)
                try:
                    self.sample_name, self.label = pickle.load(f)
                except UnicodeDecodeError:               # pickles written by python 2 (reference feeder.py:139-143)
                    f.seek(0)
                    self.sample_name, self.label = pickle.load(f, encoding='latin1')
        self.data = np.load(self.data_path, mmap_mode='r' if self.use_mmap else None)
        if self.debug:
            self.label, self.data, self.sample_name = self.label[:100], self.data[:100], self.sample_name[:100]

    def get_mean_map(self):
        """per-(channel, joint) mean over samples/frames/persons and std (reference feeder.py:166-174)"""
        data = np.asarray(self.data)
        N, C, T, V, M = data.shape
        self.mean_map = data.mean(axis=2, keepdims=True).mean(axis=4, keepdims=True).mean(axis=0)
        self.std_map = data.transpose((0, 2, 4, 1, 3)).reshape((N * T * M, C * V)).std(axis=0).reshape((C, 1, V, 1))

    def rotation_theta(self):
        """reference feeder.py:211-219"""
        if 'NTU60' in self.dataset:
            return 0.5 if 'CV' in self.dataset else 0.3
        return 0.3

    def __len__(self):
        return len(self.label)

    def __getitem__(self, index):
        clip = np.array(self.data[index])
        label = self.label[index]
        if self.device_augment:                 # transforms happen on the GPU, batch-wise (device.DeviceAugment)
            return clip.astype(np.float32, copy=False), label, index
        if self.stretch:
            clip = tools.stretch_to_maximum_length(clip)
        if self.normalization:
            clip = (clip - self.mean_map) / self.std_map
        if self.random_shift:
            clip = tools.random_shift(clip)
        if self.random_choose:
            clip = tools.random_choose(clip, self.window_size)
        elif self.window_size > 0:
            clip = tools.auto_pading(clip, self.window_size)
        if self.random_move:
            clip = tools.random_move(clip)
        if self.random_zaxis_flip:
            clip = tools.random_zaxis_flip(clip)
        if self.random_xaxis_scale:
            clip = tools.random_xaxis_scale(clip)
        if self.random_yaxis_scale:
            clip = tools.random_yaxis_scale(clip)
        if self.random_subsample is not None:
            assert 0 < self.random_subsample < 300
            clip = tools.random_subsample(clip, self.random_subsample)
        if self.random_rotation:
            clip = tools.random_rotation(clip, self.rotation_theta())
        return clip, label, index

    def top_k(self, score, top_k):
        rank = score.argsort()
        hit = [lab in rank[i, -top_k:] for i, lab in enumerate(self.label)]
        return sum(hit) * 1.0 / len(hit)
