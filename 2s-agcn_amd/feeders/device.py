"""Device side of the input path: what keeps a 700+ clips/s GPU fed.

* ``DeviceLoader``: wraps any loader of host batches ``(data, label, index)``; batches are staged through PINNED host
  buffers and copied on a side HIP stream, ``depth`` batches ahead of the consumer (double-buffered by default), so the
  11.5 MB H2D copy of a 64-clip batch (~0.2 ms over PCIe Gen5) overlaps the previous training step.
* ``DeviceAugment``: the feeder's random transforms (reference ``feeders/feeder.py:196-221``, ``feeders/tools.py``) applied
  to a whole device-resident batch at once.  The per-sample random PARAMETERS are drawn on the host with the same
  generator calls the per-clip numpy versions make (``tools.py``), so a seeded batch is reproducible; the data movement
  and arithmetic (window gather, shift, per-frame affine, rotation, flips, person scaling) run as a handful of
  batched GPU operations instead of ~B x 6 numpy passes over 180 KB clips on the host cores.
"""
import random

import numpy as np
import torch

from . import tools


class DeviceLoader:
    """Iterates ``loader`` and yields device tensors; H2D copies run ``depth`` batches ahead on their own stream."""

    def __init__(self, loader, device, depth=2):
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, int(depth))
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None
        self._pinned = {}

    def __len__(self):
        return len(self.loader)

    def _stage(self, slot, batch):
        """host batch -> (device tensors, ready event); pinned staging buffers are reused per ring slot"""
        out = []
        for j, t in enumerate(batch):
            if not torch.is_tensor(t):
                t = torch.as_tensor(np.asarray(t))
            if self.stream is None:
                out.append(t.to(self.device))
                continue
            key = (slot, j)
            buf = self._pinned.get(key)
            if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                self._pinned[key] = buf
            buf.copy_(t)
            with torch.cuda.stream(self.stream):
                out.append(buf.to(self.device, non_blocking=True))
        ev = None
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def __iter__(self):
        """A feeder thread pulls host batches, stages them in pinned buffers and enqueues the H2D copies on the side
        stream, up to ``depth`` batches ahead; the consumer only pops finished (tensors, event) pairs, so the host work
        of batch k+1 (worker hand-over, 11.5 MB memcpy into pinned memory) overlaps the training step of batch k."""
        import queue
        import threading
        q = queue.Queue(maxsize=self.depth)
        nslots = self.depth + 2
        busy = [None] * nslots          # event after which a slot's pinned buffers may be overwritten
        stop = threading.Event()

        def feed():
            try:
                if self.device.type == 'cuda':
                    torch.cuda.set_device(self.device)
                slot = 0
                for batch in self.loader:
                    if stop.is_set():
                        return
                    if busy[slot] is not None:
                        busy[slot].synchronize()       # the copy that last used these pinned buffers has finished
                    tensors, ev = self._stage(slot, batch)
                    busy[slot] = ev
                    q.put((tensors, ev))
                    slot = (slot + 1) % nslots
                q.put(None)
            except BaseException as e:                 # surface loader errors in the consumer
                q.put(e)
        th = threading.Thread(target=feed, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                tensors, ev = item
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    for t in tensors:
                        t.record_stream(cur)
                yield tuple(tensors)
        finally:
            stop.set()
            while th.is_alive():                       # unblock a feeder waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)


class DeviceAugment:
    """Batched on-device version of the feeder's transforms.  ``__call__(x)``: x (B, C, T, V, M) on the GPU ->
    (B, C, T_out, V, M); the order of the transforms is the reference's (feeder.py:196-221)."""

    def __init__(self, window_size=-1, random_choose=False, random_shift=False, random_move=False,
                 random_zaxis_flip=False, random_xaxis_scale=False, random_yaxis_scale=False, random_rotation=False,
                 rotation_theta=0.3, mean_map=None, std_map=None):
        self.window_size, self.random_choose, self.random_shift = window_size, random_choose, random_shift
        self.random_move, self.random_zaxis_flip = random_move, random_zaxis_flip
        self.random_xaxis_scale, self.random_yaxis_scale = random_xaxis_scale, random_yaxis_scale
        self.random_rotation, self.rotation_theta = random_rotation, rotation_theta
        self.mean_map, self.std_map = mean_map, std_map

    @classmethod
    def from_feeder(cls, f):
        return cls(window_size=f.window_size, random_choose=f.random_choose, random_shift=f.random_shift,
                   random_move=f.random_move, random_zaxis_flip=f.random_zaxis_flip,
                   random_xaxis_scale=f.random_xaxis_scale, random_yaxis_scale=f.random_yaxis_scale,
                   random_rotation=f.random_rotation, rotation_theta=f.rotation_theta(),
                   mean_map=getattr(f, 'mean_map', None) if f.normalization else None,
                   std_map=getattr(f, 'std_map', None) if f.normalization else None)

    # ---- the individual batched transforms (parameters explicit, so that tests can pin them) -------------------------
    @staticmethod
    def gather_frames(x, src, valid):
        """out[b, :, t] = x[b, :, src[b, t]] where valid[b, t], else 0.  src/valid: (B, T_out) on the device."""
        B, C, T, V, M = x.shape
        idx = src.clamp(0, T - 1).view(B, 1, -1, 1, 1).expand(B, C, src.shape[1], V, M)
        out = torch.gather(x, 2, idx)
        return out * valid.view(B, 1, -1, 1, 1).to(out.dtype)

    @staticmethod
    def valid_ranges(x):
        """per clip [begin, end) of the frames holding any non-zero value (tools.valid_range), on the device"""
        nz = (x != 0).flatten(3).any(3).any(1)                  # (B, T)
        T = nz.shape[1]
        ar = torch.arange(T, device=x.device)
        begin = torch.where(nz, ar, torch.full_like(ar, T)).min(1).values
        end = torch.where(nz, ar + 1, torch.zeros_like(ar)).max(1).values
        begin = torch.where(end > 0, begin, torch.zeros_like(begin))    # all-zero clip: [0, T) like numpy's argmax
        end = torch.where(end > 0, end, torch.full_like(end, T))
        return begin, end

    @staticmethod
    def affine_xy(x, a, s, tx, ty):
        """per-frame in-plane rotation/scale/translation of channels 0, 1 (tools.apply_move); a, s, tx, ty: (B, T)"""
        cos, sin = (torch.cos(a) * s)[:, :, None, None], (torch.sin(a) * s)[:, :, None, None]
        x0, x1 = x[:, 0], x[:, 1]
        n0 = cos * x0 - sin * x1 + tx[:, :, None, None]
        n1 = sin * x0 + cos * x1 + ty[:, :, None, None]
        return torch.cat([n0.unsqueeze(1), n1.unsqueeze(1), x[:, 2:]], 1)

    @staticmethod
    def person_scale(x, channel, S):
        """second person's offset from the first scaled by S (B,) along one axis (tools.random_axis_scale)"""
        x = x.clone()
        d = x[:, channel, :, :, 1] - x[:, channel, :, :, 0]
        x[:, channel, :, :, 1] = x[:, channel, :, :, 0] + d * S.view(-1, 1, 1)
        return x

    def __call__(self, x):
        B, C, T, V, M = x.shape
        dev = x.device
        if self.mean_map is not None:
            mm = torch.as_tensor(np.asarray(self.mean_map), dtype=x.dtype, device=dev)
            sm = torch.as_tensor(np.asarray(self.std_map), dtype=x.dtype, device=dev)
            x = (x - mm) / sm
        if self.random_shift:            # draws: random.randint(0, T - size) per clip (tools.random_shift)
            begin, end = self.valid_ranges(x)
            size = (end - begin).cpu().numpy()
            bias = torch.as_tensor([random.randint(0, T - int(sz)) for sz in size], device=dev)
            ar = torch.arange(T, device=dev).view(1, T)
            src = ar - bias.view(B, 1) + begin.view(B, 1)
            valid = (ar >= bias.view(B, 1)) & (ar < (bias + (end - begin)).view(B, 1))
            x = self.gather_frames(x, src, valid)
        W = self.window_size
        if self.random_choose and W > 0 and W != T:      # draws as tools.random_choose
            if T > W:
                b0 = torch.as_tensor([random.randint(0, T - W) for _ in range(B)], device=dev)
                src = torch.arange(W, device=dev).view(1, W) + b0.view(B, 1)
                x = self.gather_frames(x, src, torch.ones_like(src, dtype=torch.bool))
            else:
                b0 = torch.as_tensor([random.randint(0, W - T) for _ in range(B)], device=dev)
                ar = torch.arange(W, device=dev).view(1, W)
                x = self.gather_frames(x, ar - b0.view(B, 1), (ar >= b0.view(B, 1)) & (ar < (b0 + T).view(B, 1)))
        elif (not self.random_choose) and W > 0 and T < W:          # tools.auto_pading at the front
            x = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, W - T))
        T = x.shape[2]
        if self.random_move:
            par = np.stack([np.stack(tools.move_parameters(T)) for _ in range(B)])          # (B, 4, T)
            a, s, tx, ty = [torch.as_tensor(par[:, i], dtype=x.dtype, device=dev) for i in range(4)]
            x = self.affine_xy(x, a, s, tx, ty)
        if self.random_zaxis_flip:
            sign = torch.as_tensor([-1.0 if random.random() > 0.5 else 1.0 for _ in range(B)], dtype=x.dtype, device=dev)
            x = torch.cat([x[:, :2], x[:, 2:3] * sign.view(B, 1, 1, 1, 1), x[:, 3:]], 1)
        if self.random_xaxis_scale:
            S = torch.as_tensor(np.random.choice(tools.SCALE_CANDIDATES, B), dtype=x.dtype, device=dev)
            x = self.person_scale(x, 0, S)
        if self.random_yaxis_scale:
            S = torch.as_tensor(np.random.choice(tools.SCALE_CANDIDATES, B), dtype=x.dtype, device=dev)
            x = self.person_scale(x, 1, S)
        if self.random_rotation:
            R = np.stack([tools.rotation_matrix(np.random.uniform(-self.rotation_theta, self.rotation_theta, (1, 3))[0])
                          for _ in range(B)])
            x = torch.einsum('bij,bjtvm->bitvm', torch.as_tensor(R, dtype=x.dtype, device=dev), x)
        return x.contiguous()
