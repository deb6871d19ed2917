"""Per-clip skeleton augmentations on the host (numpy), counterparts of the reference ``feeders/tools.py:10-231``.

Every function takes one clip ``(C, T, V, M)`` and draws its random numbers with the same generator calls, in the
same order, as the reference does (``random`` / ``numpy.random`` global state), so a seeded run reproduces the
reference's augmented clips bit for bit -- ``tests/golden/feeder_tools.npz`` holds clips the reference functions produced
with fixed seeds and ``tests/test_feeders.py`` checks these against them.

The batched GPU versions of the same transforms live in ``device.py`` (``DeviceAugment``): at hundreds of clips per
second per GPU the per-sample numpy path is the bottleneck of the input side.
"""
import random

import numpy as np

SCALE_CANDIDATES = [0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1.1, 1.2, 1.3, 1.4, 1.5]   # reference tools.py:58, 63


def downsample(clip, step, random_sample=True):
    """every ``step``-th frame from a random phase (reference tools.py:10-13)"""
    begin = np.random.randint(step) if random_sample else 0
    return clip[:, begin::step, :, :]


def auto_pading(clip, size, random_pad=False):
    """zero-pad a clip shorter than ``size`` frames, at the front or at a random offset (reference tools.py:35-44)"""
    C, T, V, M = clip.shape
    if T >= size:
        return clip
    begin = random.randint(0, size - T) if random_pad else 0
    out = np.zeros((C, size, V, M))
    out[:, begin:begin + T] = clip
    return out


def random_choose(clip, size, auto_pad=True):
    """a random window of ``size`` frames; shorter clips are padded at a random offset (reference tools.py:95-108)"""
    T = clip.shape[1]
    if T == size:
        return clip
    if T < size:
        return auto_pading(clip, size, random_pad=True) if auto_pad else clip
    begin = random.randint(0, T - size)
    return clip[:, begin:begin + size]


def valid_range(clip):
    """[begin, end) of the frames that hold any non-zero coordinate (reference tools.py:203-205)"""
    valid = (clip != 0).sum(axis=3).sum(axis=2).sum(axis=0) > 0
    begin = int(valid.argmax())
    end = len(valid) - int(valid[::-1].argmax())
    return begin, end


def random_shift(clip):
    """move the non-zero segment of the clip to a random offset (reference tools.py:199-211)"""
    T = clip.shape[1]
    begin, end = valid_range(clip)
    size = end - begin
    bias = random.randint(0, T - size)
    out = np.zeros(clip.shape)
    out[:, bias:bias + size] = clip[:, begin:end]
    return out


def move_parameters(T, angle_candidate=(-10., -5., 0., 5., 10.), scale_candidate=(0.9, 1.0, 1.1),
                    transform_candidate=(-0.2, -0.1, 0.0, 0.1, 0.2), move_time_candidate=(1,)):
    """Per-frame rotation angle (rad), scale and x/y translation of ``random_move``: random values at ``move_time + 1``
    nodes, linearly interpolated in between (reference tools.py:111-146).  Returns (a, s, t_x, t_y), each (T,)."""
    move_time = random.choice(list(move_time_candidate))
    node = np.arange(0, T, T * 1.0 / move_time).round().astype(int)
    node = np.append(node, T)
    n = len(node)
    A = np.random.choice(list(angle_candidate), n)
    S = np.random.choice(list(scale_candidate), n)
    Tx = np.random.choice(list(transform_candidate), n)
    Ty = np.random.choice(list(transform_candidate), n)
    a, s, tx, ty = np.zeros(T), np.zeros(T), np.zeros(T), np.zeros(T)
    for i in range(n - 1):
        seg = node[i + 1] - node[i]
        a[node[i]:node[i + 1]] = np.linspace(A[i], A[i + 1], seg) * np.pi / 180
        s[node[i]:node[i + 1]] = np.linspace(S[i], S[i + 1], seg)
        tx[node[i]:node[i + 1]] = np.linspace(Tx[i], Tx[i + 1], seg)
        ty[node[i]:node[i + 1]] = np.linspace(Ty[i], Ty[i + 1], seg)
    return a, s, tx, ty


def apply_move(clip, a, s, tx, ty):
    """x/y of every frame rotated by a[t], scaled by s[t], translated by (tx[t], ty[t]) (reference tools.py:148-159)"""
    cos, sin = np.cos(a) * s, np.sin(a) * s
    x, y = clip[0].copy(), clip[1].copy()              # (T, V, M)
    clip[0] = cos[:, None, None] * x - sin[:, None, None] * y + tx[:, None, None]
    clip[1] = sin[:, None, None] * x + cos[:, None, None] * y + ty[:, None, None]
    return clip


def random_move(clip, **kw):
    """smoothly varying in-plane rotation / scale / translation (reference tools.py:111-161); in place like the
    reference"""
    return apply_move(clip, *move_parameters(clip.shape[1], **kw))


def random_axis_scale(clip, candidate, channel):
    """scale the second person's offset from the first along one axis (reference tools.py:47-55)"""
    S = np.random.choice(candidate, 1)
    distance = clip[channel, :, :, 1] - clip[channel, :, :, 0]
    clip[channel, :, :, 1] = clip[channel, :, :, 0] + distance * S
    return clip


def random_xaxis_scale(clip):
    return random_axis_scale(clip, SCALE_CANDIDATES, 0)


def random_yaxis_scale(clip):
    return random_axis_scale(clip, SCALE_CANDIDATES, 1)


def random_flip(clip, channel):
    """negate one coordinate with probability 1/2 (reference tools.py:74-79)"""
    if random.random() > 0.5:
        clip[channel] = -clip[channel]
    return clip


def random_xaxis_flip(clip):
    return random_flip(clip, 0)


def random_yaxis_flip(clip):
    return random_flip(clip, 1)


def random_zaxis_flip(clip):
    return random_flip(clip, 2)


def random_subsample(clip, freq):
    """one random frame out of each of ``freq`` equal segments (reference tools.py:215-221)"""
    T = clip.shape[1]
    seg = T // freq
    offsets = np.multiply(list(range(freq)), seg) + np.random.randint(seg, size=freq)
    return clip[:, offsets]


def rotation_matrix(rot):
    """R = Rz . Ry . Rx for angles rot = (rx, ry, rz) with the reference's sign convention (tools.py:164-185):
    Rx = [[1,0,0],[0,c,s],[0,-s,c]], Ry = [[c,0,-s],[0,1,0],[s,0,c]], Rz = [[c,s,0],[-s,c,0],[0,0,1]]."""
    cx, cy, cz = np.cos(rot)
    sx, sy, sz = np.sin(rot)
    rx = np.array([[1, 0, 0], [0, cx, sx], [0, -sx, cx]])
    ry = np.array([[cy, 0, -sy], [0, 1, 0], [sy, 0, cy]])
    rz = np.array([[cz, sz, 0], [-sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


def random_rotation(clip, theta=0.5):
    """one random 3-D rotation (angles uniform in [-theta, theta]) applied to every joint (reference tools.py:189-198)"""
    rot = np.random.uniform(-theta, theta, (1, 3))[0]
    R = rotation_matrix(rot)
    return np.einsum('ij,jtvm->itvm', R, clip)


def stretch_to_maximum_length(clip):
    """linearly resample the non-padded prefix of the clip to the full length (reference tools.py:224-235)"""
    C, T, V, M = clip.shape
    nz = np.where(np.flip(clip.sum((0, 2, 3))) != 0.0)[0]
    t_last = T - int(nz[0])
    src = np.transpose(clip[:, :t_last], (0, 2, 3, 1)).reshape(C * V * M, t_last)
    pos = np.linspace(0, t_last - 1, T)
    lo = np.clip(np.floor(pos).astype(int), 0, max(t_last - 2, 0))
    w = pos - lo
    hi = np.minimum(lo + 1, t_last - 1)
    out = src[:, lo] * (1.0 - w) + src[:, hi] * w
    return np.transpose(out.reshape(C, V, M, T), (0, 3, 1, 2))
