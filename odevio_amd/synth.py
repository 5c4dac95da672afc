"""Seeded synthetic KITTI-shaped inputs.  No dataset exists offline.

* ``img``  U(-0.5, 0.5)  [B,S,3,H,W] - the range ``ToTensor() - 0.5`` produces (reference src/data/utils.py:359)
* ``imu``  N(mu, sigma)  [B,10(S-1)+1,6] with the KITTI statistics of reference src/data/transforms.py:24-26
* ``timestamps`` [B,S] seconds, strictly ascending: 10 Hz regular, or the reference's frame-drop process
  (reference src/data/KITTI_dataset.py:64-74: drop interior frames i.i.d. with probability p).
"""
import numpy as np
import torch

IMU_MEAN = (-0.065, 0.079, 9.79, 1e-4, 6e-4, -6.6e-3)
IMU_STD = (1.006, 1.217, 0.403, 0.024, 0.027, 0.172)
FRAME_DT = 0.1  # KITTI camera period
IMU_PER_FRAME = 10


def _gen(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def images(B, S, H=256, W=512, seed=0):
    return torch.rand((B, S, 3, H, W), generator=_gen(seed), dtype=torch.float32) - 0.5


def imu(B, S, seed=0):
    T = IMU_PER_FRAME * (S - 1) + 1
    z = torch.randn((B, T, 6), generator=_gen(seed + 7919), dtype=torch.float32)
    return z * torch.tensor(IMU_STD) + torch.tensor(IMU_MEAN)


def timestamps(B, S, drop=0.0, seed=0, absolute=False):
    """Regular 10 Hz stamps, or stamps surviving an i.i.d. interior frame drop with probability ``drop``.

    Each row simulates its own 10 Hz stream and keeps the first ``S`` surviving frames (frame 0 and
    the frame after a kept one are treated like the reference: the first frame is never dropped).
    ``absolute=True`` adds a per-row start offset (streaming windows carry absolute time).
    """
    rng = np.random.default_rng(seed + 104729)
    out = np.zeros((B, S), dtype=np.float64)
    for b in range(B):
        kept = [0]
        k = 0
        while len(kept) < S:
            k += 1
            if drop > 0.0 and rng.random() < drop:
                continue
            kept.append(k)
        out[b] = np.asarray(kept, dtype=np.float64) * FRAME_DT
        if absolute:
            out[b] += float(rng.integers(0, 4000)) * FRAME_DT
    return torch.from_numpy(out.astype(np.float32))


def batch(B, S=11, H=256, W=512, drop=0.0, seed=0):
    return images(B, S, H, W, seed), imu(B, S, seed), timestamps(B, S, drop, seed)


def trajectory(n_frames, seed=0, noise=0.0):
    """Car-like relative poses [n_frames-1, 6] float64 in the reference's convention (angles x,y,z then translation,
    camera z forward; src/data/utils.py:44-69): ~1 m per frame forward with slowly varying yaw about y and small
    roll/pitch/side-slip.  `noise` adds an estimation error (a biased, noisy copy) - a stand-in for network output."""
    rng = np.random.default_rng(seed)
    n = n_frames - 1
    yaw_rate = 0.02 * np.sin(np.arange(n) / 37.0 + rng.uniform(0, 6)) + 0.004 * rng.standard_normal(n)
    p = np.zeros((n, 6))
    p[:, 0] = 0.002 * rng.standard_normal(n)
    p[:, 1] = yaw_rate
    p[:, 2] = 0.002 * rng.standard_normal(n)
    p[:, 3] = 0.01 * rng.standard_normal(n)
    p[:, 4] = 0.01 * rng.standard_normal(n)
    p[:, 5] = 1.0 + 0.3 * np.sin(np.arange(n) / 91.0) + 0.02 * rng.standard_normal(n)
    if noise:
        p = p * (1.0 + noise) + noise * 0.05 * rng.standard_normal(p.shape) * np.array([0.02, 0.02, 0.02, 1, 1, 1])
    return p


def drive(n_frames, H=256, W=512, seed=0, t0=0.0):
    """One synthetic drive for the streaming evaluator: (frames [N,3,H,W], imus [10(N-1)+1,6], timestamps [N] absolute
    seconds at ~10 Hz with jitter, poses_rel [N-1,6])."""
    frames = images(1, n_frames, H, W, seed)[0]
    g = _gen(seed + 7)
    z = torch.randn((10 * (n_frames - 1) + 1, 6), generator=g, dtype=torch.float32)
    imus = z * torch.tensor(IMU_STD) + torch.tensor(IMU_MEAN)
    rng = np.random.default_rng(seed + 13)
    ts = t0 + np.concatenate(([0.0], np.cumsum(0.1 + 0.004 * rng.standard_normal(n_frames - 1))))
    return frames, imus, torch.from_numpy(ts.astype(np.float32)), trajectory(n_frames, seed)
