"""Streaming evaluation of whole drives: the window / stride logic and the hidden-state carry of the reference's
tester (src/data/KITTI_eval.py:76-89 ``data_partition`` windows, 124-160 ``test_one_path``, 162-199 ``eval``).

The reference walks ONE drive at a time with batch 1.  Drives are independent, so here any number of drives advance
in lock-step as one batch: at step i every drive that still has an i-th window contributes one row, the carried
``hc`` rows of exactly those drives are gathered on the device, and the hot path runs once per (step, window length)
group.  Results per drive are identical to walking it alone (tests/test_gpu_parity.py::test_stream_*).
"""
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import metrics

IMU_PER_FRAME = 10   # IMU samples between two frames (KITTI_eval.py:29, 84-86)


def partition(n_frames: int, seq_len: int):
    """Windows (first frame, one-past-last frame) of one drive: stride ``seq_len - 1`` so consecutive windows share a
    frame, and whatever is left (2..seq_len frames) as a last, shorter window (KITTI_eval.py:76-89)."""
    if n_frames < 2 or seq_len < 2:
        raise ValueError("a drive needs at least 2 frames and seq_len >= 2")
    out, start = [], 0
    while start + seq_len < n_frames:
        out.append((start, start + seq_len))
        start += seq_len - 1
    out.append((start, n_frames))
    return out


def imu_rows(first: int, last: int):
    """IMU rows of the window of frames [first, last): 10 per interval plus the closing sample (KITTI_eval.py:84-86)."""
    return first * IMU_PER_FRAME, (last - 1) * IMU_PER_FRAME + 1


@dataclass
class Drive:
    """One drive resident in memory: frames [N,3,H,W] float32 (already resized and shifted by -0.5 like
    KITTI_eval.py:100-103), imus [10*(N-1)+1, 6], timestamps [N] seconds (absolute, as read from times.txt),
    poses_rel [N-1, 6] ground truth or None."""
    frames: torch.Tensor
    imus: torch.Tensor
    timestamps: torch.Tensor
    poses_rel: Optional[np.ndarray] = None
    name: str = ""

    def __post_init__(self):
        n = self.frames.shape[0]
        if self.timestamps.shape[0] != n or self.imus.shape[0] < (n - 1) * IMU_PER_FRAME + 1:
            raise ValueError(f"drive {self.name!r}: {n} frames need {n} timestamps and {(n - 1) * IMU_PER_FRAME + 1} IMU rows")


def drop_frames(poses_rel, timestamps, imus, dropout, rng):
    """Irregular sampling of an evaluation drive (KITTI_eval.py:58-69).  Walking i = 1, 2, ...: with probability
    `dropout` the relative poses i and i+1 are merged into one and timestamp i, frame i and IMU rows [10i, 10i+10) are
    removed (the reference's own indexing, kept as is).  `rng()` returns uniforms in [0,1) - pass ``random.random``
    after ``random.seed`` to reproduce a reference run.  -> (poses_rel, timestamps, imus, kept frame indices)."""
    poses_rel = np.array(poses_rel, dtype=np.float64)
    keep_t = np.ones(len(timestamps), dtype=bool)
    keep_imu = np.ones(len(imus), dtype=bool)
    # the walk is sequential by definition (every decision shifts the later indices); it only touches index lists,
    # the merged poses are composed afterwards in one batched matrix product per run of dropped frames
    groups = [[j] for j in range(len(poses_rel))]      # which original relative poses make up the current pose i
    alive_t = list(range(len(timestamps)))
    alive_imu_off = 0                                   # rows already deleted in front of the current position
    i = 1
    while i < len(groups) - 2:
        if rng() < dropout:
            groups[i] = groups[i] + groups[i + 1]
            del groups[i + 1]
            keep_t[alive_t[i]] = False
            del alive_t[i]
            lo = i * IMU_PER_FRAME + alive_imu_off
            keep_imu[lo:lo + IMU_PER_FRAME] = False
            alive_imu_off += IMU_PER_FRAME
        else:
            i += 1
    out = np.empty((len(groups), 6), dtype=np.float64)
    for k, g in enumerate(groups):
        out[k] = poses_rel[g[0]] if len(g) == 1 else _compose(poses_rel[g])
    return out, np.asarray(timestamps)[keep_t], np.asarray(imus)[keep_imu], np.flatnonzero(keep_t)


def training_windows(n_frames: int, seq_len: int):
    """Sliding windows (first frame, one-past-last frame) of the TRAINING set: stride 1, ``range(0, n - seq_len)`` - the
    reference leaves the last possible window out (src/data/KITTI_dataset.py:77)."""
    return [(i, i + seq_len) for i in range(0, n_frames - seq_len)]


def training_samples(poses_rel, timestamps, imus, seq_len, dropout=0.0, rng=None):
    """The training-time sample list of one drive (reference src/data/KITTI_dataset.py:64-106): the frame-drop walk
    (identical to the evaluator's: pose i+1 merged into pose i, timestamp / frame i and IMU rows [10i, 10i+10) removed),
    then one sample per sliding window: -> list of dicts {"frames": kept frame indices [seq_len], "imus"
    [10(seq_len-1)+1, 6], "gts" [seq_len-1, 6], "timestamps" [seq_len]}.  ``rng()`` returns uniforms in [0, 1)."""
    if dropout > 0.0:
        if rng is None:
            raise ValueError("training_samples: dropout needs an rng (random.random after random.seed reproduces a reference run)")
        poses_rel, timestamps, imus, kept = drop_frames(poses_rel, timestamps, imus, dropout, rng)
    else:
        poses_rel, timestamps, imus = np.asarray(poses_rel, dtype=np.float64), np.asarray(timestamps), np.asarray(imus)
        kept = np.arange(len(timestamps))
    out = []
    for a, b in training_windows(len(timestamps), seq_len):
        lo, hi = imu_rows(a, b)
        out.append({"frames": np.asarray(kept[a:b]), "imus": imus[lo:hi], "gts": poses_rel[a:b - 1], "timestamps": timestamps[a:b]})
    return out


def _euler_to_rot(theta):
    c, s = np.cos(theta), np.sin(theta)
    one, zero = np.ones_like(c[..., 0]), np.zeros_like(c[..., 0])
    rx = np.stack([one, zero, zero, zero, c[..., 0], -s[..., 0], zero, s[..., 0], c[..., 0]], -1).reshape(theta.shape[:-1] + (3, 3))
    ry = np.stack([c[..., 1], zero, s[..., 1], zero, one, zero, -s[..., 1], zero, c[..., 1]], -1).reshape(theta.shape[:-1] + (3, 3))
    rz = np.stack([c[..., 2], -s[..., 2], zero, s[..., 2], c[..., 2], zero, zero, zero, one], -1).reshape(theta.shape[:-1] + (3, 3))
    return rz @ (ry @ rx)


def _compose(poses):
    """Left-to-right composition of consecutive relative poses, re-expressed as one 6-DoF pose
    (concatenate_pose_changes applied repeatedly, src/data/utils.py:163-195, Euler extraction utils.py:24-41)."""
    eps = np.finfo(float).eps * 4.0
    cur = poses[0]
    for nxt in poses[1:]:
        r1, r2 = _euler_to_rot(cur[:3]), _euler_to_rot(nxt[:3])
        r = r1 @ r2
        t = r1 @ nxt[3:] + cur[3:]
        cy = np.hypot(r[0, 0], r[1, 0])
        ay = np.arctan2(-r[2, 0], cy)
        if abs(ay + np.pi / 2) < eps:
            ax, az = 0.0, np.arctan2(-r[1, 2], -r[0, 2])
        elif abs(ay - np.pi / 2) < eps:
            ax, az = 0.0, np.arctan2(r[1, 2], r[0, 2])
        else:
            ax, az = np.arctan2(r[2, 1], r[2, 2]), np.arctan2(r[1, 0], r[0, 0])
        cur = np.array([ax, ay, az, t[0], t[1], t[2]])
    return cur


class StreamTester:
    """``KITTI_tester`` for drives already in memory (KITTI_eval.py:113-199): ``test_paths`` streams the windows through the
    network with the hidden state carried from window to window, ``eval`` adds the KITTI metrics."""

    def __init__(self, seq_len: int, device="cuda"):
        self.seq_len = seq_len
        self.device = torch.device(device)
        self.errors, self.est = [], []

    @torch.no_grad()
    def test_paths(self, net, drives: Sequence[Drive]) -> List[np.ndarray]:
        """-> per drive the estimated relative poses [N-1, 6] (np.vstack of the windows, KITTI_eval.py:150-156)."""
        plans = [partition(d.frames.shape[0], self.seq_len) for d in drives]
        check = getattr(net, "check", None)
        if len(drives) > 1 and getattr(getattr(net, "opt", None), "model_type", "") == "cde":
            # PoseCDE integrates every row over ROW 0's timestamps (PoseCDE.py:101): lock-stepping drives with
            # different clocks would differ from walking each alone, as the reference does
            raise ValueError("model_type 'cde' streams one drive at a time (PoseCDE.py:101 uses row 0's timestamps for the whole batch)")
        hc = [None] * len(drives)                       # per drive [L,1,F] on the device, None before the first window
        chunks = [[] for _ in drives]
        for step in range(max(len(p) for p in plans)):
            by_len = {}
            for k, p in enumerate(plans):
                if step < len(p):
                    by_len.setdefault(p[step][1] - p[step][0], []).append(k)
            for n_fr, ks in sorted(by_len.items(), reverse=True):
                img = torch.stack([drives[k].frames[plans[k][step][0]:plans[k][step][1]] for k in ks])
                imu = torch.stack([drives[k].imus[slice(*imu_rows(*plans[k][step]))] for k in ks])
                ts = torch.stack([drives[k].timestamps[plans[k][step][0]:plans[k][step][1]] for k in ks])
                img = img.to(self.device, torch.float32, non_blocking=True)
                imu = imu.to(self.device, torch.float32, non_blocking=True)
                ts = ts.to(self.device, torch.float32, non_blocking=True)
                h_in = None if step == 0 else torch.cat([hc[k] for k in ks], dim=1)
                pose, h_out = net(img, imu, ts, hc=h_in)
                if check is not None:
                    check()   # a failed forward (timeout, step budget, fp16x2 range) must not reach the metrics as garbage
                for j, k in enumerate(ks):
                    hc[k] = h_out[:, j:j + 1].clone() if torch.is_tensor(h_out) else h_out
                    chunks[k].append(pose[j])
        return [torch.cat(c).float().cpu().numpy() for c in chunks]

    def eval(self, net, drives: Sequence[Drive]):
        """-> list of {"t_rel","r_rel","t_rmse","r_rmse","usage"} per drive (KITTI_eval.py:162-199)."""
        self.errors, self.est = [], []
        for d, pose_est in zip(drives, self.test_paths(net, drives)):
            if d.poses_rel is None:
                raise ValueError(f"drive {d.name!r} has no ground truth")
            est, gt, t_rel, r_rel, t_rmse, r_rmse, usage, speed = metrics.kitti_eval(pose_est, None, d.poses_rel, self.device)
            self.est.append({"pose_est_global": est, "pose_gt_global": gt, "decs": None, "probs": None, "speed": speed,
                             "pose_est": pose_est})
            self.errors.append({"t_rel": t_rel, "r_rel": r_rel, "t_rmse": t_rmse, "r_rmse": r_rmse, "usage": usage})
        return self.errors

    def save_text(self, save_dir, names: Sequence[str]):
        """KITTI pose files, 12 numbers per line (saveSequence, src/data/utils.py:289-294; KITTI_eval.py:213-220)."""
        import os
        for name, e in zip(names, self.est):
            for tag, mats in (("pred", e["pose_est_global"]), ("gt", e["pose_gt_global"])):
                with open(os.path.join(str(save_dir), f"{name}_{tag}.txt"), "w") as f:
                    for m in mats:
                        f.write(" ".join(str(v) for v in m.flatten()[:12]) + "\n")
