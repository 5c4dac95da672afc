"""ORACLE (test infrastructure, never the product): plain-loop restatement of the reference's trajectory metrics.

Only tests/ may import this.  Follows the reference line by line in behaviour (not in text):

* ``euler_to_rot`` / ``pose_to_matrix`` / ``path_accu``      - src/data/utils.py:93-117, 131-161
* ``rot_to_euler`` / ``compose_poses``                      - src/data/utils.py:24-41, 163-195
* ``rmse``                                                  - src/data/utils.py:202-208
* ``trajectory_distances`` / ``last_frame_from_length``     - src/data/utils.py:211-235
* ``rotation_error`` / ``translation_error``                - src/data/utils.py:72-90
* ``kitti_err_cal`` / ``kitti_eval``                        - src/data/KITTI_eval.py:223-284
* ``partition`` / ``drop_frames``                           - src/data/KITTI_eval.py:58-89

PARITY UNPINNED for this file: src/data/utils.py imports torchvision, which is absent from the image, so the
reference functions cannot be run here to generate fixtures, and the reference holds no golden metric outputs.
The tests pin it instead on closed-form cases (identical trajectories, a constant yaw-rate circle, a pure scale
error) and use it as the independent check of the vectorised product code.
"""
import math

import numpy as np

_EPS = np.finfo(float).eps * 4.0
LENGTHS = (100, 200, 300, 400, 500, 600, 700, 800)
STEP = 10  # frames between segment starts (10 Hz data)
IMU_PER_FRAME = 10


def euler_to_rot(theta):
    """Rz(theta[2]) . Ry(theta[1]) . Rx(theta[0])  (utils.py:93-117).  As in the reference, sin/cos are taken in the
    dtype of `theta` (float32 for network outputs) and the matrices and their products are float64 (numpy promotes the
    python-int entries of the literal 3x3 lists to int64, hence the arrays to float64)."""
    c, s = np.cos(theta).astype(np.float64), np.sin(theta).astype(np.float64)
    rx = np.array([[1, 0, 0], [0, c[0], -s[0]], [0, s[0], c[0]]])
    ry = np.array([[c[1], 0, s[1]], [0, 1, 0], [-s[1], 0, c[1]]])
    rz = np.array([[c[2], -s[2], 0], [s[2], c[2], 0], [0, 0, 1]])
    return rz @ (ry @ rx)


def pose_to_matrix(pose):
    m = np.eye(4)
    m[:3, :3] = euler_to_rot(pose[:3])
    m[:3, 3] = pose[3:6]
    return m


def path_accu(poses, start=None):
    out = [np.eye(4) if start is None else np.array(start, dtype=np.float64)]
    for p in poses:
        out.append(out[-1] @ pose_to_matrix(p))
    return out


def rot_to_euler(m):
    cy = math.sqrt(m[0, 0] * m[0, 0] + m[1, 0] * m[1, 0])
    ay = math.atan2(-m[2, 0], cy)
    if -math.pi / 2 - _EPS < ay < -math.pi / 2 + _EPS:
        ax, az = 0.0, math.atan2(-m[1, 2], -m[0, 2])
    elif math.pi / 2 - _EPS < ay < math.pi / 2 + _EPS:
        ax, az = 0.0, math.atan2(m[1, 2], m[0, 2])
    else:
        ax, az = math.atan2(m[2, 1], m[2, 2]), math.atan2(m[1, 0], m[0, 0])
    return np.array([ax, ay, az])


def compose_poses(p1, p2):
    m = pose_to_matrix(np.asarray(p1, dtype=np.float64)) @ pose_to_matrix(np.asarray(p2, dtype=np.float64))
    return np.concatenate((rot_to_euler(m[:3, :3]), m[:3, 3]))


def rmse(est, gt):
    t = np.sqrt(np.mean(np.sum((est[:, 3:] - gt[:, 3:]) ** 2, -1)))
    r = np.sqrt(np.mean(np.sum((est[:, :3] - gt[:, :3]) ** 2, -1)))
    return t, r


def trajectory_distances(mats):
    dist, speed = [0.0], [0.0]
    for i in range(len(mats) - 1):
        d = float(np.linalg.norm(mats[i][:3, 3] - mats[i + 1][:3, 3]))
        dist.append(dist[i] + d)
        speed.append(d * 10)
    return dist, speed


def last_frame_from_length(dist, first, length):
    for i in range(first, len(dist)):
        if dist[i] > dist[first] + length:
            return i
    return -1


def _rel(a, b):
    return np.linalg.inv(a) @ b


def rotation_error(a, b):
    e = _rel(a, b)
    d = 0.5 * (e[0, 0] + e[1, 1] + e[2, 2] - 1.0)
    return math.acos(max(min(d, 1.0), -1.0))


def translation_error(a, b):
    e = _rel(a, b)
    return math.sqrt(e[0, 3] ** 2 + e[1, 3] ** 2 + e[2, 3] ** 2)


def kitti_err_cal(est_mats, gt_mats):
    err = []
    dist, speed = trajectory_distances(gt_mats)
    for first in range(0, len(gt_mats), STEP):
        for length in LENGTHS:
            last = last_frame_from_length(dist, first, length)
            if last == -1 or last >= len(est_mats) or first >= len(est_mats):
                continue
            d_gt = _rel(gt_mats[first], gt_mats[last])
            d_est = _rel(est_mats[first], est_mats[last])
            err.append([first, rotation_error(d_est, d_gt) / length, translation_error(d_est, d_gt) / length, length])
    t_rel = sum(e[2] for e in err) / len(err)  # ZeroDivisionError on a drive shorter than 100 m, as in the reference
    r_rel = sum(e[1] for e in err) / len(err)
    return err, t_rel, r_rel, np.asarray(speed)


def kitti_eval(pose_est, pose_gt):
    """-> dict(t_rel [%], r_rel [deg/100 m], t_rmse, r_rmse [deg], est_mats, gt_mats, speed)  (KITTI_eval.py:223-245)."""
    t_rmse, r_rmse = rmse(pose_est, pose_gt)
    est_mats, gt_mats = path_accu(pose_est), path_accu(pose_gt)
    _, t_rel, r_rel, speed = kitti_err_cal(est_mats, gt_mats)
    return {"t_rel": t_rel * 100, "r_rel": r_rel / np.pi * 180 * 100, "t_rmse": t_rmse, "r_rmse": r_rmse / np.pi * 180,
            "est_mats": est_mats, "gt_mats": gt_mats, "speed": speed}


def partition(n_frames, seq_len):
    """Evaluation windows (first frame, one-past-last frame): stride seq_len-1, the rest in a last short window."""
    out, start = [], 0
    while start + seq_len < n_frames:
        out.append((start, start + seq_len))
        start += seq_len - 1
    out.append((start, n_frames))
    return out


def drop_frames(poses_rel, timestamps, imus, dropout, rng):
    """Irregular sampling for evaluation (KITTI_eval.py:58-69): frame i is removed with probability `dropout`, its
    relative pose merged into the previous one and its 10 IMU rows deleted.  rng() -> uniform [0,1).
    Returns (poses_rel, timestamps, imus, kept frame indices)."""
    poses_rel = [np.asarray(p, dtype=np.float64) for p in poses_rel]
    timestamps = list(timestamps)
    imus = np.asarray(imus)
    kept = list(range(len(timestamps)))
    i = 1
    while i < len(poses_rel) - 2:
        if rng() < dropout:
            poses_rel[i] = compose_poses(poses_rel[i], poses_rel[i + 1])
            del poses_rel[i + 1]
            del timestamps[i]
            del kept[i]
            imus = np.delete(imus, np.arange(i * IMU_PER_FRAME, (i + 1) * IMU_PER_FRAME), axis=0)
        else:
            i += 1
    return np.stack(poses_rel), np.asarray(timestamps), imus, kept


def training_samples(poses_rel, timestamps, imus, seq_len, dropout, rng):
    """Plain-loop restatement of the training set construction for one drive (KITTI_dataset.py:64-106): the np.delete
    drop walk, then one sample per sliding window, `range(0, len(frames) - seq_len)`.  Frames are represented by their
    original indices (the reference pops file paths)."""
    poses_rel = np.asarray(poses_rel, dtype=np.float64).copy()
    timestamps = np.asarray(timestamps).copy()
    imus = np.asarray(imus).copy()
    frames = list(range(len(timestamps)))
    i = 1
    while i < len(poses_rel) - 2:
        if dropout > 0.0 and rng() < dropout:
            poses_rel[i] = compose_poses(poses_rel[i], poses_rel[i + 1])
            poses_rel = np.delete(poses_rel, i + 1, axis=0)
            timestamps = np.delete(timestamps, i, axis=0)
            imus = np.delete(imus, np.arange(i * IMU_PER_FRAME, (i + 1) * IMU_PER_FRAME), axis=0)
            frames.pop(i)
        else:
            i += 1
    out = []
    for i in range(0, len(frames) - seq_len):
        out.append({"frames": frames[i:i + seq_len], "timestamps": timestamps[i:i + seq_len],
                    "imus": imus[i * IMU_PER_FRAME:(i + seq_len - 1) * IMU_PER_FRAME + 1], "gts": poses_rel[i:i + seq_len - 1]})
    return out
