"""Trajectory metrics of the reference's evaluator (src/data/KITTI_eval.py:223-284, src/data/utils.py:72-243).

``path_accu`` runs on the device (csrc/pose.hip through ``odevio_path_accu``): the chain of rigid transforms is an
associative scan, one workgroup per drive, float64.  The KITTI segment statistics that follow are O(frames) host work
on the resulting matrices and are written as whole-array numpy (batched 4x4 algebra, ``searchsorted`` for the segment
ends) instead of the reference's Python loops; names and return values follow the reference.
"""
import ctypes

import numpy as np
import torch

from . import _lib

LENGTHS = (100, 200, 300, 400, 500, 600, 700, 800)   # metres (KITTI_eval.py:250)
STEP = 10                                            # frames between segment starts (KITTI_eval.py:255)


def path_accu(poses, carry=None, offsets=None):
    """Global pose matrices from relative 6-DoF poses (src/data/utils.py:142-161), on the device.

    poses  : [N,6] CUDA tensor, float32 (network output) or float64 (ground truth); several drives back to back when
             `offsets` ([n_drives+1] ints, first pose of every drive) is given.
    carry  : None or [n_drives,4,4] float64 start poses (streaming: the last matrix of the previous window).
    returns: [N + n_drives, 4, 4] float64 CUDA tensor; each drive contributes its start pose followed by one matrix per
             relative pose, like the reference's list that starts with the identity.
    """
    if not (torch.is_tensor(poses) and poses.is_cuda):
        raise ValueError("path_accu runs on the device: pass a CUDA tensor")
    if poses.dim() != 2 or poses.shape[1] != 6 or poses.dtype not in (torch.float32, torch.float64):
        raise ValueError("poses must be [N,6] float32 or float64")
    poses = poses.contiguous()
    n = poses.shape[0]
    off = [0, n] if offsets is None else [int(o) for o in offsets]
    if off[0] != 0 or off[-1] != n or any(b < a for a, b in zip(off, off[1:])):
        raise ValueError("offsets must start at 0, end at N and be non-decreasing")
    nd = len(off) - 1
    off_dev = torch.tensor(off, dtype=torch.int64, device=poses.device)
    c_ptr = None
    if carry is not None:
        carry = carry.to(device=poses.device, dtype=torch.float64).reshape(nd, 4, 4).contiguous()
        c_ptr = ctypes.c_void_p(carry.data_ptr())
    out = torch.empty(n + nd, 4, 4, dtype=torch.float64, device=poses.device)
    lib = _lib.load()
    _lib.check(lib.odevio_path_accu(ctypes.c_void_p(poses.data_ptr()), int(poses.dtype == torch.float64),
                                    ctypes.c_void_p(off_dev.data_ptr()), nd, c_ptr, ctypes.c_void_p(out.data_ptr()),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def rmse_err_cal(pose_est, pose_gt):
    """RMSE of the relative translation and rotation vectors (src/data/utils.py:202-208)."""
    pose_est, pose_gt = np.asarray(pose_est), np.asarray(pose_gt)
    t = np.sqrt(np.mean(np.sum((pose_est[:, 3:] - pose_gt[:, 3:]) ** 2, -1)))
    r = np.sqrt(np.mean(np.sum((pose_est[:, :3] - pose_gt[:, :3]) ** 2, -1)))
    return t, r


def trajectory_distances(mats):
    """Cumulative path length and speed (x10 Hz) per frame (src/data/utils.py:211-227)."""
    step = np.linalg.norm(np.diff(mats[:, :3, 3], axis=0), axis=1)
    return np.concatenate(([0.0], np.cumsum(step))), np.concatenate(([0.0], step * 10))


def kitti_err_cal(est_mats, gt_mats):
    """KITTI odometry segment errors (KITTI_eval.py:248-284): for every 10th start frame and every length in 100..800 m
    the rotation / translation error of the estimated segment against ground truth, per metre.
    -> (err rows [first_frame, r_err/len, t_err/len, len], t_rel, r_rel, speed)."""
    est_mats, gt_mats = np.asarray(est_mats, dtype=np.float64), np.asarray(gt_mats, dtype=np.float64)
    dist, speed = trajectory_distances(gt_mats)
    n_gt, n_est = len(gt_mats), len(est_mats)
    first = np.repeat(np.arange(0, n_gt, STEP), len(LENGTHS))
    length = np.tile(np.asarray(LENGTHS, dtype=np.float64), len(first) // len(LENGTHS))
    # first index i >= first with dist[i] > dist[first] + length; dist is non-decreasing
    last = np.searchsorted(dist, dist[first] + length, side="right")
    ok = (last < n_gt) & (last < n_est) & (first < n_est)
    first, last, length = first[ok], last[ok], length[ok]
    if len(first) == 0:
        raise ZeroDivisionError("no 100 m segment fits in the drive")   # the reference divides by len(err) == 0
    d_gt = np.linalg.inv(gt_mats[first]) @ gt_mats[last]
    d_est = np.linalg.inv(est_mats[first]) @ est_mats[last]
    e = np.linalg.inv(d_est) @ d_gt
    r_err = np.arccos(np.clip(0.5 * (e[:, 0, 0] + e[:, 1, 1] + e[:, 2, 2] - 1.0), -1.0, 1.0))
    t_err = np.linalg.norm(e[:, :3, 3], axis=1)
    err = np.stack((first.astype(np.float64), r_err / length, t_err / length, length), axis=1)
    return err, float(np.mean(err[:, 2])), float(np.mean(err[:, 1])), speed


def kitti_eval(pose_est, dec_est, pose_gt, device="cuda"):
    """(pose_est_mat, pose_gt_mat, t_rel [%], r_rel [deg/100 m], t_rmse, r_rmse [deg], usage, speed)
    - same tuple as the reference's kitti_eval (KITTI_eval.py:223-245); `dec_est` is unused there too."""
    pose_est = np.asarray(pose_est)
    pose_gt = np.asarray(pose_gt)[:, :6]
    t_rmse, r_rmse = rmse_err_cal(pose_est, pose_gt)
    est = path_accu(torch.as_tensor(pose_est).to(device)).cpu().numpy()
    gt = path_accu(torch.as_tensor(pose_gt).to(device)).cpu().numpy()
    _, t_rel, r_rel, speed = kitti_err_cal(est, gt)
    return est, gt, t_rel * 100, r_rel / np.pi * 180 * 100, t_rmse, r_rmse / np.pi * 180, 0, speed
