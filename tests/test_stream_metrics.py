"""Streaming evaluator host logic and KITTI metrics (CPU part): the vectorised product code against the plain-loop
oracle (oracle/kitti_metrics.py) and against closed-form cases.  The device part (path_accu kernel, streaming through
the network) is in tests/test_gpu_parity.py."""
import random

import numpy as np
import pytest

from odevio_amd import metrics, stream, synth
from oracle import kitti_metrics as om


@pytest.mark.parametrize("n_frames", [2, 3, 10, 11, 12, 21, 22, 101, 4541])
@pytest.mark.parametrize("seq_len", [2, 5, 11])
def test_partition_matches_oracle_and_chains(n_frames, seq_len):
    w = stream.partition(n_frames, seq_len)
    assert w == om.partition(n_frames, seq_len)
    assert w[0][0] == 0 and w[-1][1] == n_frames
    for (a0, a1), (b0, b1) in zip(w, w[1:]):
        assert b0 == a1 - 1                       # consecutive windows share one frame: poses chain without gaps
        assert a1 - a0 == seq_len
    assert 2 <= w[-1][1] - w[-1][0] <= seq_len
    assert sum(b - a - 1 for a, b in w) == n_frames - 1   # every interval estimated exactly once


def test_partition_rejects_degenerate():
    with pytest.raises(ValueError):
        stream.partition(1, 11)
    with pytest.raises(ValueError):
        stream.partition(10, 1)


def test_imu_rows():
    assert stream.imu_rows(0, 11) == (0, 101)
    assert stream.imu_rows(10, 21) == (100, 201)
    assert stream.imu_rows(20, 23) == (200, 221)


@pytest.mark.parametrize("dropout", [0.0, 0.3, 0.7])
def test_drop_frames_matches_oracle(dropout):
    n = 120
    poses = synth.trajectory(n, seed=5)
    ts = np.cumsum(np.full(n, 0.1))
    imus = np.arange((n - 1) * 10 + 1, dtype=np.float64)[:, None].repeat(6, 1)
    random.seed(42)
    p1, t1, i1, k1 = stream.drop_frames(poses, ts, imus, dropout, random.random)
    random.seed(42)
    p2, t2, i2, k2 = om.drop_frames(poses, ts, imus, dropout, random.random)
    assert list(k1) == list(k2)
    np.testing.assert_array_equal(t1, t2)
    np.testing.assert_array_equal(i1, i2)
    np.testing.assert_allclose(p1, p2, rtol=0, atol=1e-12)
    assert len(p1) == len(t1) - 1 and len(i1) == (len(t1) - 1) * 10 + 1
    if dropout == 0.0:
        assert len(t1) == n
    else:
        assert len(t1) < n
        # merging relative poses must not move the end of the trajectory
        a, b = om.path_accu(poses)[-1], om.path_accu(p1)[-1]
        np.testing.assert_allclose(a, b, atol=1e-9)


def _mats(poses):
    return np.stack(om.path_accu(poses))


@pytest.mark.parametrize("n_frames,seed", [(400, 1), (1200, 2), (2500, 3)])
def test_kitti_err_cal_matches_oracle(n_frames, seed):
    gt = synth.trajectory(n_frames, seed)
    est = synth.trajectory(n_frames, seed, noise=0.05).astype(np.float32)
    gm, em = _mats(gt), _mats(est)
    err, t_rel, r_rel, speed = metrics.kitti_err_cal(em, gm)
    err_o, t_o, r_o, speed_o = om.kitti_err_cal(list(em), list(gm))
    assert len(err) == len(err_o) > 0
    np.testing.assert_allclose(err, np.asarray(err_o), rtol=1e-9, atol=1e-12)
    assert t_rel == pytest.approx(t_o, rel=1e-10) and r_rel == pytest.approx(r_o, rel=1e-10)
    np.testing.assert_allclose(speed, speed_o, rtol=1e-12)
    t1, r1 = metrics.rmse_err_cal(est, gt)
    t2, r2 = om.rmse(est, gt)
    assert t1 == pytest.approx(t2, rel=1e-12) and r1 == pytest.approx(r2, rel=1e-12)


def test_kitti_err_cal_shorter_estimate():
    """An estimate shorter than the ground truth only scores the segments it covers (KITTI_eval.py:262-267)."""
    gt = synth.trajectory(900, 4)
    gm = _mats(gt)
    em = _mats(gt[:600])
    err, *_ = metrics.kitti_err_cal(em, gm)
    err_o, *_ = om.kitti_err_cal(list(em), list(gm))
    assert len(err) == len(err_o)
    assert err[:, 0].max() < 600


def test_too_short_drive_raises_like_reference():
    gm = _mats(synth.trajectory(50, 1))       # ~50 m: no 100 m segment
    with pytest.raises(ZeroDivisionError):
        metrics.kitti_err_cal(gm, gm)
    with pytest.raises(ZeroDivisionError):
        om.kitti_err_cal(list(gm), list(gm))


# ---- closed-form pins of the (otherwise unpinned) oracle -----------------------------------------------------------
def test_oracle_identical_trajectories_have_zero_error():
    gt = synth.trajectory(800, 7)
    out = om.kitti_eval(gt, gt)
    assert out["t_rel"] == pytest.approx(0.0, abs=1e-9) and out["r_rel"] == pytest.approx(0.0, abs=1e-4)
    assert out["t_rmse"] == 0.0 and out["r_rmse"] == 0.0


def test_oracle_scale_error_is_t_rel():
    """Straight drive, estimate 10 % too long: every segment is 10 % off in translation, none in rotation."""
    n = 1000
    gt = np.zeros((n, 6))
    gt[:, 5] = 1.0
    est = gt.copy()
    est[:, 5] = 1.1
    for impl in (lambda e, g: om.kitti_err_cal(om.path_accu(e), om.path_accu(g)),
                 lambda e, g: metrics.kitti_err_cal(_mats(e), _mats(g))):
        err, t_rel, r_rel, _ = impl(est, gt)
        # a segment ends at the first frame strictly beyond `length` metres: (length + 1) frames here
        expect = np.mean([0.1 * (L + 1) / L for _, _, _, L in np.asarray(err)])
        assert t_rel == pytest.approx(expect, rel=1e-9)
        assert r_rel == pytest.approx(0.0, abs=1e-7)


def test_oracle_yaw_bias_is_r_rel():
    """Circle: estimate turns 1e-4 rad per frame more than ground truth -> r_err of a k-frame segment is k * 1e-4."""
    n = 1500
    gt = np.zeros((n, 6))
    gt[:, 5] = 1.0
    gt[:, 1] = 0.01
    est = gt.copy()
    est[:, 1] += 1e-4
    err, t_rel, r_rel, _ = om.kitti_err_cal(om.path_accu(est), om.path_accu(gt))
    err = np.asarray(err)
    dist, _ = om.trajectory_distances(om.path_accu(gt))
    for first, r_per_m, _, L in err[::37]:
        last = om.last_frame_from_length(dist, int(first), L)
        assert r_per_m * L == pytest.approx((last - int(first)) * 1e-4, rel=1e-6)


def test_oracle_euler_roundtrip():
    rng = np.random.default_rng(0)
    for _ in range(50):
        th = rng.uniform(-1.2, 1.2, 3)
        np.testing.assert_allclose(om.rot_to_euler(om.euler_to_rot(th)), th, atol=1e-12)
    a, b = rng.uniform(-0.3, 0.3, 6), rng.uniform(-0.3, 0.3, 6)
    np.testing.assert_allclose(om.pose_to_matrix(om.compose_poses(a, b)), om.pose_to_matrix(a) @ om.pose_to_matrix(b), atol=1e-12)


@pytest.mark.parametrize("dropout", [0.0, 0.4])
def test_training_samples_match_oracle(dropout):
    """Training-time sample construction (KITTI_dataset.py:64-106): drop walk + sliding windows, against the plain loop."""
    n, S = 60, 11
    poses = synth.trajectory(n, seed=8)
    ts = np.cumsum(np.full(n, 0.1))
    imus = np.arange((n - 1) * 10 + 1, dtype=np.float64)[:, None].repeat(6, 1)
    random.seed(7)
    got = stream.training_samples(poses, ts, imus, S, dropout, random.random)
    random.seed(7)
    ref = om.training_samples(poses, ts, imus, S, dropout, random.random)
    assert len(got) == len(ref) and len(got) > 0
    if dropout == 0.0:
        assert len(got) == n - S            # the reference leaves the last window out
    for a, b in zip(got, ref):
        assert list(a["frames"]) == list(b["frames"])
        np.testing.assert_array_equal(a["timestamps"], b["timestamps"])
        np.testing.assert_array_equal(a["imus"], b["imus"])
        np.testing.assert_allclose(a["gts"], b["gts"], rtol=0, atol=1e-12)
        assert a["imus"].shape == ((S - 1) * 10 + 1, 6) and a["gts"].shape == (S - 1, 6) and np.all(np.diff(a["timestamps"]) > 0)
    with pytest.raises(ValueError):
        stream.training_samples(poses, ts, imus, S, 0.3, None)
