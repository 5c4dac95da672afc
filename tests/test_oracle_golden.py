"""The oracle against outputs of the REAL reference modules (tests/golden, made by oracle/gen_golden.py).

CPU only.  Pins SURVEY.md section 8a rows A2-A5, A8 and the non-ODE part of A7 against the reference.
"""
import os

import numpy as np
import pytest
import torch

from odevio_amd import synth, weights
from odevio_amd.config import default_opt
from oracle import odevio_oracle as oc

TOL = 2e-5  # same ATen CPU kernels, different call decomposition / thread count


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def close(a, b, tol=TOL):
    err = oc.rel_err(torch.as_tensor(a), torch.as_tensor(b))
    assert err < tol, err


@pytest.mark.parametrize("tag", ["small", "full"])
def test_image_encoder(golden_dir, tag):
    g = load(golden_dir, f"image_encoder_{tag}.npz")
    H, W, B, S = int(g["H"]), int(g["W"]), int(g["B"]), int(g["S"])
    opt = default_opt(img_h=H, img_w=W)
    sd = weights.make_state_dict(opt, seed=int(g["wseed"]), randomize_stats=bool(g["randomize_stats"]))
    img = synth.images(B, S, H, W, seed=int(g["iseed"]))
    assert abs(float(img.double().sum()) - float(g["img_sum"])) < 1e-6
    fv, inter = oc.image_encoder(sd, img, return_intermediate=True)
    close(fv, g["fv"])
    close(inter["conv1"][:, ::8, ::8, ::8], g["conv1_sample"])
    close(inter["conv3_1"][:, ::16, ::2, ::2], g["conv3_1_sample"])
    c6 = inter["conv6"] if tag == "small" else inter["conv6"][:, ::32]
    close(c6, g["conv6"])


def test_inertial_encoder(golden_dir):
    g = load(golden_dir, "inertial_encoder.npz")
    opt = default_opt()
    sd = weights.make_state_dict(opt, seed=int(g["wseed"]), randomize_stats=True)
    real = torch.from_numpy(g["imu04"])
    for T in (11, 21, 51, 101, 105):
        fi = oc.inertial_encoder(sd, real[:, :T])
        assert fi.shape[1] == (T - 1) // 10  # tail samples ignored (T=105 -> 10 pairs)
        close(fi, g[f"fi_T{T}"])
    syn = synth.imu(3, 11, seed=5)
    assert abs(float(syn.double().sum()) - float(g["syn_sum"])) < 1e-6
    close(oc.inertial_encoder(sd, syn), g["fi_syn"])


@pytest.mark.parametrize("method", ["cat", "soft"])
def test_fusion(golden_dir, method):
    g = load(golden_dir, "fusion.npz")
    opt = default_opt(fuse_method=method)
    sd = weights.make_state_dict(opt, seed=int(g["wseed"]), randomize_stats=True)
    close(oc.fuse(sd, torch.from_numpy(g["fv"]), torch.from_numpy(g["fi"]), method), g[method])


def test_fusion_hard_has_no_restatement():
    with pytest.raises(ValueError):
        oc.fuse({}, torch.zeros(1, 1, 2), torch.zeros(1, 1, 2), "hard")


@pytest.mark.parametrize("act", ["tanh", "relu", "leaky_relu", "softplus"])
def test_odefunc(golden_dir, act):
    g = load(golden_dir, "odefunc.npz")
    y = torch.from_numpy(g["y"])
    for n in (2, 3):
        for H in (512, 1024):
            opt = default_opt(ode_activation_fn=act, ode_fn_num_layers=n, ode_hidden_dim=H)
            sd = weights.make_state_dict(opt, seed=int(g["wseed"]), randomize_stats=True)
            close(oc.ode_func(sd, y, n, act), g[f"f_{act}_{n}_{H}"])


def test_odefunc_rejects_unknown_activation():
    with pytest.raises(ValueError):
        oc._activation("gelu")  # reference ODEFunc.py:34


def test_cdefunc(golden_dir):
    g = load(golden_dir, "odefunc.npz")
    opt = default_opt(model_type="cde", cde_hidden_dim=128, v_f_len=96, i_f_len=32)
    sd = weights.make_state_dict(opt, seed=int(g["wseed"]), randomize_stats=True)
    z = torch.from_numpy(g["z"])
    out = oc.mlp_tanh_out(sd, "Pose_net.cde_func.net", 3, z, "tanh").view(3, 128, 129)
    close(out, g["cde_f"])


@pytest.mark.parametrize("rnn_type", ["rnn", "gru"])
@pytest.mark.parametrize("L", [2, 3])
@pytest.mark.parametrize("method", ["cat", "soft"])
def test_pose_rnn_skeleton(golden_dir, rnn_type, L, method):
    g = load(golden_dir, "pose_rnn.npz")
    opt = default_opt(model_type="rnn", ode_rnn_type=rnn_type, rnn_num_layers=L, fuse_method=method)
    sd = weights.make_state_dict(opt, seed=int(g["wseed"]), randomize_stats=True)
    fv, fi, ts = (torch.from_numpy(g[k]) for k in ("fv", "fi", "ts"))
    key = f"{rnn_type}_{L}_{method}"
    p1, h1 = oc.pose_rnn(sd, fv, fi, ts, None, opt)
    close(p1, g[key + "_pose1"])
    close(h1, g[key + "_h1"])
    p2, h2 = oc.pose_rnn(sd, fv.flip(0), fi.flip(0), ts, h1, opt)  # carried hidden state, layout [L,B,F]
    close(p2, g[key + "_pose2"])
    close(h2, g[key + "_h2"])


# ------------------------------------------------------------------------------------------------
# model.train() semantics of the encoders (scripts/train_model.py:219) against the REAL modules in train mode
# (tests/golden/train_mode.npz, made by oracle/gen_golden_train.py: the dropout masks the modules drew are part of the fixture)
# ------------------------------------------------------------------------------------------------
def _unpack(g, key):
    shape = tuple(int(x) for x in g[key + "_shape"])
    n = int(np.prod(shape))
    return torch.from_numpy(np.unpackbits(g[key])[:n].reshape(shape).astype(np.float32))


def test_image_encoder_train_mode(golden_dir):
    g = load(golden_dir, "train_mode.npz")
    H, W, B, S = int(g["img_H"]), int(g["img_W"]), int(g["img_B"]), int(g["img_S"])
    opt = default_opt(img_h=H, img_w=W)
    sd = weights.make_state_dict(opt, seed=int(g["img_wseed"]), randomize_stats=True)
    for step in range(2):                      # two consecutive steps: the second normalises with fresh batch statistics
        img = synth.images(B, S, H, W, seed=40 + step)   # and moves the buffers the first one left behind
        assert abs(float(img.double().sum()) - float(g[f"img{step}_sum"])) < 1e-6
        masks = [_unpack(g, f"img{step}_mask{i}") for i in range(9)]
        kept = [float(m.mean()) for m in masks]
        assert all(abs(k - 0.8) < 0.02 for k in kept[:8]) and abs(kept[8] - 0.5) < 0.05, kept   # Dropout(0.2) x 8, Dropout(0.5)
        new = {}
        fv = oc.image_encoder(sd, img, train=new, masks=masks)
        close(fv, g[f"img{step}_fv"])
        for k, v in new.items():
            ref = g[f"img{step}_buf_" + k[len("Image_net."):]]
            if "num_batches" in k:
                assert int(v) == int(ref) == step + 2           # the constructor's dummy forward counted once (Encoder.py:92-93)
            else:
                close(v, ref, tol=1e-5)
        sd = {**sd, **new}
    # eval mode afterwards reads the moved buffers: not what it computed before training
    assert oc.rel_err(oc.image_encoder(sd, img), oc.image_encoder(weights.make_state_dict(opt, seed=int(g["img_wseed"]), randomize_stats=True), img)) > 1e-3


def test_inertial_encoder_train_mode(golden_dir):
    g = load(golden_dir, "train_mode.npz")
    p = float(g["imu_p"])
    opt = default_opt(imu_dropout=p)
    sd = weights.make_state_dict(opt, seed=int(g["imu_wseed"]), randomize_stats=True)
    imu = synth.imu(3, 5, seed=9)
    assert abs(float(imu.double().sum()) - float(g["imu_sum"])) < 1e-6
    masks = [_unpack(g, f"imu_mask{i}") for i in range(3)]
    new = {}
    fi = oc.inertial_encoder(sd, imu, train=new, masks=masks, p_drop=p)
    close(fi, g["imu_fi"])
    for k, v in new.items():
        ref = g["imu_buf_" + k[len("Inertial_net."):]]
        if "num_batches" in k:
            assert int(v) == int(ref)
        else:
            close(v, ref, tol=1e-5)
