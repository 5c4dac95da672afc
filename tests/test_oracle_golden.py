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
