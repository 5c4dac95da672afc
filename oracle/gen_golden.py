"""Generate tests/golden/*.npz from the REAL reference modules (build container only).

Run:  python oracle/gen_golden.py   (needs /root/reference; never runs on the GPU box)

Imports the importable reference modules (Encoder, FusionModule, ODEFunc, PoseRNN - SURVEY.md
section 8c), loads weights from this build's own seeded generator (``odevio_amd.weights``) into them
with ``load_state_dict``, runs them in ``eval()`` mode on seeded synthetic inputs
(``odevio_amd.synth``) and stores only DATA: the expected outputs plus a checksum of the inputs so
a drift of the generators is caught.  The un-importable pieces (PoseODERNN -> torchode,
PoseCDE -> torchcde) have no fixture: their parity is unpinned.
"""
import os
import sys

import numpy as np
import scipy.io as sio
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("ODEVIO_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from odevio_amd import synth, weights  # noqa: E402
from odevio_amd.config import default_opt  # noqa: E402

from src.models.Encoder import ImageEncoder, InertialEncoder  # noqa: E402  (reference)
from src.models.FusionModule import FusionModule  # noqa: E402
from src.models.ODEFunc import CDEFunc, ODEFunc  # noqa: E402
from src.models.PoseRNN import PoseRNN  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def chk(t):
    return np.float64(t.double().sum().item())


def main():
    torch.set_num_threads(1)
    torch.manual_seed(0)
    os.makedirs(OUT, exist_ok=True)

    # ---------------- G1: ImageEncoder (reduced size, all intermediates; one full-size pair)
    for tag, H, W, B, S, rnd in (("small", 64, 128, 2, 3, True), ("full", 256, 512, 1, 2, False)):
        opt = default_opt(img_h=H, img_w=W)
        sd = weights.make_state_dict(opt, seed=11, randomize_stats=rnd)
        net = ImageEncoder(opt)
        net.load_state_dict(sub(sd, "Image_net."))
        net.eval()
        img = synth.images(B, S, H, W, seed=3)
        with torch.no_grad():
            fv = net(img)
            v = torch.cat((img[:, :-1], img[:, 1:]), dim=2).view(B * (S - 1), 6, H, W)
            c1 = net.conv1(v)
            c3_1 = net.conv3_1(net.conv3(net.conv2(c1)))
            c6 = net.encode_image(v)
        np.savez_compressed(
            os.path.join(OUT, f"image_encoder_{tag}.npz"), H=H, W=W, B=B, S=S, wseed=11, iseed=3,
            randomize_stats=rnd, img_sum=chk(img), fv=fv.numpy(),
            conv1_sample=c1[:, ::8, ::8, ::8].contiguous().numpy(),
            conv3_1_sample=c3_1[:, ::16, ::2, ::2].contiguous().numpy(),
            conv6=c6.numpy() if tag == "small" else c6[:, ::32].contiguous().numpy())

    # ---------------- G2: InertialEncoder on real KITTI IMU (04.mat) and synthetic lengths
    opt = default_opt()
    sd = weights.make_state_dict(opt, seed=12, randomize_stats=True)
    net = InertialEncoder(opt)
    net.load_state_dict(sub(sd, "Inertial_net."))
    net.eval()
    mat = sio.loadmat(os.path.join(REF, "dataset", "imus", "04.mat"))["imu_data_interp"]
    real = torch.from_numpy(mat[:105].astype(np.float32))[None]  # [1,105,6]
    rec = {"wseed": 12, "imu04": real.numpy()}
    with torch.no_grad():
        for T in (11, 21, 51, 101, 105):
            rec[f"fi_T{T}"] = net(real[:, :T]).numpy()
        syn = synth.imu(3, 11, seed=5)
        rec["syn_sum"] = chk(syn)
        rec["fi_syn"] = net(syn).numpy()
    np.savez_compressed(os.path.join(OUT, "inertial_encoder.npz"), **rec)

    # ---------------- G3: FusionModule cat / soft
    rec = {"wseed": 13}
    g = torch.Generator().manual_seed(21)
    fv = torch.randn(2, 4, 512, generator=g)
    fi = torch.randn(2, 4, 256, generator=g)
    rec["fv"], rec["fi"] = fv.numpy(), fi.numpy()
    for method in ("cat", "soft"):
        opt = default_opt(fuse_method=method)
        sd = weights.make_state_dict(opt, seed=13, randomize_stats=True)
        net = FusionModule(768, method)
        net.load_state_dict(sub(sd, "Pose_net.fuse."))
        net.eval()
        with torch.no_grad():
            rec[method] = net(fv, fi).numpy()
    np.savez_compressed(os.path.join(OUT, "fusion.npz"), **rec)

    # ---------------- G4: ODEFunc for every activation x n in {2,3}, H in {512,1024}; CDEFunc
    rec = {"wseed": 14}
    g = torch.Generator().manual_seed(22)
    y = torch.randn(5, 768, generator=g) * 0.7
    rec["y"] = y.numpy()
    for act in ("tanh", "relu", "leaky_relu", "softplus"):
        for n in (2, 3):
            for H in (512, 1024):
                opt = default_opt(ode_activation_fn=act, ode_fn_num_layers=n, ode_hidden_dim=H)
                sd = weights.make_state_dict(opt, seed=14, randomize_stats=True)
                net = ODEFunc(768, H, n, act)
                net.load_state_dict(sub(sd, "Pose_net.ode_func."))
                net.eval()
                with torch.no_grad():
                    rec[f"f_{act}_{n}_{H}"] = net(torch.tensor(0.0), y).numpy()
    opt = default_opt(model_type="cde", cde_hidden_dim=128, v_f_len=96, i_f_len=32)
    sd = weights.make_state_dict(opt, seed=14, randomize_stats=True)
    net = CDEFunc(129, 128, 3, "tanh")
    net.load_state_dict(sub(sd, "Pose_net.cde_func."))
    net.eval()
    z = torch.randn(3, 128, generator=g) * 0.5
    rec["z"] = z.numpy()
    with torch.no_grad():
        rec["cde_f"] = net(torch.tensor(0.0), z).numpy()
    np.savez_compressed(os.path.join(OUT, "odefunc.npz"), **rec)

    # ---------------- G5: PoseRNN.forward (fuse -> RNN stack -> regressor), prev None and carried
    rec = {"wseed": 15}
    g = torch.Generator().manual_seed(23)
    fv = torch.randn(3, 10, 512, generator=g)
    fi = torch.randn(3, 10, 256, generator=g)
    ts = synth.timestamps(3, 11, drop=0.3, seed=2)
    rec["fv"], rec["fi"], rec["ts"] = fv.numpy(), fi.numpy(), ts.numpy()
    for rnn_type in ("rnn", "gru"):
        for L in (2, 3):
            for method in ("cat", "soft"):
                opt = default_opt(model_type="rnn", ode_rnn_type=rnn_type, rnn_num_layers=L, fuse_method=method)
                sd = weights.make_state_dict(opt, seed=15, randomize_stats=True)
                net = PoseRNN(opt)
                net.load_state_dict(sub(sd, "Pose_net."))
                net.eval()
                with torch.no_grad():
                    p1, h1 = net(fv, fi, ts, prev=None)
                    p2, h2 = net(fv.flip(0), fi.flip(0), ts, prev=h1)
                key = f"{rnn_type}_{L}_{method}"
                rec[key + "_pose1"], rec[key + "_h1"] = p1.numpy(), h1.numpy()
                rec[key + "_pose2"], rec[key + "_h2"] = p2.numpy(), h2.numpy()
    np.savez_compressed(os.path.join(OUT, "pose_rnn.npz"), **rec)
    print("golden fixtures written to", OUT)
    for fn in sorted(os.listdir(OUT)):
        print("  %-28s %8d B" % (fn, os.path.getsize(os.path.join(OUT, fn))))


if __name__ == "__main__":
    main()
