"""Generate tests/golden/train_mode.npz from the REAL reference encoders under ``model.train()`` (build container only).

Run:  python oracle/gen_golden_train.py   (needs /root/reference; never runs on the GPU box)

The reference trains under ``model.train()`` (scripts/train_model.py:219): BatchNorm with batch statistics (and the running
statistics' update) and Dropout in both encoders.  Dropout is stochastic under torch's generator, so the masks the real modules
DREW are captured with forward hooks on their ``nn.Dropout`` children (kept = output != 0 where the input != 0) and stored,
bit-packed, beside the outputs and the updated buffers: the oracle's train-mode restatement is then checked with exactly those
masks (tests/test_oracle_golden.py).  DATA only: inputs come from this build's seeded generators (checksums stored).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("ODEVIO_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from odevio_amd import synth, weights  # noqa: E402
from odevio_amd.config import default_opt  # noqa: E402

from src.models.Encoder import ImageEncoder, InertialEncoder  # noqa: E402  (reference)

OUT = os.path.join(ROOT, "tests", "golden")


def sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def capture_masks(net):
    """Forward hooks on every nn.Dropout of `net`, in call order: keep mask = (output != 0) | (input == 0)."""
    masks, handles = [], []

    def hook(_mod, inp, out):
        masks.append(((out != 0) | (inp[0] == 0)).numpy())

    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            handles.append(m.register_forward_hook(hook))
    return masks, handles


def main():
    torch.set_num_threads(1)
    rec = {}
    # ---------------- ImageEncoder, 64 x 128, B = 2, S = 3 (4 pairs), two consecutive train-mode forwards (running stats move twice)
    H, W, B, S = 64, 128, 2, 3
    opt = default_opt(img_h=H, img_w=W)
    sd = weights.make_state_dict(opt, seed=17, randomize_stats=True)
    net = ImageEncoder(opt)
    net.load_state_dict(sub(sd, "Image_net."))
    net.train()
    masks, handles = capture_masks(net)
    torch.manual_seed(1234)
    for step in range(2):
        img = synth.images(B, S, H, W, seed=40 + step)
        del masks[:]
        with torch.no_grad():
            fv = net(img)
        rec[f"img{step}_sum"] = np.float64(img.double().sum().item())
        rec[f"img{step}_fv"] = fv.numpy()
        for i, m in enumerate(masks):
            rec[f"img{step}_mask{i}"] = np.packbits(m.reshape(-1))
            rec[f"img{step}_mask{i}_shape"] = np.asarray(m.shape)
        for k, v in net.state_dict().items():
            if "running" in k or "num_batches" in k:
                rec[f"img{step}_buf_{k}"] = v.numpy().copy()
    for h in handles:
        h.remove()
    rec.update(img_H=H, img_W=W, img_B=B, img_S=S, img_wseed=17)

    # ---------------- InertialEncoder, imu_dropout = 0.3, B = 3, S = 5 (12 pairs)
    opt = default_opt(imu_dropout=0.3)
    sd = weights.make_state_dict(opt, seed=18, randomize_stats=True)
    net = InertialEncoder(opt)
    net.load_state_dict(sub(sd, "Inertial_net."))
    net.train()
    masks, handles = capture_masks(net)
    torch.manual_seed(4321)
    imu = synth.imu(3, 5, seed=9)
    with torch.no_grad():
        fi = net(imu)
    rec["imu_sum"] = np.float64(imu.double().sum().item())
    rec["imu_fi"] = fi.numpy()
    for i, m in enumerate(masks):
        rec[f"imu_mask{i}"] = np.packbits(m.reshape(-1))
        rec[f"imu_mask{i}_shape"] = np.asarray(m.shape)
    for k, v in net.state_dict().items():
        if "running" in k or "num_batches" in k:
            rec[f"imu_buf_{k}"] = v.numpy().copy()
    for h in handles:
        h.remove()
    rec.update(imu_wseed=18, imu_p=0.3)
    path = os.path.join(OUT, "train_mode.npz")
    np.savez_compressed(path, **rec)
    print("written", path, os.path.getsize(path), "B")


if __name__ == "__main__":
    main()
