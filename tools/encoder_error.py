#!/usr/bin/env python
"""Error of the image encoder (conv1..conv6 + visual head) against an fp64 evaluation of the same network, for the
two encoder arithmetic modes and for PyTorch's own fp32 CPU path (the oracle).  Needs a GPU.
Usage: python tools/encoder_error.py [H W]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from odevio_amd import DeepVIO, default_opt, synth, weights  # noqa: E402
from oracle import odevio_oracle as oc  # noqa: E402  (a measurement tool, not the product path)

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 512)
opt = default_opt(img_h=H, img_w=W)
sd = weights.make_state_dict(opt, seed=5, randomize_stats=True)
img = synth.images(2, 3, H, W, seed=1)
truth = oc.image_encoder(sd, img, torch.float64)
print(f"image encoder {H}x{W}, 4 pairs; error = max|x - fp64| / max|fp64|")
print(f"  oracle (PyTorch CPU fp32)        {oc.rel_err(oc.image_encoder(sd, img), truth):.3e}")
for mode in ("f16x2", "f32"):
    os.environ["ODEVIO_CONV_MATH"] = mode
    m = DeepVIO(opt, seed=5)
    m.load_state_dict(sd)
    m = m.cuda()
    fv = m.image_encoder(img.cuda())
    m.check()
    print(f"  HIP, ODEVIO_CONV_MATH={mode:6s}      {oc.rel_err(fv, truth):.3e}")
