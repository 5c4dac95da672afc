#!/usr/bin/env python
"""Timing of the Neural-CDE path at BASELINE configs[4]-like shapes.

  * one vector-field evaluation (odevio_cde_func) on an odd piece of the control path (the whole [H*(H+1), H] last layer
    streams from HBM: bytes / time against the 8 TB/s roofline) and on an even piece (H rows only),
  * one PoseCDE window (pose head only: fuse -> cdeint -> regressor) with the window's timestamps starting at `t0`.
Usage: python tools/time_cde.py [hidden] [B] [solver] [t0]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from odevio_amd import DeepVIO, default_opt, synth  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
solver = sys.argv[3] if len(sys.argv) > 3 else "dopri5"
t_start = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
v = H * 3 // 4
opt = default_opt(img_h=64, img_w=128, model_type="cde", cde_hidden_dim=H, v_f_len=v, i_f_len=H - v, cde_solver=solver)
t0 = time.perf_counter()
model = DeepVIO(opt, seed=0).cuda().eval()
g = torch.Generator().manual_seed(0)
fv, fi = torch.randn(B, 10, v, generator=g) * 0.5, torch.randn(B, 10, H - v, generator=g) * 0.5
ts = synth.timestamps(B, 11, seed=1) + t_start
fv, fi, ts = fv.cuda(), fi.cuda(), ts.cuda()

# ---- one evaluation of the vector field
obs = torch.cat([ts[:, 1:, None], torch.cat([fv, fi], -1)], -1).contiguous()
z = torch.tanh(torch.randn(B, H, generator=g)).cuda()
model.cde_func(z, obs, 1)
torch.cuda.synchronize()
print(f"setup + first evaluation {time.perf_counter() - t0:.1f} s")
w_bytes = H * (H + 1) * H * 4
for seg, what, nbytes in ((1, "odd piece (feature channels move): whole last layer", w_bytes),
                          (0, "even piece (time channel moves): H rows", H * H * 4)):
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        model.cde_func(z, obs, seg)
    e0.record()
    for _ in range(n):
        model.cde_func(z, obs, seg)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"CDEFunc evaluation, hidden {H}, B={B}, {what}: {ms * 1e3:.1f} us = {nbytes / ms / 1e6:.1f} GB/s "
          f"of last-layer weights ({nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s)")

# ---- one window of the pose head
model.Pose_net.history = None
poses, z0, stats = model.pose_cde(fv, fi, ts, None, return_stats=True)   # warm-up (and the step hint of the next solve)
torch.cuda.synchronize()
n = 3
t1 = time.perf_counter()
for _ in range(n):
    model.Pose_net.history = None
    model.pose_cde(fv, fi, ts, None)
torch.cuda.synchronize()
dt = (time.perf_counter() - t1) / n
print(f"PoseCDE hidden {H}, B={B}, 10 intervals from t={t_start + 0.1:.2f}, {solver}, steps (attempted, accepted) = {stats}: "
      f"{dt * 1e3:.2f} ms per window -> {B * 11 / dt:.1f} frames/s (pose head only)")
