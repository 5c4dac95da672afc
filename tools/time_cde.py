#!/usr/bin/env python
"""Wall time of the Neural-CDE pose head (PoseCDE.forward after the encoders) at BASELINE configs[4]-like shapes.
Usage: python tools/time_cde.py [hidden] [B] [solver]"""
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
v = H * 3 // 4
opt = default_opt(img_h=64, img_w=128, model_type="cde", cde_hidden_dim=H, v_f_len=v, i_f_len=H - v, cde_solver=solver)
t0 = time.perf_counter()
model = DeepVIO(opt, seed=0).cuda().eval()
g = torch.Generator().manual_seed(0)
fv, fi = torch.randn(B, 10, v, generator=g) * 0.5, torch.randn(B, 10, H - v, generator=g) * 0.5
ts = synth.timestamps(B, 11, seed=1)
fv, fi, ts = fv.cuda(), fi.cuda(), ts.cuda()
model.Pose_net.history = None
poses, z0, stats = model.pose_cde(fv, fi, ts, None, return_stats=True)   # builds the plan, warm-up
torch.cuda.synchronize()
print(f"setup + first call {time.perf_counter() - t0:.1f} s; solver steps (total, accepted) = {stats}")
n = 3
t1 = time.perf_counter()
for _ in range(n):
    model.Pose_net.history = None
    model.pose_cde(fv, fi, ts, None)
torch.cuda.synchronize()
dt = (time.perf_counter() - t1) / n
print(f"PoseCDE hidden {H}, B={B}, 10 intervals, {solver}: {dt * 1e3:.1f} ms per window -> {B * 11 / dt:.1f} frames/s (pose head only)")
