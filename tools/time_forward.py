#!/usr/bin/env python
"""Wall time of DeepVIO.forward at the BASELINE configs[1] shape without any stage timers (pure launch stream).
Usage: python tools/time_forward.py [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from odevio_amd import DeepVIO, default_opt, synth  # noqa: E402
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
m = DeepVIO(default_opt(ode_solver="rk4"), seed=0).cuda()
img, imu, ts = [t.cuda() for t in synth.batch(16, 11, 256, 512, seed=100)]
for _ in range(5):
    out = m(img, imu, ts)
m.check()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = m(img, imu, ts)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
m.check()
print(f"{dt * 1e3:.3f} ms per forward, {16 * 11 / dt:.0f} frames/s")
