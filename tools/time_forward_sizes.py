#!/usr/bin/env python
"""DeepVIO.forward at several batch sizes (the reference's evaluator streams ONE window at a time: B = 1): wall time per forward,
frames/s and the stage times.  Usage: python tools/time_forward_sizes.py [solver]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from odevio_amd import DeepVIO, default_opt, synth  # noqa: E402
solver = sys.argv[1] if len(sys.argv) > 1 else "rk4"
m = DeepVIO(default_opt(ode_solver=solver), seed=0).cuda()
for B in (1, 2, 4, 8, 16, 48):
    img, imu, ts = [t.cuda() for t in synth.batch(B, 11, 256, 512, seed=100)]
    for _ in range(5):
        m(img, imu, ts)
    m.check()
    torch.cuda.synchronize()
    steps = 30
    t0 = time.perf_counter()
    for _ in range(steps):
        m(img, imu, ts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    m.profile_enable(True, depth=5)
    for _ in range(5):
        m(img, imu, ts)
    st = m.profile_read()
    m.profile_enable(False)
    print(f"B={B:3d}: {dt * 1e3:7.3f} ms per forward, {B * 11 / dt:8.0f} frames/s; stages " + " ".join(f"{k}={v:.3f}" for k, v in st.items()))
