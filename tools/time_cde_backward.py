#!/usr/bin/env python
"""Wall time of the Neural-CDE pose net's training forward + backward (odevio_cde_fwd / odevio_cde_bwd) at BASELINE configs[4]'s
shape: hidden 1024, 16 sequences x 10 intervals, dopri5.  Two windows: regular 10 Hz timestamps (relative time <= 1 s: the solve
never leaves piece 0 of the control path - only the time channel moves, H rows of the last layer take part) and a 50 % frame-drop
window (relative time > 1 s: odd pieces stream the whole 4.3 GB layer).  Usage: python tools/time_cde_backward.py [hidden]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from odevio_amd import DeepVIO, default_opt, synth, train  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
v = H * 3 // 4
opt = default_opt(img_h=64, img_w=128, model_type="cde", cde_hidden_dim=H, v_f_len=v, i_f_len=H - v, cde_solver="dopri5")
m = DeepVIO(opt, seed=0).cuda()
m.train()
B, P = 16, 10
g = torch.Generator().manual_seed(0)
fv = (torch.randn(B, P, v, generator=g) * 0.5).cuda().requires_grad_(True)
fi = (torch.randn(B, P, H - v, generator=g) * 0.5).cuda().requires_grad_(True)
w = torch.randn(B, P, 6, generator=g).cuda()
for name, drop in (("regular 10 Hz window (piece 0 only)", 0.0), ("50 % frame drop (crosses knots: odd pieces)", 0.5)):
    ts = synth.timestamps(B, P + 1, drop=drop, seed=1).cuda()

    def step():
        for q in m.parameters():
            q.grad = None
        poses, z0 = train.pose_cde(m, fv, fi, ts)
        (poses * w).sum().backward()

    step()
    m.check()
    torch.cuda.synchronize()
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    with torch.no_grad():
        t1 = time.perf_counter()
        for _ in range(n):
            m.pose_cde(fv.detach(), fi.detach(), ts)
        torch.cuda.synchronize()
        df = (time.perf_counter() - t1) / n
    print(f"PoseCDE hidden {H}, B={B}, {P} intervals, dopri5, {name}: window ends at t = {float(ts[0, -1] - ts[0, 0]):.1f} s; "
          f"forward {df * 1e3:.1f} ms, forward + backward {dt * 1e3:.1f} ms")
