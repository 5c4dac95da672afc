"""Times the production integrator (pose_net on features) with events; GPU box only."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from odevio_amd import DeepVIO, default_opt, synth
for solver, B, extra in (("rk4", 16, {}), ("dopri5", 16, {}), ("rk4", 1, {}), ("rk4", 26, {}), ("rk4", 16, dict(ode_substeps=4))):
    opt = default_opt(img_h=64, img_w=128, ode_solver=solver, **extra)
    m = DeepVIO(opt, seed=0).cuda()
    g = torch.Generator().manual_seed(0)
    fv, fi = torch.randn(B, 10, 512, generator=g).cuda(), torch.randn(B, 10, 256, generator=g).cuda()
    ts = synth.timestamps(B, 11).cuda()
    fused = m.fuse(fv, fi)
    for _ in range(3):
        m.pose_net(fv, fi, ts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 30
    for _ in range(n):
        m.pose_net(fv, fi, ts)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1000
    steps = 10 * extra.get("ode_substeps", 1)
    print(f"{solver} B={B} {extra}: pose_net {us:.1f} us per call" + (f" -> {steps/us*1e6:.0f} RK4 steps/s" if solver == "rk4" else ""))
