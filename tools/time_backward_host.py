#!/usr/bin/env python
"""Is the pose-path backward bound by the HOST's launch rate?  Host time until loss.backward() returns (everything enqueued)
against the time until the device is done.  Usage: python tools/time_backward_host.py [solver] [rnn]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from odevio_amd import DeepVIO, default_opt, synth, train  # noqa: E402

solver = sys.argv[1] if len(sys.argv) > 1 else "dopri5"
rnn = sys.argv[2] if len(sys.argv) > 2 else "gru"
opt = default_opt(img_h=64, img_w=128, ode_solver=solver, ode_rnn_type=rnn, freeze_encoder=True)
m = DeepVIO(opt, seed=0).cuda()
B, P = 16, 10
g = torch.Generator().manual_seed(0)
fv = torch.randn(B, P, 512, generator=g).cuda().requires_grad_(True)
fi = torch.randn(B, P, 256, generator=g).cuda().requires_grad_(True)
ts = synth.timestamps(B, P + 1, drop=0.5 if solver != "rk4" else 0.0, seed=1).cuda()
gts = torch.randn(B, P, 6, generator=g).cuda() * 0.1
for _ in range(3):
    poses, _ = train.pose_net(m, fv, fi, ts)
    train.pose_loss(poses, gts).backward()
torch.cuda.synchronize()
th = td = tf = 0.0
n = 10
for _ in range(n):
    for p in m.parameters():
        p.grad = None
    t0 = time.perf_counter()
    poses, _ = train.pose_net(m, fv, fi, ts)
    loss = train.pose_loss(poses, gts)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    tf += t1 - t0; th += t2 - t1; td += t3 - t1
print(f"{solver} {rnn}: forward {tf / n * 1e3:.2f} ms; backward: host returns after {th / n * 1e3:.2f} ms, device done after {td / n * 1e3:.2f} ms")
