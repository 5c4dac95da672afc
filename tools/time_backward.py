#!/usr/bin/env python
"""Wall time of one pose-net training step below the encoders (forward on the persistent kernel + odevio_ode_rnn_bwd +
odevio_pose_loss) at the BASELINE configs[1] shape (B=16, 10 intervals).  Usage: python tools/time_backward.py [solver] [rnn]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from odevio_amd import DeepVIO, default_opt, synth, train  # noqa: E402

solver = sys.argv[1] if len(sys.argv) > 1 else "rk4"
rnn = sys.argv[2] if len(sys.argv) > 2 else "rnn"
opt = default_opt(img_h=64, img_w=128, ode_solver=solver, ode_rnn_type=rnn, freeze_encoder=True)
m = DeepVIO(opt, seed=0).cuda()
B, P = 16, 10
g = torch.Generator().manual_seed(0)
fv = torch.randn(B, P, 512, generator=g).cuda().requires_grad_(True)
fi = torch.randn(B, P, 256, generator=g).cuda().requires_grad_(True)
ts = synth.timestamps(B, P + 1, drop=0.5 if solver != "rk4" else 0.0, seed=1).cuda()
gts = torch.randn(B, P, 6, generator=g).cuda() * 0.1


def step():
    for p in m.parameters():
        p.grad = None
    poses, _ = train.pose_net(m, fv, fi, ts)
    loss = train.pose_loss(poses, gts)
    loss.backward()
    return loss


for _ in range(3):
    step()
m.check()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"pose-net training step (fwd + bwd below the encoders), B={B}, {P} intervals, {solver}, {rnn}: {dt * 1e3:.2f} ms")

# the whole optimizer step of the reference's loop (clip_grad_norm_ + Adam + the plan's layouts refreshed): PoseNetTrainer
trainer = train.PoseNetTrainer(m)
fvd, fid = fv.detach(), fi.detach()
for _ in range(3):
    trainer.step(fvd, fid, ts, gts)
m.check()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    trainer.step(fvd, fid, ts, gts)
torch.cuda.synchronize()
dt2 = (time.perf_counter() - t0) / n
ta = 0.0
for _ in range(n):
    trainer.zero_grad()
    poses, _ = train.pose_net(m, fvd, fid, ts)
    train.pose_loss(poses, gts).backward()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    trainer.apply_gradients()
    torch.cuda.synchronize()
    ta += time.perf_counter() - t1
print(f"  of which clip + Adam over {len(trainer.params)} tensors + plan refresh: {ta / n * 1e3:.2f} ms")
print(f"PoseNetTrainer.step (fwd + bwd + clip + Adam + plan refresh): {dt2 * 1e3:.2f} ms per step")
