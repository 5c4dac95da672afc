#!/usr/bin/env python
"""Wall time of the reference's WHOLE training step (scripts/train_model.py:63-86 under model.train()) at the BASELINE configs[1]
shape: 16 sequences x 11 frames of 256x512, RK4, on the device path - stage by stage (each stage synchronised, so the sum is an
upper bound of the unsynchronised step printed last).  Usage: python tools/time_train_step.py [frozen|full] [B]
  frozen: --freeze_encoder (the reference recipe): Image_net forward without a graph;  full: Image_net's backward too."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from odevio_amd import DeepVIO, default_opt, synth, train  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "full"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S = 11
opt = default_opt(ode_solver="rk4", freeze_encoder=(mode == "frozen"))
m = DeepVIO(opt, seed=0).cuda()
m.train()
img, imu, ts = synth.batch(B, S, 256, 512, seed=1)
img, imu, ts = img.cuda(), imu.cuda(), ts.cuda()
gts = torch.randn(B, S - 1, 6, generator=torch.Generator().manual_seed(0)).cuda() * 0.1
trainer = train.PoseNetTrainer(m)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


def fwd_img_nograd():
    with torch.no_grad():
        return m.image_encoder(img)


t_f, fv0 = timed(fwd_img_nograd)
print(f"Image_net forward, train mode (batch-statistics BatchNorm + dropout), no graph: {t_f:.2f} ms")
if mode != "frozen":
    t_k, fv = timed(lambda: train.image_encoder(m, img))
    print(f"Image_net forward, train mode, kept for the backward: {t_k:.2f} ms")
    g = torch.randn_like(fv)

    def bwd():
        trainer.zero_all_grads()
        f = train.image_encoder(m, img)
        f.backward(g)
    t_b, _ = timed(bwd, n=3)
    print(f"Image_net forward + backward (all 29 parameter gradients): {t_b:.2f} ms  (backward alone ~{t_b - t_k:.2f} ms)")
t_i, fi = timed(lambda: train.imu_encoder(m, imu))
print(f"Inertial_net forward, train mode, with graph: {t_i:.2f} ms")


def pose_step():
    trainer.zero_all_grads()
    poses, _ = train.pose_net(m, fv0, train.imu_encoder(m, imu), ts)
    train.pose_loss(poses, gts).backward()
t_p, _ = timed(pose_step)
print(f"Inertial_net + Pose_net forward + backward + loss: {t_p:.2f} ms")


def whole():
    trainer.zero_all_grads()
    out = trainer.accumulate(None, None, ts, gts, imu=imu, img=img)
    trainer.apply_gradients()
    return out
t_w, out = timed(whole, n=5)
m.check()
print(f"WHOLE step ({mode}: train-mode encoders, pose net, loss, backward, clip_grad_norm_, Adam, plan refresh), B={B}: {t_w:.2f} ms "
      f"= {B * S / (t_w * 1e-3):.0f} frames/s trained;  grad norm {float(trainer.grad_norm):.4f}, loss {float(out[0]):.5f}")
