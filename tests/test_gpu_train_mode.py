"""GPU parity of the model.train() semantics.

The reference trains under ``model.train()`` (scripts/train_model.py:219, forward at :69): every BatchNorm of both encoders - the
frozen Image_net's too - normalises with batch statistics and moves its running statistics, every Dropout is on
(src/models/Encoder.py:8-22,43-57,82-90).  Dropout draws from torch's generator in the reference, so the device path (Philox) can
never be bit-equal to it: parity is checked by handing the DEVICE's masks to the oracle, whose train-mode arithmetic is itself pinned
against the real reference modules with the masks THEY drew (tests/test_oracle_golden.py, tests/golden/train_mode.npz).
"""
import numpy as np
import pytest
import torch

from odevio_amd import default_opt, synth, train
from oracle import odevio_oracle as oc
from oracle import philox as ph

from test_gpu_parity import assert_close, make_model

pytestmark = pytest.mark.gpu

IMAGE_P = oc.IMAGE_DROPOUT


def _image_masks(model, opt, seed, call0, n_pairs):
    """Keep masks (NCHW, 1 = kept) of the 9 blocks for draws call0 .. call0+8, from the device generator."""
    from odevio_amd import weights
    masks, h, w = [], opt.img_h, opt.img_w
    for i, (_, _, cout, k, s) in enumerate(weights.IMAGE_CONVS):
        h, w = weights.conv_out(h, k, s), weights.conv_out(w, k, s)
        f = model.dropout_mask(seed, call0 + i, IMAGE_P[i], (n_pairs, h, w, cout)).cpu()      # device order: NHWC
        masks.append((f != 0).float().permute(0, 3, 1, 2).contiguous())
    return masks


def _imu_masks(model, seed, call0, n_pairs, p):
    return [(model.dropout_mask(seed, call0 + j, p, (n_pairs, c, 11)).cpu() != 0).float() for j, c in enumerate((64, 128, 256))]


def test_dropout_generator_matches_the_philox_restatement():
    opt = default_opt(img_h=64, img_w=128)
    model, _ = make_model(opt, seed=70)
    for seed, call, p, n in ((0, 0, 0.2, 1003), (5, 7, 0.5, 4096), (0xABCDEF0123456789, (1 << 34) + 1, 0.3, 777), (1, 2, 0.0, 64)):
        dev = model.dropout_mask(seed, call, p, (n,)).cpu().numpy()
        ref = ph.dropout_factors(seed, call, p, n)
        assert np.array_equal(dev, ref), (seed, call, p)
    keep = float((model.dropout_mask(3, 0, 0.2, (1 << 20,)) != 0).float().mean())
    assert abs(keep - 0.8) < 2e-3, keep


def test_image_encoder_train_mode_three_steps():
    """ImageEncoder.forward under train(): fv to 1e-4 of the oracle fed the device's masks, for three consecutive steps, and the
    running statistics / num_batches_tracked afterwards equal torch's own update of the same buffers."""
    opt = default_opt(img_h=64, img_w=128)
    model, sd = make_model(opt, seed=71)
    model.train()
    model.set_seed(21)
    B, S = 3, 3
    P = B * (S - 1)
    ref_sd = dict(sd)
    for step in range(3):
        img = synth.images(B, S, 64, 128, seed=80 + step)
        seed, call0 = model.rng_state()
        assert (seed, call0) == (21, 9 * step)
        fv = model.image_encoder(img.cuda())
        model.check()
        new = {}
        ref = oc.image_encoder(ref_sd, img, train=new, masks=_image_masks(model, opt, seed, call0, P))
        assert_close(fv, ref, what=f"fv, train-mode step {step}")
        ref_sd = {**ref_sd, **new}
    now = model.state_dict()
    for k, v in ref_sd.items():
        if k.startswith("Image_net.") and ("running" in k or "num_batches" in k):
            if "num_batches" in k:
                assert int(now[k]) == int(v) == 4, k
            else:
                assert oc.rel_err(now[k], v) < 1e-5, (k, oc.rel_err(now[k], v))
    # eval mode afterwards: the plan folds the MOVED statistics (and matches the oracle on them), not the ones it was built with
    model.eval()
    fv_eval = model.image_encoder(img.cuda())
    assert_close(fv_eval, oc.image_encoder(ref_sd, img), what="fv, eval mode after training")
    assert oc.rel_err(fv_eval, oc.image_encoder(sd, img)) > 1e-3


def test_image_encoder_train_mode_full_size_batch():
    """The bench batch's shape in train mode (16 x 11 frames of 256 x 512: 5.2 M pixels per channel in conv1's statistics; split-K
    layers with the identity epilogue): finite, deterministic per seed, different per draw, and sequences 0..1 agree with the oracle
    run on the SAME batch statistics (the oracle walks the whole batch: BatchNorm couples the pairs)."""
    opt = default_opt()
    model, sd = make_model(opt, seed=72, randomize=False)
    model.train()
    B, S = 4, 3                       # 8 pairs at full resolution: the oracle's train-mode pass stays at seconds
    img = synth.images(B, S, 256, 512, seed=90)
    model.set_seed(5)
    fv = model.image_encoder(img.cuda())
    model.check()
    ref = oc.image_encoder(sd, img, train={}, masks=_image_masks(model, opt, 5, 0, B * (S - 1)))
    assert_close(fv, ref, what="fv, train mode at 256x512")
    model.set_seed(5)
    assert torch.equal(model.image_encoder(img.cuda()), fv)
    assert not torch.equal(model.image_encoder(img.cuda()), fv)
    img16 = synth.images(16, 11, 256, 512, seed=91).cuda()
    fv16 = model.image_encoder(img16)
    model.check()
    assert fv16.shape == (16, 10, 512) and torch.isfinite(fv16).all()


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_inertial_encoder_train_mode_forward_and_backward(p):
    """InertialEncoder under train(): forward (fi, running statistics) and every parameter gradient against autograd through the
    oracle with the device's masks - the batch-statistics BatchNorm backward and the masks of the same three draws."""
    opt = default_opt(img_h=64, img_w=128, imu_dropout=p)
    model, sd = make_model(opt, seed=73)
    model.train()
    model.set_seed(9)
    B, T = 3, 41
    n_pairs = B * 4
    imu = synth.imu(B, 5, seed=13)
    g = torch.Generator().manual_seed(2)
    w = torch.randn(B, 4, 256, generator=g)
    names = train.imu_param_names()
    fi = train.imu_encoder(model, imu.cuda())
    (fi * w.cuda()).sum().backward()
    model.check()
    assert model.rng_state() == (9, 3)
    leaves = {k: (v.clone().double().requires_grad_(k in names) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    new = {}
    ref = oc.inertial_encoder(leaves, imu, dtype=torch.float64, train=new, masks=_imu_masks(model, 9, 0, n_pairs, p), p_drop=p)
    (ref * w.double()).sum().backward()
    assert_close(fi, ref, what="fi (train mode)")
    params = dict(model.named_parameters())
    for n in names:
        got, want = params[n].grad, leaves[n].grad
        scale = float(want.abs().max())
        if n.endswith(("0.bias", "4.bias", "8.bias")):
            # a conv bias in front of a batch-statistics BatchNorm has a zero gradient in exact arithmetic: both sides hold rounding noise
            assert float(got.abs().max()) < 1e-4 * float(leaves[n.replace(".bias", ".weight")].grad.abs().max()) + 1e-6, n
            continue
        assert float((got.cpu().double() - want).abs().max()) <= 1e-3 * scale, (n, oc.rel_err(got, want))
    now = model.state_dict()
    for k, v in new.items():
        if "num_batches" in k:
            assert int(now[k]) == int(v)
        else:
            assert oc.rel_err(now[k], v.float()) < 1e-5, k


def test_deepvio_forward_in_train_mode_is_the_reference_training_forward():
    """model.train(); model(img, imu, ts) = what scripts/train_model.py:69 computes: train-mode encoders, then the pose net
    (which has no train-mode difference: rnn_dropout_out is unused, PoseODERNN.py:43-47)."""
    opt = default_opt(img_h=64, img_w=128, ode_solver="rk4", imu_dropout=0.2)
    model, sd = make_model(opt, seed=74)
    model.train()
    model.set_seed(31)
    img, imu, ts = synth.batch(3, 4, 64, 128, seed=23)
    with torch.no_grad():
        poses, h = model(img.cuda(), imu.cuda(), ts.cuda())
    model.check()
    assert model.rng_state() == (31, 12)                                     # 9 image-encoder draws, then 3 inertial-encoder draws
    fv = oc.image_encoder(sd, img, train={}, masks=_image_masks(model, opt, 31, 0, 9))
    fi = oc.inertial_encoder(sd, imu, train={}, masks=_imu_masks(model, 31, 9, 9, 0.2), p_drop=0.2)
    ref_p, ref_h = oc.pose_ode_rnn(sd, fv, fi, ts, None, opt)
    assert_close(poses, ref_p, what="poses (train-mode forward)")
    assert_close(h, ref_h, what="h_T (train-mode forward)")
    with pytest.raises(ValueError):
        model(torch.zeros(1, 3, 64, 128, 3, dtype=torch.uint8, device="cuda"), imu[:1, :21].cuda(), ts[:1, :3].cuda())


def _loss(poses, gts):
    return 100 * torch.nn.functional.mse_loss(poses[:, :, :3], gts[:, :, :3]) + torch.nn.functional.mse_loss(poses[:, :, 3:], gts[:, :, 3:])


@pytest.mark.parametrize("optimizer,accum", [("Adam", 1), ("SGD", 1), ("Adam", 2)])
def test_train_epoch_follows_torch_with_batchnorm_in_train_mode(optimizer, accum):
    """train_epoch against the reference's loop restated on the oracle (scripts/train_model.py:48-95 under model.train()): frozen
    Image_net in train mode, the inertial encoder in the graph with batch-statistics BatchNorm + Dropout (device masks handed over),
    clip_grad_norm_ over Pose_net + Inertial_net, torch.optim.Adam / SGD over Pose_net, gradient accumulation.  Loss trajectory,
    gradient norms and parameters after the epoch."""
    opt = default_opt(img_h=64, img_w=128, ode_solver="rk4", freeze_encoder=True, imu_dropout=0.1, optimizer=optimizer,
                      grad_accumulation_steps=accum, gradient_clip=0.05 if optimizer == "Adam" else 5.0, lr_warmup=1e-4)
    model, sd = make_model(opt, seed=75)
    model.set_seed(77)
    B, S = 3, 4
    P = B * (S - 1)
    scale = torch.tensor([0.01, 0.02, 0.01, 0.05, 0.05, 1.0])
    batches = []
    for k in range(4):
        img, imu, ts = synth.batch(B, S, 64, 128, seed=60 + k)
        g = torch.Generator().manual_seed(70 + k)
        batches.append((img, imu, torch.randn(B, S - 1, 6, generator=g) * scale, ts, "synthetic"))
    trainer = train.PoseNetTrainer(model)
    losses, norms = [], []
    mean = train.train_epoch(model, trainer, batches, log=lambda m: (losses.append(float(m.split("pose loss: ")[1].split(",")[0])),
                                                                     norms.append(float(m.split("grad norm: ")[1]))), log_every=1)
    model.check()
    # ---- the same epoch in PyTorch on the oracle, fed the device's masks (12 draws per batch: 9 image blocks, 3 inertial blocks)
    inames = train.imu_param_names()
    leaves = {k: (v.clone().float() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    params = [leaves[n].requires_grad_(True) for n in trainer.names]
    iparams = [leaves[n].requires_grad_(True) for n in inames]
    if optimizer == "Adam":
        optim = torch.optim.Adam(params, lr=opt.lr_warmup, betas=(0.9, 0.999), eps=1e-8, weight_decay=opt.weight_decay)
    else:
        optim = torch.optim.SGD(params, lr=opt.lr_warmup, momentum=0.9)
    ref_losses, ref_norms = [], []
    optim.zero_grad()
    for i, (img, imu, gts, ts, _) in enumerate(batches):
        new = {}
        with torch.no_grad():
            fv = oc.image_encoder(leaves, img, train=new, masks=_image_masks(model, opt, 77, 12 * i, P))
        fi = oc.inertial_encoder(leaves, imu, train=new, masks=_imu_masks(model, 77, 12 * i + 9, P, 0.1), p_drop=0.1)
        poses, _ = oc.pose_ode_rnn(leaves, fv, fi, ts, None, opt)
        loss = _loss(poses, gts)
        loss.backward()
        for k, v in new.items():
            leaves[k] = v
        if (i + 1) % accum == 0 or i + 1 == len(batches):
            ref_norms.append(float(torch.nn.utils.clip_grad_norm_(params + iparams, max_norm=opt.gradient_clip)))
            optim.step()
            optim.zero_grad()
            for q in iparams:
                q.grad = None
        ref_losses.append(float(loss.detach()))
    assert len(losses) == 4 and abs(mean - sum(ref_losses) / 4) <= 3e-4 * abs(mean)
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 3e-4 * abs(b) + 1e-6, (losses, ref_losses)
    dev_norms = [norms[i] for i in range(len(batches)) if (i + 1) % accum == 0 or i + 1 == len(batches)]
    for a, b in zip(dev_norms, ref_norms):
        assert abs(a - b) <= 3e-3 * b, (dev_norms, ref_norms)
    lr = opt.lr_warmup
    for n, p in zip(trainer.names, trainer.params):
        d = (p.detach().cpu() - leaves[n].detach()).abs()
        if optimizer == "Adam":
            assert float(d.max()) <= 1.0 * lr * len(ref_norms), n
            assert float((d > 0.03 * lr).float().mean()) < 3e-3, n
        else:
            moved = float((leaves[n].detach() - sd[n]).abs().max())
            ulp = 1.2e-7 * float(leaves[n].detach().abs().max())          # the update is only a few hundred ulp of the parameter itself
            assert float(d.max()) <= 2e-3 * moved + 2 * ulp, (n, float(d.max()), moved)
    now = model.state_dict()
    for k in now:
        if "running" in k:
            assert oc.rel_err(now[k], leaves[k]) < 2e-5, k


@pytest.mark.parametrize("hw,B,S", [((64, 128), 2, 3), ((96, 160), 1, 3), ((128, 256), 1, 2)])   # (128 x 256: rows of 64 / 32 pixels - the row-wise weight gradient)
def test_image_encoder_backward_matches_autograd_through_the_oracle(hw, B, S, monkeypatch):
    """odevio_image_encoder_bwd: every Image_net parameter gradient (nine Conv2d weights, BatchNorm gamma / beta, the visual head)
    against torch.autograd through the oracle's train-mode image encoder in float64, fed the device's dropout masks - Dropout,
    LeakyReLU, batch-statistics BatchNorm backward, weight gradients (contraction over all pixels) and input gradients (stride-1
    convolutions of the zero-dilated gradient with the reversed filters; 96 x 160: odd sizes in the deeper blocks)."""
    H, W = hw
    opt = default_opt(img_h=H, img_w=W)
    model, sd = make_model(opt, seed=78)
    model.train()
    model.set_seed(41)
    img = synth.images(B, S, H, W, seed=95)
    g = torch.Generator().manual_seed(4)
    wgt = torch.randn(B, S - 1, 512, generator=g)
    names = train.image_param_names()
    fv = train.image_encoder(model, img.cuda())
    (fv * wgt.cuda()).sum().backward()
    model.check()
    leaves = {k: (v.clone().double().requires_grad_(k in names) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    ref = oc.image_encoder(leaves, img, dtype=torch.float64, train={}, masks=_image_masks(model, opt, 41, 0, B * (S - 1)))
    (ref * wgt.double()).sum().backward()
    assert_close(fv, ref, what="fv (train mode, kept for the backward)")
    params = dict(model.named_parameters())
    worst = {}
    for n in names:
        got, want = params[n].grad, leaves[n].grad
        assert got is not None and got.shape == want.shape, n
        worst[n] = float((got.cpu().double() - want).abs().max() / want.abs().max().clamp_min(1e-30))
    bad = {n: e for n, e in worst.items() if e > 1e-3}
    assert not bad, bad
    # conv2 / conv3's input gradients above ran as four parity sub-convolutions of the undilated gradient; the dilated form and conv1's
    # unfolded weight gradient (64-wide channel tile per tap) must give the same numbers
    first = {n: params[n].grad.clone() for n in names}
    for n in names:
        params[n].grad = None
    monkeypatch.setenv("ODEVIO_DGRAD_DILATED", "1")
    monkeypatch.setenv("ODEVIO_WGRAD_NO_FOLD", "1")
    monkeypatch.setenv("ODEVIO_WGRAD_PER_TAP", "1")      # (and the per-tap weight-gradient kernel instead of the row-wise one)
    model.set_seed(41)
    fv2 = train.image_encoder(model, img.cuda())
    (fv2 * wgt.cuda()).sum().backward()
    model.check()
    monkeypatch.delenv("ODEVIO_DGRAD_DILATED")
    monkeypatch.delenv("ODEVIO_WGRAD_NO_FOLD")
    monkeypatch.delenv("ODEVIO_WGRAD_PER_TAP")
    assert torch.equal(fv, fv2)
    for n in names:
        e = float((params[n].grad - first[n]).abs().max() / first[n].abs().max().clamp_min(1e-30))
        assert e < 5e-5, (n, e)
    assert any(not torch.equal(params[n].grad, first[n]) for n in names)
    with pytest.raises(ValueError):                                  # a backward needs ITS forward: another shape has nothing kept
        from odevio_amd import _lib
        arr = train._tensor_array(["Image_net.visual_head.bias"], [torch.empty(512, device="cuda")])
        _lib.check(model._lib.odevio_image_encoder_bwd(model._plan, img.cuda().data_ptr(), B + 1, S, fv.data_ptr(), 512, arr, 1, model._stream()))


def test_trainer_counts_image_encoder_gradients_in_the_clip_norm_when_not_frozen():
    """freeze_encoder = False (the reference's DEFAULT flag value, scripts/config.py:32): loss.backward() reaches Image_net, whose
    gradients count in clip_grad_norm_(model.parameters()) (scripts/train_model.py:84) while the optimizer still only holds Pose_net
    (utils/utils.py:116-119).  One train_epoch step against the same loop on the oracle: the total norm (dominated by what Image_net
    adds), the losses, and Image_net's weights untouched."""
    opt = default_opt(img_h=64, img_w=128, ode_solver="rk4", freeze_encoder=False, gradient_clip=0.05, lr_warmup=1e-4)
    model, sd = make_model(opt, seed=79)
    model.set_seed(55)
    B, S = 2, 3
    P = B * (S - 1)
    img, imu, ts = synth.batch(B, S, 64, 128, seed=66)
    gts = torch.randn(B, S - 1, 6, generator=torch.Generator().manual_seed(5)) * torch.tensor([0.01, 0.02, 0.01, 0.05, 0.05, 1.0])
    trainer = train.PoseNetTrainer(model)
    out = []
    train.train_epoch(model, trainer, [(img, imu, gts, ts, "synthetic")], log=out.append, log_every=1)
    model.check()
    dev_loss, dev_norm = float(out[0].split("pose loss: ")[1].split(",")[0]), float(trainer.grad_norm)
    leaves = {k: (v.clone().float() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    names = [k for k in leaves if leaves[k].is_floating_point() and "running" not in k and k in dict(model.named_parameters())]
    for n in names:
        leaves[n].requires_grad_(True)
    fv = oc.image_encoder(leaves, img, train={}, masks=_image_masks(model, opt, 55, 0, P))
    fi = oc.inertial_encoder(leaves, imu, train={}, masks=_imu_masks(model, 55, 9, P, 0.0), p_drop=0.0)
    poses, _ = oc.pose_ode_rnn(leaves, fv, fi, ts, None, opt)
    loss = _loss(poses, gts)
    loss.backward()
    with_image = float(torch.nn.utils.clip_grad_norm_([leaves[n] for n in names], max_norm=0.05))
    without = float(torch.sqrt(sum((leaves[n].grad.double() ** 2).sum() for n in names if not n.startswith("Image_net."))))
    assert abs(dev_loss - float(loss)) <= 3e-4 * abs(float(loss)) + 1e-6
    assert abs(dev_norm - with_image) <= 3e-3 * with_image, (dev_norm, with_image, without)
    assert with_image > 1.05 * without                                # Image_net's share is visible: the frozen-encoder norm would be wrong
    now = dict(model.named_parameters())
    for n in names:
        if n.startswith(("Image_net.", "Inertial_net.")):
            assert torch.equal(now[n].detach().cpu(), sd[n]), n       # never the optimizer's


def test_the_reference_training_loop_runs_unchanged_on_this_model():
    """The drop-in property for TRAINING: the body of the reference's train() (scripts/train_model.py:54-86) with the reference's own
    optimizer factory (utils/utils.py:115-130: torch.optim.Adam over Pose_net's two parameter groups) and its --freeze_encoder lines
    (:191-194), written here exactly as the reference writes them, runs on odevio_amd.DeepVIO: model(...) in train() returns poses
    with an autograd graph whose nodes are libodevio calls, loss.backward() fills .grad, clip_grad_norm_(model.parameters()) and
    optimizer.step() are PyTorch's own - and the next forward sees the stepped weights through an in-place plan refresh (the plan
    object is never rebuilt).  Checked against the same loop on the oracle, fed the device's dropout masks."""
    opt = default_opt(img_h=64, img_w=128, ode_solver="rk4", freeze_encoder=True, weight_decay=5e-5, gradient_clip=5, lr_warmup=1e-4)
    model, sd = make_model(opt, seed=87)
    model.set_seed(3)
    B, S = 2, 4
    P = B * (S - 1)
    scale = torch.tensor([0.01, 0.02, 0.01, 0.05, 0.05, 1.0])
    train_loader = []
    for k in range(3):
        img, imu, ts = synth.batch(B, S, 64, 128, seed=110 + k)
        train_loader.append((img, imu, torch.randn(B, S - 1, 6, generator=torch.Generator().manual_seed(k)) * scale, ts, "synthetic"))
    args = opt
    # ---- scripts/train_model.py:191-194
    if args.freeze_encoder:
        for param in model.Image_net.parameters():
            param.requires_grad = False
    # ---- utils/utils.py:115-130
    param_groups = [{"params": model.Pose_net.get_other_params(), "lr": args.lr_warmup},
                    {"params": model.Pose_net.get_regressor_params(), "lr": args.lr_warmup}]
    optimizer = torch.optim.Adam(param_groups, lr=args.lr_warmup, betas=(0.9, 0.999), eps=1e-08, weight_decay=args.weight_decay)
    # ---- scripts/train_model.py:219, :54-86
    model.train()
    mse_losses, plans = [], set()
    data_len = len(train_loader)
    optimizer.zero_grad()
    for i, (imgs, imus, gts, timestamps, folder) in enumerate(train_loader):
        imgs, imus, gts, timestamps = imgs.cuda().float(), imus.cuda().float(), gts.cuda().float(), timestamps.cuda().float()
        poses, _ = model(imgs, imus, timestamps, hc=None)
        angle_loss = torch.nn.functional.mse_loss(poses[:, :, :3], gts[:, :, :3])
        translation_loss = torch.nn.functional.mse_loss(poses[:, :, 3:], gts[:, :, 3:])
        pose_loss = 100 * angle_loss + translation_loss
        loss = pose_loss
        loss.backward()
        if (i + 1) % args.grad_accumulation_steps == 0 or (i + 1) == data_len:
            if args.gradient_clip:
                torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=args.gradient_clip)
                optimizer.step()
            optimizer.zero_grad()
        mse_losses.append(pose_loss.item())
        plans.add(model._plan.value)
    model.check()
    assert len(plans) == 1, "a torch optimizer step on Pose_net must refresh the plan in place, not rebuild it"
    # ---- the same loop on the oracle
    inames = train.imu_param_names()
    leaves = {k: (v.clone().float() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    pnames = [n for n, _ in model.named_parameters() if n.startswith("Pose_net.")]
    params = [leaves[n].requires_grad_(True) for n in pnames]
    iparams = [leaves[n].requires_grad_(True) for n in inames]
    ref_opt = torch.optim.Adam(params, lr=args.lr_warmup, betas=(0.9, 0.999), eps=1e-08, weight_decay=args.weight_decay)
    ref_losses = []
    for i, (img, imu, gts, ts, _) in enumerate(train_loader):
        new = {}
        with torch.no_grad():
            fv = oc.image_encoder(leaves, img, train=new, masks=_image_masks(model, opt, 3, 12 * i, P))
        fi = oc.inertial_encoder(leaves, imu, train=new, masks=_imu_masks(model, 3, 12 * i + 9, P, 0.0), p_drop=0.0)
        poses, _ = oc.pose_ode_rnn(leaves, fv, fi, ts, None, opt)
        loss = _loss(poses, gts)
        loss.backward()
        leaves.update(new)
        torch.nn.utils.clip_grad_norm_(params + iparams, max_norm=args.gradient_clip)
        ref_opt.step()
        ref_opt.zero_grad()
        for q in iparams:
            q.grad = None
        ref_losses.append(float(loss.detach()))
    for a, b in zip(mse_losses, ref_losses):
        assert abs(a - b) <= 3e-4 * abs(b) + 1e-6, (mse_losses, ref_losses)
    now = dict(model.named_parameters())
    lr = args.lr_warmup
    for n in pnames:
        d = (now[n].detach().cpu() - leaves[n].detach()).abs()
        assert float(d.max()) <= 1.0 * lr * 3 and float((d > 0.03 * lr).float().mean()) < 3e-3, n
        assert not torch.equal(now[n].detach().cpu(), sd[n])         # it did train
    # and eval mode afterwards computes on the trained weights and the moved running statistics
    model.eval()
    img, imu, ts = train_loader[0][0], train_loader[0][1], train_loader[0][3]
    poses, _ = model(img.cuda(), imu.cuda(), ts.cuda())
    ref, _ = oc.deepvio_forward({k: v.detach().cpu() for k, v in model.state_dict().items()}, img, imu, ts, None, opt)
    assert_close(poses, ref, what="poses (eval after the reference's loop)")


def test_a_falsy_gradient_clip_means_no_update_like_the_reference():
    """scripts/train_model.py:83-86: `if args.gradient_clip: clip; optimizer.step()` - with gradient_clip = 0 the gradients are
    zeroed and nothing is updated."""
    opt = default_opt(img_h=64, img_w=128, ode_solver="rk4", freeze_encoder=True, gradient_clip=0)
    model, _ = make_model(opt, seed=76)
    g = torch.Generator().manual_seed(10)
    fv, fi = torch.randn(2, 3, 512, generator=g).cuda(), torch.randn(2, 3, 256, generator=g).cuda()
    ts, gts = synth.timestamps(2, 4, seed=3).cuda(), torch.randn(2, 3, 6, generator=g).cuda()
    trainer = train.PoseNetTrainer(model)
    before = [p.detach().clone() for p in trainer.params]
    trainer.step(fv, fi, ts, gts)
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, trainer.params)) and trainer.steps == 0
