"""Gradients of the ODE-RNN pose path (odevio_ode_rnn_bwd, odevio_pose_loss) against torch.autograd through the oracle.

The oracle is plain PyTorch, so autograd through ``oracle.pose_ode_rnn`` IS the reference's training-step gradient
(scripts/train_model.py:69-78: loss = 100 * MSE(angles) + MSE(translations), backward through the solver's operations).
It runs in float64 here so that the comparison measures the HIP path, not fp32 noise in the reference.
Bar: 1e-3 of each gradient tensor's max."""
import pytest
import torch

from odevio_amd import default_opt, synth, train
from oracle import odevio_oracle as oc

from test_gpu_parity import make_model

pytestmark = pytest.mark.gpu
GTOL = 1e-3


def _oracle_grads(sd, fv, fi, ts, hc, gts, opt, names, dtype=torch.float64):
    leaves = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    fv64, fi64 = fv.clone().to(dtype).requires_grad_(True), fi.clone().to(dtype).requires_grad_(True)   # clones: .to() of the same dtype is the tensor itself
    hc64 = None if hc is None else hc.clone().to(dtype).requires_grad_(True)
    with_ode = opt.model_type == "ode-rnn"
    # the step sizes an adaptive controller picked are constants of the differentiation (the backward replays the accepted steps)
    poses, h_T = oc.pose_ode_rnn(leaves, fv64, fi64, ts, hc64, opt, dtype=dtype, with_ode=with_ode, detach_controller=True)
    loss = 100 * torch.nn.functional.mse_loss(poses[:, :, :3], gts[:, :, :3].to(dtype)) + \
        torch.nn.functional.mse_loss(poses[:, :, 3:], gts[:, :, 3:].to(dtype))
    loss.backward()
    out = {"fv": fv64.grad, "fi": fi64.grad, "loss": loss.detach(), "poses": poses.detach()}
    if hc64 is not None:
        out["hc"] = hc64.grad
    for n in names:
        out[n] = leaves[n].grad
    return out


@pytest.mark.parametrize("cfg", [
    dict(ode_solver="rk4"),
    dict(ode_solver="rk4_classic", ode_substeps=2, ode_activation_fn="leaky_relu"),
    dict(ode_solver="rk4", rnn_num_layers=3, ode_activation_fn="softplus", ode_fn_num_layers=2, ode_hidden_dim=1024),  # the reference recipe's shapes
    dict(ode_solver="rk4", ode_activation_fn="relu", rnn_num_layers=1),
    dict(model_type="rnn"),
    dict(ode_solver="rk4", ode_rnn_type="gru"),
    dict(model_type="rnn", ode_rnn_type="gru", rnn_num_layers=3),
    dict(ode_solver="rk4", fuse_method="soft"),                       # FusionModule "soft": its Linear is a Pose_net parameter
    dict(model_type="rnn", fuse_method="soft", rnn_num_layers=1),
])
@pytest.mark.parametrize("with_hc", [False, True])
def test_ode_rnn_backward_matches_autograd_through_the_oracle(cfg, with_hc):
    opt = default_opt(img_h=64, img_w=128, **cfg)
    model, sd = make_model(opt, seed=71)
    B, P, L, F = 3, 4, opt.rnn_num_layers, 768
    g = torch.Generator().manual_seed(5)
    fv, fi = torch.randn(B, P, 512, generator=g), torch.randn(B, P, 256, generator=g)
    ts = synth.timestamps(B, P + 1, drop=0.3, seed=3, absolute=with_hc)
    hc = torch.randn(L, B, F, generator=g) * 0.3 if with_hc else None
    gts = torch.randn(B, P, 6, generator=g) * torch.tensor([0.01, 0.02, 0.01, 0.05, 0.05, 1.0])
    names = train.fuse_param_names(opt) + train.pose_param_names(opt)
    ref = _oracle_grads(sd, fv, fi, ts, hc, gts, opt, names)

    fv_d, fi_d = fv.cuda().requires_grad_(True), fi.cuda().requires_grad_(True)
    hc_d = None if hc is None else hc.cuda().requires_grad_(True)
    poses, h_T = train.pose_net(model, fv_d, fi_d, ts.cuda(), hc_d)
    loss = train.pose_loss(poses, gts.cuda())
    loss.backward()
    model.check()
    assert oc.rel_err(poses, ref["poses"]) < 1e-4
    assert abs(float(loss.detach()) - float(ref["loss"])) <= 1e-4 * abs(float(ref["loss"]))
    errs = {"fv": oc.rel_err(fv_d.grad, ref["fv"]), "fi": oc.rel_err(fi_d.grad, ref["fi"])}
    if hc is not None:
        errs["hc"] = oc.rel_err(hc_d.grad, ref["hc"])
    params = dict(model.named_parameters())
    for n in names:
        assert params[n].grad is not None, n
        errs[n] = oc.rel_err(params[n].grad, ref[n])
    bad = {k: f"{v:.2e}" for k, v in errs.items() if not v < GTOL}
    assert not bad, f"gradients off by more than {GTOL}: {bad}"


@pytest.mark.parametrize("cfg", [
    dict(ode_solver="dopri5"),
    dict(ode_solver="dopri5", rnn_num_layers=3, ode_activation_fn="softplus", ode_fn_num_layers=2, ode_hidden_dim=1024),  # the reference's training recipe (scripts/run_training.sh:6-28)
    dict(ode_solver="tsit5", ode_rnn_type="gru"),
    dict(ode_solver="heun"),
])
def test_adaptive_solver_backward_replays_the_accepted_steps(cfg):
    """Adaptive solvers: the backward replays the ACCEPTED steps of the forward (logged by the integrator kernel) with
    their sizes held constant.  Reference: autograd through the oracle in fp32 with the controller detached - fp32 so
    that both sides take the same step decisions (tests/test_gpu_parity.py pins the step sequences); a sequence that
    differs in a borderline accept/reject would change the gradient at the tolerance's own level (rtol 1e-2)."""
    opt = default_opt(img_h=64, img_w=128, **cfg)
    model, sd = make_model(opt, seed=73)
    B, P, L, F = 3, 4, opt.rnn_num_layers, 768
    g = torch.Generator().manual_seed(6)
    fv, fi = torch.randn(B, P, 512, generator=g), torch.randn(B, P, 256, generator=g)
    ts = synth.timestamps(B, P + 1, drop=0.4, seed=4, absolute=True)
    hc = torch.randn(L, B, F, generator=g) * 0.3
    gts = torch.randn(B, P, 6, generator=g) * torch.tensor([0.01, 0.02, 0.01, 0.05, 0.05, 1.0])
    names = train.pose_param_names(opt)
    ref = _oracle_grads(sd, fv, fi, ts, hc, gts, opt, names, dtype=torch.float32)
    fv_d, fi_d, hc_d = fv.cuda().requires_grad_(True), fi.cuda().requires_grad_(True), hc.cuda().requires_grad_(True)
    poses, h_T = train.pose_net(model, fv_d, fi_d, ts.cuda(), hc_d)
    train.pose_loss(poses, gts.cuda()).backward()
    model.check()
    assert oc.rel_err(poses, ref["poses"]) < 1e-4
    errs = {"fv": oc.rel_err(fv_d.grad, ref["fv"]), "fi": oc.rel_err(fi_d.grad, ref["fi"]), "hc": oc.rel_err(hc_d.grad, ref["hc"])}
    params = dict(model.named_parameters())
    for n in names:
        errs[n] = oc.rel_err(params[n].grad, ref[n])
    bad = {k: f"{v:.2e}" for k, v in errs.items() if not v < GTOL}
    assert not bad, f"gradients off by more than {GTOL}: {bad}"


@pytest.mark.parametrize("cfg,B", [(dict(ode_solver="rk4", ode_substeps=2), 5), (dict(ode_solver="dopri5", ode_rnn_type="gru"), 5),
                                   (dict(ode_solver="rk4"), 40),                       # two chunks of rows (32 + 8 sequences) per persistent launch
                                   (dict(ode_solver="heun", rnn_num_layers=3), 19)])   # three RNN layers: chunks of 16 sequences
def test_taped_backward_equals_the_plain_one_and_the_step_by_step_tape(cfg, B, monkeypatch):
    """Three routes to the same gradients: (a) the autograd function (odevio_ode_rnn_fwd_taped keeps the forward's log, _bwd_taped reads
    it: the persistent kernel runs once per step), (b) the plain pair (the backward runs the forward again to write the same log) -
    the same arithmetic on the same log, so bit-equal - and (c) ODEVIO_TAPE_IN_ORDER=1: no logged states, the tape walks the steps in
    order from its own recomputed states (the route of windows whose states would not fit) - equal to rounding."""
    import ctypes
    from odevio_amd import _lib
    opt = default_opt(img_h=64, img_w=128, **cfg)
    model, _ = make_model(opt, seed=74)
    P, L, F = 4, opt.rnn_num_layers, 768
    g = torch.Generator().manual_seed(8)
    fused = torch.randn(B, P, F, generator=g).cuda()
    ts = synth.timestamps(B, P + 1, drop=0.4, seed=5, absolute=True).cuda()
    hc = (torch.randn(L, B, F, generator=g) * 0.3).cuda()
    gp = (torch.randn(B, P, 6, generator=g) * 0.1).cuda()
    ghT = (torch.randn(L, B, F, generator=g) * 0.01).cuda()
    names = train.pose_param_names(opt)
    params = dict(model.named_parameters())
    model._ensure_plan()

    def run(route):
        out = {"fused": torch.empty_like(fused), "hc": torch.empty_like(hc)}
        grads = [torch.empty_like(params[n]) for n in names]
        arr = train._tensor_array(names, grads)
        poses, h_T = torch.empty(B, P, 6, device="cuda"), torch.empty(L, B, F, device="cuda")
        common = (model._plan, fused.data_ptr(), ts.data_ptr(), hc.data_ptr(), B, P, gp.data_ptr(), ghT.data_ptr(), out["fused"].data_ptr(),
                  out["hc"].data_ptr(), arr, len(grads))
        if route == "taped":
            n = ctypes.c_int64(0)
            _lib.check(model._lib.odevio_ode_rnn_tape_floats(model._plan, B, P, ctypes.byref(n)))
            assert n.value > 0
            tape = torch.empty(n.value, device="cuda")
            _lib.check(model._lib.odevio_ode_rnn_fwd_taped(model._plan, fused.data_ptr(), ts.data_ptr(), hc.data_ptr(), B, P, poses.data_ptr(),
                                                           h_T.data_ptr(), tape.data_ptr(), n.value, model._stream()))
            _lib.check(model._lib.odevio_ode_rnn_bwd_taped(*common, tape.data_ptr(), n.value, model._stream()))
            # a tape of another size is refused
            assert model._lib.odevio_ode_rnn_bwd_taped(*common, tape.data_ptr(), n.value - 4, model._stream()) == _lib.ERR_BAD_ARG
        else:
            _lib.check(model._lib.odevio_ode_rnn_bwd(*common, model._stream()))
        model.check()
        out.update({n_: g_ for n_, g_ in zip(names, grads)})
        return out

    a, b = run("taped"), run("plain")
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # the reverse sweep of an interval as one launch of the integrator's adjoint twin (default) against one launch per product
    monkeypatch.setenv("ODEVIO_ADJOINT_LAUNCHES", "1")
    d = run("plain")
    monkeypatch.delenv("ODEVIO_ADJOINT_LAUNCHES")
    for k in a:
        assert oc.rel_err(d[k], a[k]) < 2e-5, (k, oc.rel_err(d[k], a[k]))
    assert any(not torch.equal(a[k], d[k]) for k in a)
    monkeypatch.setenv("ODEVIO_TAPE_IN_ORDER", "1")
    c = run("plain")
    monkeypatch.delenv("ODEVIO_TAPE_IN_ORDER")
    for k in a:
        assert oc.rel_err(c[k], a[k]) < 2e-5, (k, oc.rel_err(c[k], a[k]))
    assert any(not torch.equal(a[k], c[k]) for k in a)     # (the two tapes really are different routes)


@pytest.mark.parametrize("widths,rows", [((200, 200), (32, 20)), ((512, 256), (7, 100))])
def test_soft_fusion_backward_over_many_rows(widths, rows, monkeypatch):
    """FusionModule 'soft' backward with hundreds of rows: the weight gradient is the wide-tile TN product (64 x 64 tiles, rows split over
    workgroup layers, bias sums from the same launch) incl. its ragged edges at the reference recipe's 200 + 200 = 400 columns - against
    autograd through the oracle in float64, and against the narrow-tile form of the same product."""
    from odevio_amd import _lib
    v, i = widths
    B, P = rows
    opt = default_opt(img_h=64, img_w=128, model_type="rnn", fuse_method="soft", v_f_len=v, i_f_len=i)
    model, sd = make_model(opt, seed=75)
    g = torch.Generator().manual_seed(9)
    fv, fi = torch.randn(B, P, v, generator=g), torch.randn(B, P, i, generator=g)
    gout = torch.randn(B, P, v + i, generator=g)
    names = train.fuse_param_names(opt)
    leaves = {n: sd[n].clone().double().requires_grad_(True) for n in names}
    fv64, fi64 = fv.double().requires_grad_(True), fi.double().requires_grad_(True)
    oc.fuse({**sd, **leaves}, fv64, fi64, "soft", dtype=torch.float64).backward(gout.double())
    params = dict(model.named_parameters())

    def run():
        for n in names:
            params[n].grad = None
        fvd, fid = fv.cuda().requires_grad_(True), fi.cuda().requires_grad_(True)
        fused = train._FuseFunction.apply(model, names, fvd, fid, *[params[n] for n in names])
        fused.backward(gout.cuda())
        model.check()
        return {"fv": fvd.grad, "fi": fid.grad, **{n: params[n].grad.clone() for n in names}}

    model._ensure_plan()
    got = run()
    ref = {"fv": fv64.grad, "fi": fi64.grad, **{n: leaves[n].grad for n in names}}
    bad = {k: f"{oc.rel_err(got[k], ref[k]):.2e}" for k in ref if not oc.rel_err(got[k], ref[k]) < GTOL}
    assert not bad, bad
    monkeypatch.setenv("ODEVIO_TN_NARROW", "1")
    narrow = run()
    monkeypatch.delenv("ODEVIO_TN_NARROW")
    for k in got:
        assert oc.rel_err(got[k], narrow[k]) < 2e-5, (k, oc.rel_err(got[k], narrow[k]))
    assert any(not torch.equal(got[n], narrow[n]) for n in names)     # (really two kernels)


def test_euler_backward_replays_every_dt0_step():
    """euler under torchode's controller (PoseODERNN.py:125-137) has no error estimate: every dt0 = 1e-4 step is accepted until the
    interval's end (the last one clipped) - a thousand steps per 0.1 s.  The backward replays them like any logged step sequence; short
    intervals (20 - 45 steps each, different per row) keep the test quick.  Also: a step log that overflows its first size is retried."""
    opt = default_opt(img_h=64, img_w=128, ode_solver="euler")
    model, sd = make_model(opt, seed=72)
    B, P, L, F = 2, 3, 2, 768
    g = torch.Generator().manual_seed(7)
    fv, fi = torch.randn(B, P, 512, generator=g), torch.randn(B, P, 256, generator=g)
    ts = torch.tensor([[0.0, 0.0021, 0.0052, 0.0097], [1.0, 1.0034, 1.0061, 1.0083]])
    hc = torch.randn(L, B, F, generator=g) * 0.3
    gts = torch.randn(B, P, 6, generator=g) * torch.tensor([0.01, 0.02, 0.01, 0.05, 0.05, 1.0])
    names = train.pose_param_names(opt)
    ref = _oracle_grads(sd, fv, fi, ts, hc, gts, opt, names, dtype=torch.float32)
    fv_d, fi_d, hc_d = fv.cuda().requires_grad_(True), fi.cuda().requires_grad_(True), hc.cuda().requires_grad_(True)
    poses, _ = train.pose_net(model, fv_d, fi_d, ts.cuda(), hc_d)
    train.pose_loss(poses, gts.cuda()).backward()
    model.check()
    assert oc.rel_err(poses, ref["poses"]) < 1e-4
    errs = {"fv": oc.rel_err(fv_d.grad, ref["fv"]), "fi": oc.rel_err(fi_d.grad, ref["fi"]), "hc": oc.rel_err(hc_d.grad, ref["hc"])}
    params = dict(model.named_parameters())
    for n in names:
        errs[n] = oc.rel_err(params[n].grad, ref[n])
    bad = {k: f"{v:.2e}" for k, v in errs.items() if not v < GTOL}
    assert not bad, f"gradients off by more than {GTOL}: {bad}"
    # an interval of 0.0097 s = 97 accepted steps > the adaptive solvers' first log size (64): heun with a tight tolerance overflows it
    opt2 = default_opt(img_h=64, img_w=128, ode_solver="heun")
    opt2.ode_rtol, opt2.ode_atol = 2e-6, 1e-8
    m2, _ = make_model(opt2, seed=72)
    fv2 = fv.cuda().requires_grad_(True)
    ts2 = torch.tensor([[0.0, 0.3, 0.7, 1.0], [0.0, 0.2, 0.5, 0.9]])
    p2, _, stats = m2.pose_net(fv.cuda(), fi.cuda(), ts2.cuda(), None, return_stats=True)
    assert int(stats[:, 1].max()) > 3 * 64, stats                       # more accepted steps per interval than the first log holds
    p3, _ = train.pose_net(m2, fv2, fi.cuda(), ts2.cuda())
    p3.sum().backward()
    m2.check()
    assert torch.isfinite(fv2.grad).all() and float(fv2.grad.abs().max()) > 0


def test_pose_loss_matches_the_reference_formula():
    g = torch.Generator().manual_seed(1)
    p, q = torch.randn(5, 10, 6, generator=g), torch.randn(5, 10, 6, generator=g)
    pd = p.cuda().requires_grad_(True)
    loss = train.pose_loss(pd, q.cuda())
    loss.backward()
    p64 = p.double().requires_grad_(True)
    ref = 100 * torch.nn.functional.mse_loss(p64[:, :, :3], q[:, :, :3].double()) + torch.nn.functional.mse_loss(p64[:, :, 3:], q[:, :, 3:].double())
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * float(ref)
    assert oc.rel_err(pd.grad, p64.grad) < 1e-6


# ---------------------------------------------------------------------------------------------------------------------
# the optimizer step of the reference's training loop (scripts/train_model.py:76-86, utils/utils.py:115-130)
# ---------------------------------------------------------------------------------------------------------------------
def _torch_reference_training(sd, opt, names, batches, lr, weight_decay, clip, eps):
    """The reference's loop on the oracle: torch autograd (fp32, like the reference), clip_grad_norm_, torch.optim.Adam over
    Pose_net's parameters."""
    leaves = {k: v.clone().float() for k, v in sd.items() if v.is_floating_point()}
    params = [leaves[n].requires_grad_(True) for n in names]
    optim = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.999), eps=eps, weight_decay=weight_decay)
    losses, norms = [], []
    for fv, fi, ts, gts in batches:
        optim.zero_grad()
        poses, _ = oc.pose_ode_rnn(leaves, fv, fi, ts, None, opt, with_ode=opt.model_type == "ode-rnn", detach_controller=True)
        loss = 100 * torch.nn.functional.mse_loss(poses[:, :, :3], gts[:, :, :3]) + torch.nn.functional.mse_loss(poses[:, :, 3:], gts[:, :, 3:])
        loss.backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_(params, max_norm=clip)))
        optim.step()
        losses.append(float(loss.detach()))
    return {n: p.detach() for n, p in zip(names, params)}, losses, norms


@pytest.mark.parametrize("cfg,clip", [
    (dict(ode_solver="rk4"), 5.0),                                  # the reference's defaults: clipping rarely active
    (dict(ode_solver="rk4", fuse_method="soft"), 0.05),            # clipping active in every step
    (dict(model_type="rnn", ode_rnn_type="gru"), 5.0),
])
def test_pose_net_trainer_follows_torch_adam_on_the_oracle(cfg, clip):
    opt = default_opt(img_h=64, img_w=128, freeze_encoder=True, **cfg)
    model, sd = make_model(opt, seed=81)
    B, P = 3, 4
    g = torch.Generator().manual_seed(9)
    scale = torch.tensor([0.01, 0.02, 0.01, 0.05, 0.05, 1.0])
    batches = [(torch.randn(B, P, 512, generator=g), torch.randn(B, P, 256, generator=g), synth.timestamps(B, P + 1, drop=0.2, seed=20 + k),
                torch.randn(B, P, 6, generator=g) * scale) for k in range(4)]
    lr, wd, eps = 1e-4, 5e-5, 1e-8
    trainer = train.PoseNetTrainer(model, lr=lr, weight_decay=wd, gradient_clip=clip, eps=eps)
    ref_params, ref_losses, ref_norms = _torch_reference_training(sd, opt, trainer.names, batches, lr, wd, clip, eps)
    before = {n: p.detach().clone() for n, p in zip(trainer.names, trainer.params)}
    losses, norms = [], []
    for fv, fi, ts, gts in batches:
        loss, poses, _ = trainer.step(fv.cuda(), fi.cuda(), ts.cuda(), gts.cuda())
        losses.append(float(loss.detach()))
        norms.append(float(trainer.grad_norm))
    model.check()
    for a, b in zip(losses, ref_losses):           # the loss of step k sees the parameters of steps < k: the updates took effect
        assert abs(a - b) <= 2e-4 * abs(b), (losses, ref_losses)
    for a, b in zip(norms, ref_norms):
        assert abs(a - b) <= 2e-3 * b, (norms, ref_norms)
    if clip < 1.0:
        assert min(ref_norms) > clip               # the case is meant to clip
    # Adam's update is lr * m_hat / (sqrt(v_hat) + eps): +-lr-sized whatever the gradient's size, so an element whose gradient
    # is ~0 can differ by a fraction of lr between two fp32 implementations; everything else must agree closely
    moved = 0.0
    for n, p in zip(trainer.names, trainer.params):
        d = (p.detach().cpu() - ref_params[n]).abs()
        moved = max(moved, float((p.detach().cpu() - before[n].cpu()).abs().max()))
        assert float(d.max()) <= 1.0 * lr * len(batches), f"{n}: max diff {float(d.max()):.2e}"
        assert float((d > 0.02 * lr).float().mean()) < 2e-3, f"{n}: {float((d > 0.02 * lr).float().mean()):.2e} of the elements differ by more than 2 % of lr"
    assert moved > 0.5 * lr                        # parameters did move
    # and the forward kernels see the new parameters: a plain forward equals the oracle on the reference-trained weights
    fv, fi, ts, _ = batches[0]
    new_sd = dict(sd)
    new_sd.update(ref_params)
    poses_ref, _ = oc.pose_ode_rnn(new_sd, fv, fi, ts, None, opt, with_ode=opt.model_type == "ode-rnn")
    poses_now, _ = model.pose_net(fv.cuda(), fi.cuda(), ts.cuda())
    assert oc.rel_err(poses_now, poses_ref) < 5e-4
    # the in-place refresh (device re-layout kernels) leaves the plan exactly as a fresh plan built from the same parameters
    from odevio_amd import DeepVIO
    fresh = DeepVIO(opt, seed=0, state_dict={k: v.detach().cpu().clone() for k, v in model.state_dict().items()}).cuda()
    poses_fresh, hT_fresh = fresh.pose_net(fv.cuda(), fi.cuda(), ts.cuda())
    assert torch.equal(poses_now, poses_fresh)
    # ... for the backward's copies (plain and transposed weights) too: identical gradients
    for mdl in (model, fresh):
        for p in mdl.parameters():
            p.grad = None
    outs = []
    for mdl in (model, fresh):
        fvd = fv.cuda().requires_grad_(True)
        po, _ = train.pose_net(mdl, fvd, fi.cuda(), ts.cuda())
        train.pose_loss(po, batches[0][3].cuda()).backward()
        outs.append([fvd.grad] + [dict(mdl.named_parameters())[n].grad for n in trainer.names])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("kind", ["Adam", "SGD"])
def test_optimizer_step_over_many_tensors_follows_torch(kind):
    """odevio_optimizer_step (one call, one launch per 64 tensors) against torch.optim over 70 tensors of assorted sizes, two learning
    rates, weight decay and a clip factor, three steps: parameters and optimizer state."""
    import ctypes
    from odevio_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    sizes = [int(x) for x in torch.randint(1, 5000, (70,), generator=g)]
    ref = [torch.randn(n, generator=g).cuda().requires_grad_(True) for n in sizes]
    dev = [r.detach().clone() for r in ref]
    s1 = [torch.zeros_like(p) for p in dev]
    s2 = [torch.zeros_like(p) for p in dev]
    lrs_py = [1e-2 if i % 3 else 3e-3 for i in range(len(sizes))]
    groups = [{"params": [p], "lr": lr} for p, lr in zip(ref, lrs_py)]
    wd = 1e-3
    opt = torch.optim.Adam(groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd) if kind == "Adam" else torch.optim.SGD(groups, lr=1e-2, momentum=0.9, weight_decay=wd)
    names = [f"t{i}" for i in range(len(sizes))]
    coef = torch.tensor([0.0, 0.5], device="cuda")                 # {norm (unused here), clip factor}
    lrs = (ctypes.c_float * len(sizes))(*lrs_py)
    for step in range(1, 4):
        grads = [torch.randn(n, generator=g).cuda() for n in sizes]
        for r, gr in zip(ref, grads):
            r.grad = gr * 0.5                                      # torch sees the clipped gradient
        opt.step()
        arr = lambda ts: train._tensor_array(names, ts)
        rc = lib.odevio_optimizer_step(0 if kind == "Adam" else 1, arr(dev), arr(grads), arr(s1), arr(s2) if kind == "Adam" else None, lrs, len(sizes),
                                       0.9, 0.999, 1e-8, wd, step, coef.data_ptr(), None)
        _lib.check(rc)
        torch.cuda.synchronize()
        for i, (d, r) in enumerate(zip(dev, ref)):
            assert torch.allclose(d, r.detach(), rtol=2e-6, atol=2e-7), (kind, step, i, float((d - r.detach()).abs().max()))
    st = opt.state[ref[5]]
    assert torch.allclose(s1[5], st["exp_avg"] if kind == "Adam" else st["momentum_buffer"], rtol=2e-6, atol=1e-7)
    # sizes that disagree are refused before anything is written
    bad = train._tensor_array(names, [torch.empty(3, device="cuda")] * len(sizes))
    assert lib.odevio_optimizer_step(0, train._tensor_array(names, dev), bad, train._tensor_array(names, s1), train._tensor_array(names, s2), lrs, len(sizes),
                                     0.9, 0.999, 1e-8, 0.0, 1, None, None) == _lib.ERR_BAD_ARG


def test_training_reduces_the_loss_on_a_fixed_batch():
    opt = default_opt(img_h=64, img_w=128, ode_solver="rk4", freeze_encoder=True)
    model, _ = make_model(opt, seed=82)
    g = torch.Generator().manual_seed(10)
    fv, fi = torch.randn(4, 5, 512, generator=g).cuda(), torch.randn(4, 5, 256, generator=g).cuda()
    ts = synth.timestamps(4, 6, seed=30).cuda()
    gts = (torch.randn(4, 5, 6, generator=g) * torch.tensor([0.01, 0.02, 0.01, 0.05, 0.05, 1.0])).cuda()
    trainer = train.PoseNetTrainer(model, lr=1e-3)
    losses = [float(trainer.step(fv, fi, ts, gts)[0]) for _ in range(12)]
    model.check()
    assert losses[-1] < 0.7 * losses[0], losses


def test_inertial_encoder_backward_matches_autograd_through_the_oracle():
    opt = default_opt(img_h=64, img_w=128)
    model, sd = make_model(opt, seed=83)
    B, T = 3, 41
    g = torch.Generator().manual_seed(11)
    imu = torch.randn(B, T, 6, generator=g)
    gfi = torch.randn(B, (T - 1) // 10, 256, generator=g)
    names = train.imu_param_names()
    leaves = {k: v.clone().double().requires_grad_(k in names) for k, v in sd.items() if v.is_floating_point()}   # (running statistics are buffers)
    fi_ref = oc.inertial_encoder(leaves, imu, dtype=torch.float64)
    (fi_ref * gfi.double()).sum().backward()
    fi = train.imu_encoder(model, imu.cuda())
    (fi * gfi.cuda()).sum().backward()
    model.check()
    assert oc.rel_err(fi, fi_ref.detach()) < 1e-4
    params = dict(model.named_parameters())
    errs = {n: oc.rel_err(params[n].grad, leaves[n].grad) for n in names}
    bad = {k: f"{v:.2e}" for k, v in errs.items() if not v < GTOL}
    assert not bad, f"gradients off by more than {GTOL}: {bad}"


def test_trainer_with_the_inertial_encoder_in_the_graph():
    """The reference's recipe freezes Image_net only: Inertial_net's gradients exist and count in clip_grad_norm_(model.parameters()),
    while the optimizer holds Pose_net alone (utils/utils.py:116-119)."""
    opt = default_opt(img_h=64, img_w=128, ode_solver="rk4", freeze_encoder=True)
    model, sd = make_model(opt, seed=84)
    B, P = 3, 4
    g = torch.Generator().manual_seed(12)
    scale = torch.tensor([0.01, 0.02, 0.01, 0.05, 0.05, 1.0])
    batches = [(torch.randn(B, P, 512, generator=g), torch.randn(B, 10 * P + 1, 6, generator=g), synth.timestamps(B, P + 1, seed=40 + k),
                torch.randn(B, P, 6, generator=g) * scale) for k in range(3)]
    lr, wd, eps, clip = 1e-4, 5e-5, 1e-8, 0.05
    trainer = train.PoseNetTrainer(model, lr=lr, weight_decay=wd, gradient_clip=clip, eps=eps)
    inames = train.imu_param_names()
    leaves = {k: v.clone().float() for k, v in sd.items() if v.is_floating_point()}
    params = [leaves[n].requires_grad_(True) for n in trainer.names]
    iparams = [leaves[n].requires_grad_(True) for n in inames]
    optim = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.999), eps=eps, weight_decay=wd)
    ref_norms, ref_losses = [], []
    for fv, imu, ts, gts in batches:
        for p in params + iparams:
            p.grad = None
        fi = oc.inertial_encoder(leaves, imu)
        poses, _ = oc.pose_ode_rnn(leaves, fv, fi, ts, None, opt)
        loss = 100 * torch.nn.functional.mse_loss(poses[:, :, :3], gts[:, :, :3]) + torch.nn.functional.mse_loss(poses[:, :, 3:], gts[:, :, 3:])
        loss.backward()
        ref_norms.append(float(torch.nn.utils.clip_grad_norm_(params + iparams, max_norm=clip)))
        optim.step()
        ref_losses.append(float(loss.detach()))
    pose_only = []
    for fv, imu, ts, gts in batches:
        loss, _, _ = trainer.step(fv.cuda(), None, ts.cuda(), gts.cuda(), imu=imu.cuda())
        assert abs(float(loss) - ref_losses[len(pose_only)]) <= 2e-4 * abs(ref_losses[len(pose_only)])
        pose_only.append(float(trainer.grad_norm))
    model.check()
    for a, b in zip(pose_only, ref_norms):
        assert abs(a - b) <= 2e-3 * b, (pose_only, ref_norms)
    for n, p in zip(trainer.names, trainer.params):
        d = (p.detach().cpu() - leaves[n].detach()).abs()
        assert float(d.max()) <= 1.0 * lr * len(batches), n
        assert float((d > 0.02 * lr).float().mean()) < 2e-3, n
    now = dict(model.named_parameters())
    for n in inames:                                   # Inertial_net is not the optimizer's: unchanged
        assert torch.equal(now[n].detach().cpu(), sd[n])


def test_train_epoch_runs_the_reference_loop_end_to_end():
    """train.train_epoch = scripts/train_model.py:48-95 on the device path: frames and IMU samples in, encoders, pose net,
    loss, backward, clip, Adam per batch - under model.train() like the reference's epoch loop (:219): batch-statistics BatchNorm and
    Dropout in both encoders.  A fixed pair of batches, several epochs: the loss goes down, nothing fails."""
    opt = default_opt(img_h=64, img_w=128, ode_solver="rk4", freeze_encoder=True)
    model, _ = make_model(opt, seed=85)
    batches = []
    for k in range(2):
        img, imu, ts = synth.batch(2, 4, 64, 128, seed=50 + k)
        g = torch.Generator().manual_seed(60 + k)
        gts = torch.randn(2, 3, 6, generator=g) * torch.tensor([0.01, 0.02, 0.01, 0.05, 0.05, 1.0])
        batches.append((img, imu, gts, ts, "synthetic"))
    trainer = train.PoseNetTrainer(model, lr=1e-3)
    lines = []
    means = [train.train_epoch(model, trainer, batches, log=lines.append, log_every=1) for _ in range(6)]
    model.check()
    assert model.training and all(m == m for m in means) and means[-1] < 0.9 * means[0], means
    assert len(lines) == 12 and "pose loss" in lines[0]
    assert int(model.Image_net.conv1[1].num_batches_tracked) == 1 + 12          # the constructor's dummy forward + 12 train-mode batches
    model.eval()
    img, imu, ts = batches[0][0], batches[0][1], batches[0][3]
    poses, _ = model(img.cuda(), imu.cuda(), ts.cuda())                          # eval afterwards folds the MOVED running statistics
    ref, _ = oc.deepvio_forward({k: v.detach().cpu() for k, v in model.state_dict().items()}, img, imu, ts, None, opt)
    assert oc.rel_err(poses, ref) < 1e-4


def test_hard_fusion_straight_through_backward_with_the_same_noise():
    """FusionModule "hard" in training: F.gumbel_softmax(..., hard=True) returns y_hard - y_soft.detach() + y_soft, so the forward
    value is the one-hot mask and the gradient is y_soft's.  The reference draws its Gumbel noise from torch's generator; here the
    oracle is handed the noise the device drew (odevio_debug_gumbel for the same seed and draw index) - with equal noise the mask
    and every gradient must agree."""
    import ctypes
    from odevio_amd import _lib
    opt = default_opt(img_h=64, img_w=128, fuse_method="hard", model_type="rnn", rnn_num_layers=1)
    model, sd = make_model(opt, seed=86)
    B, P, F = 3, 4, 768
    g = torch.Generator().manual_seed(13)
    fv, fi = torch.randn(B, P, 512, generator=g), torch.randn(B, P, 256, generator=g)
    ts = synth.timestamps(B, P + 1, seed=5)
    gts = torch.randn(B, P, 6, generator=g) * torch.tensor([0.01, 0.02, 0.01, 0.05, 0.05, 1.0])
    model.set_seed(21)
    seed, call = model.rng_state()
    noise = torch.empty(B * P * F, 2, device="cuda")
    _lib.check(model._lib.odevio_debug_gumbel(seed, call, B * P * F, noise.data_ptr(), model._stream()))
    noise = noise.cpu().double().reshape(B, P, F, 2)
    # reference arithmetic (FusionModule.py:24-29 + torch.nn.functional.gumbel_softmax, tau = 1) with that noise, float64
    names = train.fuse_param_names(opt) + train.pose_param_names(opt)
    leaves = {k: v.clone().double().requires_grad_(k in names) for k, v in sd.items() if v.is_floating_point()}
    fv64, fi64 = fv.double().requires_grad_(True), fi.double().requires_grad_(True)
    fused = oc.fuse(leaves, fv64, fi64, "hard", dtype=torch.float64, noise=noise)
    cat_opt = default_opt(img_h=64, img_w=128, fuse_method="cat", model_type="rnn", rnn_num_layers=1)
    poses, _ = oc.pose_ode_rnn(leaves, fused[..., :512], fused[..., 512:], ts, None, cat_opt, dtype=torch.float64, with_ode=False)
    loss = 100 * torch.nn.functional.mse_loss(poses[:, :, :3], gts[:, :, :3].double()) + torch.nn.functional.mse_loss(poses[:, :, 3:], gts[:, :, 3:].double())
    loss.backward()
    # device path, same seed -> same draw
    model.set_seed(21)
    fv_d, fi_d = fv.cuda().requires_grad_(True), fi.cuda().requires_grad_(True)
    poses_d, _ = train.pose_net(model, fv_d, fi_d, ts.cuda())
    train.pose_loss(poses_d, gts.cuda()).backward()
    model.check()
    assert oc.rel_err(poses_d, poses.detach()) < 1e-4                      # same mask, same forward
    errs = {"fv": oc.rel_err(fv_d.grad, fv64.grad), "fi": oc.rel_err(fi_d.grad, fi64.grad)}
    params = dict(model.named_parameters())
    for n in names:
        errs[n] = oc.rel_err(params[n].grad, leaves[n].grad)
    bad = {k: f"{v:.2e}" for k, v in errs.items() if not v < GTOL}
    assert not bad, f"gradients off by more than {GTOL}: {bad}"
