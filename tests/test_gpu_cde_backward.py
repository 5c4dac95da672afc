"""GPU tests of the Neural-CDE path's backward (odevio_cde_bwd).

Reference semantics: PoseCDE.forward in training mode (src/models/PoseCDE.py:76-103: relative timestamps, no window history) with
``adjoint = False`` (:98-101) = plain autograd through torchcde's cdeint -> torchdiffeq's odeint.  Both libraries are absent offline,
so - like the forward - the check is torch.autograd through the ORACLE's restatement of their published algorithm (parity with the
real libraries UNPINNED), in fp32 for the adaptive solver so that both sides take the same step decisions (the forward tests pin
the step sequences), with the step sizes constants of the differentiation on both sides.
"""
import pytest
import torch

from odevio_amd import default_opt, synth, train
from oracle import odevio_oracle as oc

from test_gpu_parity import make_model

pytestmark = pytest.mark.gpu
GTOL = 2e-3


def _oracle(sd, opt, names, fv, fi, ts, prev, w_p, w_z, dtype):
    leaves = {k: (v.clone().to(dtype).requires_grad_(k in names) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    fv_r, fi_r = fv.clone().to(dtype).requires_grad_(True), fi.clone().to(dtype).requires_grad_(True)
    prev_r = None if prev is None else prev.clone().to(dtype).requires_grad_(True)
    tr = {}
    poses, z0, _ = oc.pose_cde(leaves, fv_r, fi_r, ts, prev_r, None, opt, dtype=dtype, training=True, trace=tr)
    ((poses * w_p.to(dtype)).sum() + (z0 * w_z.to(dtype)).sum()).backward()
    out = {"poses": poses.detach(), "z0": z0.detach(), "fv": fv_r.grad, "fi": fi_r.grad, "trace": tr}
    if prev_r is not None:
        out["prev"] = prev_r.grad
    for n in names:
        out[n] = leaves[n].grad
    return out


@pytest.mark.parametrize("cfg,with_prev", [
    (dict(cde_solver="dopri5"), False),
    (dict(cde_solver="dopri5", fuse_method="soft", cde_activation_fn="softplus", cde_fn_num_layers=2), True),
    (dict(cde_solver="rk4"), False),
    (dict(cde_solver="euler", cde_activation_fn="relu"), True),
    (dict(cde_solver="dopri5", cde_hidden_dim=400, v_f_len=200, i_f_len=200, cde_activation_fn="softplus", cde_fn_num_layers=2), False),  # the reference's CDE recipe shapes
])
def test_cde_backward_matches_autograd_through_the_oracle(cfg, with_prev):
    """Every gradient of the training forward: encoder features (through fusion, the control path's dX/dt and z0), a carried state,
    CDEFunc's hidden layers and its [H (H+1), H] last layer, the initial layer, the regressor, the fusion Linear.  The windows
    (frame drop 0.6: relative time up to ~2 s) cross knots of the control path: even pieces (only the time channel moves), odd
    pieces (every feature channel), steps clipped at a knot with f re-evaluated behind it, rejected steps, dense output."""
    kw = dict(cde_hidden_dim=128, v_f_len=96, i_f_len=32)
    kw.update(cfg)
    opt = default_opt(img_h=64, img_w=128, model_type="cde", **kw)
    model, sd = make_model(opt, seed=91)
    model.train()
    H, v, i = opt.cde_hidden_dim, opt.v_f_len, opt.i_f_len
    B, P = 3, 6
    g = torch.Generator().manual_seed(8)
    fv, fi = torch.randn(B, P, v, generator=g) * 0.5, torch.randn(B, P, i, generator=g) * 0.5
    ts = synth.timestamps(B, P + 1, drop=0.6, seed=1 if with_prev else 10, absolute=True)      # row 0: 0.6 1.0 1.3 1.6 1.8 2.5 | 0.8 1.0 1.7 2.1 2.5 2.8
    prev = torch.tanh(torch.randn(B, H, generator=g)) if with_prev else None
    w_p, w_z = torch.randn(B, P, 6, generator=g), torch.randn(B, H, generator=g) * 0.1
    names = train.fuse_param_names(opt) + train.cde_param_names(opt)
    if with_prev:
        names = [n for n in names if not n.startswith("Pose_net.initial.")]       # a carried state bypasses the initial layer
    dtype = torch.float32 if opt.cde_solver == "dopri5" else torch.float64
    ref = _oracle(sd, opt, names, fv, fi, ts, prev, w_p, w_z, dtype)
    assert float((ts[0, -1] - ts[0, 0])) > 1.0, "the window must cross a knot of the control path"

    fv_d, fi_d = fv.cuda().requires_grad_(True), fi.cuda().requires_grad_(True)
    prev_d = None if prev is None else prev.cuda().requires_grad_(True)
    poses, z0 = train.pose_cde(model, fv_d, fi_d, ts.cuda(), prev_d)
    ((poses * w_p.cuda()).sum() + (z0 * w_z.cuda()).sum()).backward()
    model.check()
    assert oc.rel_err(poses, ref["poses"]) < 1e-4 and oc.rel_err(z0, ref["z0"]) < 1e-4
    errs = {"fv": oc.rel_err(fv_d.grad, ref["fv"]), "fi": oc.rel_err(fi_d.grad, ref["fi"])}
    if prev is not None:
        errs["prev"] = oc.rel_err(prev_d.grad, ref["prev"])
    params = dict(model.named_parameters())
    for n in names:
        assert params[n].grad is not None, n
        errs[n] = oc.rel_err(params[n].grad, ref[n])
    bad = {k: f"{e:.2e}" for k, e in errs.items() if not e < GTOL}
    assert not bad, f"gradients off by more than {GTOL}: {bad}"
    if opt.cde_solver == "dopri5":
        steps = ref["trace"]["steps"]
        assert any(not acc for _, _, acc in steps) or len(steps) > 4          # the replay had something to skip / several steps to walk


def test_deepvio_forward_cde_in_train_mode_carries_a_graph():
    """model.train(); model(img, imu, ts) for model_type cde: train-mode encoders, then the Neural-CDE pose net with its backward -
    loss.backward() reaches CDEFunc, the initial layer, the regressor and Inertial_net."""
    opt = default_opt(img_h=64, img_w=128, model_type="cde", cde_hidden_dim=128, v_f_len=96, i_f_len=32, freeze_encoder=True)
    model, _ = make_model(opt, seed=92)
    for q in model.Image_net.parameters():
        q.requires_grad = False
    model.train()
    img, imu, ts = synth.batch(2, 4, 64, 128, seed=31)
    poses, z0 = model(img.cuda(), imu.cuda(), ts.cuda())
    assert poses.requires_grad and poses.shape == (2, 3, 6)
    poses.square().sum().backward()
    model.check()
    got = {n for n, q in model.named_parameters() if q.grad is not None and float(q.grad.abs().max()) > 0}
    for n in ("Pose_net.cde_func.net.6.weight", "Pose_net.cde_func.net.0.bias", "Pose_net.initial.0.weight", "Pose_net.regressor.2.bias",
              "Inertial_net.proj.weight", "Inertial_net.encoder_conv.0.weight"):
        assert n in got, (n, sorted(got))
    assert all(q.grad is None for q in model.Image_net.parameters())
