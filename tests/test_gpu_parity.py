"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the golden fixtures.

Every test here needs a real MI355X (``-m gpu``).  The bar is the north star's: fp32 results within
1e-4 relative (max|gpu - oracle| / max|oracle| per tensor); the tolerance is stated at each assert.
"""
import os

import numpy as np
import pytest
import torch

from odevio_amd import default_opt, synth, weights
from oracle import odevio_oracle as oc

pytestmark = pytest.mark.gpu
TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def make_model(opt, seed, randomize=True):
    from odevio_amd import DeepVIO
    sd = weights.make_state_dict(opt, seed=seed, randomize_stats=randomize)
    model = DeepVIO(opt, seed=seed, state_dict=sd)
    return model.cuda(), sd


def assert_close(got, ref, tol=TOL, what=""):
    """max|got - ref| / max|ref| < tol.  6-DoF poses ([..., 6] = Euler angles in rad || translation in m, reference
    src/data/utils.py:44-69) are additionally held to the same bar on the rotation and the translation columns
    separately, each on its own scale: the two halves differ in unit and magnitude."""
    err = oc.rel_err(got, ref)
    assert err < tol, f"{what}: rel err {err:.3e} >= {tol}"
    if ref.dim() >= 2 and ref.shape[-1] == 6:
        er, et = oc.rel_err(got[..., :3], ref[..., :3]), oc.rel_err(got[..., 3:], ref[..., 3:])
        assert er < tol and et < tol, f"{what}: rotation rel err {er:.3e}, translation rel err {et:.3e} (tol {tol})"
        err = max(err, er, et)
    return err


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


# ------------------------------------------------------------------------------------------------
# kernel level
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hw", [(64, 128), (96, 160), (80, 208), (66, 130), (256, 512)])
def test_conv_blocks_one_by_one(dev, hw):
    """Every conv block alone, fed with the ORACLE's input for that block (errors cannot compound).  Sizes: multiples
    of the tile grids, ragged ones (partial 8x32 conv1 tiles, partial 256-pixel GEMM tiles, W % 4 != 0: the ingest's scalar
    path, odd spatial sizes in every layer; with B*(S-1) = 4 pairs the last image / last pixel rows end inside a tile) and
    the KITTI size.  The same shapes run against the bounds-audit build."""
    H, W = hw
    opt = default_opt(img_h=H, img_w=W)
    model, sd = make_model(opt, seed=21)
    B, S = (1, 2) if H == 256 else (2, 3)
    img = synth.images(B, S, H, W, seed=5)
    _, inter = oc.image_encoder(sd, img, return_intermediate=True)
    names = [n for n, _, _ in oc.IMAGE_CONVS]
    x = img.cuda()
    for i, name in enumerate(names):
        out = model.conv_block(i, x, B, S)
        assert_close(out, nhwc(inter[name]), what=name)
        x = nhwc(inter[name]).cuda()
    model.check()


def test_image_encoder_ragged_sizes(dev):
    # image sizes that are not multiples of the tile sizes exercise every bounds path
    opt = default_opt(img_h=72, img_w=136)
    model, sd = make_model(opt, seed=22)
    img = synth.images(1, 4, 72, 136, seed=6)
    assert_close(model.image_encoder(img.cuda()), oc.image_encoder(sd, img), what="fv")


@pytest.mark.parametrize("force", [
    "1:n192,2:n192,3:n192,4:n192,5:n192,6:n192,7:n192,8:n192",          # 192 x 128 tiles everywhere
    "1:n,2:w192,3:w192,4:w192,5:w192,6:w192,7:w192,8:w192",              # 192 x 256 tiles wherever Cout % 256 == 0
    "2:w,3:w,4:w:s2,5:w192:s3,6:n:s4,7:w:s2,8:w192:s2",                   # split-K on every shape
])
def test_every_tile_shape_of_the_fp16x2_kernel(dev, monkeypatch, force):
    # the planner picks among four tile shapes per layer; forced here so that small inputs reach all of them (ragged
    # image: partial tiles, rows past M, the 192-pixel tile's uneven epilogue passes)
    opt = default_opt(img_h=72, img_w=136)
    model, sd = make_model(opt, seed=23)
    img = synth.images(2, 4, 72, 136, seed=7)
    ref = oc.image_encoder(sd, img)
    monkeypatch.setenv("ODEVIO_CONV_FORCE", ",".join(f"{i}:n" for i in range(1, 9)))   # 256 x 128 tiles, no split-K
    base = model.image_encoder(img.cuda())
    monkeypatch.setenv("ODEVIO_CONV_FORCE", force)
    got = model.image_encoder(img.cuda())
    model.check()
    assert_close(got, ref, what="fv, forced tile shapes")
    if ":s" not in force:   # same K order per output, whatever the tile: bit-identical
        assert torch.equal(got, base)


@pytest.mark.parametrize("B,S,H,W", [(1, 2, 64, 128), (3, 3, 80, 208), (5, 2, 128, 256), (2, 6, 96, 160), (7, 3, 64, 192), (1, 9, 128, 128)])
def test_tile_planner_over_many_shapes(dev, B, S, H, W):
    """The per-layer tile planner decides from (pixels, channels, K-tiles, CUs): batch and image sizes that give every layer a
    different pixel count - partial last tiles, layers smaller than one round, XCD map on and off, split-K or not - all have to
    land on the oracle (the planner's own choices, nothing forced)."""
    opt = default_opt(img_h=H, img_w=W)
    model, sd = make_model(opt, seed=90 + B)
    img = synth.images(B, S, H, W, seed=B * 10 + S)
    fv = model.image_encoder(img.cuda())
    model.check()
    assert_close(fv, oc.image_encoder(sd, img), what=f"fv at B={B} S={S} {H}x{W}")


def test_two_phase_tile_plan_matches_single_phase(dev, monkeypatch):
    # a layer split between two tile shapes (R full rounds of the chip in one, the remaining pixels in another), at a size
    # where the split falls inside the batch: conv2 of 8 frame pairs = 65,536 pixels; one round of 192 x 128 tiles covers
    # n_cu x 192 of them (49,152 on 256 CUs), 256 x 128 tiles take the rest.  (The bench batch's own two-phase plans run in
    # test_baseline_config1_full_batch.)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    if n_cu * 192 >= 65536:
        pytest.skip("device too large for the split to fall inside this batch")
    opt = default_opt()
    model, sd = make_model(opt, seed=24)
    img = synth.images(2, 5, 256, 512, seed=8)
    monkeypatch.setenv("ODEVIO_CONV_FORCE", "1:n")
    single = model.image_encoder(img.cuda())
    monkeypatch.setenv("ODEVIO_CONV_FORCE", "1:n192:1:n")
    two = model.image_encoder(img.cuda())
    model.check()
    assert torch.equal(two, single)
    assert_close(two, oc.image_encoder(sd, img), what="fv, two-phase plan")


def test_64_bit_dma_addressing_form_of_the_conv_kernel(dev, tmp_path):
    """The production form of conv_f16x2_kernel addresses its DMA sources as scalar base + 32-bit offset; buffers of 4 GB and
    more (very large batches) fall back to the 64-bit form (two pointers, range compares), which no default-size test reaches.
    ODEVIO_CONV_OFF64 selects it; the variable is read once per process, hence the child process."""
    import subprocess
    import sys
    out = str(tmp_path / "fv.pt")
    code = (
        "import sys, torch\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "from odevio_amd import default_opt, synth\n"
        "from test_gpu_parity import make_model\n"
        "opt = default_opt(img_h=72, img_w=136)\n"
        "model, sd = make_model(opt, seed=22)\n"
        "img = synth.images(2, 4, 72, 136, seed=6)\n"
        "fv = model.image_encoder(img.cuda()); model.check()\n"
        f"torch.save(fv.cpu(), {out!r})\n")
    env = dict(os.environ, ODEVIO_CONV_OFF64="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    opt = default_opt(img_h=72, img_w=136)
    model, sd = make_model(opt, seed=22)
    img = synth.images(2, 4, 72, 136, seed=6)
    fv64 = torch.load(out, weights_only=True)
    assert_close(fv64, oc.image_encoder(sd, img), what="fv, 64-bit addressing form")
    assert torch.equal(fv64, model.image_encoder(img.cuda()).cpu())        # same arithmetic, another address computation


def test_image_encoder_golden_full_size(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "image_encoder_full.npz"))
    opt = default_opt()
    model, _ = make_model(opt, seed=int(g["wseed"]), randomize=bool(g["randomize_stats"]))
    img = synth.images(int(g["B"]), int(g["S"]), 256, 512, seed=int(g["iseed"]))
    assert_close(model.image_encoder(img.cuda()), torch.from_numpy(g["fv"]), what="fv vs reference")


def test_image_encoder_golden_small(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "image_encoder_small.npz"))
    opt = default_opt(img_h=int(g["H"]), img_w=int(g["W"]))
    model, _ = make_model(opt, seed=int(g["wseed"]), randomize=True)
    img = synth.images(int(g["B"]), int(g["S"]), int(g["H"]), int(g["W"]), seed=int(g["iseed"]))
    assert_close(model.image_encoder(img.cuda()), torch.from_numpy(g["fv"]), what="fv vs reference")
    c1 = model.conv_block(0, img.cuda(), int(g["B"]), int(g["S"]))  # NHWC
    assert_close(c1.permute(0, 3, 1, 2)[:, ::8, ::8, ::8], torch.from_numpy(g["conv1_sample"]), what="conv1 vs reference")


def test_hard_fusion_is_a_gumbel_argmax_drawn_on_the_device(dev):
    """FusionModule "hard" (FusionModule.py:24-29): mask = gumbel_softmax(logits.view(..., F, 2), tau=1, hard=True)[..., 0], i.e. feature j
    is kept with probability sigmoid(l0 - l1).  Stochastic under torch's generator in the reference, so there is no bit-level
    parity: checked here are the arithmetic around the mask, the distribution, and reproducibility per seed."""
    opt = default_opt(img_h=64, img_w=128, fuse_method="hard")
    model, _ = make_model(opt, seed=31)
    F = 768
    d = torch.linspace(-3.0, 3.0, F)
    with torch.no_grad():
        model.Pose_net.fuse.net[0].weight.zero_()               # logits = bias: (d_j, 0) per feature, independent of the input
        b = torch.zeros(2 * F)
        b[0::2] = d
        model.Pose_net.fuse.net[0].bias.copy_(b)
    g = torch.Generator().manual_seed(3)
    fv, fi = torch.randn(64, 16, 512, generator=g).cuda(), torch.randn(64, 16, 256, generator=g).cuda()
    cat = torch.cat((fv, fi), -1)
    model.set_seed(7)
    a = model.fuse(fv, fi)
    keep = a != 0
    assert torch.equal(a, torch.where(keep, cat, torch.zeros_like(cat)))       # fused = cat * one-hot mask, nothing else
    freq = keep.float().mean(dim=(0, 1)).cpu()
    p = torch.sigmoid(d)
    sigma = (p * (1 - p) / (64 * 16)).sqrt()
    assert float(((freq - p).abs() / sigma).max()) < 5.5, float(((freq - p).abs() / sigma).max())
    assert abs(float((freq - p).mean())) < 3e-3                                  # no bias over the 768 features
    model.set_seed(7)
    assert torch.equal(model.fuse(fv, fi), a)                                    # same seed, same mask
    b2 = model.fuse(fv, fi)
    assert not torch.equal(b2, a)                                                # the next call draws fresh noise
    model.set_seed(8)
    assert not torch.equal(model.fuse(fv, fi), a)
    # and the whole forward runs on the device path with it
    img, imu, ts = synth.batch(2, 3, 64, 128, seed=4)
    poses, h = model(img.cuda(), imu.cuda(), ts.cuda())
    model.check()
    assert torch.isfinite(poses).all() and torch.isfinite(h).all()


def test_device_generator_is_philox4x32_10(dev):
    """The noise fuse_method "hard" draws on the device is Philox 4x32-10 keyed by the plan's seed: bit-equal uniforms, i.e. Gumbel
    values equal to the oracle's restatement (pinned by the published known-answer vectors, tests/test_oracle_philox.py) up to the
    rounding of two float logarithms."""
    import ctypes
    from odevio_amd import _lib
    from oracle import philox as ph
    lib = _lib.load()
    for seed, call, n in ((0, 0, 1001), (21, 0, 4096), (0x1234567890ABCDEF, 5, 777), (7, (1 << 33) + 2, 64)):
        out = torch.empty(n, 2, device="cuda")
        _lib.check(lib.odevio_debug_gumbel(seed, call, n, out.data_ptr(), None))
        torch.cuda.synchronize()
        ref = torch.from_numpy(ph.gumbel_pairs(seed, call, n))
        assert float((out.cpu() - ref).abs().max()) < 2e-5, (seed, call)      # |g| <= 17: a few ulp of the nested logarithms


def test_inertial_encoder(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "inertial_encoder.npz"))
    opt = default_opt(img_h=64, img_w=128)
    model, sd = make_model(opt, seed=int(g["wseed"]))
    real = torch.from_numpy(g["imu04"])
    for T in (11, 21, 51, 101, 105):
        fi = model.imu_encoder(real[:, :T].cuda())
        assert fi.shape == (1, (T - 1) // 10, 256)
        assert_close(fi, torch.from_numpy(g[f"fi_T{T}"]), what=f"fi T={T} vs reference")
    syn = synth.imu(3, 11, seed=5)
    assert_close(model.imu_encoder(syn.cuda()), torch.from_numpy(g["fi_syn"]), what="fi synthetic vs reference")
    big = synth.imu(16, 11, seed=9)
    assert_close(model.imu_encoder(big.cuda()), oc.inertial_encoder(sd, big), what="fi B=16 vs oracle")


@pytest.mark.parametrize("method", ["cat", "soft"])
def test_fusion(dev, golden_dir, method):
    g = np.load(os.path.join(golden_dir, "fusion.npz"))
    opt = default_opt(img_h=64, img_w=128, fuse_method=method)
    model, _ = make_model(opt, seed=int(g["wseed"]))
    out = model.fuse(torch.from_numpy(g["fv"]).cuda(), torch.from_numpy(g["fi"]).cuda())
    assert_close(out, torch.from_numpy(g[method]), what=f"fuse {method} vs reference")


@pytest.mark.parametrize("act", ["tanh", "relu", "leaky_relu", "softplus"])
@pytest.mark.parametrize("n,H", [(3, 512), (2, 1024)])
def test_odefunc(dev, golden_dir, act, n, H):
    g = np.load(os.path.join(golden_dir, "odefunc.npz"))
    opt = default_opt(img_h=64, img_w=128, ode_activation_fn=act, ode_fn_num_layers=n, ode_hidden_dim=H)
    model, sd = make_model(opt, seed=int(g["wseed"]))
    y = torch.from_numpy(g["y"])
    assert_close(model.ode_func(y.cuda()), torch.from_numpy(g[f"f_{act}_{n}_{H}"]), what="f(y) vs reference")
    yy = torch.randn(37, 768, generator=torch.Generator().manual_seed(3)) * 0.8  # ragged row count, several groups
    assert_close(model.ode_func(yy.cuda()), oc.ode_func(sd, yy, n, act), what="f(y) 37 rows vs oracle")
    model.check()


# ------------------------------------------------------------------------------------------------
# integrator
# ------------------------------------------------------------------------------------------------
def _rows_problem(rows, seed):
    g = torch.Generator().manual_seed(seed)
    y0 = torch.randn(rows, 768, generator=g) * 0.5
    t0 = torch.rand(rows, generator=g) * 3.0
    gaps = torch.tensor([0.1, 0.2, 0.3, 0.5])[torch.randint(0, 4, (rows,), generator=g)]
    return y0, t0, t0 + gaps


@pytest.mark.parametrize("solver,substeps", [("rk4", 1), ("rk4", 3), ("rk4_classic", 2), ("dopri5", 1), ("tsit5", 1),
                                             ("heun", 1)])
@pytest.mark.parametrize("rows", [32, 5])
def test_ode_steps(dev, solver, substeps, rows):
    opt = default_opt(img_h=64, img_w=128, ode_solver=solver, ode_substeps=substeps)
    model, sd = make_model(opt, seed=31)
    y0, t0, t1 = _rows_problem(rows, seed=rows)
    got, stats = model.ode_steps(y0.cuda(), t0.cuda(), t1.cuda(), return_stats=True)
    model.check()
    tr = {}
    f = lambda y: oc.ode_func(sd, y, opt.ode_fn_num_layers, opt.ode_activation_fn)
    ref = oc.evolve_state(f, y0, t0, t1, solver, substeps, trace=tr)
    assert_close(got, ref, what=f"{solver} state")
    # identical step sequences: attempted and accepted step counts per row
    assert torch.equal(stats[:, 0].cpu().long(), tr["n_steps"]), (stats[:, 0].cpu(), tr["n_steps"])
    assert torch.equal(stats[:, 1].cpu().long(), tr["n_accepted"])


def test_safe_and_local_handoff_agree(dev, monkeypatch):
    """The placement-independent write-through hand-off and the verified same-XCD one give the same bits."""
    opt = default_opt(img_h=64, img_w=128, ode_solver="dopri5")
    model, _ = make_model(opt, seed=34)
    y0, t0, t1 = _rows_problem(32, seed=7)
    monkeypatch.setenv("ODEVIO_SAFE_HANDOFF", "1")
    a, sa = model.ode_steps(y0.cuda(), t0.cuda(), t1.cuda(), return_stats=True)
    model.check()
    monkeypatch.setenv("ODEVIO_SAFE_HANDOFF", "0")
    b, sb = model.ode_steps(y0.cuda(), t0.cuda(), t1.cuda(), return_stats=True)
    model.check()
    assert torch.equal(a, b) and torch.equal(sa, sb)


def test_euler_reference_semantics(dev):
    # torchode's controller never changes dt without an error estimate: 1e-4 steps to the end
    opt = default_opt(img_h=64, img_w=128, ode_solver="euler")
    model, sd = make_model(opt, seed=32)
    y0 = torch.randn(4, 768, generator=torch.Generator().manual_seed(1)) * 0.5
    t0 = torch.zeros(4)
    t1 = torch.tensor([0.0105, 0.02, 0.0033, 0.01])
    got, stats = model.ode_steps(y0.cuda(), t0.cuda(), t1.cuda(), return_stats=True)
    model.check()
    f = lambda y: oc.ode_func(sd, y, 3, "tanh")
    tr = {}
    ref = oc.evolve_state(f, y0, t0, t1, "euler", trace=tr)
    assert_close(got, ref, what="euler state")
    assert torch.equal(stats[:, 0].cpu().long(), tr["n_steps"])


def test_ode_steps_linearity_of_time_shift(dev):
    # autonomous field: shifting both ends of every interval by a constant changes nothing but rounding of t
    opt = default_opt(img_h=64, img_w=128, ode_solver="dopri5")
    model, _ = make_model(opt, seed=33)
    y0, t0, t1 = _rows_problem(16, seed=2)
    a = model.ode_steps(y0.cuda(), (t0 * 0).cuda(), (t1 - t0).cuda())
    b = model.ode_steps(y0.cuda(), (t0 * 0 + 64.0).cuda(), (t1 - t0 + 64.0).cuda())
    assert_close(a, b, tol=1e-3, what="time-shift invariance")


@pytest.mark.parametrize("rnn_type", ["rnn", "gru"])
@pytest.mark.parametrize("L", [2, 3])
@pytest.mark.parametrize("method", ["cat", "soft"])
def test_pose_rnn_golden(dev, golden_dir, rnn_type, L, method):
    """fuse -> RNN stack -> regressor (no ODE) against the REAL reference PoseRNN, incl. carried hc."""
    g = np.load(os.path.join(golden_dir, "pose_rnn.npz"))
    opt = default_opt(img_h=64, img_w=128, model_type="rnn", ode_rnn_type=rnn_type, rnn_num_layers=L, fuse_method=method)
    model, _ = make_model(opt, seed=int(g["wseed"]))
    fv, fi, ts = (torch.from_numpy(g[k]).cuda() for k in ("fv", "fi", "ts"))
    key = f"{rnn_type}_{L}_{method}"
    p1, h1 = model.pose_net(fv, fi, ts, None)
    assert_close(p1, torch.from_numpy(g[key + "_pose1"]), what="pose1 vs reference")
    assert_close(h1, torch.from_numpy(g[key + "_h1"]), what="h1 vs reference")
    p2, h2 = model.pose_net(fv.flip(0), fi.flip(0), ts, h1)
    assert_close(p2, torch.from_numpy(g[key + "_pose2"]), what="pose2 vs reference")
    assert_close(h2, torch.from_numpy(g[key + "_h2"]), what="h2 vs reference")
    model.check()


@pytest.mark.parametrize("cfg", [
    dict(ode_solver="rk4"),
    dict(ode_solver="dopri5"),
    dict(ode_solver="dopri5", ode_rnn_type="gru", fuse_method="soft"),
    dict(ode_solver="tsit5", rnn_num_layers=3, ode_activation_fn="softplus", ode_fn_num_layers=2, ode_hidden_dim=1024,
         fuse_method="soft"),  # the reference's own training recipe (scripts/run_training.sh:6-28) with tsit5
    dict(ode_solver="dopri5", rnn_num_layers=3, ode_activation_fn="softplus", ode_fn_num_layers=2, ode_hidden_dim=1024,
         fuse_method="soft"),  # ... and with the recipe's real solver flag (dopri5, run_training.sh:17)
    dict(ode_solver="heun", rnn_num_layers=1),
])
@pytest.mark.parametrize("B,drop", [(16, 0.0), (3, 0.5)])
def test_pose_ode_rnn(dev, cfg, B, drop):
    """The whole ODE-RNN loop on encoder features: regular and irregular (50 % drop) timestamps, ragged batch."""
    opt = default_opt(img_h=64, img_w=128, **cfg)
    model, sd = make_model(opt, seed=41)
    g = torch.Generator().manual_seed(B)
    fv = torch.randn(B, 10, 512, generator=g)
    fi = torch.randn(B, 10, 256, generator=g)
    ts = synth.timestamps(B, 11, drop=drop, seed=B, absolute=True)
    poses, h, stats = model.pose_net(fv.cuda(), fi.cuda(), ts.cuda(), None, return_stats=True)
    model.check()
    tr = {}
    ref_p, ref_h = oc.pose_ode_rnn(sd, fv, fi, ts, None, opt, trace=tr)
    assert_close(poses, ref_p, what="poses")
    assert_close(h, ref_h, what="h_T")
    # Step sequences agree except where the embedded error estimate sits at the fp32 rounding floor
    # (e.g. the all-zero initial state): there 0.9*ratio^(-1/5) moves between ~8 and the clamp of 10 with
    # the last bits of tanh, so an interval may take a step more or fewer (the states still agree to 1e-4,
    # asserted above).  Bound the drift: on average under one step per row, never more than 15 % of a row's steps.
    want = sum(t["n_steps"] for t in tr["intervals"])
    diff = (stats[:, 0].cpu().long() - want).abs()
    assert float(diff.float().mean()) <= 1.0 and int(diff.max()) <= 0.15 * int(want.max()), (stats[:, 0].cpu(), want)
    # streaming: carry h_T into the next window with absolute timestamps (reference KITTI_eval.py:141)
    # Both sides start the second window from the SAME carried state (the device's h_T), so the bar stays 1e-4; chaining each
    # side's own state instead compounds the first window's difference (checked too, at twice the bar).
    ts2 = ts + 1.0
    p2, h2 = model.pose_net(fv.flip(1).cuda(), fi.flip(1).cuda(), ts2.cuda(), h)
    r2, rh2 = oc.pose_ode_rnn(sd, fv.flip(1), fi.flip(1), ts2, h.cpu(), opt)
    assert_close(p2, r2, what="poses (carried hc, same state in)")
    assert_close(h2, rh2, what="h_T (carried hc, same state in)")
    r3, rh3 = oc.pose_ode_rnn(sd, fv.flip(1), fi.flip(1), ts2, ref_h, opt)
    assert_close(p2, r3, tol=2e-4, what="poses (each side chains its own state)")
    assert_close(h2, rh3, tol=2e-4, what="h_T (each side chains its own state)")


# ------------------------------------------------------------------------------------------------
# Neural-CDE path
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [dict(), dict(cde_solver="rk4"), dict(cde_solver="euler"),
                                 dict(cde_activation_fn="softplus", cde_fn_num_layers=2, fuse_method="soft")])
@pytest.mark.parametrize("training", [True, False])
def test_pose_cde(dev, cfg, training):
    """PoseCDE.forward on encoder features: training mode (relative time) and eval mode (raw time, carried history)."""
    opt = default_opt(img_h=64, img_w=128, model_type="cde", cde_hidden_dim=128, v_f_len=96, i_f_len=32, **cfg)
    model, sd = make_model(opt, seed=61)
    model.train(training)
    B = 5
    g = torch.Generator().manual_seed(3)
    fv, fi = torch.randn(B, 10, 96, generator=g), torch.randn(B, 10, 32, generator=g)
    ts = synth.timestamps(B, 11, drop=0.3, seed=4) + (0.0 if training else 3.0)
    poses, z0, (n_steps, n_acc) = model.pose_cde(fv.cuda(), fi.cuda(), ts.cuda(), None, return_stats=True)
    tr = {}
    ref_p, ref_z0, hist = oc.pose_cde(sd, fv, fi, ts, None, None, opt, training=training, trace=tr)
    assert_close(z0, ref_z0, what="z0")
    assert_close(poses, ref_p, what="poses")
    if opt.cde_solver == "dopri5":
        assert (n_steps, n_acc) == (tr["n_steps"], tr["n_accepted"])
    if not training:  # second window: carried state + history (reference KITTI_eval.py:141 with PoseCDE)
        p2, z2 = model.pose_cde(fv.flip(0).cuda(), fi.flip(0).cuda(), (ts + 1.0).cuda(), z0)
        r2, rz2, _ = oc.pose_cde(sd, fv.flip(0), fi.flip(0), ts + 1.0, ref_z0, hist, opt)
        assert_close(p2, r2, tol=2e-4, what="poses (window 2)")
        assert model.Pose_net.history.shape == (B, 20, 129)
    model.eval()


def test_deepvio_forward_cde(dev):
    opt = default_opt(img_h=64, img_w=128, model_type="cde", cde_hidden_dim=128, v_f_len=96, i_f_len=32)
    model, sd = make_model(opt, seed=62)
    img, imu, ts = synth.batch(2, 5, 64, 128, seed=12)
    poses, z0 = model(img.cuda(), imu.cuda(), (ts + 2.0).cuda())
    fv, fi = oc.image_encoder(sd, img), oc.inertial_encoder(sd, imu)
    ref_p, ref_z0, _ = oc.pose_cde(sd, fv, fi, ts + 2.0, None, None, opt)
    assert poses.shape == (2, 4, 6) and z0.shape == (2, 128)
    assert_close(poses, ref_p, what="poses")
    assert_close(z0, ref_z0, what="z0")


# ------------------------------------------------------------------------------------------------
# whole path
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("solver", ["rk4", "dopri5"])
def test_deepvio_forward_small(dev, solver):
    opt = default_opt(img_h=64, img_w=128, ode_solver=solver)
    model, sd = make_model(opt, seed=51)
    img, imu, ts = synth.batch(3, 5, 64, 128, drop=0.3, seed=8)
    poses, h = model(img.cuda(), imu.cuda(), ts.cuda())
    model.check()
    ref_p, ref_h = oc.deepvio_forward(sd, img, imu, ts, None, opt)
    assert poses.shape == (3, 4, 6) and h.shape == (2, 3, 768)
    assert_close(poses, ref_p, what="poses")
    assert_close(h, ref_h, what="h_T")


def test_deepvio_forward_full_size(dev):
    """KITTI-sized frames (256x512), B=2, S=4, RK4: the north star's parity bar on the real shapes."""
    opt = default_opt(ode_solver="rk4")
    model, sd = make_model(opt, seed=52, randomize=False)
    img, imu, ts = synth.batch(2, 4, 256, 512, seed=9)
    poses, h = model(img.cuda(), imu.cuda(), ts.cuda())
    model.check()
    ref_p, ref_h = oc.deepvio_forward(sd, img, imu, ts, None, opt)
    e = assert_close(poses, ref_p, what="poses")
    assert_close(h, ref_h, what="h_T")
    # both fp32 paths should sit equally close to the fp64 truth
    tru_p, _ = oc.deepvio_forward(sd, img, imu, ts, None, opt, dtype=torch.float64)
    assert oc.rel_err(poses, tru_p) < TOL and oc.rel_err(ref_p, tru_p) < TOL, e


def test_forward_is_deterministic_and_batch_independent(dev):
    # size-independent properties: same inputs -> same bits; a sequence's poses do not depend on its batch mates
    opt = default_opt(img_h=64, img_w=128, ode_solver="dopri5")
    model, _ = make_model(opt, seed=53)
    img, imu, ts = synth.batch(5, 4, 64, 128, drop=0.5, seed=10)
    p1, h1 = model(img.cuda(), imu.cuda(), ts.cuda())
    p2, h2 = model(img.cuda(), imu.cuda(), ts.cuda())
    assert torch.equal(p1, p2) and torch.equal(h1, h2)
    p3, _ = model(img[1:3].cuda(), imu[1:3].cuda(), ts[1:3].cuda())
    assert_close(p3, p1[1:3], tol=1e-5, what="batch independence")


def test_errors_are_loud(dev):
    from odevio_amd import DeepVIO
    with pytest.raises(ValueError):
        DeepVIO(default_opt(ode_solver="rk45"))
    with pytest.raises(ValueError):
        DeepVIO(default_opt(ode_rnn_type="lstm"))
    with pytest.raises(ValueError):
        DeepVIO(default_opt(ode_activation_fn="gelu"))
    with pytest.raises(NotImplementedError):
        DeepVIO(default_opt(model_type="ltc"))
    model = DeepVIO(default_opt(img_h=64, img_w=128)).cuda()
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 2, 3, 64, 128), torch.zeros(1, 11, 6), torch.zeros(1, 2))  # CPU tensors: no CPU path
    with pytest.raises(ValueError):
        model(torch.zeros(1, 3, 3, 64, 128).cuda(), torch.zeros(1, 11, 6).cuda(), torch.zeros(1, 3).cuda())  # imu too short


# ------------------------------------------------------------------------------------------------
# streaming evaluator: device pose accumulation, hidden-state carry over windows, KITTI metrics
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 7, 255, 256, 257, 1000, 4540])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_path_accu_matches_oracle(dev, n, dtype):
    """Scan on the device vs the reference's sequential product; float64, so 1e-9 of the trajectory extent."""
    from odevio_amd import metrics
    from oracle import kitti_metrics as om
    poses = torch.from_numpy(synth.trajectory(n + 1, seed=n)).to(dtype)
    got = metrics.path_accu(poses.cuda()).cpu().numpy()
    ref = np.stack(om.path_accu(poses.numpy()))
    assert got.shape == (n + 1, 4, 4)
    scale = max(1.0, np.abs(ref[:, :3, 3]).max())
    # float32 input: device and numpy sinf/cosf may differ in the last bit of a factor, which the product carries along
    tol = 1e-9 if dtype == torch.float64 else 2e-6
    assert np.abs(got - ref).max() / scale < tol
    np.testing.assert_array_equal(got[:, 3], np.tile([0.0, 0.0, 0.0, 1.0], (n + 1, 1)))


def test_path_accu_carry_and_many_drives(dev):
    from odevio_amd import metrics
    from oracle import kitti_metrics as om
    lens = [300, 1, 777, 64]
    poses = [synth.trajectory(n + 1, seed=30 + i) for i, n in enumerate(lens)]
    cat = torch.from_numpy(np.concatenate(poses)).cuda()
    off = np.concatenate(([0], np.cumsum(lens)))
    got = metrics.path_accu(cat, offsets=off).cpu().numpy()
    pos = 0
    for p, n in zip(poses, lens):
        ref = np.stack(om.path_accu(p))
        assert np.abs(got[pos:pos + n + 1] - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
        pos += n + 1
    # streaming: windows chained through the carry equal one pass over the drive
    p = torch.from_numpy(poses[2]).cuda()
    whole = metrics.path_accu(p)
    carry, parts = None, []
    for a in range(0, 777, 10):
        m = metrics.path_accu(p[a:a + 10], carry=carry)
        carry = m[-1:]
        parts.append(m[1:] if a else m)
    assert (torch.cat(parts) - whole).abs().max().item() < 1e-9 * whole.abs().max().item()
    with pytest.raises(ValueError):
        metrics.path_accu(p.cpu())
    with pytest.raises(ValueError):
        metrics.path_accu(p, offsets=[0, 5])


def test_kitti_eval_matches_oracle(dev):
    from odevio_amd import metrics
    from oracle import kitti_metrics as om
    gt = synth.trajectory(1500, seed=3)
    est = synth.trajectory(1500, seed=3, noise=0.04).astype(np.float32)
    est_m, gt_m, t_rel, r_rel, t_rmse, r_rmse, usage, speed = metrics.kitti_eval(est, None, gt)
    ref = om.kitti_eval(est, gt)
    assert t_rel == pytest.approx(ref["t_rel"], rel=1e-4) and r_rel == pytest.approx(ref["r_rel"], rel=1e-4)
    assert t_rmse == pytest.approx(ref["t_rmse"], rel=1e-12) and r_rmse == pytest.approx(ref["r_rmse"], rel=1e-12)
    assert t_rel > 1.0 and usage == 0
    np.testing.assert_allclose(speed, ref["speed"], rtol=1e-9)
    assert np.abs(gt_m - np.stack(ref["gt_mats"])).max() < 1e-8


@pytest.mark.parametrize("model_type,solver", [("ode-rnn", "rk4"), ("ode-rnn", "dopri5"), ("rnn", "rk4")])
def test_stream_windows_carry_hidden_state(dev, model_type, solver):
    """Whole drives streamed window by window (stride S-1, short last window, absolute timestamps, hc carried):
    three drives of different lengths in lock-step == the oracle walking each drive alone like the reference's
    test_one_path (KITTI_eval.py:124-160)."""
    from odevio_amd import stream
    H, W, S = 64, 128, 5
    opt = default_opt(img_h=H, img_w=W, ode_solver=solver, model_type=model_type, seq_len=S)
    model, sd = make_model(opt, seed=61)
    drives = []
    for i, n in enumerate([14, 9, 6]):          # 4 windows (last of 2 frames), 2 windows (5+5), 2 windows (5+2)
        fr, im, ts, gt = synth.drive(n, H, W, seed=70 + i, t0=100.0 * (i + 1))
        drives.append(stream.Drive(fr, im, ts, gt, name=f"d{i}"))
    tester = stream.StreamTester(S)
    est = tester.test_paths(model, drives)
    model.check()
    for d, e in zip(drives, est):
        n = d.frames.shape[0]
        assert e.shape == (n - 1, 6)
        hc, ref = None, []
        for a, b in stream.partition(n, S):
            lo, hi = stream.imu_rows(a, b)
            p, hc = oc.deepvio_forward(sd, d.frames[a:b][None], d.imus[lo:hi][None], d.timestamps[a:b][None], hc, opt)
            ref.append(p[0])
        # (each side chains its OWN carried state over up to four windows: with the adaptive solver the windows' differences
        # compound, hence twice the bar there; test_pose_ode_rnn holds a window that starts from the same state to 1e-4)
        assert_close(torch.from_numpy(e), torch.cat(ref), tol=2e-4 if solver == "dopri5" else TOL, what=f"{d.name} streamed poses")
    # a drive streamed alone gives the same poses as in the lock-step batch
    alone = tester.test_paths(model, drives[1:2])[0]
    assert_close(torch.from_numpy(alone), torch.from_numpy(est[1]), tol=1e-5, what="lock-step vs alone")


def test_forward_from_uint8_frames(dev):
    """the loader's uint8 HWC frames go in directly; byte / 255 - 0.5 (ToTensor() - 0.5, reference
    src/data/utils.py:359, KITTI_eval.py:100-103) is fused into the encoder's ingest pass.  Must equal the float path fed
    with the same normalisation done by torch, and the oracle."""
    opt = default_opt(img_h=64, img_w=128, ode_solver="rk4")
    model, sd = make_model(opt, seed=81)
    B, S = 3, 4
    g = torch.Generator().manual_seed(5)
    frames = torch.randint(0, 256, (B, S, 64, 128, 3), generator=g, dtype=torch.uint8)
    _, imu, ts = synth.batch(B, S, 64, 128, drop=0.3, seed=12)
    as_float = frames.permute(0, 1, 4, 2, 3).float().div(255) - 0.5          # what the reference's loader produces
    p_u8, h_u8 = model(frames.cuda(), imu.cuda(), ts.cuda())
    p_f, h_f = model(as_float.cuda(), imu.cuda(), ts.cuda())
    model.check()
    assert torch.equal(p_u8, p_f) and torch.equal(h_u8, h_f)                 # same fp32 values enter the same kernels
    ref_p, ref_h = oc.deepvio_forward(sd, as_float, imu, ts, None, opt)
    assert_close(p_u8, ref_p, what="poses from uint8 frames")
    assert_close(h_u8, ref_h, what="h_T from uint8 frames")
    with pytest.raises(ValueError):
        model(frames.permute(0, 1, 4, 2, 3).contiguous().cuda(), imu.cuda(), ts.cuda())   # CHW uint8: the entry takes HWC frames


def test_resize_matches_pillow(dev, golden_dir):
    """the loader's TF.resize on the device, bit-identical to Pillow (integer arithmetic): the golden
    vectors were written by Pillow itself (oracle/gen_golden_resize.py); the oracle restatement covers batched frames."""
    import hashlib
    from odevio_amd.deepvio import resize_frames
    from oracle import pil_resize as pr
    from oracle.gen_golden_resize import CASES, frame
    g = np.load(os.path.join(golden_dir, "resize.npz"))
    for i, (hi, wi, ho, wo, seed, kind) in enumerate(CASES):
        got = resize_frames(torch.from_numpy(frame(hi, wi, seed, kind)).cuda(), ho, wo).cpu().numpy()
        np.testing.assert_array_equal(got[:8], g[f"top{i}"])
        np.testing.assert_array_equal(got[-8:], g[f"bot{i}"])
        assert hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest() == str(g[f"sha{i}"]), (i, kind)
    batch = torch.randint(0, 256, (2, 3, 94, 310, 3), generator=torch.Generator().manual_seed(3), dtype=torch.uint8)
    got = resize_frames(batch.cuda(), 64, 128).cpu().numpy()
    np.testing.assert_array_equal(got, pr.resize_bilinear_u8(batch.numpy(), 64, 128))
    with pytest.raises(ValueError):
        resize_frames(batch.float().cuda(), 64, 128)


def test_forward_from_camera_sized_frames(dev):
    """Camera-sized uint8 frames go straight in: resize (PIL-exact) + normalisation + encoder all on the device; must
    equal the forward fed with the frames the reference's loader would have produced (oracle resize, ToTensor() - 0.5)."""
    from oracle import pil_resize as pr
    opt = default_opt(img_h=64, img_w=128, ode_solver="rk4")
    model, sd = make_model(opt, seed=82)
    B, S = 2, 3
    raw = torch.randint(0, 256, (B, S, 94, 310, 3), generator=torch.Generator().manual_seed(6), dtype=torch.uint8)
    _, imu, ts = synth.batch(B, S, 64, 128, seed=13)
    p_raw, h_raw = model(raw.cuda(), imu.cuda(), ts.cuda())
    model.check()
    small = torch.from_numpy(pr.resize_bilinear_u8(raw.numpy(), 64, 128))
    as_float = small.permute(0, 1, 4, 2, 3).float().div(255) - 0.5
    p_f, h_f = model(as_float.cuda(), imu.cuda(), ts.cuda())
    assert torch.equal(p_raw, p_f) and torch.equal(h_raw, h_f)
    ref_p, ref_h = oc.deepvio_forward(sd, as_float, imu, ts, None, opt)
    assert_close(p_raw, ref_p, what="poses from camera-sized frames")
    assert_close(h_raw, ref_h, what="h_T from camera-sized frames")


def test_baseline_config0_single_clip(dev):
    """BASELINE configs[0]'s shape: one clip, batch 1, seq-len 11, 256x512, RK4 fixed step, fp32."""
    opt = default_opt(ode_solver="rk4")
    model, sd = make_model(opt, seed=91, randomize=False)
    img, imu, ts = synth.batch(1, 11, 256, 512, seed=21)
    poses, h = model(img.cuda(), imu.cuda(), ts.cuda())
    model.check()
    ref_p, ref_h = oc.deepvio_forward(sd, img, imu, ts, None, opt)
    assert poses.shape == (1, 10, 6)
    assert_close(poses, ref_p, what="poses")
    assert_close(h, ref_h, what="h_T")


def test_pose_cde_wide_hidden(dev):
    """Towards BASELINE configs[4] (hidden 1024): the CDE vector field at hidden 512 (its last layer is a
    [512*513, 512] matrix, 0.54 GB), fixed-grid solver so that the CPU oracle stays within seconds."""
    opt = default_opt(img_h=64, img_w=128, model_type="cde", cde_hidden_dim=512, v_f_len=384, i_f_len=128,
                      cde_solver="euler", cde_fn_num_layers=1)
    model, sd = make_model(opt, seed=63)
    B = 2
    g = torch.Generator().manual_seed(7)
    fv, fi = torch.randn(B, 2, 384, generator=g) * 0.5, torch.randn(B, 2, 128, generator=g) * 0.5
    ts = synth.timestamps(B, 3, seed=5)
    poses, z0 = model.pose_cde(fv.cuda(), fi.cuda(), ts.cuda(), None)
    ref_p, ref_z0, _ = oc.pose_cde(sd, fv, fi, ts, None, None, opt, training=False)
    assert_close(z0, ref_z0, what="z0")
    assert_close(poses, ref_p, what="poses")


def test_f32_mfma_encoder_mode(dev, monkeypatch):
    """ODEVIO_CONV_MATH=f32 (read at plan creation) runs the encoder on the fp32-input MFMA kernels; same parity bar."""
    monkeypatch.setenv("ODEVIO_CONV_MATH", "f32")
    opt = default_opt(img_h=64, img_w=128, ode_solver="rk4")
    model, sd = make_model(opt, seed=92)
    img, imu, ts = synth.batch(2, 4, 64, 128, seed=14)
    poses, h = model(img.cuda(), imu.cuda(), ts.cuda())
    model.check()
    ref_p, ref_h = oc.deepvio_forward(sd, img, imu, ts, None, opt)
    assert_close(poses, ref_p, what="poses (f32 MFMA encoder)")
    assert_close(h, ref_h, what="h_T (f32 MFMA encoder)")
    monkeypatch.setenv("ODEVIO_CONV_MATH", "fp8")
    from odevio_amd import DeepVIO
    bad = DeepVIO(opt, seed=1).cuda()
    with pytest.raises(ValueError):
        bad(img.cuda(), imu.cuda(), ts.cuda())


def test_f16x2_range_guard_is_loud(dev):
    """An encoder activation beyond the fp16 range cannot be carried as two fp16 pieces: the epilogue raises the
    status word and check() fails instead of returning inf/nan poses silently.  The plan's per-layer activation exponents
    (a power of two per layer, from a variance estimate at plan creation) absorb up to 2^24 of mis-scaling first: a
    checkpoint whose conv1 outputs are ~1e5 - beyond fp16 as they stand - still runs, and accurately."""
    from odevio_amd._lib import OdevioError
    opt = default_opt(img_h=64, img_w=128, ode_solver="rk4")
    model, sd = make_model(opt, seed=93)
    img, imu, ts = synth.batch(1, 3, 64, 128, seed=15)
    big = {k: v.clone() for k, v in sd.items()}
    big["Image_net.conv1.0.weight"] *= 3e6          # conv1 outputs ~1e5, brought back by conv2's BatchNorm statistics
    big["Image_net.conv2.1.running_var"] *= 9e12
    big["Image_net.conv2.1.running_mean"] *= 3e6
    model.load_state_dict(big)
    poses, _ = model(img.cuda(), imu.cuda(), ts.cuda())
    model.check()
    ref_p, _ = oc.deepvio_forward(big, img, imu, ts, None, opt)
    assert_close(poses, ref_p, what="poses with conv1 activations ~1e5 (absorbed by the activation exponent)")
    huge = {k: v.clone() for k, v in sd.items()}
    huge["Image_net.conv1.0.weight"] *= 1e20        # beyond what a 2^24 exponent can absorb
    model.load_state_dict(huge)
    model(img.cuda(), imu.cuda(), ts.cuda())
    with pytest.raises(OdevioError, match="fp16x2 range"):
        model.check()
    model.load_state_dict(sd)                        # the plan is rebuilt; the flag was cleared by check()
    poses, _ = model(img.cuda(), imu.cuda(), ts.cuda())
    model.check()
    ref_p, _ = oc.deepvio_forward(sd, img, imu, ts, None, opt)
    assert_close(poses, ref_p, what="poses after recovery")


def test_bounds_audit_catches_a_planted_violation(dev, monkeypatch):
    """Only with the bounds-audit build (ODEVIO_LIB=.../libodevio_audit.so): the audit must fire when an extent is
    declared too short (ODEVIO_AUDIT_SELFTEST shaves 256 bytes off the input extent handed to the conv kernel) - and
    redirect the access instead of faulting."""
    from odevio_amd import _lib
    if "audit" not in os.path.basename(_lib.LIB_PATH):
        pytest.skip("production library: the kernels carry no bounds checks")
    opt = default_opt(img_h=64, img_w=128)
    model, sd = make_model(opt, seed=21)
    img = synth.images(2, 3, 64, 128, seed=5)
    _, inter = oc.image_encoder(sd, img, return_intermediate=True)
    model.conv_block(1, nhwc(inter["conv1"]).cuda(), 2, 3)
    model.check()                                                  # clean run: nothing to report
    monkeypatch.setenv("ODEVIO_AUDIT_SELFTEST", "1")
    model.conv_block(1, nhwc(inter["conv1"]).cuda(), 2, 3)
    with pytest.raises(_lib.OdevioError, match="outside its buffers"):
        model.check()
    monkeypatch.delenv("ODEVIO_AUDIT_SELFTEST")
    out = model.conv_block(1, nhwc(inter["conv1"]).cuda(), 2, 3)
    model.check()
    assert_close(out, nhwc(inter["conv2"]), what="conv2 after the self-test")


def test_failed_forward_surfaces_without_check(dev):
    """A failed forward must not hand garbage on silently when nobody calls check(): the status words travel to pinned
    host memory behind every forward, and the next entry point reports the failure of the one before it."""
    from odevio_amd._lib import OdevioError
    opt = default_opt(img_h=64, img_w=128, ode_solver="rk4")
    model, sd = make_model(opt, seed=93)
    big = {k: v.clone() for k, v in sd.items()}
    big["Image_net.conv1.0.weight"] *= 1e20
    model.load_state_dict(big)
    img, imu, ts = synth.batch(1, 3, 64, 128, seed=15)
    model(img.cuda(), imu.cuda(), ts.cuda())          # raises the range word on the device; returns normally
    torch.cuda.synchronize()                          # (the test's only reason to wait: the copy must have landed)
    with pytest.raises(OdevioError, match="fp16x2 range"):
        model(img.cuda(), imu.cuda(), ts.cuda())
    model.load_state_dict(sd)
    poses, _ = model(img.cuda(), imu.cuda(), ts.cuda())
    model.check()
    ref_p, _ = oc.deepvio_forward(sd, img, imu, ts, None, opt)
    assert_close(poses, ref_p, what="poses after recovery")


def test_baseline_config1_full_batch(dev):
    """BASELINE configs[1] at its real size (16 sequences x 11 frames of 256x512, RK4): the layer shapes, tile counts
    and split-K factors of the bench.  The oracle checks sequences 0 and 15 (sequences are independent end to end)."""
    opt = default_opt(ode_solver="rk4")
    model, sd = make_model(opt, seed=94, randomize=False)
    img, imu, ts = synth.batch(16, 11, 256, 512, drop=0.2, seed=31)
    poses, h = model(img.cuda(), imu.cuda(), ts.cuda())
    model.check()
    assert poses.shape == (16, 10, 6) and h.shape == (2, 16, 768)
    assert torch.isfinite(poses).all()
    for b in (0, 15):
        ref_p, ref_h = oc.deepvio_forward(sd, img[b:b + 1], imu[b:b + 1], ts[b:b + 1], None, opt)
        assert_close(poses[b:b + 1], ref_p, what=f"poses of sequence {b}")
        assert_close(h[:, b:b + 1], ref_h, what=f"h_T of sequence {b}")


def test_large_batch_crosses_the_4gb_and_chunking_boundaries(dev):
    """48 sequences x 11 frames of 256x512 on one GPU (the per-GPU load of BASELINE configs[3] is 16; three times that here): conv1's
    output is 4.03 GB, so conv2 takes the 64-bit DMA addressing form in production conditions (no environment variable), the tile
    planner sees three times the pixels, and the integrator walks the batch in two launches (32 sequences at 8 rows per group,
    then 16 at 4).  The oracle checks the first, the 33rd (first of the second launch) and the last sequence."""
    opt = default_opt(ode_solver="rk4")
    model, sd = make_model(opt, seed=95, randomize=False)
    B = 48
    img, imu, ts = synth.batch(B, 11, 256, 512, drop=0.2, seed=33)
    poses, h = model(img.cuda(), imu.cuda(), ts.cuda())
    model.check()
    assert poses.shape == (B, 10, 6) and h.shape == (2, B, 768)
    assert torch.isfinite(poses).all() and torch.isfinite(h).all()
    for b in (0, 32, B - 1):
        ref_p, ref_h = oc.deepvio_forward(sd, img[b:b + 1], imu[b:b + 1], ts[b:b + 1], None, opt)
        assert_close(poses[b:b + 1], ref_p, what=f"poses of sequence {b}")
        assert_close(h[:, b:b + 1], ref_h, what=f"h_T of sequence {b}")


def test_encoder_error_against_fp64_truth(dev, monkeypatch, capsys):
    """How far the image encoder (conv1..conv6 + visual head, 256x512) is from an fp64 evaluation of the same network,
    for both arithmetic modes.  The fp16x2 operand split must be as accurate as the fp32-input MFMA path (DESIGN.md 5.1;
    measured: oracle fp32 9.3e-7, fp32 MFMA 1.37e-6, fp16x2 1.24e-6)."""
    from odevio_amd import DeepVIO
    opt = default_opt()
    sd = weights.make_state_dict(opt, seed=5, randomize_stats=True)
    img = synth.images(2, 3, 256, 512, seed=1)
    truth = oc.image_encoder(sd, img, torch.float64)
    errs = {"oracle_fp32": oc.rel_err(oc.image_encoder(sd, img), truth)}
    for mode in ("f16x2", "f32"):
        monkeypatch.setenv("ODEVIO_CONV_MATH", mode)
        m = DeepVIO(opt, seed=5)
        m.load_state_dict(sd)
        m = m.cuda()
        fv = m.image_encoder(img.cuda())
        m.check()
        errs[mode] = oc.rel_err(fv, truth)
    with capsys.disabled():
        print("\nimage encoder vs fp64 truth (max err / max):", {k: f"{v:.3e}" for k, v in errs.items()})
    assert errs["f16x2"] < 5e-6 and errs["f32"] < 5e-6
    assert errs["f16x2"] < 2.0 * errs["f32"]


def test_reduced_precision_mode_reports_its_error(dev, monkeypatch, capsys):
    """ODEVIO_CONV_MATH=f16 (BASELINE configs[2]'s reduced-precision flavour: fp16 encoder operands, fp32 accumulation,
    fp32 integrator) is OUTSIDE the 1e-4 parity claim: it reports its error against the fp32 oracle and must stay within
    what an 11-bit significand allows."""
    monkeypatch.setenv("ODEVIO_CONV_MATH", "f16")
    opt = default_opt(img_h=64, img_w=128, ode_solver="dopri5")
    model, sd = make_model(opt, seed=95)
    img, imu, ts = synth.batch(3, 5, 64, 128, drop=0.5, seed=16)
    poses, h = model(img.cuda(), imu.cuda(), ts.cuda())
    model.check()
    ref_p, ref_h = oc.deepvio_forward(sd, img, imu, ts, None, opt)
    ep, eh = oc.rel_err(poses, ref_p), oc.rel_err(h, ref_h)
    with capsys.disabled():
        print(f"\nODEVIO_CONV_MATH=f16: poses rel err {ep:.2e}, h_T rel err {eh:.2e} (fp32 parity bar: 1e-4)")
    assert ep < 5e-3 and eh < 5e-3


# ------------------------------------------------------------------------------------------------
# precision hardening of the fp16x2 operand split
# ------------------------------------------------------------------------------------------------
def _structured_frames(B, S, H, W, seed):
    """uint8 frames with what camera frames have and noise does not: flat regions (incl. pure black / white = the ends of
    the normalised range, -0.5 and +0.5 - 1/255...), smooth gradients, hard edges, a little sensor noise."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W]
    out = np.zeros((B, S, H, W, 3), dtype=np.uint8)
    for b in range(B):
        for s in range(S):
            f = np.stack([(x + 3 * s) * 255 // (W + 3 * S), (y + b) * 255 // (H + B), ((x // 8 + y // 8 + s) % 2) * 255], -1).astype(np.int32)
            f[H // 6:H // 3, W // 5:W // 2] = 255                       # saturated
            f[H // 2:2 * H // 3, W // 2:4 * W // 5] = 0                 # black
            f[2 * H // 3:, :W // 4] = 128                               # flat grey: the l piece of byte/255 - 0.5 is tiny here
            f += rng.integers(-2, 3, f.shape)
            out[b, s] = np.clip(f, 0, 255).astype(np.uint8)
    return torch.from_numpy(out)


def test_structured_uint8_frames(dev, capsys):
    """Structured frames (flat, saturated and black regions, gradients, checkerboards) through odevio_forward_u8: the parity
    bar of the noise inputs must hold, and the encoder features must sit as close to the fp64 truth as for noise."""
    opt = default_opt(img_h=64, img_w=128, ode_solver="rk4")
    model, sd = make_model(opt, seed=83)
    B, S = 2, 4
    frames = _structured_frames(B, S, 64, 128, seed=2)
    _, imu, ts = synth.batch(B, S, 64, 128, seed=18)
    as_float = frames.permute(0, 1, 4, 2, 3).float().div(255) - 0.5
    poses, h = model(frames.cuda(), imu.cuda(), ts.cuda())
    model.check()
    ref_p, ref_h = oc.deepvio_forward(sd, as_float, imu, ts, None, opt)
    assert_close(poses, ref_p, what="poses (structured frames)")
    assert_close(h, ref_h, what="h_T (structured frames)")
    fv = model.image_encoder(as_float.cuda())
    truth = oc.image_encoder(sd, as_float, torch.float64)
    e_hip, e_cpu = oc.rel_err(fv, truth), oc.rel_err(oc.image_encoder(sd, as_float), truth)
    with capsys.disabled():
        print(f"\nstructured frames, encoder vs fp64 truth: HIP f16x2 {e_hip:.2e}, CPU fp32 {e_cpu:.2e}")
    assert e_hip < 5e-6


@pytest.mark.parametrize("case", ["tiny", "huge"])
def test_extreme_batchnorm_statistics(dev, monkeypatch, case, capsys):
    """Checkpoints whose BatchNorm statistics push a layer's activations far from O(1): `tiny` makes conv2's outputs ~1e-4
    (their low fp16 pieces would be subnormal: absolute resolution 2^-25) and lets conv3's BatchNorm scale them back,
    `huge` makes conv4's outputs ~3e3 (towards the fp16 range) with conv4_1's statistics normalising them.  The plan's
    per-layer activation exponents must keep the fp16x2 encoder as accurate as the fp32-input MFMA path."""
    from odevio_amd import DeepVIO
    opt = default_opt(img_h=64, img_w=128)
    sd = weights.make_state_dict(opt, seed=84, randomize_stats=True)
    if case == "tiny":
        sd["Image_net.conv2.1.weight"] *= 2e-4
        sd["Image_net.conv2.1.bias"] *= 2e-4
        sd["Image_net.conv3.1.running_var"] *= 1e-9      # eps = 1e-5 caps the gain at 316
        sd["Image_net.conv3.1.running_mean"] *= 2e-4
        sd["Image_net.conv3.1.weight"] *= 30.0
    else:
        sd["Image_net.conv4.1.weight"] *= 3000.0
        sd["Image_net.conv4.1.bias"] *= 3000.0
        sd["Image_net.conv4_1.1.running_var"] *= 9e6
        sd["Image_net.conv4_1.1.running_mean"] *= 3000.0
    img = synth.images(2, 3, 64, 128, seed=7)
    truth = oc.image_encoder(sd, img, torch.float64)
    assert 0.05 < float(truth.abs().max()) < 1e3       # the modification is compensated: the features stay ordinary
    errs = {}
    for mode in ("f16x2", "f32"):
        monkeypatch.setenv("ODEVIO_CONV_MATH", mode)
        m = DeepVIO(opt, state_dict=sd).cuda()
        fv = m.image_encoder(img.cuda())
        m.check()
        errs[mode] = oc.rel_err(fv, truth)
    with capsys.disabled():
        print(f"\nextreme BatchNorm statistics ({case}): encoder vs fp64 truth", {k: f"{v:.2e}" for k, v in errs.items()})
    assert errs["f16x2"] < 1e-5 and errs["f16x2"] < 4.0 * errs["f32"] + 1e-6


def test_elementwise_error_histogram(dev, monkeypatch, capsys):
    """Element-wise (not tensor-max) error of conv3_1 and conv6 against an fp64 evaluation of the same block fed with the
    same input, for the fp16x2 and the fp32-input MFMA encoders: percentiles of |err| / max|ref| and, for elements above
    1 % of the tensor's max, of the element's own relative error."""
    from odevio_amd import DeepVIO
    opt = default_opt(img_h=64, img_w=128)
    sd = weights.make_state_dict(opt, seed=85, randomize_stats=True)
    img = synth.images(2, 3, 64, 128, seed=9)
    _, inter = oc.image_encoder(sd, img, torch.float64, return_intermediate=True)
    names = [n for n, _, _ in oc.IMAGE_CONVS]
    rows = []
    for mode in ("f16x2", "f32"):
        monkeypatch.setenv("ODEVIO_CONV_MATH", mode)
        m = DeepVIO(opt, state_dict=sd).cuda()
        for layer in ("conv3_1", "conv6"):
            i = names.index(layer)
            x = nhwc(inter[names[i - 1]]).float()
            ref = nhwc(inter[layer])                       # fp64, from the fp64 input (x is its fp32 rounding)
            got = m.conv_block(i, x.cuda(), 2, 3).double().cpu()
            err = (got - ref).abs()
            scale = float(ref.abs().max())
            big = ref.abs() > 0.01 * scale
            q = lambda t, p: float(torch.quantile(t.flatten()[:1_000_000], p))
            rows.append((mode, layer, q(err / scale, 0.5), q(err / scale, 0.999), float(err.max() / scale),
                         q((err / ref.abs())[big], 0.5), q((err / ref.abs())[big], 0.999)))
        m.check()
    with capsys.disabled():
        print("\nmode   layer     |err|/max: p50      p99.9    max      own-relative (>1% of max): p50      p99.9")
        for r in rows:
            print(f"{r[0]:6s} {r[1]:8s}            {r[2]:.2e} {r[3]:.2e} {r[4]:.2e}                              {r[5]:.2e} {r[6]:.2e}")
    for r in rows:
        assert r[4] < 1e-5 and r[6] < 1e-4, r


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("fp32_mfma", 1e-4), ("fp16", 5e-3), ("bf16", 5e-3)])
def test_dtype_flag_selects_the_arithmetic(dev, dtype, tol):
    """--dtype (build extension; the reference is fp32-only) picks the encoder's arithmetic through the config, not an
    environment variable: fp32 (two fp16 pieces per operand) and fp32_mfma hold the 1e-4 bar, fp16 / bf16 are the reduced mode."""
    opt = default_opt(img_h=64, img_w=128, ode_solver="rk4", dtype=dtype)
    model, sd = make_model(opt, seed=86)
    img, imu, ts = synth.batch(2, 4, 64, 128, seed=19)
    poses, h = model(img.cuda(), imu.cuda(), ts.cuda())
    model.check()
    ref_p, ref_h = oc.deepvio_forward(sd, img, imu, ts, None, opt)
    assert oc.rel_err(poses, ref_p) < tol and oc.rel_err(h, ref_h) < tol
    if dtype == "bf16":
        from odevio_amd import DeepVIO
        with pytest.raises(ValueError):
            DeepVIO(default_opt(img_h=64, img_w=128, dtype="fp8"))
