"""CPU restatement of the ODE-VIO hot path - TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker / the timed CPU baseline.  The product path (``odevio_amd``) never
imports it and fails loudly when the HIP library is missing.

What it restates (every function cites the reference file:line it follows; paths are relative to
the reference checkout, mc1017/ODE-VIO):

* ``image_encoder``    - src/models/Encoder.py:97-122 (+ ``conv`` block :8-22); with ``train=`` the model.train() semantics the
                         reference trains under (scripts/train_model.py:219): batch-statistics BatchNorm + Dropout for given masks
* ``inertial_encoder`` - src/models/Encoder.py:60-74 (same ``train=`` switch)
* ``fuse``             - src/models/FusionModule.py:17-23 (``cat`` and ``soft``)
* ``ode_func``         - src/models/ODEFunc.py:9-15,38-39
* ``evolve_state``     - src/models/PoseODERNN.py:70-75 + torchode 0.2.0 (NOT in the checkout)
* ``rnn_stack``        - torch.nn.RNN / torch.nn.GRU as configured in PoseODERNN.py:139-148
* ``pose_ode_rnn``     - src/models/PoseODERNN.py:88-123
* ``pose_rnn``         - src/models/PoseRNN.py:53-73 (importable in the build container: pins the
                         fuse -> RNN -> regressor skeleton against the real reference)
* ``cde_func`` / ``pose_cde`` - src/models/ODEFunc.py:44-83, src/models/PoseCDE.py:76-103 + torchcde
                         0.2.5 / torchdiffeq 0.2.3 (NOT in the checkout)
* ``deepvio_forward``  - src/models/DeepVIO.py:61-68

Pinning status
--------------
* Encoders, fusion, ODEFunc/CDEFunc, and the RNN + regressor skeleton are pinned against outputs
  of the real reference modules, captured in the build container by ``oracle/gen_golden.py`` and
  committed under ``tests/golden/`` (checked by ``tests/test_oracle_golden.py``).
* The integrator arithmetic lives in third-party ``torchode==0.2.0`` (ODE-RNN) and
  ``torchcde==0.2.5`` -> ``torchdiffeq==0.2.3`` (CDE); neither is installed here nor vendored in
  the reference, and the reference has no tests or golden vectors for them.  **Parity of
  ``evolve_state`` / ``pose_cde`` with the real libraries is UNPINNED**; they follow the written
  spec in DESIGN.md section 3 (the libraries' published algorithms) and are cross-checked against
  independent known answers (SciPy RK45 with the same controller constants, matrix exponentials,
  convergence orders).  RK4 is a BASELINE-defined extension the reference does not offer.

The arithmetic is plain PyTorch CPU ops in the requested ``dtype`` (float32 = the reference's CPU
path, same ATen kernels; float64 = high-accuracy truth for error budgeting).
"""
import math

import torch
import torch.nn.functional as F_

IMAGE_CONVS = [  # (name, kernel, stride) - reference Encoder.py:82-90
    ("conv1", 7, 2), ("conv2", 5, 2), ("conv3", 5, 2), ("conv3_1", 3, 1), ("conv4", 3, 2),
    ("conv4_1", 3, 1), ("conv5", 3, 2), ("conv5_1", 3, 1), ("conv6", 3, 2),
]
BN_EPS = 1e-5  # nn.BatchNorm default


def _sd(sd, dtype):
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


# ----------------------------------------------------------------------------------------------
# encoders
# ----------------------------------------------------------------------------------------------
IMAGE_DROPOUT = (0.2,) * 8 + (0.5,)  # Encoder.py:82-90: Dropout(0.2) after conv1 .. conv5_1, Dropout(0.5) after conv6
BN_MOMENTUM = 0.1  # nn.BatchNorm default


def _bn(sd, q, y, train):
    """BatchNorm in eval mode (running statistics) or, with ``train`` = a dict that collects the updated buffers, exactly what
    the module does under ``model.train()``: F.batch_norm(training=True, momentum=0.1) on CLONES of the running statistics
    (batch statistics normalise, the clones receive torch's own update) and num_batches_tracked + 1."""
    if train is None:
        return F_.batch_norm(y, sd[q + ".running_mean"], sd[q + ".running_var"], sd[q + ".weight"], sd[q + ".bias"], False, 0.0, BN_EPS)
    rm, rv = sd[q + ".running_mean"].clone(), sd[q + ".running_var"].clone()
    out = F_.batch_norm(y, rm, rv, sd[q + ".weight"], sd[q + ".bias"], True, BN_MOMENTUM, BN_EPS)
    train[q + ".running_mean"], train[q + ".running_var"] = rm, rv
    if q + ".num_batches_tracked" in sd:
        train[q + ".num_batches_tracked"] = sd[q + ".num_batches_tracked"] + 1
    return out


def _dropout(x, mask, p):
    """nn.Dropout(p) in train mode for a GIVEN keep mask (1 = kept): x * mask / (1 - p).  torch draws the mask from its own
    generator (no bit-level parity possible); the GPU tests hand the device's mask in (odevio_debug_dropout)."""
    if mask is None:
        return x
    return x * (mask.to(x.dtype) * (1.0 / (1.0 - p)))


def conv_block(sd, prefix, x, k, stride, train=None, mask=None, p_drop=0.0):
    """Conv2d(bias=False, pad=(k-1)//2) -> BN2d -> LeakyReLU(0.1) -> Dropout; Encoder.py:8-22.  eval (train=None): running
    statistics, dropout = id.  train mode: batch statistics (+ the buffers' update collected in ``train``), dropout with ``mask``."""
    y = F_.conv2d(x, sd[prefix + ".0.weight"], None, stride=stride, padding=(k - 1) // 2)
    y = F_.leaky_relu(_bn(sd, prefix + ".1", y, train), 0.1)
    return _dropout(y, mask, p_drop) if train is not None else y


def image_encoder(sd, img, dtype=torch.float32, return_intermediate=False, train=None, masks=None):
    """Encoder.py:97-122.  img [B,S,3,H,W] -> fv [B,S-1,v_f_len].  ``train`` (a dict) switches to model.train() semantics:
    batch-statistics BatchNorm, the updated buffers returned in the dict, Dropout with the given ``masks`` (9 NCHW keep masks)."""
    sd = _sd(sd, dtype)
    img = img.to(dtype)
    v = torch.cat((img[:, :-1], img[:, 1:]), dim=2)  # :101 pair concat on the channel axis
    B, P = v.shape[0], v.shape[1]
    x = v.reshape(B * P, v.shape[2], v.shape[3], v.shape[4])
    inter = {}
    for i, (name, k, s) in enumerate(IMAGE_CONVS):
        x = conv_block(sd, "Image_net." + name, x, k, s, train, None if masks is None else masks[i], IMAGE_DROPOUT[i])
        if return_intermediate:
            inter[name] = x
    flat = x.reshape(B, P, -1)  # :110 flatten in (C,H,W) order
    fv = F_.linear(flat, sd["Image_net.visual_head.weight"], sd["Image_net.visual_head.bias"])
    return (fv, inter) if return_intermediate else fv


def inertial_encoder(sd, imu, dtype=torch.float32, train=None, masks=None, p_drop=0.0):
    """Encoder.py:60-74.  imu [B,T,6] -> fi [B,(T-1)//10,i_f_len]; windows of 11 samples, stride 10.  ``train`` / ``masks``
    (3 keep masks [B*P, C, 11]) / ``p_drop`` = opt.imu_dropout: model.train() semantics as in image_encoder."""
    sd = _sd(sd, dtype)
    imu = imu.to(dtype)
    B = imu.shape[0]
    n_pairs = (imu.shape[1] - 1) // 10
    win = torch.stack([imu[:, 10 * i:10 * i + 11, :] for i in range(n_pairs)], dim=1)  # [B,P,11,6]
    x = win.reshape(B * n_pairs, 11, 6).permute(0, 2, 1)  # [BP,6,11]
    for j, idx in enumerate((0, 4, 8)):
        p = f"Inertial_net.encoder_conv.{idx}"
        q = f"Inertial_net.encoder_conv.{idx + 1}"
        x = F_.conv1d(x, sd[p + ".weight"], sd[p + ".bias"], padding=1)
        x = F_.leaky_relu(_bn(sd, q, x, train), 0.1)
        if train is not None:
            x = _dropout(x, None if masks is None else masks[j], p_drop)
    out = F_.linear(x.reshape(x.shape[0], -1), sd["Inertial_net.proj.weight"], sd["Inertial_net.proj.bias"])
    return out.reshape(B, n_pairs, -1)


def fuse(sd, fv, fi, method, dtype=torch.float32, noise=None):
    """FusionModule.py:17-29.  ``hard`` is stochastic in the reference (F.gumbel_softmax under torch's generator): it is restated
    here for GIVEN Gumbel noise ``noise`` [..., F, 2] - torch.nn.functional.gumbel_softmax(logits, tau=1, hard=True) with its
    ``-empty_like(logits).exponential_().log()`` replaced by the argument: y_soft = softmax(logits + noise),
    ret = y_hard - y_soft.detach() + y_soft (straight-through), fused = cat * ret[..., 0]."""
    sd = _sd(sd, dtype)
    c = torch.cat((fv.to(dtype), fi.to(dtype)), -1)
    if method == "cat":
        return c
    if method == "soft":
        return c * F_.linear(c, sd["Pose_net.fuse.net.0.weight"], sd["Pose_net.fuse.net.0.bias"])
    if method == "hard" and noise is not None:
        logits = F_.linear(c, sd["Pose_net.fuse.net.0.weight"], sd["Pose_net.fuse.net.0.bias"]).view(*c.shape, 2)
        y_soft = (logits + noise.to(dtype)).softmax(-1)
        y_hard = torch.zeros_like(y_soft).scatter_(-1, y_soft.max(-1, keepdim=True)[1], 1.0)
        ret = y_hard - y_soft.detach() + y_soft
        return c * ret[..., 0]
    raise ValueError(f"fuse method {method!r} has no deterministic restatement (hard needs the noise)")


# ----------------------------------------------------------------------------------------------
# vector fields
# ----------------------------------------------------------------------------------------------
def _activation(name):
    """ODEFunc.py:23-36 (LeakyReLU with the *default* slope 0.01, Softplus beta=1 threshold=20)."""
    if name == "tanh":
        return torch.tanh
    if name == "relu":
        return torch.relu
    if name == "leaky_relu":
        return lambda x: F_.leaky_relu(x, 0.01)
    if name == "softplus":
        return F_.softplus
    raise ValueError(f"Activation function {name} not supported")


def mlp_tanh_out(sd, prefix, n_hidden, x, act):
    """Linear, act, (n_hidden-1) x [Linear, act], Linear, Tanh - ODEFunc.py:9-15 / :52-58."""
    fn = _activation(act)
    for li in range(n_hidden + 1):
        x = F_.linear(x, sd[f"{prefix}.{2 * li}.weight"], sd[f"{prefix}.{2 * li}.bias"])
        x = fn(x) if li < n_hidden else torch.tanh(x)
    return x


def ode_func(sd, y, n_hidden, act, dtype=torch.float32):
    """ODEFunc.forward (ODEFunc.py:38-39); autonomous: t is ignored."""
    return mlp_tanh_out(_sd(sd, dtype), "Pose_net.ode_func.net", n_hidden, y.to(dtype), act)


# ----------------------------------------------------------------------------------------------
# explicit Runge-Kutta tableaux
# ----------------------------------------------------------------------------------------------
class Tableau:
    def __init__(self, name, a, b, b_err, order, fsal):
        self.name, self.a, self.b, self.b_err, self.order, self.fsal = name, a, b, b_err, order, fsal
        self.stages = len(b)


DOPRI5 = Tableau(
    "dopri5",
    a=[[],
       [1 / 5],
       [3 / 40, 9 / 40],
       [44 / 45, -56 / 15, 32 / 9],
       [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
       [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
       [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]],
    b=[35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0],
    b_err=[35 / 384 - 5179 / 57600, 0.0, 500 / 1113 - 7571 / 16695, 125 / 192 - 393 / 640,
           -2187 / 6784 + 92097 / 339200, 11 / 84 - 187 / 2100, -1 / 40],
    order=5, fsal=True)

TSIT5 = Tableau(
    "tsit5",
    a=[[],
       [0.161],
       [-0.008480655492356989, 0.335480655492357],
       [2.8971530571054935, -6.359448489975075, 4.3622954328695815],
       [5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525],
       [5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401,
        -0.028269050394068383],
       [0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081,
        2.324710524099774]],
    b=[0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081,
       2.324710524099774, 0.0],
    b_err=[-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995,
           -0.1447110071732629, 0.5823571654525552, -0.45808210592918697, 1 / 66],
    order=5, fsal=True)

HEUN = Tableau("heun", a=[[], [1.0]], b=[0.5, 0.5], b_err=[-0.5, 0.5], order=2, fsal=False)
EULER = Tableau("euler", a=[[]], b=[1.0], b_err=None, order=1, fsal=False)
# torchdiffeq's method="rk4" is the 3/8 rule (rk4_alt_step_func); rk4_classic is the textbook one.
RK4_38 = Tableau("rk4", a=[[], [1 / 3], [-1 / 3, 1.0], [1.0, -1.0, 1.0]],
                 b=[1 / 8, 3 / 8, 3 / 8, 1 / 8], b_err=None, order=4, fsal=False)
RK4_CLASSIC = Tableau("rk4_classic", a=[[], [0.5], [0.0, 0.5], [0.0, 0.0, 1.0]],
                      b=[1 / 6, 1 / 3, 1 / 3, 1 / 6], b_err=None, order=4, fsal=False)

TABLEAUX = {t.name: t for t in (DOPRI5, TSIT5, HEUN, EULER, RK4_38, RK4_CLASSIC)}
TABLEAUX["runge_kutta"] = RK4_38
REFERENCE_SOLVERS = ("dopri5", "heun", "tsit5", "euler")  # PoseODERNN.py:125-137
FIXED_STEP_SOLVERS = ("rk4", "runge_kutta", "rk4_classic")

# torchode IntegralController(atol=1e-6, rtol=1e-2) defaults, PoseODERNN.py:57,72
ATOL, RTOL, DT0 = 1e-6, 1e-2, 1e-4
SAFETY, FACTOR_MIN, FACTOR_MAX = 0.9, 0.2, 10.0
MAX_STEPS = 1_000_000


def rk_stages(f, tab, y, dt, k1=None):
    """One explicit RK step for a batch of rows with per-row ``dt`` [R].  Returns (y1, err, k_last)."""
    dtc = dt[:, None]
    ks = [f(y) if k1 is None else k1]
    for i in range(1, tab.stages):
        acc = None
        for j, aij in enumerate(tab.a[i]):
            if aij == 0.0:
                continue
            term = ks[j] * aij
            acc = term if acc is None else acc + term
        ks.append(f(y + dtc * acc))
    if tab.fsal:
        # the last stage is evaluated AT y1 (a[-1] == b[:-1]), so y1 is that stage's argument
        acc = None
        for j, bj in enumerate(tab.b[:-1]):
            if bj == 0.0:
                continue
            term = ks[j] * bj
            acc = term if acc is None else acc + term
        y1 = y + dtc * acc
    else:
        acc = None
        for j, bj in enumerate(tab.b):
            if bj == 0.0:
                continue
            term = ks[j] * bj
            acc = term if acc is None else acc + term
        y1 = y + dtc * acc
    err = None
    if tab.b_err is not None:
        acc = None
        for j, ej in enumerate(tab.b_err):
            if ej == 0.0:
                continue
            term = ks[j] * ej
            acc = term if acc is None else acc + term
        err = dtc * acc
    return y1, err, ks[-1]


def evolve_state(f, y0, t0, t1, method, substeps=1, atol=ATOL, rtol=RTOL, dt0=DT0, trace=None, detach_controller=False):
    """Solve dy/dt = f(y) from per-row ``t0`` to ``t1`` (PoseODERNN.py:70-75 + DESIGN.md section 3).

    Rows are independent (own t, dt, accept flag), exactly as in torchode.  ``method`` in the
    reference's set uses the adaptive I-controller; ``rk4``/``rk4_classic`` take ``substeps`` equal
    steps.  ``trace`` (optional dict) receives per-row step counts and the dt sequence.
    ``detach_controller=True`` (gradient reference of the backward tests): the step sizes the controller picks are
    constants of the differentiation - autograd then differentiates the accepted steps as they were taken.
    """
    tab = TABLEAUX[method]
    R = y0.shape[0]
    dtype = y0.dtype
    t0 = t0.to(dtype)
    t1 = t1.to(dtype)
    y = y0.clone()
    if method in FIXED_STEP_SOLVERS:
        h = (t1 - t0) / substeps
        for _ in range(substeps):
            y, _, _ = rk_stages(f, tab, y, h)
        if trace is not None:
            trace["n_steps"] = torch.full((R,), substeps, dtype=torch.int64)
            trace["n_accepted"] = torch.full((R,), substeps, dtype=torch.int64)
        return y

    t = t0.clone()
    span = t1 - t
    dt_next = torch.full((R,), dt0, dtype=dtype)
    last = dt_next >= span
    dt = torch.where(last, span, dt_next)
    running = t < t1
    n_steps = torch.zeros(R, dtype=torch.int64)
    n_acc = torch.zeros(R, dtype=torch.int64)
    dts = [[] for _ in range(R)]
    k1 = f(y) if tab.fsal else None
    it = 0
    while bool(running.any()):
        it += 1
        if it > MAX_STEPS:
            raise RuntimeError("evolve_state: step budget exhausted")
        y1, err, klast = rk_stages(f, tab, y, dt, k1)
        if err is not None:
            with torch.set_grad_enabled(torch.is_grad_enabled() and not detach_controller):
                ye, y1e, ee = (y.detach(), y1.detach(), err.detach()) if detach_controller else (y, y1, err)
                bound = atol + rtol * torch.maximum(ye.abs(), y1e.abs())
                ratio = torch.sqrt(torch.mean((ee / bound) ** 2, dim=1))
                accept = ratio < 1.0
                factor = torch.clamp(SAFETY * ratio ** (-1.0 / tab.order), FACTOR_MIN, FACTOR_MAX)
                dt_next = dt * factor
        else:
            accept = torch.ones(R, dtype=torch.bool)
            dt_next = dt.clone()
        upd = accept & running
        if trace is not None:
            for r in range(R):
                if bool(running[r]):
                    dts[r].append((float(dt[r]), bool(accept[r])))
        n_steps += running.to(torch.int64)
        n_acc += upd.to(torch.int64)
        t = torch.where(upd, torch.where(last, t1, t + dt), t)
        y = torch.where(upd[:, None], y1, y)
        if tab.fsal:
            k1 = torch.where(upd[:, None], klast, k1)
        running = t < t1
        span = t1 - t
        last = dt_next >= span
        dt = torch.where(last, span, dt_next)
    if trace is not None:
        trace["n_steps"], trace["n_accepted"], trace["dts"] = n_steps, n_acc, dts
    return y


# ----------------------------------------------------------------------------------------------
# RNN stack, regressor, pose nets
# ----------------------------------------------------------------------------------------------
def rnn_stack(sd, rnn_type, n_layers, x, h):
    """One time step of nn.RNN(tanh)/nn.GRU, ``batch_first``, as built in PoseODERNN.py:139-148.

    x [B,F]; h [L,B,F] -> (out [B,F] = top layer, h' [L,B,F]).
    """
    new_h = []
    inp = x
    for l in range(n_layers):
        w_ih, w_hh = sd[f"Pose_net.rnn.weight_ih_l{l}"], sd[f"Pose_net.rnn.weight_hh_l{l}"]
        b_ih, b_hh = sd[f"Pose_net.rnn.bias_ih_l{l}"], sd[f"Pose_net.rnn.bias_hh_l{l}"]
        gi = F_.linear(inp, w_ih, b_ih)
        gh = F_.linear(h[l], w_hh, b_hh)
        if rnn_type == "rnn":
            hn = torch.tanh(gi + gh)
        elif rnn_type == "gru":
            i_r, i_z, i_n = gi.chunk(3, -1)
            h_r, h_z, h_n = gh.chunk(3, -1)
            r = torch.sigmoid(i_r + h_r)
            z = torch.sigmoid(i_z + h_z)
            n = torch.tanh(i_n + r * h_n)
            hn = (1.0 - z) * n + z * h[l]
        else:
            raise ValueError(f"RNN type {rnn_type} not supported")
        new_h.append(hn)
        inp = hn
    return inp, torch.stack(new_h, 0)


def regressor(sd, x):
    """Linear(F,128) -> LeakyReLU(0.1) -> Linear(128,6); PoseODERNN.py:64-68."""
    y = F_.leaky_relu(F_.linear(x, sd["Pose_net.regressor.0.weight"], sd["Pose_net.regressor.0.bias"]), 0.1)
    return F_.linear(y, sd["Pose_net.regressor.2.weight"], sd["Pose_net.regressor.2.bias"])


def pose_ode_rnn(sd, fv, fi, ts, prev, opt, dtype=torch.float32, trace=None, with_ode=True, detach_controller=False):
    """PoseODERNN.forward (PoseODERNN.py:88-123); ``with_ode=False`` gives PoseRNN.forward (PoseRNN.py:53-73)."""
    sd = _sd(sd, dtype)
    fused = fuse(sd, fv, fi, opt.fuse_method, dtype)
    B, P, Fdim = fused.shape
    L = opt.rnn_num_layers
    h = torch.zeros(L, B, Fdim, dtype=dtype) if prev is None else prev.to(dtype).clone()
    ts = ts.to(dtype)
    ts_diff = ts - ts[:, :1] if prev is None else ts  # :100
    f = lambda y: mlp_tanh_out(sd, "Pose_net.ode_func.net", opt.ode_fn_num_layers, y, opt.ode_activation_fn)
    outs = []
    for i in range(P):
        if with_ode:
            # :109-111 - every layer's state is evolved over [t_i, t_{i+1}] with the same ODEFunc;
            # rows are independent in torchode, so stacking the L layers into L*B rows is exact.
            rows = h.reshape(L * B, Fdim)
            t0 = ts_diff[:, i].repeat(L)
            t1 = ts_diff[:, i + 1].repeat(L)
            tr = {} if trace is not None else None
            rows = evolve_state(f, rows, t0, t1, opt.ode_solver, getattr(opt, "ode_substeps", 1), trace=tr,
                                detach_controller=detach_controller)
            if trace is not None:
                trace.setdefault("intervals", []).append(tr)
            h = rows.reshape(L, B, Fdim)
        out, h = rnn_stack(sd, opt.ode_rnn_type, L, fused[:, i], h)  # :114
        outs.append(out)
    output = torch.stack(outs, 1)
    return regressor(sd, output), h


def pose_rnn(sd, fv, fi, ts, prev, opt, dtype=torch.float32):
    return pose_ode_rnn(sd, fv, fi, ts, prev, opt, dtype, with_ode=False)


def deepvio_forward(sd, img, imu, ts, hc, opt, dtype=torch.float32, trace=None):
    """DeepVIO.forward (DeepVIO.py:61-68) for model_type ode-rnn / rnn."""
    fv = image_encoder(sd, img, dtype)
    fi = inertial_encoder(sd, imu, dtype)
    if opt.model_type == "ode-rnn":
        return pose_ode_rnn(sd, fv, fi, ts, hc, opt, dtype, trace)
    if opt.model_type == "rnn":
        return pose_rnn(sd, fv, fi, ts, hc, opt, dtype)
    raise NotImplementedError(f"model_type {opt.model_type!r}")


def rel_err(a, b):
    """max|a-b| / max|b| - the tensor-scale relative error used for the 1e-4 parity bar."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# ----------------------------------------------------------------------------------------------
# Neural-CDE path (PoseCDE.py:76-103) - torchcde 0.2.5 / torchdiffeq 0.2.3 restated (UNPINNED)
# ----------------------------------------------------------------------------------------------
DPS_C_MID = [6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
             187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]
CDE_ATOL, CDE_RTOL = 1e-6, 1e-4  # PoseCDE.py:101


def rectilinear_coeffs(obs):
    """torchcde.linear_interpolation_coeffs(obs, rectilinear=0): [(t1,x1),(t2,x1),(t2,x2),(t3,x2),...] (2L-1 knots at 0..2L-2)."""
    B, L, C = obs.shape
    out = obs.new_empty(B, 2 * L - 1, C)
    out[:, 0::2] = obs
    lag = obs[:, :-1].clone()
    lag[:, :, 0] = obs[:, 1:, 0]
    out[:, 1::2] = lag
    return out


def control_segment(t_f32, n_knots):
    """Index of the linear piece torchcde's LinearInterpolation uses at time t (bucketize(t) - 1, clamped)."""
    idx = int(math.ceil(t_f32)) - 1  # knots are the integers 0..n_knots-1; t on a knot belongs to the piece on its LEFT
    return max(0, min(idx, n_knots - 2))


def cde_field(sd, opt, coeffs, dtype):
    """f(t, z) = CDEFunc(z).view(B, H, H+1) @ dX/dt(t)   (torchcde _VectorField, ODEFunc.py:81-83)."""
    Hc = opt.cde_hidden_dim

    def f(t_f32, z):
        i = control_segment(t_f32, coeffs.shape[1])
        g = coeffs[:, i + 1] - coeffs[:, i]  # knot spacing 1
        vf = mlp_tanh_out(sd, "Pose_net.cde_func.net", opt.cde_fn_num_layers, z, opt.cde_activation_fn).view(-1, Hc, Hc + 1)
        return (vf @ g.unsqueeze(-1)).squeeze(-1).to(dtype)
    return f


def _rms(x):
    return float(x.abs().pow(2).mean().sqrt())


def _f32_prev(t):
    return float(torch.nextafter(torch.tensor(t, dtype=torch.float32), torch.tensor(t - 1.0, dtype=torch.float32)))


def _f32_next(t):
    return float(torch.nextafter(torch.tensor(t, dtype=torch.float32), torch.tensor(t + 1.0, dtype=torch.float32)))


def odeint_dopri5(f, y0, ts, jump_t, rtol=CDE_RTOL, atol=CDE_ATOL, trace=None):
    """torchdiffeq 0.2.3 odeint(method='dopri5') with a `jump_t` option, restated: ONE shared step size for the whole
    batch (RMS norm over every element), Hairer initial step, 4th-order dense output at the requested times, steps
    clipped at the jump points where f is re-evaluated on the far side.  Time is float64, the state `y0.dtype`;
    f receives time as float32 (torchdiffeq casts it to the state's dtype), perturbed one ulp backwards at a step end."""
    tab = DOPRI5
    f32 = lambda t: float(torch.tensor(t, dtype=torch.float32))
    t0 = float(ts[0])
    fy0 = f(f32(t0), y0)
    # _select_initial_step(order = 4)
    scale = atol + y0.abs() * rtol
    d0, d1 = _rms(y0 / scale), _rms(fy0 / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    h0 = float(torch.tensor(h0, dtype=y0.dtype))
    y1 = y0 + h0 * fy0
    f1 = f(f32(f32(t0) + h0), y1)
    d2 = abs(_rms((f1 - fy0) / scale) / h0)
    if d1 <= 1e-15 and d2 <= 1e-15:
        h1 = max(1e-6, h0 * 1e-3)
    else:
        h1 = float(torch.tensor(0.01 / max(d1, d2), dtype=y0.dtype) ** (1.0 / 5.0))
    dt = float(min(100 * h0, h1))
    jumps = sorted(j for j in jump_t if j > t0)
    ji = 0
    y, fy, tcur, tprev = y0, fy0, t0, t0
    interp = [y0] * 5
    out = [y0]
    n_steps = n_acc = 0
    for target in [float(t) for t in ts[1:]]:
        while target > tcur:
            n_steps += 1
            if n_steps > MAX_STEPS:
                raise RuntimeError("odeint_dopri5: step budget exhausted")
            step = dt
            t1 = tcur + step
            on_jump = False
            if jumps and tcur < jumps[ji] < tcur + step:
                on_jump = True
                t1 = jumps[ji]
                step = t1 - tcur
            dtf = float(torch.tensor(step, dtype=y0.dtype))
            ks = [fy]
            for i in range(1, 7):
                acc = None
                for j, aij in enumerate(tab.a[i]):
                    term = ks[j] * (aij * dtf)
                    acc = term if acc is None else acc + term
                ti = _f32_prev(f32(t1)) if i == 6 else f32(f32(tcur) + (sum(tab.a[i])) * dtf)  # alpha_i = sum of row i
                ks.append(f(ti, y + acc))
                if i == 6:
                    y1 = y + acc  # FSAL: the last stage is evaluated at y1
            err = None
            for j, ej in enumerate(tab.b_err):
                term = ks[j] * (ej * dtf)
                err = term if err is None else err + term
            tol = atol + rtol * torch.maximum(y.abs(), y1.abs())
            ratio = _rms(err / tol)
            accept = ratio <= 1.0
            if trace is not None:
                trace.setdefault("steps", []).append((tcur, step, accept))
            if accept:
                n_acc += 1
                ymid = None
                for j, mj in enumerate(DPS_C_MID):
                    term = ks[j] * (mj * dtf)
                    ymid = term if ymid is None else ymid + term
                ymid = y + ymid
                fa, fb = ks[0], ks[6]
                interp = [y, dtf * fa, dtf * (fb - 4 * fa) - 11 * y - 5 * y1 + 16 * ymid,
                          dtf * (5 * fa - 3 * fb) + 18 * y + 14 * y1 - 32 * ymid,
                          2 * dtf * (fb - fa) - 8 * (y1 + y) + 16 * ymid]
                tprev, tcur, y, fy = tcur, t1, y1, ks[6]
                if on_jump:
                    if ji != len(jumps) - 1:
                        ji += 1
                    fy = f(_f32_next(f32(tcur)), y)  # the far side of the discontinuity
            # _optimal_step_size
            if ratio == 0:
                factor = 10.0
            else:
                dfac = 1.0 if ratio < 1 else 0.2
                factor = min(10.0, max(0.9 / ratio ** 0.2, dfac))
            dt = step * factor
        x = float(torch.tensor((target - tprev) / (tcur - tprev), dtype=y0.dtype))
        total, xp = interp[0] + x * interp[1], x
        for cf in interp[2:]:
            xp = xp * x
            total = total + xp * cf
        out.append(total)
    if trace is not None:
        trace["n_steps"], trace["n_accepted"] = n_steps, n_acc
    return torch.stack(out, 0)


def odeint_fixed(f, y0, ts, method):
    """torchdiffeq fixed-grid solvers with the output times as the grid (step_size=None): euler, rk4 (3/8 rule)."""
    f32 = lambda t: float(torch.tensor(t, dtype=torch.float32))
    out, y = [y0], y0
    for a, b in zip(ts[:-1], ts[1:]):
        t0, t1 = float(a), float(b)
        dt = float(torch.tensor(t1 - t0, dtype=y0.dtype))
        k1 = f(f32(t0), y)
        if method == "euler":
            y = y + dt * k1
        else:
            k2 = f(f32(t0 + dt / 3), y + dt * k1 / 3)
            k3 = f(f32(t0 + dt * 2 / 3), y + dt * (k2 - k1 / 3))
            k4 = f(_f32_prev(f32(t1)), y + dt * (k1 - k2 + k3))
            y = y + (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
        out.append(y)
    return torch.stack(out, 0)


def pose_cde(sd, fv, fi, ts, prev, history, opt, dtype=torch.float32, training=False, trace=None):
    """PoseCDE.forward (PoseCDE.py:76-103).  Returns (poses [B,P,6], z0 [B,H], new history).

    Faithful to the reference, including its quirks: eval mode uses the raw timestamps (:81), the control path's knots
    are the integers 0..2L-2 while the integration runs over the real times `ts[0, 1:]` of ROW 0 (:101), the first
    output is z0 itself, and z0 (not the final state) is returned (:103).  `reduction_net` is never applied (:53-58).
    """
    sd = _sd(sd, dtype)
    fused = fuse(sd, fv, fi, opt.fuse_method, dtype)
    ts = ts.to(dtype)
    tsd = ts - ts[:, :1] if training else ts
    x = torch.cat([tsd[:, 1:, None], fused], dim=-1)
    if not training:
        history = torch.cat([history, x], dim=1) if prev is not None else x
        obs = history
    else:
        history, obs = None, x
    coeffs = rectilinear_coeffs(obs)
    if prev is None:
        z0 = torch.tanh(F_.linear(coeffs[:, 0], sd["Pose_net.initial.0.weight"], sd["Pose_net.initial.0.bias"]))
    else:
        z0 = prev.to(dtype)
    f = cde_field(sd, opt, coeffs, dtype)
    tt = [float(v) for v in tsd[0, 1:]]
    if opt.cde_solver == "dopri5":
        zs = odeint_dopri5(f, z0, tt, jump_t=[float(k) for k in range(coeffs.shape[1])], trace=trace)
    elif opt.cde_solver in ("euler", "rk4", "runge_kutta"):
        zs = odeint_fixed(f, z0, tt, "euler" if opt.cde_solver == "euler" else "rk4")
    else:
        raise ValueError(f"Solver {opt.cde_solver} not supported")
    return regressor(sd, zs.transpose(0, 1)), z0, history
