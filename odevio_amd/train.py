"""Training-side surface of the hot path: gradients through the ODE-RNN pose net.

The reference trains with plain autograd (scripts/train_model.py:69-78): ``poses, _ = model(...)``, the loss
``100 * MSE(angles) + MSE(translations)``, ``loss.backward()`` through torchode's AutoDiffAdjoint, i.e. backpropagation
through the solver's own arithmetic.  Here the same chain is two ``torch.autograd.Function``s whose forward AND backward
run in libodevio (``odevio_ode_rnn_fwd`` / ``odevio_ode_rnn_bwd``, ``odevio_pose_loss``); PyTorch only carries the graph.

What ``odevio_ode_rnn_bwd`` covers: fixed-step solvers (rk4, rk4_classic) and adaptive ones (dopri5, tsit5, heun: the
forward's accepted steps are replayed, their sizes held constant), ``nn.RNN`` and ``nn.GRU``; ``odevio_fuse_bwd`` covers
``cat`` and ``soft`` fusion, ``odevio_fuse_hard_bwd`` the straight-through estimator of ``hard``.  Gradients reach the encoder FEATURES (fv, fi), the carried state ``hc`` and every parameter
of ``Pose_net`` (fusion, ODEFunc, RNN, regressor) - exactly the parameters the reference's optimizer holds
(utils/utils.py:115-119: ``Pose_net.get_other_params()`` + ``get_regressor_params()``; the encoders are not in it).
``PoseNetTrainer`` is that optimizer step on the device: ``clip_grad_norm_`` + ``torch.optim.Adam`` (or ``SGD``) as kernels
(``odevio_grad_clip``, ``odevio_adam_step``, ``odevio_sgd_step``) and ``odevio_plan_update`` to put the new parameters in front of
the forward kernels.

Train-mode semantics.  The reference trains under ``model.train()`` (scripts/train_model.py:219): BatchNorm with batch statistics
and Dropout in BOTH encoders, the frozen ``Image_net`` included.  With ``model.train()`` set, ``model.image_encoder`` /
``imu_encoder`` compute exactly that (``odevio_image_encoder_fwd_train`` / ``odevio_imu_encoder_fwd_train``: running statistics
moved in place, masks from the model's Philox stream) and ``imu_encoder`` / ``image_encoder`` below back-propagate through the
batch-statistics BatchNorm and the same masks (``odevio_imu_encoder_bwd_train``, ``odevio_image_encoder_bwd``).  The reference's own
recipe freezes ``Image_net`` (``--freeze_encoder``, scripts/run_training.sh:24); with the flag off its gradients are computed and
count in ``clip_grad_norm_`` exactly as in the reference, whose optimizer nevertheless never updates the encoders
(utils/utils.py:116-119).
"""
import ctypes

import torch

from . import _lib


def pose_param_names(opt):
    """Reference ``state_dict`` keys of the parameters ``odevio_ode_rnn_bwd`` produces gradients for, in a fixed order."""
    names = []
    if opt.model_type == "ode-rnn":
        for l in range(opt.ode_fn_num_layers + 1):
            names += [f"Pose_net.ode_func.net.{2 * l}.weight", f"Pose_net.ode_func.net.{2 * l}.bias"]
    for k in range(opt.rnn_num_layers):
        names += [f"Pose_net.rnn.{w}_l{k}" for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    names += [f"Pose_net.regressor.{i}.{w}" for i in (0, 2) for w in ("weight", "bias")]
    return names


def fuse_param_names(opt):
    """Parameters of ``Pose_net.fuse`` (FusionModule.py:11-14): a Linear for ``soft`` (and ``hard``), none for ``cat``."""
    return ["Pose_net.fuse.net.0.weight", "Pose_net.fuse.net.0.bias"] if opt.fuse_method in ("soft", "hard") else []


def imu_param_names():
    """Parameters of ``Inertial_net`` (Encoder.py:41-56): three Conv1d + BatchNorm1d pairs and the projection."""
    names = []
    for i in (0, 4, 8):
        names += [f"Inertial_net.encoder_conv.{i}.weight", f"Inertial_net.encoder_conv.{i}.bias",
                  f"Inertial_net.encoder_conv.{i + 1}.weight", f"Inertial_net.encoder_conv.{i + 1}.bias"]
    return names + ["Inertial_net.proj.weight", "Inertial_net.proj.bias"]


def _versions(params):
    """What changes when a parameter is written to (in place or replaced): the identity of a forward's weights."""
    return tuple((p.data_ptr(), p._version) for p in params)


def _tensor_array(names, tensors):
    arr = (_lib.OdevioTensor * len(tensors))()
    for i, (n, t) in enumerate(zip(names, tensors)):
        arr[i].name, arr[i].data, arr[i].numel = n.encode(), t.data_ptr(), t.numel()
    return arr


class _FuseFunction(torch.autograd.Function):
    """(fv [B,P,v], fi [B,P,i], *fusion parameters) -> fused [B,P,v+i]  (FusionModule.forward, cat / soft)."""

    @staticmethod
    def forward(ctx, model, names, fv, fi, *params):
        fv, fi = fv.detach().contiguous().float(), fi.detach().contiguous().float()
        ctx.rng = model.rng_state() if model.opt.fuse_method == "hard" else None   # the draw this forward's Gumbel mask uses
        fused = model.fuse(fv, fi)
        ctx.model, ctx.names = model, names
        ctx.param_shapes = [tuple(p.shape) for p in params]
        ctx.save_for_backward(fv, fi)
        return fused

    @staticmethod
    def backward(ctx, g_fused):
        model = ctx.model
        fv, fi = ctx.saved_tensors
        g_fused = g_fused.contiguous().float()
        g_fv, g_fi = torch.empty_like(fv), torch.empty_like(fi)
        grads = [torch.empty(s, device=fv.device, dtype=torch.float32) for s in ctx.param_shapes]
        arr = _tensor_array(ctx.names, grads)
        model._ensure_plan()
        if ctx.rng is not None:   # "hard": straight-through gradient for the mask of that draw
            _lib.check(model._lib.odevio_fuse_hard_bwd(model._plan, fv.data_ptr(), fi.data_ptr(), fv.shape[0] * fv.shape[1], ctx.rng[0], ctx.rng[1],
                                                       g_fused.data_ptr(), g_fv.data_ptr(), g_fi.data_ptr(), arr, len(grads), model._stream()))
        else:
            _lib.check(model._lib.odevio_fuse_bwd(model._plan, fv.data_ptr(), fi.data_ptr(), fv.shape[0] * fv.shape[1], g_fused.data_ptr(),
                                                  g_fv.data_ptr(), g_fi.data_ptr(), arr, len(grads), model._stream()))
        return (None, None, g_fv, g_fi, *grads)


class _OdeRnnFunction(torch.autograd.Function):
    """(fused [B,P,F], ts [B,P+1], hc [L,B,F] | None, *pose parameters) -> (poses [B,P,6], h_T [L,B,F])."""

    @staticmethod
    def forward(ctx, model, names, fused, ts, hc, *params):
        fused = fused.detach().contiguous()
        ts = ts.detach().to(torch.float32).contiguous()
        hcd = None if hc is None else hc.detach().contiguous()
        B, P, F = fused.shape
        L = model.opt.rnn_num_layers
        poses = torch.empty(B, P, 6, device=fused.device, dtype=torch.float32)
        h_T = torch.empty(L, B, F, device=fused.device, dtype=torch.float32)
        # the log of this forward's steps, kept for backward() the way autograd keeps saved tensors: the training step then runs the
        # persistent kernel once (without it odevio_ode_rnn_bwd runs the forward again to write the same log)
        n_tape = ctypes.c_int64(0)
        _lib.check(model._lib.odevio_ode_rnn_tape_floats(model._plan, B, P, ctypes.byref(n_tape)))
        ctx.tape = None
        hcp = None if hcd is None else hcd.data_ptr()
        if n_tape.value > 0 and any(ctx.needs_input_grad):
            ctx.tape = torch.empty(n_tape.value, device=fused.device, dtype=torch.float32)
            ctx.tape_weights = _versions(params)
            _lib.check(model._lib.odevio_ode_rnn_fwd_taped(model._plan, fused.data_ptr(), ts.data_ptr(), hcp, B, P, poses.data_ptr(),
                                                           h_T.data_ptr(), ctx.tape.data_ptr(), n_tape.value, model._stream()))
        else:
            _lib.check(model._lib.odevio_ode_rnn_fwd(model._plan, fused.data_ptr(), ts.data_ptr(), hcp, B, P, poses.data_ptr(), h_T.data_ptr(),
                                                     None, model._stream()))
        ctx.params = params
        ctx.model, ctx.names, ctx.has_hc = model, names, hc is not None
        ctx.save_for_backward(fused, ts, *([hcd] if hcd is not None else []))
        ctx.param_shapes = [tuple(p.shape) for p in params]
        return poses, h_T

    @staticmethod
    def backward(ctx, g_poses, g_hT):
        model = ctx.model
        saved = ctx.saved_tensors
        fused, ts = saved[0], saved[1]
        hc = saved[2] if ctx.has_hc else None
        B, P, F = fused.shape
        g_poses = torch.zeros(B, P, 6, device=fused.device) if g_poses is None else g_poses.contiguous().float()
        g_hT = None if g_hT is None else g_hT.contiguous().float()
        g_fused = torch.empty_like(fused)
        g_hc = torch.empty_like(hc) if hc is not None else None
        grads = [torch.empty(s, device=fused.device, dtype=torch.float32) for s in ctx.param_shapes]
        arr = (_lib.OdevioTensor * len(grads))()
        for i, (n, g) in enumerate(zip(ctx.names, grads)):
            arr[i].name, arr[i].data, arr[i].numel = n.encode(), g.data_ptr(), g.numel()
        model._ensure_plan()
        common = (model._plan, fused.data_ptr(), ts.data_ptr(), None if hc is None else hc.data_ptr(), B, P, g_poses.data_ptr(),
                  None if g_hT is None else g_hT.data_ptr(), g_fused.data_ptr(), None if g_hc is None else g_hc.data_ptr(), arr, len(grads))
        # (a tape is only the log of THESE weights' forward: parameters changed in between -> the plain backward writes a fresh one)
        if ctx.tape is not None and ctx.tape_weights == _versions(ctx.params):
            _lib.check(model._lib.odevio_ode_rnn_bwd_taped(*common, ctx.tape.data_ptr(), ctx.tape.numel(), model._stream()))
        else:
            _lib.check(model._lib.odevio_ode_rnn_bwd(*common, model._stream()))
        ctx.tape = None
        return (None, None, g_fused, None, g_hc, *grads)


class _ImuEncoderFunction(torch.autograd.Function):
    """imu [B,T,6] (+ the Inertial_net parameters) -> fi [B,(T-1)/10,i_f_len]  (InertialEncoder.forward).  ``model.training``
    decides the semantics of forward AND backward: batch-statistics BatchNorm + Dropout(opt.imu_dropout), or eval-mode BatchNorm."""

    @staticmethod
    def forward(ctx, model, names, imu, *params):
        imu = imu.detach().contiguous().float()
        ctx.train_rng = model.rng_state() if model.training else None   # the three dropout draws this forward is about to make
        fi = model.imu_encoder(imu)
        ctx.model, ctx.names = model, names
        ctx.param_shapes = [tuple(p.shape) for p in params]
        ctx.save_for_backward(imu)
        return fi

    @staticmethod
    def backward(ctx, g_fi):
        model = ctx.model
        (imu,) = ctx.saved_tensors
        g_fi = g_fi.contiguous().float()
        grads = [torch.empty(s, device=imu.device, dtype=torch.float32) for s in ctx.param_shapes]
        model._ensure_plan()
        if ctx.train_rng is not None:
            _lib.check(model._lib.odevio_imu_encoder_bwd_train(model._plan, imu.data_ptr(), imu.shape[0], imu.shape[1], float(model.opt.imu_dropout),
                                                               ctx.train_rng[0], ctx.train_rng[1], g_fi.data_ptr(),
                                                               _tensor_array(ctx.names, grads), len(grads), model._stream()))
        else:
            _lib.check(model._lib.odevio_imu_encoder_bwd(model._plan, imu.data_ptr(), imu.shape[0], imu.shape[1], g_fi.data_ptr(),
                                                         _tensor_array(ctx.names, grads), len(grads), model._stream()))
        return (None, None, None, *grads)


def image_param_names():
    """Parameters of ``Image_net`` (Encoder.py:82-95): nine Conv2d (no bias) + BatchNorm2d pairs and the visual head."""
    from . import weights
    names = []
    for name, _, _, _, _ in weights.IMAGE_CONVS:
        names += [f"Image_net.{name}.0.weight", f"Image_net.{name}.1.weight", f"Image_net.{name}.1.bias"]
    return names + ["Image_net.visual_head.weight", "Image_net.visual_head.bias"]


class _ImageEncoderFunction(torch.autograd.Function):
    """img [B,S,3,H,W] (+ the Image_net parameters) -> fv [B,S-1,v_f_len]  (ImageEncoder.forward under ``model.train()``: batch-statistics
    BatchNorm + Dropout) with the backward to every Image_net parameter (``odevio_image_encoder_bwd``).  The frames get no gradient."""

    @staticmethod
    def forward(ctx, model, names, img, *params):
        img = img.detach().contiguous().float()
        fv = model.image_encoder(img, keep=True)
        ctx.model, ctx.names = model, names
        ctx.param_shapes = [tuple(p.shape) for p in params]
        ctx.needs = [p.requires_grad for p in params]
        ctx.save_for_backward(img)
        return fv

    @staticmethod
    def backward(ctx, g_fv):
        model = ctx.model
        (img,) = ctx.saved_tensors
        g_fv = g_fv.contiguous().float()
        grads = [torch.empty(s, device=img.device, dtype=torch.float32) if need else None for s, need in zip(ctx.param_shapes, ctx.needs)]
        pairs = [(n, g) for n, g in zip(ctx.names, grads) if g is not None]
        model._ensure_plan()
        _lib.check(model._lib.odevio_image_encoder_bwd(model._plan, img.data_ptr(), img.shape[0], img.shape[1], g_fv.data_ptr(), g_fv.shape[-1],
                                                       _tensor_array([n for n, _ in pairs], [g for _, g in pairs]), len(pairs), model._stream()))
        return (None, None, None, *grads)


def image_encoder(model, img):
    """``model.Image_net`` forward WITH an autograd graph to its parameters - train mode only (the reference never back-propagates
    through an eval-mode encoder: its loop runs under ``model.train()``, scripts/train_model.py:219)."""
    if not model.training:
        raise RuntimeError("odevio_amd.train.image_encoder: the image encoder's backward exists for train mode (model.train()) - the mode the "
                           "reference trains in; eval-mode features carry no graph (model.image_encoder)")
    model._ensure_plan()
    names = image_param_names()
    params = dict(model.named_parameters())
    return _ImageEncoderFunction.apply(model, names, img, *[params[n] for n in names])


def imu_encoder(model, imu):
    """``model.Inertial_net`` forward WITH an autograd graph to its parameters (the raw IMU samples get no gradient)."""
    model._ensure_plan()
    names = imu_param_names()
    params = dict(model.named_parameters())
    return _ImuEncoderFunction.apply(model, names, imu, *[params[n] for n in names])


class _PoseLossFunction(torch.autograd.Function):
    """100 * MSE(angles) + MSE(translations) (scripts/train_model.py:72-77) with its gradient from the same kernel."""

    @staticmethod
    def forward(ctx, poses, gts):
        poses = poses.contiguous().float()
        gts = gts.detach().contiguous().float()
        loss3 = torch.empty(3, device=poses.device, dtype=torch.float32)
        grad = torch.empty_like(poses)
        lib = _lib.load()
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.odevio_pose_loss(poses.data_ptr(), gts.data_ptr(), poses.numel() // 6, loss3.data_ptr(), grad.data_ptr(), stream))
        ctx.save_for_backward(grad)
        ctx.parts = loss3
        return loss3[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


def pose_loss(poses, gts):
    """The reference's training loss on device tensors [B,P,6]; differentiable w.r.t. ``poses``."""
    if not poses.is_cuda:
        raise RuntimeError("pose_loss runs on the GPU (no CPU path)")
    return _PoseLossFunction.apply(poses, gts)


def pose_net(model, fv, fi, timestamps, hc=None):
    """``model.Pose_net`` forward (PoseODERNN.forward, reference PoseODERNN.py:88-123) WITH an autograd graph: the
    returned ``poses`` / ``h_T`` back-propagate to ``fv``, ``fi``, ``hc`` and the parameters of ``model.Pose_net``."""
    opt = model.opt
    if opt.model_type not in ("ode-rnn", "rnn"):
        raise ValueError("odevio_amd.train.pose_net: model_type must be ode-rnn or rnn")
    model._ensure_plan()
    names = pose_param_names(opt)
    params = dict(model.named_parameters())
    plist = [params[n] for n in names]
    fnames = fuse_param_names(opt)
    fused = _FuseFunction.apply(model, fnames, fv, fi, *[params[n] for n in fnames])
    return _OdeRnnFunction.apply(model, names, fused, timestamps, hc, *plist)


def cde_param_names(opt):
    """Parameters of the Neural-CDE pose net that are on its path (PoseCDE.py:59-72; ``reduction_net`` is constructed but never applied)."""
    names = []
    for l in range(opt.cde_fn_num_layers + 1):
        names += [f"Pose_net.cde_func.net.{2 * l}.weight", f"Pose_net.cde_func.net.{2 * l}.bias"]
    names += ["Pose_net.initial.0.weight", "Pose_net.initial.0.bias"]
    return names + [f"Pose_net.regressor.{i}.{w}" for i in (0, 2) for w in ("weight", "bias")]


class _CdeFunction(torch.autograd.Function):
    """(obs [B,L,1+F], z0_in [B,H] | None, *parameters) -> (poses [B,n_out,6], z0 [B,H]): cdeint + regressor (PoseCDE.py:94-103) with the
    backward of ``odevio_cde_bwd`` (the solve re-run with a tape, then swept in reverse).  ``t_out``: float64 host tensor."""

    @staticmethod
    def forward(ctx, model, names, t_out, obs, z0_in, *params):
        obs = obs.detach().contiguous().float()
        z0d = None if z0_in is None else z0_in.detach().contiguous().float()
        B, L, C = obs.shape
        hc = model.opt.cde_hidden_dim
        n_out = t_out.numel()
        poses = torch.empty(B, n_out, 6, device=obs.device, dtype=torch.float32)
        z0 = torch.empty(B, hc, device=obs.device, dtype=torch.float32)
        stats = (ctypes.c_int32 * 2)()
        _lib.check(model._lib.odevio_cde_fwd(model._plan, obs.data_ptr(), B, L, t_out.data_ptr(), n_out, None if z0d is None else z0d.data_ptr(),
                                             poses.data_ptr(), z0.data_ptr(), ctypes.cast(stats, ctypes.c_void_p), model._stream()))
        ctx.model, ctx.names, ctx.t_out, ctx.has_z0 = model, names, t_out, z0d is not None
        ctx.param_shapes = [tuple(p.shape) for p in params]
        ctx.save_for_backward(obs, *([z0d] if z0d is not None else []))
        return poses, z0

    @staticmethod
    def backward(ctx, g_poses, g_z0):
        model = ctx.model
        saved = ctx.saved_tensors
        obs = saved[0]
        z0d = saved[1] if ctx.has_z0 else None
        B, L, C = obs.shape
        n_out = ctx.t_out.numel()
        g_poses = torch.zeros(B, n_out, 6, device=obs.device) if g_poses is None else g_poses.contiguous().float()
        g_z0 = None if g_z0 is None else g_z0.contiguous().float()
        g_obs = torch.empty_like(obs)
        g_z0_in = torch.empty_like(z0d) if z0d is not None else None
        grads = [torch.empty(s, device=obs.device, dtype=torch.float32) for s in ctx.param_shapes]
        model._ensure_plan()
        _lib.check(model._lib.odevio_cde_bwd(model._plan, obs.data_ptr(), B, L, ctx.t_out.data_ptr(), n_out, None if z0d is None else z0d.data_ptr(),
                                             g_poses.data_ptr(), None if g_z0 is None else g_z0.data_ptr(), g_obs.data_ptr(),
                                             None if g_z0_in is None else g_z0_in.data_ptr(), _tensor_array(ctx.names, grads), len(grads), None,
                                             model._stream()))
        return (None, None, None, g_obs, g_z0_in, *grads)


def pose_cde(model, fv, fi, timestamps, prev=None):
    """``model.Pose_net`` forward for ``model_type cde`` (PoseCDE.forward, reference PoseCDE.py:76-103) in TRAINING mode WITH an autograd
    graph: relative timestamps, no window history (:81-92), gradients to ``fv``, ``fi``, ``prev`` and every parameter on the path."""
    opt = model.opt
    if opt.model_type != "cde":
        raise ValueError("odevio_amd.train.pose_cde: model_type must be cde")
    if not model.training:
        raise RuntimeError("odevio_amd.train.pose_cde: training mode only (eval mode accumulates a window history and carries no graph)")
    model._ensure_plan()
    params = dict(model.named_parameters())
    fnames = fuse_param_names(opt)
    fused = _FuseFunction.apply(model, fnames, fv, fi, *[params[n] for n in fnames])
    ts = timestamps.detach().to(torch.float32)
    tsd = ts - ts[:, :1]
    obs = torch.cat([tsd[:, 1:, None].to(fused.device), fused], dim=-1)        # (plumbing: the concat's backward is a slice)
    t_out = tsd[0, 1:].double().cpu().contiguous()
    model.Pose_net.history = None
    names = cde_param_names(opt)
    return _CdeFunction.apply(model, names, t_out, obs, prev, *[params[n] for n in names])


class PoseNetTrainer:
    """The reference's optimizer step for ``Pose_net`` on the device (scripts/train_model.py:48-95, utils/utils.py:115-130).

    The reference builds ``torch.optim.Adam(betas=(0.9, 0.999), eps=1e-8, weight_decay=args.weight_decay)`` over
    ``Pose_net``'s parameters only, and per batch runs forward -> ``100 * MSE(angles) + MSE(translations)`` ->
    ``backward`` -> ``clip_grad_norm_(max_norm=args.gradient_clip)`` -> ``optimizer.step()`` -> ``zero_grad()``.  Here every
    one of those stages is a libodevio call; PyTorch carries the graph between them.  ``step`` / ``accumulate`` take the encoder
    FEATURES, or the raw ``imu`` / ``img`` - then the encoders are part of the graph (train-mode BatchNorm and Dropout when the model is
    in ``train()``; ``Image_net`` only when ``freeze_encoder`` is off): their gradients count in the clipping norm as in the
    reference, whose optimizer nevertheless never holds them.  The optimizer step itself is ONE call over all tensors
    (``odevio_optimizer_step``), the plan's copies of the parameters are refreshed in place (``odevio_plan_update``).
    """

    def __init__(self, model, lr=None, betas=(0.9, 0.999), eps=1e-8, weight_decay=None, gradient_clip=None, process_group=None,
                 optimizer=None):
        opt = model.opt
        # --freeze_encoder (scripts/train_model.py:191-194: requires_grad = False on Image_net; the recipe's setting): Image_net runs
        # without a graph.  Otherwise its gradients exist and count in clip_grad_norm_(model.parameters()) (:84) although the optimizer
        # never holds them (utils/utils.py:116-119) - reproduced through odevio_amd.train.image_encoder
        self.freeze_encoder = bool(getattr(opt, "freeze_encoder", False))
        for q in model.Image_net.parameters():
            q.requires_grad_(not self.freeze_encoder)
        self.optimizer = str(getattr(opt, "optimizer", "Adam") if optimizer is None else optimizer)
        if self.optimizer not in ("Adam", "SGD"):
            raise ValueError(f"optimizer {self.optimizer!r} not supported: Adam or SGD (utils/utils.py:120-129)")
        # data parallel over the GPUs of a node (one process per GPU): every rank steps on its own sequences, the loss is scaled by
        # 1 / world_size and the gradients are summed in ONE RCCL all-reduce before clip + Adam (odevio_amd.dist.allreduce_gradients)
        self.process_group = process_group
        if opt.model_type not in ("ode-rnn", "rnn"):
            raise ValueError("PoseNetTrainer: model_type must be ode-rnn or rnn")
        self.model = model
        # Two parameter groups like utils/utils.py:116-119: group 0 = Pose_net without the regressor, group 1 = the regressor.
        # The reference's epoch loop re-assigns the learning rate of group 0 ONLY (train_model.py:211-216, the line for group 1
        # is commented out), so the regressor keeps lr_warmup for the whole run; set_epoch() reproduces that.
        self.lr = float(opt.lr_warmup if lr is None else lr)
        self.lr_regressor = self.lr
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        self.weight_decay = float(opt.weight_decay if weight_decay is None else weight_decay)
        self.gradient_clip = float(opt.gradient_clip if gradient_clip is None else gradient_clip)
        self.names = fuse_param_names(opt) + pose_param_names(opt)
        params = dict(model.named_parameters())
        self.params = [params[n] for n in self.names]
        self._name_set = set(self.names)
        for p in self.params:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise RuntimeError("PoseNetTrainer: parameters must be contiguous fp32 tensors on the GPU (model.cuda())")
        self.exp_avg = [torch.zeros_like(p) for p in self.params]          # SGD: the momentum buffers
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params] if self.optimizer == "Adam" else []
        if self._world() > 1:
            # every rank draws its OWN dropout / Gumbel noise: fold the rank into the Philox key (same seed, different streams)
            import torch.distributed as tdist
            seed, _ = model.rng_state()
            model.set_seed((seed + 0x9E3779B97F4A7C15 * (tdist.get_rank(process_group) + 1)) & 0xFFFFFFFFFFFFFFFF)
        self.norm_coef = torch.zeros(2, device=self.params[0].device, dtype=torch.float32)   # {total grad norm, clip factor}
        self.steps = 0
        # what never changes between steps (parameters and optimizer state are updated IN PLACE): the tensor tables of the one-call
        # optimizer step and of the plan refresh, and the list of every parameter of the model (the clipping norm looks at all of them)
        self._param_table = _tensor_array(self.names, [p.detach() for p in self.params])
        self._state1_table = _tensor_array(self.names, self.exp_avg)
        self._state2_table = _tensor_array(self.names, self.exp_avg_sq) if self.exp_avg_sq else None
        self._param_ptrs = [p.data_ptr() for p in self.params]
        self._all_named = [(n, p) for n, p in model.named_parameters() if n not in self._name_set]

    def set_epoch(self, ep):
        """The reference's per-epoch schedule (train_model.py:25-35, 211-215): lr_warmup for the first epochs_warmup epochs,
        lr_joint for the next epochs_joint, lr_fine afterwards - applied to parameter group 0 only, as the reference does."""
        o = self.model.opt
        if ep < o.epochs_warmup:
            self.lr = float(o.lr_warmup)
        elif ep < o.epochs_warmup + o.epochs_joint:
            self.lr = float(o.lr_joint)
        else:
            self.lr = float(o.lr_fine)
        return self.lr

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def apply_gradients(self):
        """clip_grad_norm_ + Adam on the gradients in ``param.grad`` + the plan's copies refreshed (device re-layout kernels);
        nothing here synchronises with the host."""
        model = self.model
        lib = model._lib
        grads = []
        for n, p in zip(self.names, self.params):
            if p.grad is None:
                raise RuntimeError(f"PoseNetTrainer: no gradient for {n}")
            grads.append(p.grad.contiguous().float())
        stream = model._stream()
        model._ensure_plan()
        extra_pairs = [(n, p) for n, p in self._all_named if p.grad is not None]
        if self._world() > 1:
            from . import dist as _dist
            for _, p in extra_pairs:
                p.grad = p.grad.contiguous().float()
            _dist.allreduce_gradients(grads + [p.grad for _, p in extra_pairs], self.process_group)
        # clip_grad_norm_(model.parameters()): every parameter that has a gradient counts in the norm - with the recipe's frozen
        # Image_net that is Pose_net (updated below) and Inertial_net (in the norm only: the reference's optimizer does not hold it)
        extra_names = [n for n, _ in extra_pairs]
        extra = [p.grad.contiguous().float() for _, p in extra_pairs]
        if not self.gradient_clip:
            # the reference only steps inside `if args.gradient_clip:` (scripts/train_model.py:83-85): a falsy clip value means the
            # gradients are dropped by the zero_grad that follows and NO update happens - reproduced, not "fixed"
            return False
        _lib.check(lib.odevio_grad_clip(model._plan, _tensor_array(self.names + extra_names, grads + extra), len(grads) + len(extra),
                                        self.gradient_clip, self.norm_coef.data_ptr(), stream))
        self.steps += 1
        # optimizer.step() over every Pose_net tensor in one call (two learning-rate groups, utils/utils.py:116-119)
        lrs = (ctypes.c_float * len(grads))(*[self.lr_regressor if n.startswith("Pose_net.regressor.") else self.lr for n in self.names])
        if [p.data_ptr() for p in self.params] != self._param_ptrs:   # (somebody replaced a parameter's storage: rebuild the table)
            self._param_table = _tensor_array(self.names, [p.detach() for p in self.params])
            self._param_ptrs = [p.data_ptr() for p in self.params]
        pa, s1 = self._param_table, self._state1_table
        ga = _tensor_array(self.names, grads)
        if self.optimizer == "Adam":
            _lib.check(lib.odevio_optimizer_step(0, pa, ga, s1, self._state2_table, lrs, len(grads), self.betas[0], self.betas[1],
                                                 self.eps, self.weight_decay, self.steps, self.norm_coef.data_ptr(), stream))
        else:   # torch.optim.SGD(param_groups, lr=1e-4, momentum=0.9): the groups' lr_warmup overrides 1e-4, no weight decay
            _lib.check(lib.odevio_optimizer_step(1, pa, ga, s1, None, lrs, len(grads), 0.9, 0.0, 0.0, 0.0, self.steps, self.norm_coef.data_ptr(), stream))
        # the kernels read their own layouts of these parameters (column shards, transposes): refresh them in place
        rc = lib.odevio_plan_update(model._plan, pa, len(self.params), stream)
        if rc == _lib.ERR_UNSUPPORTED:      # widths the integrator pads (F = 400, H = 200): no in-place re-layout - rebuild the plan at the next forward
            model._plan_sig = None
            return True
        _lib.check(rc)
        model._plan_sig = model._signature()
        return True

    def _world(self):
        import torch.distributed as tdist
        return tdist.get_world_size(self.process_group) if (tdist.is_available() and tdist.is_initialized()) else 1

    def zero_all_grads(self):
        for p in self.model.parameters():
            p.grad = None

    def accumulate(self, fv, fi, timestamps, gts, hc=None, imu=None, img=None):
        """forward -> loss -> backward of one batch, ADDING to the gradients already in ``param.grad`` (the reference's loop calls
        ``loss.backward()`` every batch and steps every ``grad_accumulation_steps``-th, scripts/train_model.py:78-86).  ``img`` instead
        of ``fv``: the image encoder is part of the graph (train mode; ``freeze_encoder`` off)."""
        if img is not None:
            if fv is not None:
                raise ValueError("PoseNetTrainer.accumulate: pass fv or img, not both")
            if self.freeze_encoder:
                with torch.no_grad():
                    fv = self.model.image_encoder(img)
            else:
                fv = image_encoder(self.model, img)
        if imu is not None:
            if fi is not None:
                raise ValueError("PoseNetTrainer.accumulate: pass fi or imu, not both")
            fi = imu_encoder(self.model, imu)
        poses, h_T = pose_net(self.model, fv, fi, timestamps, hc)
        loss = pose_loss(poses, gts)
        world = self._world()
        (loss if world == 1 else loss * (1.0 / world)).backward()
        return loss.detach(), poses.detach(), h_T.detach()

    def step(self, fv, fi, timestamps, gts, hc=None, imu=None):
        """One training step on a batch of features; returns (loss, poses, h_T) - ``loss`` a device scalar
        (``float(loss)`` synchronises, like the reference's ``pose_loss.item()``).  With ``imu`` [B,T,6] instead of ``fi``
        the inertial encoder is part of the graph: its gradients enter the clipping norm as in the reference's step."""
        self.zero_all_grads()
        out = self.accumulate(fv, fi, timestamps, gts, hc=hc, imu=imu)
        self.apply_gradients()
        self.zero_all_grads()
        return out

    @property
    def grad_norm(self):
        """Total gradient norm of the last step (device scalar)."""
        return self.norm_coef[0]


def train_epoch(model, trainer, loader, log=None, log_every=20):
    """One epoch of the reference's ``train()`` (scripts/train_model.py:48-95) on libodevio: ``loader`` yields
    ``(imgs [B,S,3,H,W], imus [B,10(S-1)+1,6], gts [B,S-1,6], timestamps [B,S], folder)`` like the reference's ``DataLoader``.
    Like the reference's epoch loop (:219) the model is put in ``train()``: both encoders run with batch-statistics BatchNorm and
    Dropout (``Image_net`` under ``no_grad`` when ``opt.freeze_encoder``, else inside the graph so that its gradients count in the
    clipping norm; the inertial encoder always inside the graph); gradients accumulate over
    ``opt.grad_accumulation_steps`` batches (and the last batch) before clip + step (:82-86); ``opt.optimizer`` picks Adam or SGD.
    Returns the mean pose loss like the reference.  ``log(message)`` receives the reference's per-iteration line."""
    losses = []
    model.train()
    accum = max(1, int(getattr(model.opt, "grad_accumulation_steps", 1)))
    batches = list(loader) if not hasattr(loader, "__len__") else loader
    n = len(batches)
    trainer.zero_all_grads()
    for i, (imgs, imus, gts, timestamps, _folder) in enumerate(batches):
        dev = next(model.parameters()).device
        imgs, imus = imgs.to(dev).float(), imus.to(dev).float()
        gts, timestamps = gts.to(dev).float(), timestamps.to(dev).float()
        loss, _, _ = trainer.accumulate(None, None, timestamps, gts, imu=imus, img=imgs)   # Image_net: no graph when frozen, else its backward
        if (i + 1) % accum == 0 or (i + 1) == n:
            trainer.apply_gradients()
            trainer.zero_all_grads()
        if log is not None and i % log_every == 0:
            log(f"iters: {i + 1}/{n}, pose loss: {float(loss):.6f}, grad norm: {float(trainer.grad_norm):.4f}")
        losses.append(loss)
    return float(torch.stack(losses).mean()) if losses else float("nan")
