"""Host-side mirror of the reference's model surface for the hot path.

``DeepVIO(opt).forward(img, imu, timestamps, hc=None) -> (poses, h_T)`` has the signature, tensor
layouts, attribute names (``Image_net``, ``Inertial_net``, ``Pose_net``, ``opt``) and ``state_dict``
keys of the reference class (reference src/models/DeepVIO.py:37-68), so the reference's callers
(scripts/train_model.py:69 in eval, src/data/KITTI_eval.py:141) can use it unchanged and reference
checkpoints load with ``load_state_dict``.  The sub-modules are *parameter containers only*: all
arithmetic runs in libodevio.so (hand-written HIP for gfx950) through the C ABI of
``include/odevio.h``.  There is no PyTorch compute path and no CPU path: without the library, or
on CPU tensors, ``forward`` raises.

Error behaviour follows the reference: ``ValueError`` for an unknown solver / RNN type /
activation (PoseODERNN.py:136,146; ODEFunc.py:34), ``NotImplementedError`` for ``ltc``
(DeepVIO.py:59); an unknown ``model_type`` raises ``ValueError`` instead of silently leaving
``Pose_net = None``.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib, weights


def resize_frames(frames, out_h, out_w):
    """uint8 HWC frames [..., H, W, 3] on the device -> [..., out_h, out_w, 3], bit-identical to the reference loader's
    ``TF.resize`` of the PIL image (reference src/data/KITTI_eval.py:101; PIL BILINEAR with antialiasing)."""
    if frames.dtype != torch.uint8 or not frames.is_cuda or frames.shape[-1] != 3:
        raise ValueError("resize_frames takes uint8 HWC frames on the device")
    lib = _lib.load()
    frames = frames.contiguous()
    lead, (h, w) = frames.shape[:-3], frames.shape[-3:-1]
    n = 1
    for d in lead:
        n *= d
    out = torch.empty(*lead, out_h, out_w, 3, device=frames.device, dtype=torch.uint8)
    tmp = torch.empty(n * h * out_w * 3, device=frames.device, dtype=torch.uint8) if (h != out_h and w != out_w) else None
    with torch.cuda.device(frames.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.odevio_resize_u8(frames.data_ptr(), n, h, w, out.data_ptr(), out_h, out_w, None if tmp is None else tmp.data_ptr(), stream))
    return out


def _conv_block(cin, cout, k, stride):
    # same child indices as the reference's conv() (Encoder.py:8-22): 0 = Conv2d(no bias), 1 = BatchNorm2d
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, bias=False), nn.BatchNorm2d(cout),
                         nn.LeakyReLU(0.1), nn.Dropout(0.0))


class _ImageNet(nn.Module):
    def __init__(self, opt):
        super().__init__()
        for name, cin, cout, k, s in weights.IMAGE_CONVS:
            setattr(self, name, _conv_block(cin, cout, k, s))
        oh, ow = weights.encoder_out_hw(opt.img_h, opt.img_w)
        self.visual_head = nn.Linear(1024 * oh * ow, opt.v_f_len)


class _InertialNet(nn.Module):
    def __init__(self, opt):
        super().__init__()
        layers = []
        for _, cin, cout in weights.IMU_CONVS:
            layers += [nn.Conv1d(cin, cout, 3, padding=1), nn.BatchNorm1d(cout), nn.LeakyReLU(0.1), nn.Dropout(0.0)]
        self.encoder_conv = nn.Sequential(*layers)
        self.proj = nn.Linear(256 * weights.IMU_WINDOW, opt.i_f_len)


class _Fuse(nn.Module):
    def __init__(self, f_len, method):
        super().__init__()
        if method == "soft":
            self.net = nn.Sequential(nn.Linear(f_len, f_len))
        elif method == "hard":
            self.net = nn.Sequential(nn.Linear(f_len, 2 * f_len))


class _OdeFunc(nn.Module):
    def __init__(self, f_len, hidden, n_hidden, activation):
        super().__init__()
        if activation not in _lib.ACTIVATIONS:
            raise ValueError(f"Activation function {activation} not supported")
        dims = [f_len] + [hidden] * n_hidden + [f_len]
        layers = []
        for i in range(n_hidden + 1):
            layers += [nn.Linear(dims[i], dims[i + 1]), nn.Identity()]  # odd slots: activations (no parameters)
        self.net = nn.Sequential(*layers)


class _PoseNet(nn.Module):
    """Parameter container for PoseODERNN / PoseRNN (reference PoseODERNN.py:39-68, PoseRNN.py:38-51)."""

    def __init__(self, opt, with_ode):
        super().__init__()
        self.f_len = opt.v_f_len + opt.i_f_len
        if with_ode:
            if opt.ode_solver not in _lib.SOLVERS:
                raise ValueError(f"Solver {opt.ode_solver} not supported")
            self.ode_func = _OdeFunc(self.f_len, opt.ode_hidden_dim, opt.ode_fn_num_layers, opt.ode_activation_fn)
        if opt.ode_rnn_type == "rnn":
            self.rnn = nn.RNN(self.f_len, self.f_len, opt.rnn_num_layers, batch_first=True)
        elif opt.ode_rnn_type == "gru":
            self.rnn = nn.GRU(self.f_len, self.f_len, opt.rnn_num_layers, batch_first=True)
        else:
            raise ValueError(f"RNN type {opt.ode_rnn_type} not supported")
        self.fuse = _Fuse(self.f_len, opt.fuse_method)
        self.regressor = nn.Sequential(nn.Linear(self.f_len, 128), nn.LeakyReLU(0.1), nn.Linear(128, 6))

    def get_regressor_params(self):
        return self.regressor.parameters()

    def get_other_params(self):
        return [p for n, p in self.named_parameters() if not n.startswith("regressor")]


class _PoseCDENet(nn.Module):
    """Parameter container + window history for PoseCDE (reference PoseCDE.py:41-73)."""

    def __init__(self, opt):
        super().__init__()
        self.f_len = opt.v_f_len + opt.i_f_len
        hc = opt.cde_hidden_dim
        if opt.cde_activation_fn not in _lib.ACTIVATIONS:
            raise ValueError(f"Activation function {opt.cde_activation_fn} not supported")
        if opt.cde_solver not in ("dopri5", "rk4", "runge_kutta", "euler"):
            raise ValueError(f"Solver {opt.cde_solver} not supported")
        self.fuse = _Fuse(self.f_len, opt.fuse_method)
        # constructed but never applied by the reference (PoseCDE.py:53-58); kept for state_dict compatibility
        self.reduction_net = nn.Sequential(nn.Linear(self.f_len, self.f_len // 2), nn.LeakyReLU(0.1), nn.Linear(self.f_len // 2, hc))
        self.initial = nn.Sequential(nn.Linear(hc + 1, hc), nn.Tanh())
        layers = []
        dims = [hc] * (opt.cde_fn_num_layers + 1) + [hc * (hc + 1)]
        for i in range(opt.cde_fn_num_layers + 1):
            layers += [nn.Linear(dims[i], dims[i + 1]), nn.Identity()]
        self.cde_func = nn.Module()
        self.cde_func.net = nn.Sequential(*layers)
        self.regressor = nn.Sequential(nn.Linear(hc, 128), nn.LeakyReLU(0.1), nn.Linear(128, 6))
        self.history = None  # eval-mode observations of the windows so far (PoseCDE.py:88-92)

    def get_reduction_net_params(self):
        return self.reduction_net.parameters()

    def get_regressor_params(self):
        return self.regressor.parameters()

    def get_other_params(self):
        return [p for n, p in self.named_parameters() if not n.startswith("regressor")]


class DeepVIO(nn.Module):
    def __init__(self, opt, seed=None, state_dict=None):
        super().__init__()
        if opt.model_type == "ltc":
            raise NotImplementedError("LTC model not implemented yet")
        if opt.model_type == "rde":
            raise NotImplementedError("model_type 'rde': PoseRDE is experimental and broken upstream (out of scope, SURVEY.md section 2)")
        if opt.model_type not in ("ode-rnn", "rnn", "cde"):
            raise ValueError(f"model_type {opt.model_type!r} not supported")
        if opt.fuse_method not in ("cat", "soft", "hard"):
            raise ValueError(f"fuse_method {opt.fuse_method!r} not supported")
        if getattr(opt, "dtype", "fp32") not in _lib.DTYPES:
            raise ValueError(f"--dtype {opt.dtype!r} not supported: one of {sorted(_lib.DTYPES)} (only fp32 / fp32_mfma carry the 1e-4 parity claim)")
        self.opt = opt
        self.Image_net = _ImageNet(opt)
        self.Inertial_net = _InertialNet(opt)
        if opt.model_type == "cde":
            self.Pose_net = _PoseCDENet(opt)
        else:
            self.Pose_net = _PoseNet(opt, with_ode=(opt.model_type == "ode-rnn"))
        self._plan = None
        self._plan_sig = None
        self._rng = None
        self._bn_dirty = False   # a train-mode forward moved the BatchNorm running statistics the plan's eval-mode constants were folded from
        self._warned_train = False
        self._lib = _lib.load()  # raises if the HIP library is missing: no silent fallback
        # the reference constructor leaves a random model behind (DeepVIO.py:43); ours is seeded
        # (or carries `state_dict`, which saves drawing weights that a caller would overwrite at once)
        sd = state_dict if state_dict is not None else weights.make_state_dict(opt, seed=getattr(opt, "seed", 0) if seed is None else seed)
        self.load_state_dict(sd, strict=True)
        self.eval()

    # ------------------------------------------------------------------ plan management
    def load_state_dict(self, state_dict, strict=True, **kw):
        """Accepts reference checkpoints: drops the ``Pose_net.solver.*`` aliases and a ``module.`` prefix."""
        self._plan_sig = None
        return super().load_state_dict(weights.filter_reference_state_dict(state_dict), strict=strict, **kw)

    def _signature(self):
        sig = []
        for t in list(self.parameters()) + list(self.buffers()):
            sig.append((t.data_ptr(), t._version))
        return tuple(sig)

    def _refresh_pose_net(self, old_sig, sig):
        """True when the ONLY difference between two signatures is the version of Pose_net parameters (an in-place optimizer step
        of torch.optim on them - what the reference's loop does, scripts/train_model.py:85) and the plan could be refreshed in place
        (``odevio_plan_update``: device re-layout kernels, no host round trip) instead of being rebuilt."""
        if old_sig is None or len(old_sig) != len(sig) or self.opt.model_type == "cde":
            return False
        named = list(self.named_parameters()) + list(self.named_buffers())
        for (name, _), a, b in zip(named, old_sig, sig):
            if a == b:
                continue
            if a[0] != b[0] or not name.startswith("Pose_net."):
                return False                       # a tensor moved, or an encoder weight changed: the plan folds those at creation
        from .train import fuse_param_names, pose_param_names, _tensor_array
        names = fuse_param_names(self.opt) + pose_param_names(self.opt)
        params = dict(self.named_parameters())
        tensors = [params[n].detach() for n in names]
        if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in tensors):
            return False
        rc = self._lib.odevio_plan_update(self._plan, _tensor_array(names, tensors), len(tensors), self._stream())
        return rc == _lib.ODEVIO_OK

    def _config(self):
        o = self.opt
        c = _lib.OdevioConfig()
        c.struct_size = ctypes.sizeof(_lib.OdevioConfig)
        c.model_type = _lib.MODEL_TYPES[o.model_type]
        c.img_h, c.img_w, c.v_f_len, c.i_f_len = o.img_h, o.img_w, o.v_f_len, o.i_f_len
        c.fuse_method = _lib.FUSE_METHODS[o.fuse_method]
        c.ode_hidden_dim, c.ode_fn_num_layers = o.ode_hidden_dim, o.ode_fn_num_layers
        c.ode_activation = _lib.ACTIVATIONS[o.ode_activation_fn]
        c.ode_solver = _lib.SOLVERS[o.ode_solver]
        c.ode_substeps = getattr(o, "ode_substeps", 1)
        c.rnn_type = _lib.RNN_TYPES[o.ode_rnn_type]
        c.rnn_num_layers = o.rnn_num_layers
        # torchode IntegralController(atol=1e-6, rtol=1e-2), dt0 = 1e-4 (PoseODERNN.py:57,72)
        c.atol, c.rtol, c.dt0 = getattr(o, "ode_atol", 1e-6), getattr(o, "ode_rtol", 1e-2), getattr(o, "ode_dt0", 1e-4)
        c.max_steps = getattr(o, "ode_max_steps", 200000)
        c.cde_hidden_dim, c.cde_fn_num_layers = o.cde_hidden_dim, o.cde_fn_num_layers
        c.cde_activation = _lib.ACTIVATIONS.get(o.cde_activation_fn, 0)
        c.cde_solver = _lib.SOLVERS.get(o.cde_solver, 0)
        c.arith = _lib.DTYPES[getattr(o, "dtype", "fp32")]
        return c

    def _ensure_plan(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("odevio_amd.DeepVIO runs on an MI355X only: move the model with .cuda() (no CPU path)")
        sig = self._signature()
        stale_bn = self._bn_dirty and not self.training
        if self._plan is not None and sig == self._plan_sig and not stale_bn:
            return
        if self._plan is not None and not stale_bn and self._refresh_pose_net(self._plan_sig, sig):
            self._plan_sig = sig
            return
        self._bn_dirty = False
        # the random stream (hard fusion's Gumbel noise, train-mode dropout) belongs to the MODEL, not to one plan: a rebuild after
        # load_state_dict / .cuda() / a torch optimizer step must go on drawing fresh noise, not replay the stream from draw 0
        rng = self._rng_of_plan()
        self._destroy_plan()
        sd = {k: v for k, v in self.state_dict().items() if v.is_floating_point()}
        keep = []  # keep contiguous fp32 views alive during the call
        arr = (_lib.OdevioTensor * len(sd))()
        for i, (k, v) in enumerate(sd.items()):
            t = v.detach().to(torch.float32).contiguous()
            keep.append(t)
            arr[i].name = k.encode()
            arr[i].data = t.data_ptr()
            arr[i].numel = t.numel()
        plan = ctypes.c_void_p()
        cfg = self._config()
        with torch.cuda.device(dev):
            torch.cuda.current_stream().synchronize()
            _lib.check(self._lib.odevio_plan_create(ctypes.byref(cfg), arr, len(sd), self._stream(), ctypes.byref(plan)))
        self._plan, self._plan_sig = plan, sig
        if rng is not None:
            _lib.check(self._lib.odevio_set_rng_state(self._plan, rng[0], rng[1]))
            self._rng = rng

    def _rng_of_plan(self):
        if getattr(self, "_plan", None) is None:
            return getattr(self, "_rng", None)
        seed, calls = ctypes.c_uint64(), ctypes.c_uint64()
        _lib.check(self._lib.odevio_rng_state(self._plan, ctypes.byref(seed), ctypes.byref(calls)))
        return int(seed.value), int(calls.value)

    def _destroy_plan(self):
        if getattr(self, "_plan", None) is not None:
            try:
                self._rng = self._rng_of_plan()   # survives an explicit destroy too
            except Exception:
                pass
            self._lib.odevio_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self._destroy_plan()
        except Exception:
            pass

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _dev(t, name):
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a device tensor (no CPU path)")
        return t.detach().to(torch.float32).contiguous()

    # ------------------------------------------------------------------ the reference surface
    def forward(self, img, imu, timestamps, hc=None):
        """img [B,S,3,H,W], imu [B,10(S-1)+1(+tail),6], timestamps [B,S], hc None | [L,B,F] -> (poses [B,S-1,6], h_T [L,B,F])."""
        self._ensure_plan()
        if self.training:
            # model.train() (scripts/train_model.py:69,219): both encoders normalise with BATCH statistics, move their running
            # statistics and apply Dropout - computed here as the reference computes it.  With gradients enabled the result carries
            # an autograd graph whose nodes run in libodevio (odevio_amd.train), so the reference's own loop - loss.backward(),
            # clip_grad_norm_(model.parameters()), a torch.optim step on Pose_net - works on this model unchanged.
            if img.dtype == torch.uint8:
                raise ValueError("train mode takes the loader's float frames [B,S,3,H,W] (uint8 frames: eval mode)")
            if torch.is_grad_enabled():
                from . import train as _train
                if any(q.requires_grad for q in self.Image_net.parameters()):
                    fv = _train.image_encoder(self, img)
                else:                               # --freeze_encoder (scripts/train_model.py:191-194): no graph through Image_net
                    with torch.no_grad():
                        fv = self.image_encoder(img)
                fi = _train.imu_encoder(self, imu) if any(q.requires_grad for q in self.Inertial_net.parameters()) else self.imu_encoder(imu)
                if self.opt.model_type == "cde":
                    if getattr(self.opt, "dtype", "fp32") in ("fp16", "bf16"):
                        raise ValueError("training the Neural-CDE path needs --dtype fp32 (the bf16 weight stream has no backward)")
                    return _train.pose_cde(self, fv, fi, timestamps, hc)
                return _train.pose_net(self, fv, fi, timestamps, hc)
            fv, fi = self.image_encoder(img), self.imu_encoder(imu)
            if self.opt.model_type == "cde":
                return self.pose_cde(fv, fi, timestamps, hc)
            return self.pose_net(fv, fi, timestamps, hc)
        if img.dtype == torch.uint8 and self.opt.model_type == "cde":
            raise ValueError("uint8 frames are supported for model_type ode-rnn / rnn")
        if self.opt.model_type == "cde":
            return self.pose_cde(self.image_encoder(img), self.imu_encoder(imu), timestamps, hc)
        u8 = img.dtype == torch.uint8
        if u8:
            # the loader's frames before ToTensor() - 0.5 (reference src/data/KITTI_eval.py:97-110): [B,S,H,W,3] uint8;
            # the normalisation is fused into the encoder's ingest pass (odevio_forward_u8)
            if not img.is_cuda:
                raise RuntimeError("img must be a device tensor (no CPU path)")
            if img.dim() != 5 or img.shape[-1] != 3:
                raise ValueError(f"uint8 frames must be [B,S,H,W,3] (HWC), got {tuple(img.shape)}")
            if tuple(img.shape[2:4]) != (self.opt.img_h, self.opt.img_w):
                # camera-sized frames: the loader's TF.resize (KITTI_eval.py:101) runs on the device, PIL-exact
                img = resize_frames(img.detach(), self.opt.img_h, self.opt.img_w)
            img = img.detach().contiguous()
        else:
            img = self._dev(img, "img")
        imu, ts = self._dev(imu, "imu"), self._dev(timestamps, "timestamps")
        B, S = img.shape[0], img.shape[1]
        L, F = self.opt.rnn_num_layers, self.opt.v_f_len + self.opt.i_f_len
        hcp = None
        if hc is not None:
            hc = self._dev(hc, "hc")
            if tuple(hc.shape) != (L, B, F):
                raise ValueError(f"hc must be [{L},{B},{F}], got {tuple(hc.shape)}")
            hcp = hc.data_ptr()
        poses = torch.empty(B, S - 1, 6, device=img.device, dtype=torch.float32)
        h_T = torch.empty(L, B, F, device=img.device, dtype=torch.float32)
        with torch.cuda.device(img.device):
            fwd = self._lib.odevio_forward_u8 if u8 else self._lib.odevio_forward
            _lib.check(fwd(self._plan, img.data_ptr(), imu.data_ptr(), imu.shape[1], ts.data_ptr(),
                           hcp, B, S, poses.data_ptr(), h_T.data_ptr(), None, self._stream()))
        return poses, h_T

    # ------------------------------------------------------------------ component entry points (tests, bench)
    def _bn_buffers(self, net, prefix):
        """(names, tensors) of the running statistics of `net`'s BatchNorm layers (updated in place by the train-mode kernels) and
        the list of their num_batches_tracked counters."""
        names, tensors, counters = [], [], []
        for k, b in net.named_buffers():
            if k.endswith("running_mean") or k.endswith("running_var"):
                if not (b.is_cuda and b.dtype == torch.float32 and b.is_contiguous()):
                    raise RuntimeError("BatchNorm buffers must be contiguous fp32 tensors on the GPU (model.cuda())")
                names.append(prefix + k)
                tensors.append(b)
            elif k.endswith("num_batches_tracked"):
                counters.append(b)
        return names, tensors, counters

    def _after_train_forward(self, counters):
        for c in counters:
            c += 1                              # what nn.BatchNorm does in train mode (the kernels moved mean / var in place)
        self._plan_sig = self._signature()      # the plan's train-mode inputs are these very buffers: nothing to rebuild ...
        self._bn_dirty = True                   # ... until an eval-mode forward needs the running statistics folded again

    def image_encoder(self, img, keep=False):
        """ImageEncoder.forward (Encoder.py:97-122).  In ``train()`` mode: BatchNorm with batch statistics, the modules' running
        statistics updated in place, Dropout(0.2 / 0.5) with masks from the model's random stream (``set_seed`` / ``rng_state``);
        ``keep=True`` leaves what ``odevio_image_encoder_bwd`` needs in the plan (``odevio_amd.train.image_encoder`` uses it)."""
        self._ensure_plan()
        img = self._dev(img, "img")
        B, S = img.shape[0], img.shape[1]
        fv = torch.empty(B, S - 1, self.opt.v_f_len, device=img.device, dtype=torch.float32)
        if self.training:
            names, tensors, counters = self._bn_buffers(self.Image_net, "Image_net.")
            arr = (_lib.OdevioTensor * len(names))()
            for i, (n, t) in enumerate(zip(names, tensors)):
                arr[i].name, arr[i].data, arr[i].numel = n.encode(), t.data_ptr(), t.numel()
            _lib.check(self._lib.odevio_image_encoder_fwd_train(self._plan, img.data_ptr(), B, S, fv.data_ptr(), self.opt.v_f_len, arr,
                                                                len(names), 1 if keep else 0, self._stream()))
            self._after_train_forward(counters)
            return fv
        _lib.check(self._lib.odevio_image_encoder_fwd(self._plan, img.data_ptr(), B, S, fv.data_ptr(), self.opt.v_f_len,
                                                      self._stream()))
        return fv

    def conv_block(self, layer, x, B, S):
        """One conv block (kernel-level parity): layer 0 takes img [B,S,3,H,W]; others take NHWC activations."""
        self._ensure_plan()
        x = self._dev(x, "x")
        h, w = self.opt.img_h, self.opt.img_w
        for _, _, cout, k, s in weights.IMAGE_CONVS[:layer + 1]:
            h, w = weights.conv_out(h, k, s), weights.conv_out(w, k, s)
        out = torch.empty(B * (S - 1), h, w, cout, device=x.device, dtype=torch.float32)
        _lib.check(self._lib.odevio_conv_block_fwd(self._plan, layer, x.data_ptr(), B, S, out.data_ptr(), self._stream()))
        return out

    def imu_encoder(self, imu):
        """InertialEncoder.forward (Encoder.py:60-74); ``train()`` mode as in ``image_encoder`` with Dropout(opt.imu_dropout)."""
        self._ensure_plan()
        imu = self._dev(imu, "imu")
        B, T = imu.shape[0], imu.shape[1]
        fi = torch.empty(B, (T - 1) // 10, self.opt.i_f_len, device=imu.device, dtype=torch.float32)
        if self.training:
            names, tensors, counters = self._bn_buffers(self.Inertial_net, "Inertial_net.")
            arr = (_lib.OdevioTensor * len(names))()
            for i, (n, t) in enumerate(zip(names, tensors)):
                arr[i].name, arr[i].data, arr[i].numel = n.encode(), t.data_ptr(), t.numel()
            _lib.check(self._lib.odevio_imu_encoder_fwd_train(self._plan, imu.data_ptr(), B, T, float(self.opt.imu_dropout), arr, len(names),
                                                              fi.data_ptr(), self.opt.i_f_len, self._stream()))
            self._after_train_forward(counters)
            return fi
        _lib.check(self._lib.odevio_imu_encoder_fwd(self._plan, imu.data_ptr(), B, T, fi.data_ptr(), self.opt.i_f_len,
                                                    self._stream()))
        return fi

    def dropout_mask(self, seed, call, p, shape):
        """Test hook: the factor (0 or 1 / (1 - p)) the train-mode kernels' Dropout(p) applies under draw ``call`` of ``seed``, for a
        tensor of ``shape`` in the kernels' element order (image encoder: NHWC; inertial encoder: [pair, channel, time])."""
        n = 1
        for d in shape:
            n *= int(d)
        out = torch.empty(n, device=next(self.parameters()).device, dtype=torch.float32)
        _lib.check(self._lib.odevio_debug_dropout(int(seed), int(call), float(p), n, out.data_ptr(), self._stream()))
        return out.view(*shape)

    def set_seed(self, seed):
        """Seed of the plan's random stream (fuse_method "hard" draws its Gumbel noise from it): same seed, same masks."""
        self._ensure_plan()
        _lib.check(self._lib.odevio_set_seed(self._plan, int(seed) & 0xFFFFFFFFFFFFFFFF))

    def rng_state(self):
        """(seed, draws so far) of the plan's random stream: the next "hard" fusion forward uses draw index = the second value."""
        self._ensure_plan()
        seed, calls = ctypes.c_uint64(), ctypes.c_uint64()
        _lib.check(self._lib.odevio_rng_state(self._plan, ctypes.byref(seed), ctypes.byref(calls)))
        return int(seed.value), int(calls.value)

    def fuse(self, fv, fi):
        self._ensure_plan()
        fv, fi = self._dev(fv, "fv"), self._dev(fi, "fi")
        # ("hard": the Gumbel mask of FusionModule.py:24-29 drawn on the device from the plan's own generator - set_seed())
        P = fv.shape[0] * fv.shape[1]
        out = torch.empty(fv.shape[0], fv.shape[1], fv.shape[2] + fi.shape[2], device=fv.device, dtype=torch.float32)
        _lib.check(self._lib.odevio_fuse_fwd(self._plan, fv.data_ptr(), fi.data_ptr(), P, out.data_ptr(), self._stream()))
        return out

    def ode_func(self, y):
        self._ensure_plan()
        y = self._dev(y, "y")
        out = torch.empty_like(y)
        _lib.check(self._lib.odevio_ode_func(self._plan, y.data_ptr(), y.shape[0], out.data_ptr(), self._stream()))
        return out

    def ode_steps(self, y, t0, t1, solver=None, substeps=0, return_stats=False):
        """Integrate rows of y from t0[r] to t1[r] (PoseODERNN.evolve_state)."""
        self._ensure_plan()
        y, t0, t1 = self._dev(y, "y"), self._dev(t0, "t0"), self._dev(t1, "t1")
        if solver is not None and solver not in _lib.SOLVERS:
            raise ValueError(f"Solver {solver} not supported")
        out = torch.empty_like(y)
        stats = torch.zeros(y.shape[0], 2, device=y.device, dtype=torch.int32)
        _lib.check(self._lib.odevio_ode_steps(self._plan, y.data_ptr(), t0.data_ptr(), t1.data_ptr(), y.shape[0],
                                              -1 if solver is None else _lib.SOLVERS[solver], substeps, out.data_ptr(),
                                              stats.data_ptr(), self._stream()))
        return (out, stats) if return_stats else out

    def pose_net(self, fv, fi, timestamps, hc=None, return_stats=False):
        """PoseODERNN.forward / PoseRNN.forward on encoder features."""
        self._ensure_plan()
        fused = self.fuse(fv, fi)
        ts = self._dev(timestamps, "timestamps")
        B, P, F = fused.shape
        L = self.opt.rnn_num_layers
        hcp = None
        if hc is not None:
            hc = self._dev(hc, "hc")
            hcp = hc.data_ptr()
        poses = torch.empty(B, P, 6, device=fused.device, dtype=torch.float32)
        h_T = torch.empty(L, B, F, device=fused.device, dtype=torch.float32)
        stats = torch.zeros(L * B, 2, device=fused.device, dtype=torch.int32)
        # cat/soft were applied by fuse(); hand the fused rows to the integrator
        _lib.check(self._lib.odevio_ode_rnn_fwd(self._plan, fused.data_ptr(), ts.data_ptr(), hcp, B, P, poses.data_ptr(),
                                                h_T.data_ptr(), stats.data_ptr(), self._stream()))
        return (poses, h_T, stats) if return_stats else (poses, h_T)

    def pose_cde(self, fv, fi, timestamps, prev=None, return_stats=False):
        """PoseCDE.forward (reference PoseCDE.py:76-103) on encoder features.

        Host side (plumbing only): time-channel concat and the eval-mode window history; everything else - z0, the
        control-path derivative, CDEFunc, the solver, the regressor - runs in libodevio.  Like the reference, eval mode
        uses raw timestamps and accumulates history when ``prev`` is given, the output times are ROW 0's timestamps,
        and the INITIAL state z0 is what is returned as the second value.
        """
        self._ensure_plan()
        net = self.Pose_net
        fused = self.fuse(fv, fi)
        ts = self._dev(timestamps, "timestamps")
        tsd = ts - ts[:, :1] if self.training else ts
        x = torch.cat([tsd[:, 1:, None], fused], dim=-1)
        if not self.training:
            net.history = torch.cat([net.history, x], dim=1) if prev is not None else x
            obs = net.history
        else:
            net.history = None
            obs = x
        obs = obs.contiguous()
        B, L, C = obs.shape
        t_out = tsd[0, 1:].double().cpu().contiguous()
        n_out = t_out.numel()
        hc = self.opt.cde_hidden_dim
        z0_in = None
        if prev is not None:
            prev = self._dev(prev, "hc")
            z0_in = prev.data_ptr()
        poses = torch.empty(B, n_out, 6, device=obs.device, dtype=torch.float32)
        z0 = torch.empty(B, hc, device=obs.device, dtype=torch.float32)
        stats = (ctypes.c_int32 * 2)()
        _lib.check(self._lib.odevio_cde_fwd(self._plan, obs.data_ptr(), B, L, t_out.data_ptr(), n_out, z0_in, poses.data_ptr(),
                                            z0.data_ptr(), ctypes.cast(stats, ctypes.c_void_p), self._stream()))
        return (poses, z0, (int(stats[0]), int(stats[1]))) if return_stats else (poses, z0)

    def cde_func(self, z, obs, seg):
        """One evaluation of the Neural-CDE vector field on piece ``seg`` of the rectilinear control path built from
        ``obs`` [B,L,1+F]: reshape(CDEFunc(z), [B,H,H+1]) @ dX/dt (reference ODEFunc.py:76-83 + torchcde's cdeint)."""
        self._ensure_plan()
        z, obs = self._dev(z, "z"), self._dev(obs, "obs")
        out = torch.empty_like(z)
        _lib.check(self._lib.odevio_cde_func(self._plan, z.data_ptr(), obs.data_ptr(), obs.shape[0], obs.shape[1], int(seg),
                                             out.data_ptr(), self._stream()))
        return out

    def cde_last_ms(self):
        """Duration (ms) of the last layer of the most recent ``cde_func`` call (stage timers must be on)."""
        ms = ctypes.c_float()
        _lib.check(self._lib.odevio_cde_last_ms(self._plan, ctypes.cast(ctypes.pointer(ms), ctypes.c_void_p)))
        return float(ms.value)

    STAGES = ("conv1", "conv2_6", "visual_head", "imu_fuse", "integrator", "regressor")

    def profile_enable(self, on=True, depth=1):
        """Record HIP events at the stage boundaries of every following forward (on the current stream).  ``depth``
        forwards may be issued back to back before ``profile_read`` (which then returns their average)."""
        self._ensure_plan()
        _lib.check(self._lib.odevio_profile_enable(self._plan, max(1, int(depth)) if on else 0))

    def profile_read(self):
        """Stage durations (ms), averaged over the forwards since the last read (at most ``depth``), keyed by ``STAGES``."""
        ms = (ctypes.c_float * len(self.STAGES))()
        _lib.check(self._lib.odevio_profile_read(self._plan, ctypes.cast(ms, ctypes.c_void_p)))
        return dict(zip(self.STAGES, [float(x) for x in ms]))

    def check(self):
        """Synchronise and raise if the last integrator launch reported a timeout / step-budget error."""
        self._ensure_plan()
        _lib.check(self._lib.odevio_check(self._plan, self._stream()))
