"""Seeded synthetic weights for the hot path, keyed exactly like the reference's ``state_dict``.

There are no checkpoints offline (reference README.md:69 is a Drive link), so benches and parity
tests draw weights from the *distribution* the reference's constructor leaves behind
(SURVEY.md section 8a row A10):

* every Conv/Linear: ``kaiming_normal_`` (fan_in, gain sqrt(2)) and zero bias
  (reference ``src/models/DeepVIO.py:77-87`` - this overrides ODEFunc's own N(0, 0.1) init),
* BatchNorm: gamma 1, beta 0 (``DeepVIO.py:120-122``); ``running_mean = 0``,
  ``running_var = 0.9``, ``num_batches_tracked = 1`` - the state the ImageEncoder ctor's dummy
  training-mode forward on zeros leaves (``src/models/Encoder.py:92-93``).  The reference's
  InertialEncoder never runs such a forward, so its BN1d stats stay at (0, 1, 0),
* ``nn.RNN`` / ``nn.GRU``: PyTorch default U(-1/sqrt(F), 1/sqrt(F)) (``initialization`` has no branch
  for them).

Key names are the reference's.  Each tensor is drawn from its own CPU
generator seeded by (seed, key), so the dict does not depend on creation order and is identical in
the build container and on the GPU box (same torch build).
"""
import math
import zlib

import torch

# (name, cin, cout, k, stride) of the FlowNetS-style stack, reference Encoder.py:82-90
IMAGE_CONVS = [
    ("conv1", 6, 64, 7, 2),
    ("conv2", 64, 128, 5, 2),
    ("conv3", 128, 256, 5, 2),
    ("conv3_1", 256, 256, 3, 1),
    ("conv4", 256, 512, 3, 2),
    ("conv4_1", 512, 512, 3, 1),
    ("conv5", 512, 512, 3, 2),
    ("conv5_1", 512, 512, 3, 1),
    ("conv6", 512, 1024, 3, 2),
]
# (sequential index of the Conv1d, cin, cout) of the IMU stack, reference Encoder.py:43-56
IMU_CONVS = [(0, 6, 64), (4, 64, 128), (8, 128, 256)]
IMU_WINDOW = 11  # samples per frame pair, reference Encoder.py:57,63-66


def conv_out(n, k, s):
    return (n + 2 * ((k - 1) // 2) - k) // s + 1


def encoder_out_hw(img_h, img_w):
    h, w = img_h, img_w
    for _, _, _, k, s in IMAGE_CONVS:
        h, w = conv_out(h, k, s), conv_out(w, k, s)
    return h, w


def _gen(seed, key):
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2**63 - 1))
    return g


def _kaiming(shape, seed, key):
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    std = math.sqrt(2.0 / fan_in)
    return torch.randn(shape, generator=_gen(seed, key), dtype=torch.float32) * std


def _uniform(shape, bound, seed, key):
    return (torch.rand(shape, generator=_gen(seed, key), dtype=torch.float32) * 2.0 - 1.0) * bound


def _linear(sd, prefix, n_out, n_in, seed):
    sd[prefix + ".weight"] = _kaiming((n_out, n_in), seed, prefix + ".weight")
    sd[prefix + ".bias"] = torch.zeros(n_out)


def _bn(sd, prefix, c, var, nbt):
    sd[prefix + ".weight"] = torch.ones(c)
    sd[prefix + ".bias"] = torch.zeros(c)
    sd[prefix + ".running_mean"] = torch.zeros(c)
    sd[prefix + ".running_var"] = torch.full((c,), float(var))
    sd[prefix + ".num_batches_tracked"] = torch.tensor(nbt, dtype=torch.long)


def ode_linear_indices(n_hidden_layers):
    """Sequential indices of the Linears inside ``ODEFunc.net`` (reference ODEFunc.py:9-15)."""
    return [2 * i for i in range(n_hidden_layers + 1)]


def make_state_dict(opt, seed=0, randomize_stats=False):
    """Return a CPU fp32 ``state_dict`` for ``DeepVIO(opt)`` with the reference's key names.

    ``randomize_stats=True`` perturbs biases and BN statistics (still seeded) so that parity tests
    exercise the bias / running-stat arithmetic that the constructor distribution leaves at 0/1.
    """
    sd = {}
    F = opt.v_f_len + opt.i_f_len
    # --- Image_net
    for name, cin, cout, k, _ in IMAGE_CONVS:
        sd[f"Image_net.{name}.0.weight"] = _kaiming((cout, cin, k, k), seed, f"Image_net.{name}.0.weight")
        _bn(sd, f"Image_net.{name}.1", cout, 0.9, 1)
    oh, ow = encoder_out_hw(opt.img_h, opt.img_w)
    _linear(sd, "Image_net.visual_head", opt.v_f_len, 1024 * oh * ow, seed)
    # --- Inertial_net
    for idx, cin, cout in IMU_CONVS:
        sd[f"Inertial_net.encoder_conv.{idx}.weight"] = _kaiming(
            (cout, cin, 3), seed, f"Inertial_net.encoder_conv.{idx}.weight")
        sd[f"Inertial_net.encoder_conv.{idx}.bias"] = torch.zeros(cout)
        _bn(sd, f"Inertial_net.encoder_conv.{idx + 1}", cout, 1.0, 0)
    _linear(sd, "Inertial_net.proj", opt.i_f_len, 256 * IMU_WINDOW, seed)
    # --- Pose_net
    if opt.fuse_method == "soft":
        _linear(sd, "Pose_net.fuse.net.0", F, F, seed)
    elif opt.fuse_method == "hard":
        _linear(sd, "Pose_net.fuse.net.0", 2 * F, F, seed)
    if opt.model_type in ("ode-rnn", "rnn"):
        if opt.model_type == "ode-rnn":
            H = opt.ode_hidden_dim
            dims = [F] + [H] * opt.ode_fn_num_layers + [F]
            for li, idx in enumerate(ode_linear_indices(opt.ode_fn_num_layers)):
                _linear(sd, f"Pose_net.ode_func.net.{idx}", dims[li + 1], dims[li], seed)
        gates = {"rnn": 1, "gru": 3}.get(opt.ode_rnn_type)
        if gates is None:
            raise ValueError(f"RNN type {opt.ode_rnn_type} not supported")
        bound = 1.0 / math.sqrt(F)
        for layer in range(opt.rnn_num_layers):
            for nm, shape in (("weight_ih", (gates * F, F)), ("weight_hh", (gates * F, F)),
                              ("bias_ih", (gates * F,)), ("bias_hh", (gates * F,))):
                key = f"Pose_net.rnn.{nm}_l{layer}"
                sd[key] = _uniform(shape, bound, seed, key)
        reg_in = F
    elif opt.model_type == "cde":
        Hc = opt.cde_hidden_dim
        _linear(sd, "Pose_net.reduction_net.0", F // 2, F, seed)
        _linear(sd, "Pose_net.reduction_net.2", Hc, F // 2, seed)
        _linear(sd, "Pose_net.initial.0", Hc, Hc + 1, seed)
        dims = [Hc] * (opt.cde_fn_num_layers + 1) + [Hc * (Hc + 1)]
        for li, idx in enumerate(ode_linear_indices(opt.cde_fn_num_layers)):
            _linear(sd, f"Pose_net.cde_func.net.{idx}", dims[li + 1], dims[li], seed)
        reg_in = Hc
    else:
        raise ValueError(f"model_type {opt.model_type!r} not supported by the hot path")
    _linear(sd, "Pose_net.regressor.0", 128, reg_in, seed)
    _linear(sd, "Pose_net.regressor.2", 6, 128, seed)

    if randomize_stats:
        for key in list(sd.keys()):
            t = sd[key]
            if key.endswith("num_batches_tracked"):
                continue
            if key.endswith("running_var"):
                sd[key] = 0.5 + torch.rand(t.shape, generator=_gen(seed, key + "#r"))
            elif key.endswith("running_mean") or (key.endswith(".bias") and ".rnn." not in key):
                sd[key] = 0.1 * torch.randn(t.shape, generator=_gen(seed, key + "#r"))
            elif key.endswith(".1.weight") or (".encoder_conv." in key and t.dim() == 1
                                               and key.endswith(".weight")):
                sd[key] = 0.75 + 0.5 * torch.rand(t.shape, generator=_gen(seed, key + "#r"))
    return sd


def filter_reference_state_dict(sd):
    """Drop alias keys a reference checkpoint carries but the hot path does not own.

    ``PoseODERNN.solver`` is a ``torch.compile``-wrapped torchode module holding the same
    ``ode_func`` (reference PoseODERNN.py:58-60), so checkpoints repeat its weights under
    ``Pose_net.solver.*``; a FlowNet checkpoint nests everything under ``"state_dict"``
    (reference scripts/train_model.py:181-187).
    """
    if "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    out = {}
    for k, v in sd.items():
        if k.startswith("module."):
            k = k[len("module."):]
        if k.startswith("Pose_net.solver."):
            continue
        out[k] = v
    return out


def load_flownet_checkpoint(model, checkpoint):
    """Initialise ``model.Image_net`` from a FlowNet checkpoint the way the reference does
    (scripts/train_model.py:180-188): take ``checkpoint["state_dict"]``, keep only the keys ``Image_net`` owns (FlowNetS
    carries decoder / flow-prediction layers the encoder does not have, and the encoder's ``visual_head`` is not in
    FlowNet), overlay them on the current ``Image_net`` state and load that.  Returns the sorted list of keys taken.
    Shapes are checked by ``load_state_dict`` exactly as in the reference."""
    sd = checkpoint["state_dict"] if isinstance(checkpoint, dict) and "state_dict" in checkpoint else checkpoint
    own = model.Image_net.state_dict()
    take = {k: v for k, v in sd.items() if k in own}
    own.update(take)
    model.Image_net.load_state_dict(own)
    model._plan_sig = None          # the device plan re-lays the weights at the next forward
    return sorted(take)
