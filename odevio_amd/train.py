"""Training-side surface of the hot path: gradients through the ODE-RNN pose net.

The reference trains with plain autograd (scripts/train_model.py:69-78): ``poses, _ = model(...)``, the loss
``100 * MSE(angles) + MSE(translations)``, ``loss.backward()`` through torchode's AutoDiffAdjoint, i.e. backpropagation
through the solver's own arithmetic.  Here the same chain is two ``torch.autograd.Function``s whose forward AND backward
run in libodevio (``odevio_ode_rnn_fwd`` / ``odevio_ode_rnn_bwd``, ``odevio_pose_loss``); PyTorch only carries the graph.

What ``odevio_ode_rnn_bwd`` covers: fixed-step solvers (rk4, rk4_classic) and adaptive ones (dopri5, tsit5, heun: the
forward's accepted steps are replayed, their sizes held constant), ``nn.RNN`` and ``nn.GRU``, ``cat`` fusion;
gradients reach the encoder FEATURES (fv, fi), the carried state ``hc`` and every parameter of ``Pose_net``
(ODEFunc, RNN, regressor).  The encoders' own backward is not built yet.
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
        _lib.check(model._lib.odevio_ode_rnn_fwd(model._plan, fused.data_ptr(), ts.data_ptr(), None if hcd is None else hcd.data_ptr(),
                                                 B, P, poses.data_ptr(), h_T.data_ptr(), None, model._stream()))
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
        _lib.check(model._lib.odevio_ode_rnn_bwd(
            model._plan, fused.data_ptr(), ts.data_ptr(), None if hc is None else hc.data_ptr(), B, P, g_poses.data_ptr(),
            None if g_hT is None else g_hT.data_ptr(), g_fused.data_ptr(), None if g_hc is None else g_hc.data_ptr(),
            arr, len(grads), model._stream()))
        return (None, None, g_fused, None, g_hc, *grads)


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
    if opt.fuse_method != "cat":
        raise ValueError("odevio_amd.train.pose_net: only fuse_method 'cat' has a backward so far")
    model._ensure_plan()
    names = pose_param_names(opt)
    params = dict(model.named_parameters())
    plist = [params[n] for n in names]
    fused = torch.cat((fv, fi), dim=-1)          # plumbing: autograd splits the feature gradient back into fv / fi
    return _OdeRnnFunction.apply(model, names, fused, timestamps, hc, *plist)
