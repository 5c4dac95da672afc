"""Multi-GPU layer of the hot path: shard sequences, gather poses.

Sequences (batch rows) are independent end to end - per-row ODE time/step state (reference
src/models/PoseODERNN.py:72-75) and eval-mode BatchNorm - so rank r simply takes a contiguous slice
of the batch with a full weight replica, and nothing is exchanged during compute.  The only
collective is one all-gather of the per-sequence poses [B/N, S-1, 6] (and of h_T on its batch axis,
dim 1, when the caller streams) - RCCL over xGMI on the GPUs (backend "nccl"), gloo in the CPU tests.
Training adds the one real exchange step of data parallelism: the gradients of a step, summed over the ranks in ONE
bucket (``allreduce_gradients``: Pose_net + Inertial_net are 5.6 M floats = 22 MB - a single RCCL all-reduce per step; xGMI
rings are per-link bound, so fewer and larger collectives are the right shape).
The reference itself has no distributed code (single-device nn.DataParallel only).
"""
import torch
import torch.distributed as dist


def shard_range(B, rank, world):
    """Contiguous, balanced slice [lo, hi) of B sequences for ``rank`` (first B % world ranks get one more)."""
    base, extra = divmod(B, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_inputs(img, imu, ts, hc, rank, world):
    lo, hi = shard_range(img.shape[0], rank, world)
    return img[lo:hi], imu[lo:hi], ts[lo:hi], (None if hc is None else hc[:, lo:hi].contiguous())


def _all_gather_var(x, dim, sizes, group=None):
    """all-gather tensors whose ``dim`` extent differs per rank (ragged last shard)."""
    world = dist.get_world_size(group)
    mx = max(sizes)
    pad_shape = list(x.shape)
    pad_shape[dim] = mx
    buf = x.new_zeros(pad_shape)
    buf.narrow(dim, 0, x.shape[dim]).copy_(x)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf.contiguous(), group=group)
    return torch.cat([o.narrow(dim, 0, n) for o, n in zip(outs, sizes)], dim=dim)


def gather_outputs(poses, h_T, B, group=None):
    """Local (poses [b,S-1,6], h_T [L,b,F]) -> global (poses [B,S-1,6], h_T [L,B,F]) on every rank."""
    world = dist.get_world_size(group)
    sizes = [shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world)]
    if len(set(sizes)) == 1 and poses.is_cuda:
        out = poses.new_empty((B,) + tuple(poses.shape[1:]))
        dist.all_gather_into_tensor(out, poses.contiguous(), group=group)
        return out, _all_gather_var(h_T, 1, sizes, group)
    return _all_gather_var(poses, 0, sizes, group), _all_gather_var(h_T, 1, sizes, group)


def forward_sharded(model, img, imu, ts, hc=None, group=None):
    """``model(img, imu, ts, hc)`` with the batch split over the ranks of ``group``; every rank gets the full result.

    ``model`` is any callable with the DeepVIO.forward signature (the HIP model on GPUs).
    """
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    B = img.shape[0]
    a, b, c, d = shard_inputs(img, imu, ts, hc, rank, world)
    if a.shape[0] == 0:
        raise ValueError(f"batch {B} is smaller than the world size {world}")
    poses, h_T = model(a, b, c, d)
    return gather_outputs(poses, h_T, B, group)


def allreduce_gradients(grads, group=None):
    """Sum the gradient tensors over the ranks of ``group`` in place, as ONE flat bucket (one all-reduce).

    The caller scales its loss by 1 / world_size before ``backward`` so that the sum is the data-parallel mean - no
    arithmetic happens here, only packing (copies) around the collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    flat = torch.empty(sum(g.numel() for g in grads), device=grads[0].device, dtype=grads[0].dtype)
    off = 0
    for g in grads:
        flat[off:off + g.numel()].copy_(g.reshape(-1))
        off += g.numel()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
