"""The N>1 path on CPU: 2 gloo ranks shard the sequences, run the pose net, all-gather; must equal N=1."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from odevio_amd import default_opt, synth, weights
    from odevio_amd import dist as od
    from oracle import odevio_oracle as oc
    opt = default_opt(model_type="rnn", fuse_method="soft")
    sd = weights.make_state_dict(opt, seed=7, randomize_stats=True)
    g = torch.Generator().manual_seed(0)
    fv, fi = torch.randn(B, 4, 512, generator=g), torch.randn(B, 4, 256, generator=g)
    ts = synth.timestamps(B, 5, drop=0.3, seed=1)
    hc = torch.randn(2, B, 768, generator=g) * 0.1
    # stand-in with the DeepVIO.forward signature (features play the part of img/imu): the product model needs a GPU
    model = lambda a, b, c, d: oc.pose_rnn(sd, a, b, c, d, opt)
    poses, h = od.forward_sharded(model, fv, fi, ts, hc)
    ref_p, ref_h = oc.pose_rnn(sd, fv, fi, ts, hc, opt)
    ok = bool(poses.shape == ref_p.shape and h.shape == ref_h.shape and
              oc.rel_err(poses, ref_p) < 1e-5 and oc.rel_err(h, ref_h) < 1e-5)
    lo, hi = od.shard_range(B, rank, world)
    q.put((rank, ok, lo, hi))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 5])  # even and ragged shards
def test_two_rank_gloo_equals_single_rank(B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    assert res[0][2] == 0 and res[0][3] == res[1][2] and res[1][3] == B  # contiguous cover


def test_shard_range_covers_every_sequence_once():
    from odevio_amd.dist import shard_range
    for B in (1, 7, 16, 128):
        for world in (1, 2, 3, 8):
            cuts = [shard_range(B, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from odevio_amd import dist as od
    g = torch.Generator().manual_seed(100 + rank)
    shapes = [(7, 5), (13,), (3, 4, 2), (1,)]
    grads = [torch.randn(*s, generator=g) for s in shapes]
    mine = [x.clone() for x in grads]
    od.allreduce_gradients(grads)
    # what every rank must now hold: the sum of all ranks' tensors (each rank can regenerate the others' from their seeds)
    want = [torch.zeros(*s) for s in shapes]
    for r in range(world):
        gr = torch.Generator().manual_seed(100 + r)
        for w, s in zip(want, shapes):
            w += torch.randn(*s, generator=gr)
    ok = all(torch.allclose(a, b, atol=1e-6) for a, b in zip(grads, want)) and all(a.shape == b.shape for a, b in zip(grads, mine))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_bucket_allreduce_two_ranks():
    """The training step's one exchange: every gradient tensor summed over the ranks through ONE flat bucket."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)
