"""The multi-GPU layer with the HIP model and RCCL: `dist.forward_sharded` around the real
`DeepVIO` on as many GPUs as the box has.  world_size 1 runs everywhere (the RCCL all_gather_into_tensor path with one
rank); world_size 2 needs two GPUs and is skipped on the one-GPU test box.  Each rank is its own process (one process
per GPU), started before anything in it touches the GPU."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from odevio_amd import DeepVIO, default_opt, synth, weights
    from odevio_amd import dist as od
    from oracle import odevio_oracle as oc
    opt = default_opt(img_h=64, img_w=128, ode_solver="dopri5")
    sd = weights.make_state_dict(opt, seed=97, randomize_stats=True)
    model = DeepVIO(opt, state_dict=sd).cuda()
    img, imu, ts = synth.batch(B, 4, 64, 128, drop=0.3, seed=17)
    hc = torch.randn(2, B, 768, generator=torch.Generator().manual_seed(2)) * 0.1
    poses, h = od.forward_sharded(model, img.cuda(), imu.cuda(), ts.cuda(), hc.cuda())
    model.check()
    ref_p, ref_h = oc.deepvio_forward(sd, img, imu, ts, hc, opt)
    ep, eh = oc.rel_err(poses, ref_p), oc.rel_err(h, ref_h)
    q.put((rank, tuple(poses.shape), tuple(h.shape), ep, eh))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(1, 3), (2, 4), (2, 5)])
def test_forward_sharded_hip_model_rccl(world, B):
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ps, hs, ep, eh in res:   # every rank holds the full result
        assert ps == (B, 3, 6) and hs == (2, B, 768)
        assert ep < 2e-4 and eh < 2e-4, (rank, ep, eh)   # carried hc + dopri5: the 2e-4 bar of test_pose_ode_rnn
