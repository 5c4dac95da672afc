"""Diagnostic: phase shares of the persistent integrator (needs libodevio_stamps.so; GPU box only)."""
import ctypes, os, sys
os.environ["ODEVIO_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "odevio_amd", "libodevio_stamps.so")
import torch
from odevio_amd import DeepVIO, default_opt, synth, _lib
for solver, B, safe in (("rk4", 16, "0"), ("rk4", 16, "1"), ("dopri5", 16, "0"), ("rk4", 1, "0")):
    os.environ["ODEVIO_SAFE_HANDOFF"] = safe
    opt = default_opt(img_h=64, img_w=128, ode_solver=solver)
    m = DeepVIO(opt, seed=0).cuda()
    g = torch.Generator().manual_seed(0)
    fv, fi = torch.randn(B, 10, 512, generator=g).cuda(), torch.randn(B, 10, 256, generator=g).cuda()
    ts = synth.timestamps(B, 11).cuda()
    for _ in range(3):
        m.pose_net(fv, fi, ts)
    torch.cuda.synchronize()
    out = (ctypes.c_uint64 * 8)()
    _lib.check(m._lib.odevio_debug_stamps(m._plan, ctypes.cast(out, ctypes.c_void_p), None))
    tot, tg, tl, tr, ng = [int(x) for x in out[:5]]
    local = [(int(out[5]) >> (8 * g)) & 1 for g in range(8)]
    print(f"{solver} B={B} safe={safe}: kernel {tot/100:.1f} us; gathers {ng} total {tg/100:.1f} us ({tg/max(ng,1)/100:.2f} us each); "
          f"ode layers {tl/100:.1f} us; rnn layers {tr/100:.1f} us; other {(tot-tg-tl-tr)/100:.1f} us; L2-local groups {local}")
