"""Diagnostic: phase shares of the persistent integrator (needs libodevio_stamps.so; GPU box only)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, os, sys
os.environ["ODEVIO_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "odevio_amd", "libodevio_stamps.so")
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
    out = (ctypes.c_uint64 * 12)()
    _lib.check(m._lib.odevio_debug_stamps(m._plan, ctypes.cast(out, ctypes.c_void_p), None))
    tot, tg, tl, tr, ng = [int(x) for x in out[:5]]
    local = [(int(out[5]) >> (8 * g)) & 1 for g in range(8)]
    print(f"   post-layer barrier {100*int(out[6])/tot:.1f}%  owner epilogue {100*int(out[7])/tot:.1f}%")
    pro, fev, nrm, rnp = [int(x) for x in out[8:12]]
    print(f"   by section: launch -> first interval {100*pro/tot:.1f}%  vector-field evaluations {100*fev/tot:.1f}%  error norm + controller {100*nrm/tot:.1f}%  "
          f"RNN phases {100*rnp/tot:.1f}%  Runge-Kutta arithmetic and the rest {100*(tot-pro-fev-nrm-rnp)/tot:.1f}%")
    # production library timing of the same call (events on the current stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        m.pose_net(fv, fi, ts)
    e1.record(); torch.cuda.synchronize()
    print(f"   pose_net (stamps build) {e0.elapsed_time(e1)/20*1000:.1f} us per call")
    print(f"{solver} B={B} safe={safe}: kernel {tot} cyc; gathers {ng} total {100*tg/tot:.1f}% ({tg/max(ng,1):.0f} cyc each); "
          f"ode layers {100*tl/tot:.1f}%; rnn layers {100*tr/tot:.1f}%; other {100*(tot-tg-tl-tr)/tot:.1f}%; L2-local groups {local}")
