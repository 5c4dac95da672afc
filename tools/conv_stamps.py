#!/usr/bin/env python
"""Phase stamps of workgroup 0 of the fp16x2 conv kernel (diagnostic build: make -C odevio_amd/csrc STAMPS=1): cycles spent
in the prologue (until the first K-tile has landed), the K loop and the epilogue, per layer at the bench shape.
Usage: python tools/conv_stamps.py [layers...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ODEVIO_LIB"] = os.path.join(ROOT, "odevio_amd", "libodevio_stamps.so")
import torch
from odevio_amd import DeepVIO, default_opt, weights, _lib

opt = default_opt()
m = DeepVIO(opt, seed=0).cuda()
B, S = 16, 11
P = B * (S - 1)
layers = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 5, 6, 7, 8]
h, w = opt.img_h, opt.img_w
shapes = []
for name, cin, cout, k, s in weights.IMAGE_CONVS:
    shapes.append((name, cin, hi := h, wi := w))
    h, w = weights.conv_out(h, k, s), weights.conv_out(w, k, s)
for i in layers:
    name, cin, hi, wi = shapes[i]
    x = torch.randn(P, hi, wi, cin, device="cuda")
    for _ in range(2):
        m.conv_block(i, x, B, S)
    torch.cuda.synchronize()
    out = (ctypes.c_uint64 * 8)()
    _lib.check(m._lib.odevio_debug_stamps(m._plan, ctypes.cast(out, ctypes.c_void_p), m._stream()))
    t0, t1, t2, t3, nt = out[0], out[1], out[2], out[3], out[4]
    print(f"{name:8s} K-tiles {nt:4d}: prologue {t1 - t0:7d}  K loop {t2 - t1:8d} ({(t2 - t1) / max(nt, 1):7.0f} per K-tile)  epilogue {t3 - t2:7d} cycles"
          f"  -> per-tile overhead {(t1 - t0 + t3 - t2) / max(t3 - t0, 1) * 100:.1f} % of the workgroup's life")
