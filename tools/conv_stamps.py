#!/usr/bin/env python
"""Phase stamps of workgroup 0 of the fp16x2 conv kernel (diagnostic build: make -C odevio_amd/csrc STAMPS=1): cycles spent
in the prologue (until the first K-tile has landed), the K loop and the epilogue of ONE layer at the bench shape, inside
the production chain (P2 in, P2 out): ODEVIO_STAMP_LAYER=i picks the layer, the image encoder runs, the stamps are read.
Usage: python tools/conv_stamps.py   (spawns itself once per layer)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ODEVIO_LIB"] = os.environ.get("ODEVIO_STAMPS_LIB", os.path.join(ROOT, "odevio_amd", "libodevio_stamps.so"))
import torch
from odevio_amd import DeepVIO, default_opt, weights, _lib

if "ODEVIO_STAMP_LAYER" not in os.environ:
    import subprocess
    for i in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 5, 6, 7, 8]:
        for wg in os.environ.get("ODEVIO_STAMP_WGS", "0,300").split(","):   # a first-round workgroup and a steady-state one
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, ODEVIO_STAMP_LAYER=str(i), ODEVIO_STAMP_WG=wg), check=True)
    sys.exit(0)

i = int(os.environ["ODEVIO_STAMP_LAYER"])
opt = default_opt()
m = DeepVIO(opt, seed=0).cuda()
B, S = 16, 11
img = torch.rand(B, S, 3, opt.img_h, opt.img_w, device="cuda") - 0.5
for _ in range(int(os.environ.get("ODEVIO_STAMP_WARM", "300"))):   # back-to-back launches first: the clock the chip holds under this load
    m.image_encoder(img)
torch.cuda.synchronize()
out = (ctypes.c_uint64 * 12)()
_lib.check(m._lib.odevio_debug_stamps(m._plan, ctypes.cast(out, ctypes.c_void_p), m._stream()))
t0, t1, t2, t3, nt = out[0], out[1], out[2], out[3], out[4]
name = weights.IMAGE_CONVS[i][0]
if t3 == 0:
    print(f"{name:8s} workgroup {os.environ.get('ODEVIO_STAMP_WG', '0'):>5s}: no such workgroup in this layer's grid", flush=True)
    sys.exit(0)
print(f"{name:8s} workgroup {os.environ.get('ODEVIO_STAMP_WG', '0'):>5s} K-tiles {nt:4d}: prologue {t1 - t0:7d}  K loop {t2 - t1:8d} ({(t2 - t1) / max(nt, 1):7.0f} per K-tile)  epilogue {t3 - t2:7d} cycles"
      f"  -> per-tile overhead {(t1 - t0 + t3 - t2) / max(t3 - t0, 1) * 100:.1f} % of the workgroup's life;"
      f" shader clock held {(t3 - t0) / max(out[6] - out[5], 1) * 0.1:.2f} GHz", flush=True)
