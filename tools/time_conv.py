"""Times single conv blocks (production kernels) at the bench shapes; GPU box only.  ODEVIO_LIB selects a build."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, torch
from odevio_amd import DeepVIO, default_opt, weights
opt = default_opt()
m = DeepVIO(opt, seed=0).cuda()
B, S = 16, 11
P = B * (S - 1)
layers = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 5, 6, 7, 8]
h, w = opt.img_h, opt.img_w
shapes = []
for name, cin, cout, k, s in weights.IMAGE_CONVS:
    shapes.append((name, cin, cout, k, s, h, w))
    h, w = weights.conv_out(h, k, s), weights.conv_out(w, k, s)
tot = 0.0
for i in layers:
    name, cin, cout, k, s, hi, wi = shapes[i]
    if i == 0:
        x = torch.rand(B, S, 3, hi, wi, device="cuda") - 0.5
    else:
        x = torch.randn(P, hi, wi, cin, device="cuda")
    for _ in range(2):
        m.conv_block(i, x, B, S)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        m.conv_block(i, x, B, S)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    ho, wo = weights.conv_out(hi, k, s), weights.conv_out(wi, k, s)
    fl = 2.0 * P * ho * wo * cout * cin * k * k
    tot += ms
    print(f"{name:8s} {ms:7.3f} ms  {fl/ms/1e9:6.1f} TFLOP/s")
print(f"total {tot:.3f} ms")
