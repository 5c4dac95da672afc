#!/usr/bin/env python
"""Numerics of Winograd F(2x2, 3x3) under the two-fp16-piece operand scheme of conv_f16x2.hip, on a conv4_1-sized layer (512 -> 512
channels, 16 x 32 pixels), CPU emulation with the exact piece arithmetic (pieces rounded to fp16, products and sums in fp32), against
an fp64 evaluation of the same layer.  Evidence for DESIGN.md section 5.7: the error is fine (about 2 x the direct product's); the
data flow is what rules the scheme out.  Output committed as profiles/r03_winograd_numerics.txt."""
import math
import torch

torch.manual_seed(0)


def split(x):
    h = x.to(torch.float16).to(torch.float32)
    return h, (x - h).to(torch.float16).to(torch.float32)


def prescale(w):   # largest magnitude into [2^13, 2^14), like split_conv_weights (api.hip)
    return 2.0 ** (14 - math.frexp(float(w.abs().max()))[1])


N, C, K, H, W = 2, 512, 512, 16, 32
x = torch.nn.functional.leaky_relu(torch.randn(N, C, H, W), 0.1)
w = torch.randn(K, C, 3, 3) * math.sqrt(2.0 / (C * 9))
truth = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
scale = truth.abs().max()
err = lambda y: float((y.double() - truth).abs().max() / scale)
conv = lambda a, b: torch.nn.functional.conv2d(a, b, padding=1)
print(f"layer: {C} -> {K} channels, {H} x {W} pixels, 3x3 stride 1; error = max|y - y64| / max|y64|")
print(f"direct, fp32 (PyTorch CPU)                   {err(conv(x, w)):.2e}")
ps = prescale(w)
xh, xl = split(x)
wh, wl = split(w * ps)
print(f"direct, two fp16 pieces (h h + h l + l h)      {err((conv(xl, wh) + conv(xh, wl) + conv(xh, wh)) / ps):.2e}")
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
d = torch.nn.functional.pad(x, (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2)     # 4 x 4 input tiles, stride 2
V = torch.einsum('ij,ncabjk,lk->ncabil', BT, d, BT)                               # input transform in fp32, before the split
U = torch.einsum('ij,kcjl,ml->kcim', G, w.double(), G).float()                    # filter transform in fp64, rounded once
for mode in ("fp32", "two fp16 pieces"):
    if mode == "fp32":
        M = torch.einsum('ncabil,kcil->nkabil', V, U)
    else:
        psu = prescale(U)
        Vh, Vl = split(V)
        Uh, Ul = split(U * psu)
        M = (torch.einsum('ncabil,kcil->nkabil', Vl, Uh) + torch.einsum('ncabil,kcil->nkabil', Vh, Ul) + torch.einsum('ncabil,kcil->nkabil', Vh, Uh)) / psu
    Y = torch.einsum('ij,nkabjl,ml->nkabim', AT, M, AT)
    print(f"Winograd F(2x2,3x3), {mode:16s}        {err(Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, H, W)):.2e}")
