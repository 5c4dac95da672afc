"""TEST INFRASTRUCTURE (oracle): numpy restatement of Pillow's 8-bit BILINEAR resize, the resampling the reference's
loader applies to every frame (reference src/data/KITTI_eval.py:101 and src/data/utils.py:366-371: torchvision
``TF.resize`` of a PIL image = ``PIL.Image.resize(size, BILINEAR)``).

Third-party algorithm: Pillow (``pillow`` in the reference's environment; 12.2.0 is importable in the build container, so
this restatement is PINNED against the real library: tests/test_oracle_resize.py and the golden vectors in
tests/golden/resize.npz, produced by oracle/gen_golden_resize.py).  Follows Pillow ``src/libImaging/Resample.c``:
``precompute_coeffs`` (triangle filter, support stretched by the scale when shrinking), ``normalize_coeffs_8bpc``
(22-bit fixed point), ``ImagingResampleHorizontal_8bpc`` then ``ImagingResampleVertical_8bpc`` (int32 accumulation from
2^21, ``clip8`` of the sum >> 22), a pass being skipped when its size does not change.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def coeffs(in_size, out_size):
    """-> (bounds [out,2] int (first input index, count), kk [out,ksize] int32)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = []
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w.append(1.0 - a if a < 1.0 else 0.0)
        ww = sum(w)              # Pillow adds in index order in double, like this
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One resampling pass over `axis` of a uint8 array [..., H, W, C]."""
    img = np.moveaxis(img, axis, -2).astype(np.int64)          # [..., other, in, C]
    out = np.empty(img.shape[:-2] + (bounds.shape[0], img.shape[-1]), dtype=np.uint8)
    for xx in range(bounds.shape[0]):
        lo, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        s = (1 << (PRECISION_BITS - 1)) + np.tensordot(img[..., lo:lo + n, :], kk[xx, :n], axes=([-2], [0]))
        out[..., xx, :] = np.clip(s >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, -2, axis)


def resize_bilinear_u8(img, out_h, out_w):
    """img uint8 [..., H, W, C] -> uint8 [..., out_h, out_w, C], bit-identical to PIL.Image.resize((out_w, out_h), BILINEAR)."""
    h, w = img.shape[-3], img.shape[-2]
    if w != out_w:
        img = _pass(img, *coeffs(w, out_w), axis=img.ndim - 2)
    if h != out_h:
        img = _pass(img, *coeffs(h, out_h), axis=img.ndim - 3)
    return img
