"""Generates tests/golden/resize.npz with Pillow itself (the reference's loader dependency, importable in the build
container): seeded uint8 frames, their PIL BILINEAR resizes, nothing else.  Run once: python oracle/gen_golden_resize.py"""
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def frame(h, w, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    # structured: gradients, flat and saturated regions, a sharp edge (what real frames have and noise does not)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x + y) % 256)], -1).astype(np.uint8)
    img[h // 4:h // 2, w // 3:w // 2] = 255
    img[h // 2:3 * h // 4, w // 2:2 * w // 3] = 0
    img[:, w // 5] = rng.integers(0, 256, (h, 3), dtype=np.uint8)
    return img


CASES = [  # (Hin, Win, Hout, Wout, seed, kind)
    (376, 1241, 256, 512, 1, "noise"),      # KITTI frame -> network input (KITTI_eval.py:101)
    (376, 1241, 256, 512, 2, "structured"),
    (370, 1226, 256, 512, 3, "noise"),      # the other KITTI frame size
    (64, 100, 96, 160, 4, "noise"),         # upscale: plain bilinear taps
    (100, 128, 64, 128, 5, "structured"),   # vertical pass only
    (64, 200, 64, 128, 6, "noise"),         # horizontal pass only
]

if __name__ == "__main__":
    import hashlib
    out = {"cases": np.asarray([c[:5] for c in CASES], dtype=np.int64), "kinds": np.asarray([c[5] for c in CASES])}
    for i, (hi, wi, ho, wo, seed, kind) in enumerate(CASES):
        img = frame(hi, wi, seed, kind)
        full = np.ascontiguousarray(np.asarray(Image.fromarray(img).resize((wo, ho), Image.BILINEAR)))
        # the whole output as a digest (noise does not compress), its first and last 8 rows verbatim
        out[f"sha{i}"] = np.asarray(hashlib.sha256(full.tobytes()).hexdigest())
        out[f"top{i}"], out[f"bot{i}"] = full[:8], full[-8:]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "resize.npz"), **out)
    print({k: v.shape for k, v in out.items()})
