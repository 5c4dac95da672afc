"""The resize oracle (numpy restatement of Pillow's 8-bit BILINEAR resampling) against Pillow's own outputs: the committed
golden vectors (tests/golden/resize.npz, written by oracle/gen_golden_resize.py with the real library) and, where Pillow
is importable, the library itself on fresh inputs.  Bit-exact: it is integer arithmetic."""
import hashlib
import os

import numpy as np
import pytest

from oracle import pil_resize as pr
from oracle.gen_golden_resize import CASES, frame


def test_oracle_matches_pillow_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "resize.npz"))
    assert [tuple(int(v) for v in r) for r in g["cases"]] == [c[:5] for c in CASES]
    for i, (hi, wi, ho, wo, seed, kind) in enumerate(CASES):
        got = np.ascontiguousarray(pr.resize_bilinear_u8(frame(hi, wi, seed, kind), ho, wo))
        assert got.shape == (ho, wo, 3) and got.dtype == np.uint8
        np.testing.assert_array_equal(got[:8], g[f"top{i}"])
        np.testing.assert_array_equal(got[-8:], g[f"bot{i}"])
        assert hashlib.sha256(got.tobytes()).hexdigest() == str(g[f"sha{i}"]), f"case {i} ({kind} {hi}x{wi} -> {ho}x{wo})"


@pytest.mark.parametrize("shape", [(37, 53, 20, 31), (20, 31, 37, 53), (48, 64, 48, 33), (50, 70, 25, 70), (3, 5, 7, 2)])
def test_oracle_matches_pillow_live(shape):
    Image = pytest.importorskip("PIL.Image")
    hi, wi, ho, wo = shape
    img = np.random.default_rng(sum(shape)).integers(0, 256, (hi, wi, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((wo, ho), Image.BILINEAR))
    np.testing.assert_array_equal(pr.resize_bilinear_u8(img, ho, wo), ref)


def test_batched_frames_resize_like_single_ones():
    rng = np.random.default_rng(0)
    batch = rng.integers(0, 256, (2, 3, 40, 60, 3), dtype=np.uint8)
    out = pr.resize_bilinear_u8(batch, 16, 32)
    assert out.shape == (2, 3, 16, 32, 3)
    np.testing.assert_array_equal(out[1, 2], pr.resize_bilinear_u8(batch[1, 2], 16, 32))


@pytest.mark.parametrize("sizes", [(1241, 512), (376, 256), (1226, 512), (64, 96), (7, 7), (1, 3)])
def test_library_host_table_equals_oracle(sizes):
    """The library's host-side coefficient table (what odevio_resize_u8 uploads) against the oracle's, entry by entry."""
    import ctypes
    from odevio_amd import _lib
    lib = _lib.load()
    n_in, n_out = sizes
    bounds, kk = pr.coeffs(n_in, n_out)
    ks = ctypes.c_int32()
    b = (ctypes.c_int32 * (2 * n_out))()
    k = (ctypes.c_int32 * (n_out * kk.shape[1]))()
    rc = lib.odevio_resize_table(n_in, n_out, ctypes.cast(ctypes.pointer(ks), ctypes.c_void_p), ctypes.cast(b, ctypes.c_void_p),
                                 ctypes.cast(k, ctypes.c_void_p), n_out * kk.shape[1])
    assert rc == 0 and ks.value == kk.shape[1]
    np.testing.assert_array_equal(np.asarray(b[:]).reshape(n_out, 2), bounds)
    np.testing.assert_array_equal(np.asarray(k[:]).reshape(n_out, -1), kk)
