"""Philox 4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) in numpy, and the Gumbel
noise the device draws from it for FusionModule "hard" (odevio_amd/csrc/pose.hip: hard_mask_kernel).  TEST INFRASTRUCTURE ONLY.

Pinned by the known-answer vectors of the authors' Random123 distribution (tests/test_oracle_philox.py); the GPU test then holds
the device's generator to this restatement bit for bit (the uniform bits) and to float rounding (the logarithms)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter [...,4] uint32, key [...,2] uint32 (broadcastable) -> [...,4] uint32."""
    c = np.array(counter, dtype=np.uint64) & MASK
    k = np.array(key, dtype=np.uint64) & MASK
    k = np.broadcast_to(k, c.shape[:-1] + (2,))
    k0, k1 = k[..., 0].copy(), k[..., 1].copy()
    for _ in range(10):
        p0 = M0 * c[..., 0]
        p1 = M1 * c[..., 2]
        n0 = ((p1 >> np.uint64(32)) ^ c[..., 1] ^ k0) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c[..., 3] ^ k1) & MASK
        n3 = p0 & MASK
        c = np.stack([n0, n1, n2, n3], axis=-1)
        k0 = (k0 + np.uint64(W0)) & MASK
        k1 = (k1 + np.uint64(W1)) & MASK
    return c.astype(np.uint32)


def gumbel_pairs(seed, call, n):
    """The device's noise for draw `call` of `seed`: out [n, 2] float32.  Element pair i = (2i, 2i+1) shares counter
    (i lo, i hi, call lo, call hi); the four words give (g0, g1) of element 2i and (g0, g1) of element 2i+1;
    u = ((bits >> 8) + 0.5) / 2^24, g = -log(-log(u)) in float32."""
    i = np.arange((n + 1) // 2, dtype=np.uint64)
    ctr = np.stack([i & MASK, i >> np.uint64(32), np.full_like(i, call & 0xFFFFFFFF), np.full_like(i, (call >> 32) & 0xFFFFFFFF)], axis=-1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    bits = philox4x32_10(ctr, key).reshape(-1)[:2 * n]
    u = ((bits >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    g = -np.log(-np.log(u, dtype=np.float32), dtype=np.float32)
    return g.reshape(n, 2)


def dropout_factors(seed, call, p, n):
    """What the device's train-mode Dropout(p) multiplies each of n elements by under draw `call` of `seed` (odevio_amd/csrc/philox.h):
    element e uses word e % 4 of counter block (e // 4 lo, e // 4 hi, call lo, call hi); kept iff that word >= floor(p * 2^32);
    kept elements are scaled by float32(1) / (float32(1) - float32(p)).  out [n] float32."""
    if p <= 0:
        return np.ones(n, dtype=np.float32)
    i = np.arange((n + 3) // 4, dtype=np.uint64)
    ctr = np.stack([i & MASK, i >> np.uint64(32), np.full_like(i, call & 0xFFFFFFFF), np.full_like(i, (call >> 32) & 0xFFFFFFFF)], axis=-1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    bits = philox4x32_10(ctr, key).reshape(-1)[:n]
    thr = np.uint32(min(int(float(np.float32(p)) * 4294967296.0), 0xFFFFFFFF))
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return np.where(bits >= thr, scale, np.float32(0.0)).astype(np.float32)
