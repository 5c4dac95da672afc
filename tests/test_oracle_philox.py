"""Known-answer test of the oracle's Philox 4x32-10 (oracle/philox.py) against the vectors published with the algorithm
(Random123's kat_vectors: zero counter/key, all-ones, and the digits-of-pi case).  The GPU suite then compares the device's
generator (fuse_method "hard") with this restatement (tests/test_gpu_parity.py)."""
import numpy as np

from oracle import philox as ph


def _run(c, k):
    return [int(x) for x in ph.philox4x32_10(np.array(c, dtype=np.uint64), np.array(k, dtype=np.uint64))]


def test_philox4x32_10_known_answers():
    assert _run([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert _run([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert _run([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_gumbel_pairs_are_gumbel_distributed_and_reproducible():
    g = ph.gumbel_pairs(7, 3, 200000)
    assert g.shape == (200000, 2) and g.dtype == np.float32 and np.isfinite(g).all()
    assert abs(float(g.mean()) - 0.5772) < 0.01 and abs(float(g.var()) - np.pi ** 2 / 6) < 0.03      # Gumbel(0, 1): Euler-Mascheroni, pi^2 / 6
    assert np.array_equal(g[:1000], ph.gumbel_pairs(7, 3, 1000))
    assert not np.array_equal(g[:1000], ph.gumbel_pairs(7, 4, 1000)) and not np.array_equal(g[:1000], ph.gumbel_pairs(8, 3, 1000))
    assert np.array_equal(ph.gumbel_pairs(7, 3, 999), g[:999])                                       # an odd count ends inside a block
