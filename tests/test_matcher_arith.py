"""CPU: integer identities the matcher's kernels rely on, restated in numpy.

matcher.hip splits a pair index p into (keyframe i, source cell s) = divmod(p, n_src) with one multiplication by
floor((2^32 - 1) / n_src) and ONE correction step; the bound behind "one" (p < 2^20: at most 15 keyframes x 65 535 cells + a
1024-lane look-ahead) is checked here over every divisor class that matters."""
import numpy as np


def _split(p, n_src):
    magic = np.uint64(0xFFFFFFFF // n_src)
    q = (p.astype(np.uint64) * magic) >> np.uint64(32)                 # __umulhi
    r = p.astype(np.int64) - q.astype(np.int64) * n_src
    fix = r >= n_src
    return q.astype(np.int64) + fix, r - fix * n_src


def test_pair_index_split_is_exact_below_2_pow_20():
    rng = np.random.default_rng(0)
    divisors = sorted(set([1, 2, 3, 7, 64, 255, 256, 257, 335, 1024, 1366, 4095, 4096, 32768, 65534, 65535] +
                          rng.integers(1, 65536, size=200).tolist()))
    edge = np.array([0, 1, 2 ** 20 - 2, 2 ** 20 - 1], np.int64)
    for d in divisors:
        p = np.concatenate([edge, rng.integers(0, 2 ** 20, size=4096), np.arange(0, min(2 ** 20, 40 * d), max(1, d // 3))]).astype(np.int64)
        i, s = _split(p, d)
        np.testing.assert_array_equal(i, p // d, err_msg="n_src %d" % d)
        np.testing.assert_array_equal(s, p % d, err_msg="n_src %d" % d)


def test_pair_index_split_exhaustive_for_the_usual_sizes():
    p = np.arange(2 ** 20, dtype=np.int64)
    for d in (204, 316, 335, 336, 436, 1366, 2300):
        i, s = _split(p, d)
        assert (i == p // d).all() and (s == p % d).all(), d
