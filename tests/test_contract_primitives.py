"""The written contracts the oracle and the GPU both implement (DESIGN.md sections 4-6): Philox4x32-10
known answers, uniform/normal conversions, the elementary functions, index sampling, the reduction
order.  CPU only; exercised on the GPU through the bit-exact trace comparisons."""
import math

import numpy as np

from oracle import oracle as O


def test_philox4x32_10_known_answers():
    """Random123 kat_vectors (philox4x32, 10 rounds)."""
    assert [hex(x) for x in O.philox(0, 0, 0, 0, 0)] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    assert [hex(x) for x in O.philox(0xffffffffffffffff, *[0xffffffff] * 4)] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(x) for x in O.philox(0x299f31d0a4093822, 0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344)] == \
        ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_uniform_conversions():
    assert O.u53(0, 0) == 0.0 and O.u53(0xffffffff, 0xffffffff) == 1.0 - 2.0 ** -53
    assert O.u32(0) == 2.0 ** -33 and O.u32(0xffffffff) == 1.0 - 2.0 ** -33
    assert O.stream_id(O.K_DIM, 3, 1) == (2 | 3 << 4 | 1 << 12)


def test_exp_log_accuracy_and_special_values():
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-700, 700, 20000), rng.uniform(-3, 3, 20000)])
    ulp = max(abs(O.exp(x) - math.exp(x)) / (math.ulp(math.exp(x))) for x in xs)
    assert ulp <= 1.5
    xs = np.concatenate([np.exp(rng.uniform(-700, 700, 20000)), rng.uniform(0.5, 2, 20000), [5e-324, 1e-310]])
    ulp = max(abs(O.log(x) - math.log(x)) / math.ulp(math.log(x)) for x in xs if x != 1.0)
    assert ulp <= 2.0
    assert O.exp(-np.inf) == 0.0 and O.exp(1000.0) == np.inf and np.isnan(O.exp(np.nan)) and O.exp(0.0) == 1.0
    assert O.log(0.0) == -np.inf and O.log(1.0) == 0.0 and np.isnan(O.log(-1.0)) and O.log(np.inf) == np.inf


def _normal_pair_exact(w1, w2):
    """contract v2 of the normal pair, evaluated in binary64: u = ((w1 >> 8) | 1) 2^-24, theta = pi/4 + (pi/2) y with
    y = (w2 >> 9) 2^-23 - 1/2, signs from bit 0 of w2 (cosine branch) and bit 0 of w1 (sine branch)"""
    w1, w2 = np.asarray(w1, dtype=np.uint64), np.asarray(w2, dtype=np.uint64)
    u = ((w1 >> np.uint64(8)) | np.uint64(1)).astype(float) * 2.0 ** -24
    r = np.sqrt(-2 * np.log(u))
    th = np.pi / 4 + np.pi / 2 * ((w2 >> np.uint64(9)).astype(float) * 2.0 ** -23 - 0.5)
    s0 = np.where(w2 & np.uint64(1), -1.0, 1.0)
    s1 = np.where(w1 & np.uint64(1), -1.0, 1.0)
    return s0 * r * np.cos(th), s1 * r * np.sin(th)


def test_normal32_is_a_standard_normal():
    rng = np.random.default_rng(2)
    w = rng.integers(0, 2 ** 32, size=(100000, 2), dtype=np.uint64)
    w[:500, 0] = 2 ** 32 - 1 - rng.integers(0, 2 ** 14, 500)             # u -> 1: the radius goes to 0 without cancellation
    w[500:1000, 0] = rng.integers(0, 2 ** 12, 500)                       # u -> 2^-24: the far tail
    z = np.array([O.normal32(a, b) for a, b in w])
    zs = np.array([O.normal32_sin(a, b) for a, b in w])
    e0, e1 = _normal_pair_exact(w[:, 0], w[:, 1])
    assert np.all(np.isfinite(z)) and np.all(np.isfinite(zs))
    np.testing.assert_allclose(z, e0, atol=1e-6)                         # binary32 grade: a few ulp of the largest values
    np.testing.assert_allclose(zs, e1, atol=1e-6)
    assert np.abs(np.concatenate([z, zs])).max() <= np.sqrt(2 * 24 * np.log(2)) + 1e-5
    z, zs = z[1000:], zs[1000:]
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01 and abs(((z - z.mean()) ** 4).mean() / z.var() ** 2 - 3) < 0.1
    # the sine branch serves the odd dimension of a pair: same law, uncorrelated with the cosine branch
    assert abs(zs.mean()) < 0.01 and abs(zs.std() - 1) < 0.01 and abs(np.corrcoef(z, zs)[0, 1]) < 0.01
    assert abs(np.corrcoef(z ** 2, zs ** 2)[0, 1]) < 0.01


def test_normal32_distribution_at_scale():
    """2^25 values from consecutive Philox counters: moments, Kolmogorov-Smirnov distance and tail masses of N(0,1)."""
    import ctypes as C
    from scipy import stats
    L = O.lib()
    L.orc_normal32_fill.argtypes = [C.c_uint64, C.c_int64, C.c_void_p]
    n = 1 << 24
    out = np.empty(2 * n, np.float32)
    L.orc_normal32_fill(20260929, n, out.ctypes.data)
    z = out.astype(np.float64)
    m = len(z)
    assert abs(z.mean()) < 5 / np.sqrt(m) and abs(z.var() - 1) < 5 * np.sqrt(2. / m)
    assert abs(stats.skew(z)) < 5 * np.sqrt(6. / m) and abs(stats.kurtosis(z)) < 5 * np.sqrt(24. / m)
    assert stats.kstest(z[:4000000], 'norm').statistic < 1.63 / np.sqrt(4000000)          # the 1 % critical value
    for t in (1, 2, 3, 4):
        pr = 2 * stats.norm.sf(t)
        assert abs((np.abs(z) > t).mean() - pr) < 5 * np.sqrt(pr / m)
    assert abs(np.corrcoef(z[0::2], z[1::2])[0, 1]) < 5 / np.sqrt(n)


def test_uniform16_is_low_plus_span_times_centred_uniform():
    """e ~ U(-lamb, lamb) (Dream.py:696) from a 16-bit draw: one fma, within an ulp of low + (high - low)(h + 1/2) 2^-16, inside
    the open interval, symmetric about its centre."""
    for lamb in (.05, .3):
        h = np.arange(65536)
        e = np.array([O.uniform16(x, -lamb, lamb) for x in h])
        np.testing.assert_allclose(e, -lamb + 2 * lamb * (h + 0.5) / 65536.0, rtol=0, atol=2 * np.spacing(lamb))
        assert e.min() > -lamb and e.max() < lamb and np.all(np.diff(e) > 0)
        np.testing.assert_allclose(e + e[::-1], 0, atol=4 * np.spacing(lamb))


def test_u16_is_a_centred_16_bit_uniform():
    """the crossover uniforms U_j and the e_j of Dream.py:696-700 take 16 bits each; P(U_j < CR) must equal CR
    exactly for the CR values the sampler uses (m/nCR with the centred grid, for any nCR dividing 2^16 or not
    the error is < 2^-16)."""
    h = np.arange(65536)
    u = np.array([O.u16(x) for x in h])
    np.testing.assert_array_equal(u, (h + 0.5) / 65536.0)
    for ncr in (1, 2, 3, 4, 5, 7):
        for m in range(1, ncr + 1):
            assert abs((u < m / ncr).mean() - m / ncr) < 2.0 ** -16


def test_sample_distinct_is_uniform_without_replacement():
    """stands in for random.sample(range(M), n) (Dream.py:662); the reference's own test of this
    property is test_dream.py:160-200 (M=2 returns both rows)."""
    rng = np.random.default_rng(3)
    assert sorted(O.sample_distinct(rng.integers(0, 2 ** 32, 2, dtype=np.uint64), 2)) == [0, 1]
    counts = np.zeros((5, 5))
    for _ in range(20000):
        a, b = O.sample_distinct(rng.integers(0, 2 ** 32, 2, dtype=np.uint64), 5)
        assert a != b
        counts[a, b] += 1
    off = counts[~np.eye(5, dtype=bool)]
    assert np.all(np.abs(off / 20000 - 1 / 20.) < 0.01)
    s = O.sample_distinct(rng.integers(0, 2 ** 32, 6, dtype=np.uint64), 6)
    assert sorted(s) == list(range(6))


def test_invcdf_and_wave_dot():
    assert O.invcdf([0.1, 0.9], 0.05) == 0 and O.invcdf([0.1, 0.9], 0.5) == 1 and O.invcdf([0.3, 0.3, 0.4], 0.999999) == 2
    rng = np.random.default_rng(4)
    for d in (1, 7, 100, 129, 1000):
        a, b = rng.normal(size=d), rng.normal(size=d)
        assert abs(O.wave_dot(a, b) - float(np.dot(a, b))) <= 1e-12 * max(1.0, abs(np.dot(a, b))) + 1e-13 * d
