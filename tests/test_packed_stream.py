"""CPU: the packed residual-stream format (oracle/packed_stream.py = the bit-level statement of mvlpt_amd/csrc/common.h respk_*)."""
import numpy as np

from oracle import packed_stream as P


def _values(seed=0, n=200_000):
    r = np.random.default_rng(seed)
    x = np.concatenate([r.standard_normal(n) * 3.0, r.standard_normal(n) * 300.0, r.standard_normal(n) * 1e-3,
                        r.standard_normal(1000) * 1e-6, [0.0, -0.0, 1.0, -1.0, 65504.0, -65504.0, 2.0 ** -14, 2.0 ** -24, 1.0 + 2.0 ** -11,
                                                         1.0 - 2.0 ** -12, 2048.0 + 1.0, -(1024.0 + 0.5)]]).astype(np.float32)
    return x


def test_round_trip_is_within_two_to_minus_nine_of_an_fp16_ulp():
    x = _values()
    hi, lo = P.pack(x)
    y = P.unpack(hi, lo)
    err = np.abs(y.astype(np.float64) - x.astype(np.float64))
    normal = np.abs(x.astype(np.float64)) >= 2.0 ** -14
    ulp = 2.0 ** (np.floor(np.log2(np.abs(hi[normal].astype(np.float64)))) - 10)
    assert float((err[normal] / ulp).max()) <= 2.0 ** -8           # the byte is floor(d / 32): at most one step of 2^-8 ulp
    assert float((err[normal] / np.abs(x[normal])).max()) < 2.0 ** -18
    assert float(err[~normal].max()) <= 2.0 ** -25                   # fp16 subnormals / zero: hi alone, to half of 2^-24
    assert np.all(np.abs(y[normal]) <= np.abs(x[normal]))            # truncation: never larger in magnitude than x


def test_hi_plane_is_the_rounded_value_and_lo_is_signed():
    x = _values(1)
    hi, lo = P.pack(x)
    assert np.array_equal(hi, x.astype(np.float16))
    assert lo.dtype == np.int8 and lo.min() >= -128 and lo.max() <= 127
    # the byte is the signed distance to the rounded value: negative exactly where round16 went up in magnitude
    up = np.abs(hi.astype(np.float32)) > np.abs(x)
    assert np.all(lo[up] < 0) and np.all(lo[~up] >= 0)


def test_exactly_representable_values_survive_unchanged():
    x = np.arange(-2048, 2049, dtype=np.float32)
    hi, lo = P.pack(x)
    assert np.all(lo == 0) and np.array_equal(P.unpack(hi, lo), x)


def test_random_walk_of_packed_updates_stays_at_fp32_level():
    """24 residual updates (12 blocks) through the format: the accumulated error stays ~2^-15 (24 truncations of <= 2^-18 each, all
    toward zero), forty times below an fp16 stream's."""
    r = np.random.default_rng(3)
    x = r.standard_normal((64, 768)).astype(np.float32)
    exact = x.astype(np.float64)
    hi, lo = P.pack(x)
    h16 = x.astype(np.float16)
    for _ in range(24):
        dlt = (r.standard_normal(x.shape) * 0.3).astype(np.float32)
        exact = exact + dlt
        hi, lo = P.pack(P.unpack(hi, lo) + dlt)
        h16 = (h16.astype(np.float32) + dlt).astype(np.float16)
    e_packed = np.abs(P.unpack(hi, lo) - exact).max() / np.abs(exact).max()
    e_half = np.abs(h16.astype(np.float64) - exact).max() / np.abs(exact).max()
    assert e_packed < 4e-5 and e_half > 20 * e_packed


def test_fold_weight_matches_layernorm_algebra():
    r = np.random.default_rng(4)
    K, N = 256, 128
    W = (r.standard_normal((N, K)) * K ** -0.5).astype(np.float16)
    g = (1.0 + 0.2 * r.standard_normal(K)).astype(np.float32)
    b = (0.1 * r.standard_normal(K)).astype(np.float32)
    x = (r.standard_normal((50, K)) * 2.0 + 0.7).astype(np.float32)
    Wg, cs = P.fold_weight(W, g)
    mean = x.astype(np.float64).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(x.astype(np.float64).var(-1, keepdims=True) + 1e-5)
    want = ((x - mean) * rstd * g + b) @ W.astype(np.float64).T
    got = rstd * (x.astype(np.float64) @ Wg.astype(np.float64).T - mean * cs) + W.astype(np.float64) @ b
    assert np.abs(got - want).max() < 2e-3 * np.abs(want).max()


def test_values_beyond_the_fp16_range_saturate_instead_of_becoming_infinite():
    """|x| > 65504 used to give hi = inf and a reconstructed 3.4e38; the format saturates x first (a fp16 tower cannot carry
    such a value anyway: it is the next GEMM's A operand)."""
    x = np.array([7e4, -7e4, 1e9, -3e38, 65504.0, 65519.9, np.inf, -np.inf], dtype=np.float32)
    hi, lo = P.pack(x)
    assert np.all(np.isfinite(hi.astype(np.float32))) and np.all(np.abs(hi.astype(np.float32)) == 65504.0)
    assert np.array_equal(P.unpack(hi, lo), np.sign(x) * np.float32(65504.0))
    hi, lo = P.pack(np.array([np.nan], dtype=np.float32))
    assert np.isnan(hi[0])
