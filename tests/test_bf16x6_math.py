"""The arithmetic of the small-product engine ("bf16x6"; csrc/mfma_tile.h: split_pair_bf16, mma_frags),
checked in numpy without a GPU: an fp32 number IS the sum of three bf16 terms when each is rounded to nearest even, the three
cross terms the kernel drops are below 2^-24 of |a||b|, and the six-term evaluation accumulated in fp32 is as close to the
float64 product as a plain fp32 evaluation -- with no scale anywhere, over the whole exponent range.  (Device-side counterparts:
tests/test_gpu_ops.py::test_small_product_engine_copies_24_bit_operands_exactly / ..._error_is_fp32_rounding.)"""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bf16 (ties to even), returned as float32: what v_cvt_pk_bf16_f32 does to each half"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    b1 = bf16_rne(x)
    r1 = (x - b1).astype(np.float32)              # exact in fp32
    b2 = bf16_rne(r1)
    r2 = (r1 - b2).astype(np.float32)             # exact
    b3 = bf16_rne(r2)
    return b1, b2, b3, r2


def sample(n, seed, lo=-30, hi=30):
    rs = np.random.RandomState(seed)
    x = (rs.randn(n) * np.exp2(rs.uniform(lo, hi, n))).astype(np.float32)
    x[:8] = [1.0, -1.0, 3.0, 1.0 + 2.0 ** -23, 16777215.0, 2.0 ** -100, -7.25e-12, 0.0]
    return x


def test_three_bf16_terms_reproduce_an_fp32_number_exactly():
    x = sample(200000, 3)
    b1, b2, b3, r2 = split3(x)
    assert np.array_equal(b3, r2), 'the second remainder has more than 8 significant bits somewhere'
    assert np.array_equal((b1.astype(np.float64) + b2.astype(np.float64) + b3.astype(np.float64)).astype(np.float32), x)
    assert np.array_equal(b1.astype(np.float64) + b2.astype(np.float64) + b3.astype(np.float64), x.astype(np.float64))
    # the terms shrink by 2^-8 each (round to nearest: half an ulp of an 8-bit significand)
    nz = x != 0
    assert np.all(np.abs(b2[nz]) <= 2.0 ** -8 * np.abs(x[nz]) * (1 + 2.0 ** -7))
    assert np.all(np.abs(b3[nz]) <= 2.0 ** -16 * np.abs(x[nz]) * (1 + 2.0 ** -6))
    # every term is a bf16 value: the low 16 bits of its pattern are zero
    for t in (b1, b2, b3):
        assert not np.any(t.view(np.uint32) & 0xffff)


def test_dropped_cross_terms_are_below_fp32_rounding():
    a, b = sample(100000, 5, -20, 20), sample(100000, 6, -20, 20)
    a1, a2, a3, _ = split3(a)
    b1, b2, b3, _ = split3(b)
    kept = (a3.astype(np.float64) * b1 + a1.astype(np.float64) * b3 + a2.astype(np.float64) * b2 + a2.astype(np.float64) * b1 +
            a1.astype(np.float64) * b2 + a1.astype(np.float64) * b1)                     # the kernel's six terms (each product exact in fp32)
    exact = a.astype(np.float64) * b.astype(np.float64)
    nz = exact != 0
    assert np.all(np.abs(kept - exact)[nz] <= 2.0 ** -23 * np.abs(exact)[nz])
    assert np.median(np.abs(kept - exact)[nz] / np.abs(exact)[nz]) < 2.0 ** -26


def test_six_term_product_is_as_accurate_as_fp32_without_any_scale():
    """rows whose magnitudes span sixty binary orders: the f16x3 evaluation needs a power-of-two scale per row for this, the
    bf16 terms carry fp32's exponent"""
    rs = np.random.RandomState(9)
    M, N, K = 24, 20, 1024
    a = (rs.randn(M, K) * np.exp2(rs.randint(-30, 30, (M, 1)))).astype(np.float32)
    b = (rs.randn(N, K) * np.exp2(rs.randint(-30, 30, (N, 1)))).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64).T
    pa, pb = split3(a)[:3], split3(b)[:3]
    acc = np.zeros((M, N), dtype=np.float32)
    for k0 in range(0, K, 16):                                  # per k-tile: six "MFMAs", smallest term first, fp32 accumulate
        for ia, ib in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):
            part = (pa[ia][:, k0:k0 + 16].astype(np.float64) @ pb[ib][:, k0:k0 + 16].astype(np.float64).T)
            acc = (acc.astype(np.float64) + part).astype(np.float32)
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64).T
    err = np.abs(acc.astype(np.float64) - ref) / scale
    err32 = np.abs((a @ b.T).astype(np.float64) - ref) / scale
    # (this emulation rounds the accumulator to fp32 after every one of the 6 x 64 partial products; numpy's fp32 matmul, the
    # yardstick, accumulates blocks: the two are the same class -- a small multiple of 2^-24 of sum |a||b|)
    assert err.max() < 3e-7 and err.max() < 4.0 * err32.max() + 1e-9
    assert np.sqrt(np.mean(err ** 2)) < 4.0 * np.sqrt(np.mean(err32 ** 2)) + 1e-9 and np.sqrt(np.mean(err ** 2)) < 2.0 ** -24
