"""The arithmetic behind the default GEMM / conv inner loop ("bf16x6"), checked in numpy without any
GPU: the three-way truncation split of an fp32 number into bf16 terms is EXACT, every bf16 x bf16 product is exact in
fp32, and the six cross terms the kernel keeps reproduce the fp32 product to 2^-24.5 relative on average (2^-21 worst case,
towards zero) with the default truncation split and to 2^-28 (2^-24 worst case, zero mean) with the round-to-nearest
split of the MH_SPLIT_RN=1 build.  (The device-side counterpart is tests/test_gpu_ops.py::test_gemm_keeps_all_24_mantissa_bits /
test_gemm_error_is_fp32_rounding.)"""
import numpy as np


def split3(x):
    """the kernel's split_pair arithmetic (mfma_tile.h): hi = x & 0xffff0000, r = x - hi (exact), mid = r & 0xffff0000,
    lo = r - mid"""
    x = np.asarray(x, dtype=np.float32)
    hi = (x.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
    r = (x - hi).astype(np.float32)
    mid = (r.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
    lo = (r - mid).astype(np.float32)
    return hi, mid, lo


def is_bf16(v):
    return np.all((np.asarray(v, dtype=np.float32).view(np.uint32) & np.uint32(0xffff)) == 0)


def sample(n, seed):
    rs = np.random.RandomState(seed)
    x = (rs.randn(n) * np.exp(rs.uniform(-20, 20, n))).astype(np.float32)
    x[:8] = [1.0, -1.0, 3.0, 1.0 + 2.0 ** -23, 16777215.0, 2.0 ** -100, -7.25e-12, 0.0]
    return x


def test_split_is_exact_and_every_term_is_a_bf16():
    a = sample(200000, 0)
    hi, mid, lo = split3(a)
    assert is_bf16(hi) and is_bf16(mid) and is_bf16(lo)            # 8 significant bits each: lo needs no rounding
    np.testing.assert_array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), a.astype(np.float64))
    nz = a != 0
    assert np.all(np.abs(mid[nz]) <= np.abs(a[nz]) * 2.0 ** -7) and np.all(np.abs(lo[nz]) <= np.abs(a[nz]) * 2.0 ** -15)


def test_bf16_products_are_exact_in_fp32():
    a, b = split3(sample(50000, 1)), split3(sample(50000, 2))
    for x in a:
        for y in b:
            exact = x.astype(np.float64) * y.astype(np.float64)        # 8 x 8 significant bits: 16-bit product
            ok = np.isfinite(exact) & (np.abs(exact) < 3e38) & ((exact == 0) | (np.abs(exact) > 1e-37))
            np.testing.assert_array_equal((x * y).astype(np.float64)[ok], exact[ok])


def test_six_terms_reproduce_the_fp32_product():
    a, b = sample(200000, 3), sample(200000, 4)
    ok = (a != 0) & (b != 0) & (np.abs(a.astype(np.float64) * b.astype(np.float64)) < 1e38) & \
         (np.abs(a.astype(np.float64) * b.astype(np.float64)) > 1e-30)
    a, b = a[ok], b[ok]
    (a1, a2, a3), (b1, b2, b3) = [t.astype(np.float64) for t in split3(a)], [t.astype(np.float64) for t in split3(b)]
    kept = a3 * b1 + a1 * b3 + a2 * b2 + a2 * b1 + a1 * b2 + a1 * b1       # the kernel's six MFMA terms, smallest first
    dropped = a2 * b3 + a3 * b2 + a3 * b3
    exact = a.astype(np.float64) * b.astype(np.float64)
    np.testing.assert_allclose(kept + dropped, exact, rtol=1e-15)
    rel = np.abs(kept - exact) / np.abs(exact)
    assert rel.max() < 2.0 ** -21 and np.all(kept[exact > 0] <= exact[exact > 0])   # <= 2*2^-7*2^-15 + 2^-30 of |ab|, towards zero
    assert 2.0 ** -25 < np.mean(rel) < 2.0 ** -24           # typical: one fp32 rounding (2^-24.5), one-signed
    three = a2 * b1 + a1 * b2 + a1 * b1                    # "bf16x3": what is NOT shipped
    assert (np.abs(three - exact) / np.abs(exact)).max() > 2.0 ** -16


def rne_bf16(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32).view(np.float32)      # v_cvt_pk_bf16_f32


def split3_rne(x):
    """split_pair of the MH_SPLIT_RN=1 build: hi = rne(x), r = x - hi (exact), mid = rne(r), lo = r - mid (a bf16)"""
    x = np.asarray(x, dtype=np.float32)
    hi = rne_bf16(x)
    r = (x - hi).astype(np.float32)
    mid = rne_bf16(r)
    return hi, mid, (r - mid).astype(np.float32)


def test_round_to_nearest_split_is_exact_and_tighter():
    a, b = sample(200000, 5), sample(200000, 6)
    exact = a.astype(np.float64) * b.astype(np.float64)
    ok = (a != 0) & (b != 0) & (np.abs(exact) < 1e38) & (np.abs(exact) > 1e-30)
    a, b, exact = a[ok], b[ok], exact[ok]
    sa, sb = split3_rne(a), split3_rne(b)
    assert all(is_bf16(t) for t in sa)
    np.testing.assert_array_equal(sum(t.astype(np.float64) for t in sa), a.astype(np.float64))
    assert np.all(np.abs(sa[1]) <= np.abs(a) * 2.0 ** -8) and np.all(np.abs(sa[2]) <= np.abs(a) * 2.0 ** -16)
    (a1, a2, a3), (b1, b2, b3) = [t.astype(np.float64) for t in sa], [t.astype(np.float64) for t in sb]
    kept = a3 * b1 + a1 * b3 + a2 * b2 + a2 * b1 + a1 * b2 + a1 * b1
    rel = (kept - exact) / exact
    assert np.abs(rel).max() < 2.0 ** -23.9 and np.abs(rel).mean() < 2.0 ** -27.5 and abs(rel.mean()) < 2.0 ** -33


def test_f16x3_row_scaled_split_study():
    """tools/fp16_split_study.py, the numerics case for the next inner loop: three f16 MFMAs with a
    power-of-two scale per operand row are as accurate as the shipped six bf16 MFMAs (both far below the fp32
    accumulation error); a single scale per tensor is not robust to rows that differ by many decades"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('fp16_split_study', os.path.join(root, 'tools', 'fp16_split_study.py'))
    st = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(st)
    rs = np.random.RandomState(0)
    for kind in ('normal', 'wide', 'relu', 'outlier'):
        a, b = st.operands(kind, rs, M=32, N=32, K=2048)
        ref = a.astype(np.float64) @ b.astype(np.float64)
        e_fp32 = st.errors((a @ b).astype(np.float64), ref)
        e_bf = st.errors(st.bf16x6(a, b), ref)
        e_row = st.errors(st.f16x3(a, b, True), ref)
        e_tensor = st.errors(st.f16x3(a, b, False), ref)
        assert e_row[1] < 2.0 * e_bf[1] + 1e-12 and e_row[1] < 0.5 * e_fp32[1]            # rms: same class as bf16x6, below fp32
        assert e_row[2] < 2.0 * e_bf[2] and e_row[2] < 1.5e-7                             # median relative error ~ 2^-24
        if kind == 'outlier':
            assert e_tensor[2] > 1e-4                                                     # per-tensor scaling loses the bulk
    # the split itself: h1 + h2 reproduces a*s to 2^-22 relative, and h1, h2 are exact f16 values below 65504
    a = sample(100000, 7)
    a = a[(np.abs(a) > 1e-30) & (np.abs(a) < 1e30)]
    s = st.pow2_scale(np.abs(a).max())
    big = a[np.abs(a) * s > 2.0 ** -2]                       # elements whose second term stays a normal f16
    h1, h2 = st.split_f16(big, s)
    assert np.abs(h1).max() <= 65504
    assert np.all(np.abs(h1 + h2 - big.astype(np.float64) * float(s)) <= 2.0 ** -22 * np.abs(big.astype(np.float64)) * float(s))


def test_row_exponent_bit_formula_of_the_f16x3_engine():
    """mfma_tile.h::row_exponent (experimental MH_SPLIT_F16 build) takes the exponent from the float's bit pattern; it
    must put every row maximum into [2^14, 2^15) -- below the f16 overflow threshold even after rounding -- and agree
    with the floor(log2) form the numerics study uses"""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, 'neural-motifs_amd', 'csrc', 'mfma_tile.h')).read()
    body = hdr[hdr.index('int row_exponent(unsigned absmax_bits)'):]
    body = body[:body.index('\n}')]
    assert 'const int biased = (int)(absmax_bits >> 23) & 0xff;' in body
    assert 'if (biased == 0 || biased == 0xff) return 0;' in body and 'return 14 - (biased - 127);' in body

    def row_exponent(x):
        bits = np.abs(np.asarray(x, dtype=np.float32)).view(np.uint32)
        biased = ((bits >> 23) & 0xff).astype(np.int64)
        return np.where((biased == 0) | (biased == 0xff), 0, 14 - (biased - 127))

    m = np.abs(sample(100000, 11))
    m = m[(m > 1e-37) & np.isfinite(m)]
    e = row_exponent(m)
    scaled = np.ldexp(m.astype(np.float64), e)
    assert scaled.min() >= 2.0 ** 14 and scaled.max() < 2.0 ** 15
    assert np.all(np.isfinite(np.ldexp(m, e).astype(np.float16))) and np.ldexp(m, e).astype(np.float16).max() <= 32768
    np.testing.assert_array_equal(e, 14 - np.floor(np.log2(m.astype(np.float64))).astype(np.int64))
    assert row_exponent(np.float32(0)) == 0 and row_exponent(np.float32(np.inf)) == 0 and row_exponent(np.float32(1e-45)) == 0
