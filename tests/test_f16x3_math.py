"""The arithmetic behind the matrix-product engines ("f16x3"), checked in numpy without any GPU: the
two-term f16 split of a row-scaled fp32 number reproduces it to 2^-22, three f16 MFMAs with a power-of-two scale per
operand row are as accurate as six bf16 MFMAs on a three-term split (round 1's engine, kept here only as the numpy yardstick
of tools/fp16_split_study.py) and far below the fp32 accumulation error, and the exponent both engines derive from the row
maximum's bit pattern (mfma_tile.h and pl_tile.h: row_exponent) puts every row maximum into [2^14, 2^15).  (Device-side
counterparts: tests/test_gpu_ops.py::test_gemm_copies_23_bit_operands_exactly / test_gemm_error_is_fp32_rounding.)"""
import numpy as np


def sample(n, seed):
    rs = np.random.RandomState(seed)
    x = (rs.randn(n) * np.exp(rs.uniform(-20, 20, n))).astype(np.float32)
    x[:8] = [1.0, -1.0, 3.0, 1.0 + 2.0 ** -23, 16777215.0, 2.0 ** -100, -7.25e-12, 0.0]
    return x


def test_f16x3_row_scaled_split_study():
    """tools/fp16_split_study.py, the numerics case for the f16x3 arithmetic: three f16 MFMAs with a
    power-of-two scale per operand row are as accurate as six bf16 MFMAs on a three-term split (both far below the fp32
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
    """row_exponent (mfma_tile.h: in-loop engine, pl_tile.h: plane engine -- the same three lines) takes the exponent from the
    float's bit pattern; it must put every row maximum into [2^14, 2^15) -- below the f16 overflow threshold even after
    rounding -- and agree with the floor(log2) form the numerics study uses"""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ('mfma_tile.h', 'pl_tile.h'):
        hdr = open(os.path.join(root, 'neural-motifs_amd', 'csrc', name)).read()
        body = hdr[hdr.index('int row_exponent(unsigned absmax_bits)'):]
        body = body[:body.index('\n}')]
        assert 'const int biased = (int)(absmax_bits >> 23) & 0xff;' in body, name
        assert 'if (biased == 0 || biased == 0xff) return 0;' in body and 'return 14 - (biased - 127);' in body, name

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
