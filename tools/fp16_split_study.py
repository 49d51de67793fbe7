#!/usr/bin/env python3
"""Numerics study for the next inner loop of the tile engine: an fp32 product out of THREE
f16 MFMAs instead of six bf16 MFMAs.

    a*s = h1 + h2 + r,  h1 = f16(a*s), h2 = f16(a*s - h1)  (round to nearest),  |r| <= 2^-24 |a*s|
    a*b ~ (h1(a) h1(b) + h1(a) h2(b) + h2(a) h1(b)) / (s_a s_b)            [h2 h2 dropped: <= 2^-24 |ab|]

f16 has 11 significant bits, so two terms carry 22-24 bits and every f16 x f16 product is exact in fp32; its 5-bit
exponent needs a power-of-two scale s per operand ROW (constant along K, removed exactly in the epilogue).  Everything
here is exact emulation in numpy (float16 conversion rounds to nearest even; products and sums in float64), so the
figures isolate the error of the SPLIT; the fp32 accumulation error of the matrix cores comes on top for every variant
alike.  Run: python tools/fp16_split_study.py"""
import numpy as np


def pow2_scale(absmax, top=14):
    return np.exp2(top - np.floor(np.log2(np.maximum(absmax, 1e-37)))).astype(np.float32)


def split_f16(x, s):
    xs = (x * s).astype(np.float32)
    h1 = xs.astype(np.float16)
    r = (xs - h1.astype(np.float32)).astype(np.float32)
    h2 = r.astype(np.float16)
    return h1.astype(np.float64), h2.astype(np.float64)


def f16x3(a, b, rowwise=True):
    if rowwise:
        sa, sb = pow2_scale(np.abs(a).max(1, keepdims=True)), pow2_scale(np.abs(b).max(0, keepdims=True))
    else:
        sa, sb = pow2_scale(np.abs(a).max()), pow2_scale(np.abs(b).max())
    (a1, a2), (b1, b2) = split_f16(a, sa), split_f16(b, sb)
    return (a2 @ b1 + a1 @ b2 + a1 @ b1) / (np.float64(1) * sa * sb)


def split_bf16(x):
    hi = (x.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
    r = (x - hi).astype(np.float32)
    mid = (r.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
    return hi.astype(np.float64), mid.astype(np.float64), (r - mid).astype(np.float64)


def bf16x6(a, b):
    (a1, a2, a3), (b1, b2, b3) = split_bf16(a), split_bf16(b)
    return a3 @ b1 + a1 @ b3 + a2 @ b2 + a2 @ b1 + a1 @ b2 + a1 @ b1


def operands(kind, rs, M=64, N=64, K=4096):
    a, b = rs.randn(M, K).astype(np.float32), rs.randn(K, N).astype(np.float32)
    if kind == 'wide':            # 2^16 dynamic range inside every row
        a *= np.exp2(rs.uniform(-8, 8, (M, K))).astype(np.float32)
        b *= np.exp2(rs.uniform(-8, 8, (K, N))).astype(np.float32)
    elif kind == 'relu':          # post-ReLU activations x small weights
        a, b = np.maximum(a, 0) * 3, b * np.float32(0.02)
    elif kind == 'pixels':        # conv-like: post-ReLU rows (pixels) whose magnitudes span five decades, one 30x louder than all
        a = np.maximum(a, 0) * np.power(10.0, rs.uniform(-5, 0, (M, 1))).astype(np.float32)
        a[0, :] *= np.float32(30.0)
        b = b * np.float32((2.0 / K) ** 0.5)
    elif kind == 'outlier':       # bulk at 1e-6, one row of A and one column of B nine decades above it
        a, b = a * np.float32(1e-6), b * np.float32(1e-6)
        a[3, :] *= np.float32(1e9)
        b[:, 5] *= np.float32(1e9)
    return a, b


def errors(c, ref):
    e = c - ref
    rms = np.sqrt((ref ** 2).mean())
    return np.abs(e).max() / rms, np.sqrt((e ** 2).mean()) / rms, float(np.median(np.abs(e) / np.abs(ref)))


def worst_vs_bound(c, a, b, ref):
    """max over outputs of |error| / sum_k |a||b| -- the scale of fp32's own rounding bound for that output, so that quiet
    rows are judged against their own magnitude (the per-tensor scale of the activation planes)"""
    den = np.abs(a.astype(np.float64)) @ np.abs(b.astype(np.float64)) + 1e-300
    return float((np.abs(c - ref) / den).max())


def main():
    rs = np.random.RandomState(0)
    print('%-9s %-28s %10s %10s %12s %14s' % ('operands', 'evaluation', 'max/rms', 'rms/rms', 'median rel', 'max/sum|a||b|'))
    for kind in ('normal', 'wide', 'relu', 'pixels', 'outlier'):
        a, b = operands(kind, rs)
        ref = a.astype(np.float64) @ b.astype(np.float64)
        for name, c in (('fp32 matmul (host BLAS)', (a @ b).astype(np.float64)), ('bf16x6 truncation (shipped)', bf16x6(a, b)),
                        ('f16x3, scale per row', f16x3(a, b, True)), ('f16x3, scale per tensor', f16x3(a, b, False))):
            print('%-9s %-28s %10.2e %10.2e %12.2e %14.2e' % ((kind, name) + errors(c, ref) + (worst_vs_bound(c, a, b, ref),)))


if __name__ == '__main__':
    main()
