"""The addressing of the conv3x3 implicit GEMM (neural-motifs_amd/csrc/conv.hip, conv3x3_nhwc_kernel::load_tiles),
re-derived in Python for the shipped K order (16-channel chunk outer, nine taps inner) and for the tap-major order of
the MH_CONV_TAP_MAJOR=1 build:

* every (tap, channel chunk) is visited exactly once per block, for any split of the k range over blockIdx.y
  (split-K), with k-tiles past the end dead -- so both orders compute the same sum;
* planned offset + scalar offset of a staged pixel addresses exactly input[b, y+dy, x+dx, c0 + 4q .. +3], relative
  to the block origin one halo before the tile's first pixel, never negative; taps outside the image are masked;
* the reuse distance (bytes of other A data a block reads between two visits of the same 64-byte pixel segment) is
  what DESIGN.md §5 says: one k-tile in the shipped order, a whole pass over the channels in the tap-major order.
The formulas are checked against the kernel text, so the emulation cannot drift from the source silently."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, 'neural-motifs_amd', 'csrc', 'conv.hip')).read()
KBK, BM = 16, 128


def test_formulas_are_the_ones_in_the_kernel():
    assert 'const int g16 = kt / 9, tap = kt - 9 * g16, c0 = g16 * kBK;' in SRC                     # shipped order
    assert 'const int tap = min(kt / kt_per_tap, 8);' in SRC and 'const int g16 = kt - tap * kt_per_tap, c0 = g16 * kBK;' in SRC
    assert 'const int halo = (p.W + 1) * p.Cin;' in SRC
    assert 'a_off[j] = (unsigned)(r * p.Cin + 4 * ((tid + kThreads * j) & 3)) * 4u;' in SRC
    assert '(unsigned)(halo + (dy * p.W + dx) * p.Cin + c0) * 4u' in SRC
    assert re.search(r'#define MH_CONV_TAP_MAJOR 0\b', SRC)                                           # the default build


def decode(kt, kt_per_tap, tap_major):
    if tap_major:
        tap = min(kt // kt_per_tap, 8)
        return tap, kt - tap * kt_per_tap
    g16 = kt // 9
    return kt - 9 * g16, g16


@pytest.mark.parametrize('cin', [16, 32, 64, 256, 512])
@pytest.mark.parametrize('tap_major', [False, True])
def test_every_tap_and_chunk_once_under_any_split(cin, tap_major):
    kt_per_tap = cin // KBK
    total = 9 * kt_per_tap
    for splitk in (1, 2, 3, 5, 7):
        per = -(-total // splitk)
        seen = []
        for y in range(-(-total // per)):
            lo, hi = y * per, min(total, (y + 1) * per)
            kt = lo
            while kt < hi + 2:                     # the pipeline also issues the two k-tiles after the end: dead
                if kt < hi:
                    seen.append(decode(kt, kt_per_tap, tap_major))
                else:
                    tap, _ = decode(kt, kt_per_tap, tap_major)
                    assert 0 <= tap <= 8           # `1u << tap` stays defined for dead tiles
                kt += 1
        assert sorted(seen) == [(t, g) for t in range(9) for g in range(kt_per_tap)]


@pytest.mark.parametrize('B,H,W,cin', [(2, 5, 7, 16), (1, 37, 37, 32), (3, 14, 14, 64)])
def test_offsets_address_the_shifted_pixel(B, H, W, cin):
    x = np.arange(B * H * W * cin, dtype=np.int64)            # element index as value
    halo = (W + 1) * cin
    mtot = B * H * W
    kt_per_tap = cin // KBK
    for m0 in range(0, mtot, BM):
        base = m0 * cin - halo                                # element offset of the block origin (may be < 0)
        for r in (0, 1, W - 1, W, min(BM, mtot - m0) - 1, BM - 1):
            pix = m0 + r
            row_ok = pix < mtot
            rem = (pix if row_ok else 0) % (H * W)
            py, px = rem // W, rem % W
            for q in range(4):
                a_off = r * cin + 4 * q
                for kt in range(9 * kt_per_tap):
                    tap, g16 = decode(kt, kt_per_tap, False)
                    dy, dx = tap // 3 - 1, tap % 3 - 1
                    inside = row_ok and 0 <= py + dy < H and 0 <= px + dx < W
                    soff = halo + (dy * W + dx) * cin + g16 * KBK
                    assert soff >= 0 and a_off + soff >= 0    # unsigned arithmetic in the kernel
                    if inside:
                        b = pix // (H * W)
                        want = ((b * H + py + dy) * W + px + dx) * cin + g16 * KBK + 4 * q
                        assert x[base + a_off + soff] == want


def test_reuse_distance():
    """bytes of A a block stages between two reads of the same pixel's 64-byte segment (taps dx=-1 -> dx=0 of one row)"""
    for cin in (64, 256, 512):
        kt_per_tap = cin // KBK
        a_tile = BM * KBK * 4
        for tap_major, want in ((False, a_tile), (True, kt_per_tap * a_tile)):
            order = [decode(kt, kt_per_tap, tap_major) for kt in range(9 * kt_per_tap)]
            i0, i1 = order.index((0, 0)), order.index((1, 0))           # tap 0 and tap 1 of chunk 0
            assert (i1 - i0) * a_tile == want
    assert 512 // KBK * BM * KBK * 4 * 64 > 4 << 20      # tap-major, 512 channels, 64 blocks per XCD: beyond the 4 MB L2
