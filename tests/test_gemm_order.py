"""The block -> (tile, K slice) mapping of the plane GEMMs (csrc/pl_gemm.hip: gemm_item, plan_order, plan_launch), replayed on the
host through the C ABI (mh_debug_pl_item: the kernels' own __host__ __device__ function, no device needed).

Round 6 deals the work items to the eight XCDs so that blocks which run at the same time share operand panels in that XCD's L2:
K slices on their own XCDs (>= 8 slices), bands of the tile space otherwise.  Whatever the order, every (tile, slice) must be
worked on exactly once -- a hole is a wrong product, a duplicate is a race on C.
"""
import ctypes

import pytest


SHAPES = [
    (1536, 4096, 25088, 0),      # fc6 forward (union boxes), planner's split
    (1536, 4096, 25088, 8),
    (1536, 4096, 25088, 4),
    (1536, 4096, 25088, 5),      # forced odd split: falls back to the round-3 numbering
    (1536, 25088, 4096, 0),      # fc6 input gradient
    (4096, 25088, 1536, 0),      # fc6 weight gradient
    (1536, 4096, 4096, 0),       # fc7
    (1536, 4096, 4096, 2),
    (120, 4096, 25088, 16),      # object fc6: one row of tiles
    (4096, 4096, 4096, 1),
    (300, 260, 2048, 0), (257, 513, 592, 1), (1000, 1000, 4096, 2), (130, 60, 1024, 0), (2000, 129, 5000, 4), (129, 2000, 5000, 4),
]


def _items(L, M, N, K, sk):
    out = (ctypes.c_int * 10)()
    assert L.mh_debug_pl_item(M, N, K, sk, 0, out) == 0
    gx, gy = out[0], out[1]
    rows = []
    for b in range(gx * gy):
        assert L.mh_debug_pl_item(M, N, K, sk, b, out) == 0
        rows.append(tuple(out))
    return rows


@pytest.mark.parametrize('order', [1, 0])
@pytest.mark.parametrize('M,N,K,sk', SHAPES)
def test_every_tile_and_slice_exactly_once(so_path, M, N, K, sk, order):
    L = ctypes.CDLL(so_path)           # the library itself: another test of the session may have put the CPU shim behind _hip.lib()
    L.mh_debug_pl_item.argtypes = [ctypes.c_int] * 4 + [ctypes.c_longlong, ctypes.POINTER(ctypes.c_int)]
    L.mh_debug_pl_order(order)
    try:
        rows = _items(L, M, N, K, sk)
    finally:
        L.mh_debug_pl_order(1)
    gx, gy, _, _, _, _, tiles_m, tiles_n, splitk, used_order = rows[0]
    seen = {}
    for b, r in enumerate(rows):
        if not r[5]:
            assert used_order != 0          # only the banded orders launch blocks without work
            continue
        tm, tn, z = r[2], r[3], r[4]
        assert 0 <= tm < tiles_m and 0 <= tn < tiles_n and 0 <= z < splitk
        assert (tm, tn, z) not in seen, 'work item %s claimed by blocks %d and %d' % ((tm, tn, z), seen[(tm, tn, z)], b)
        seen[(tm, tn, z)] = b
    assert len(seen) == tiles_m * tiles_n * splitk
    if order == 0:
        assert used_order == 0
    if used_order == 1:                     # a K slice lives on one XCD
        for (tm, tn, z), b in seen.items():
            assert (b % gx) % 8 == z % 8
    if used_order == 2:                     # an XCD works inside one slice, on one contiguous band of tiles
        per_xcd = {}
        for (tm, tn, z), b in seen.items():
            per_xcd.setdefault((b % gx) % 8, []).append((tm, tn, z))
        for xcd, its in per_xcd.items():
            assert len({z for _, _, z in its}) == 1
            tms, tns = sorted({i[0] for i in its}), sorted({i[1] for i in its})
            assert tms == list(range(tms[0], tms[-1] + 1)) and tns == list(range(tns[0], tns[-1] + 1))
            assert len(its) == len(tms) * len(tns)
        # idle blocks cost a dispatch each: the busiest XCD has at most one band row / column more than the lightest
        assert gx * gy <= 8 * (len(seen) // 8 + max(tiles_m, tiles_n))


def test_concurrent_blocks_share_panels(so_path):
    """the point of the order: the first 32 blocks an XCD receives (what it runs at once with 256x256 tiles) touch few operand
    panels -- fc6 forward: 12 tiles of one slice used to span 12 panels per slice, now 32 tiles span <= 13
    (a 6 x 5 patch and the first two tiles of the next one)."""
    L = ctypes.CDLL(so_path)           # the library itself: another test of the session may have put the CPU shim behind _hip.lib()
    L.mh_debug_pl_item.argtypes = [ctypes.c_int] * 4 + [ctypes.c_longlong, ctypes.POINTER(ctypes.c_int)]
    rows = _items(L, 1536, 4096, 25088, 8)
    gx = rows[0][0]
    if rows[0][6] * rows[0][7] != 96:       # the planner picked another tile shape: the panel count below is for 256x256
        pytest.skip('planner did not choose 256x256 tiles')
    for xcd in range(8):
        first = [r for b, r in enumerate(rows) if (b % gx) % 8 == xcd and r[5]][:32]
        assert len({r[4] for r in first}) == 1
        assert len({r[2] for r in first}) + len({r[3] for r in first}) <= 13
