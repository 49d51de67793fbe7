"""The bf16-plane LDS image of the tile engine (neural-motifs_amd/csrc/mfma_tile.h), re-derived in Python:

* index arithmetic: what the staging threads write (store_wm / store_km) is exactly what the MFMA lanes read
  (fetch_frags), for every operand width and both global orientations;
* bank behaviour under the service-group rules of MI355X_MICROARCH.md §LDS: ds_read_b128 is served in four fixed
  16-lane groups over 64 banks, ds_write_b64 in four contiguous 16-lane groups over 32 banks, ds_write_b32 in two
  32-lane halves over 32 banks; a conflict = two different addresses of one group on one bank.

These are the claims DESIGN.md §3.2 makes (fragment reads and K-contiguous writes conflict-free, k-major writes at
most 2-way); the constants below must match the header (checked against its text)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = open(os.path.join(ROOT, 'neural-motifs_amd', 'csrc', 'mfma_tile.h')).read()
K_ROW_DW = int(re.search(r'constexpr int kRowDw = (\d+);', HDR).group(1))

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
               [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS = B128_GROUPS + [[l + 32 for l in g] for g in B128_GROUPS]


def swz(r):
    assert 'return (r >> 3) & 1;' in HDR            # plane_swz in the header
    return (r >> 3) & 1


def slot_addr(row, plane, half):
    """dword address of the 16-byte slot holding k = 8*half .. 8*half+7 of `plane` in `row`"""
    return row * K_ROW_DW + 4 * ((2 * plane + half) ^ swz(row))


def km_row(w):
    return 64 * (w >> 6) + 32 * (w & 1) + ((w & 63) >> 1)


def worst_conflict(groups, addr_of_lane, width_dw, nbanks):
    worst = 1
    for grp in groups:
        per_bank = {}
        for lane in grp:
            a = addr_of_lane(lane)
            if a is None:
                continue
            for d in range(width_dw):
                per_bank.setdefault((a + d) % nbanks, set()).add(a)
        worst = max(worst, max(len(v) for v in per_bank.values()))
    return worst


def test_row_is_three_planes_of_eight_dwords():
    assert K_ROW_DW == 24
    assert 'struct PlaneFrags' in HDR and 'ds_read_b128' in HDR


def _image(WD, wm):
    """simulate the staging writes of one k-tile: returns {dword address: (tile row or column, k_even, k_odd, plane)}"""
    lds = {}

    def put(addr, val):
        assert addr not in lds, 'two threads write dword %d' % addr
        lds[addr] = val
    if wm:        # store_wm: float4 f = tid + 256 j -> row f >> 2, k-quad f & 3; two packed dwords per plane
        for tid in range(256):
            for j in range(WD * 16 // 1024):
                f = tid + 256 * j
                r, kq = f >> 2, f & 3
                for p in range(3):
                    base = slot_addr(r, p, kq >> 1) + 2 * (kq & 1)
                    put(base, (r, 4 * kq, 4 * kq + 1, p))
                    put(base + 1, (r, 4 * kq + 2, 4 * kq + 3, p))
    else:         # store_km: task t -> w-quad q = 8*(t>>6) + (t&7), k-pair kp = (t>>3)&7; one dword per (w, plane)
        for tid in range(256):
            for jt in range((2 * WD + 255) // 256):
                t = tid + 256 * jt
                if 2 * WD < 256 and t >= 2 * WD:
                    continue
                q, kp = 8 * (t >> 6) + (t & 7), (t >> 3) & 7
                for j in range(4):
                    w = 4 * q + j
                    for p in range(3):
                        put(slot_addr(km_row(w), p, kp >> 2) + (kp & 3), (w, 2 * kp, 2 * kp + 1, p))
    return lds


def test_what_is_staged_is_what_the_mfma_lanes_read():
    for WD in (64, 128, 256):
        for wm in (True, False):
            lds = _image(WD, wm)
            assert len(lds) == WD * 24
            for w0 in range(0, WD, 64):                       # a wave's 64 rows / columns
                for lane in range(64):
                    i, g = lane & 31, lane >> 5
                    for s in range(2):                         # the two 32-wide MFMA sub-tiles
                        for p in range(3):
                            a = slot_addr(w0 + 32 * s + i, p, g)
                            expect_coord = (w0 + i + 32 * s) if wm else (w0 + 2 * i + s)      # tile_coord<WM>
                            for d in range(4):
                                coord, k0, k1, plane = lds[a + d]
                                assert (coord, k0, k1, plane) == (expect_coord, 8 * g + 2 * d, 8 * g + 2 * d + 1, p)


def test_fragment_reads_are_conflict_free():
    for row0 in (0, 32, 64, 96, 128, 192):
        for p in range(3):
            assert worst_conflict(B128_GROUPS, lambda l: slot_addr(row0 + (l & 31), p, l >> 5), 4, 64) == 1


def test_k_contiguous_staging_writes_are_conflict_free():
    groups = [list(range(16 * k, 16 * k + 16)) for k in range(4)]
    for t0 in range(0, 1024, 64):                              # every wave of every float4 slice of a 256-row tile
        for p in range(3):
            def addr(l, t0=t0, p=p):
                f = t0 + l
                return slot_addr(f >> 2, p, (f & 3) >> 1) + 2 * (f & 1)
            assert worst_conflict(groups, addr, 2, 32) == 1


def test_k_major_staging_writes_are_at_most_two_way():
    halves = [list(range(32)), list(range(32, 64))]
    worst = 1
    for wave in range(8):
        for j in range(4):
            for p in range(3):
                def addr(l, wave=wave, j=j, p=p):
                    t = 64 * wave + l
                    q, kp = 8 * (t >> 6) + (t & 7), (t >> 3) & 7
                    return slot_addr(km_row(4 * q + j), p, kp >> 2) + (kp & 3)
                worst = max(worst, worst_conflict(halves, addr, 1, 32))
    assert worst == 2


def test_packed_plane_chunk_writes_tile_the_banks_in_f16x3():
    """conv weights arrive as planes and are staged with ds_write_b128 (plan_planes / store_planes): eight contiguous lanes
    per LDS cycle over 32 banks.  An f16x3 row uses 64 of its 96 bytes, so consecutive rows (0 and 96 mod 128) overlap on 8
    banks; the row permutation of the header pairs rows whose spans tile the 128 bytes (PMC before: SQ_LDS_BANK_CONFLICT =
    20 % of the conv's LDS cycles)."""
    m = re.search(r'r = \(r & ~7\) \| \(\(0x([0-9a-f]+)u >> \(4 \* \(r & 7\)\)\) & 7\);', HDR)
    assert m, 'row permutation of plan_planes not found'
    code = int(m.group(1), 16)
    perm = [(code >> (4 * i)) & 7 for i in range(8)]
    assert sorted(perm) == list(range(8))
    chunks = 4                                         # kPlaneChunks of the f16x3 build: h1 | h2, 2 x 16 B each
    assert 'constexpr int kPlaneChunks = 2 * kNumPlanes;' in HDR and 'constexpr int kNumPlanes = 2;' in HDR
    groups = [list(range(8 * g, 8 * g + 8)) for g in range(8)]

    def worst(permute, WD):
        w = 1
        for j in range(WD * chunks // 256):
            for wave in range(4):
                def addr(lane):
                    e = 64 * wave + lane + 256 * j
                    r, c = e // chunks, e % chunks
                    if permute:
                        r = (r & ~7) | perm[r & 7]
                    return r * K_ROW_DW + 4 * (c ^ swz(r))
                w = max(w, worst_conflict(groups, addr, 4, 32))
        return w

    assert worst(True, 128) == 1 and worst(True, 64) == 1
    assert worst(False, 128) == 2                      # the consecutive-row assignment this replaces
    rows = sorted((e // chunks & ~7) | perm[(e // chunks) & 7] for e in range(128 * chunks) if e % chunks == 0)
    assert rows == list(range(128))                    # still every row exactly once
