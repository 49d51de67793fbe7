"""The ring K loop's index arithmetic (neural-motifs_amd/csrc/pl_ring.h, the ring kernels of pl_gemm.hip / pl_conv.hip), replayed in Python:

* LDS-DMA staging: a wave instruction deposits lane l at byte 16 l of its 1 KiB piece; with the SOURCE-side swizzle of `plan_dma`
  chunk c of tile row r ends up at `lds_chunk(r, c)` -- exactly where the fragment reads of `rplan_frags` look for it;
* the fragment reads stay conflict-free for ds_read_b128's fixed 16-lane service groups (bank model of MI355X_MICROARCH.md);
* the TRANSPOSED accumulator map of `rmma` / `racc_quads` (weights as the matrix core's A operand): every (row, column) of the block tile
  exactly once, four consecutive columns per register quad;
* the epilogue patches: fp32 [32][36] floats written with ds_write_b128 (conflict-free) and read back row-major; image cells at an
  80-byte pitch written with ds_write_b64 (two-way at worst); every element of a 32 x 32 accumulator lands in the right cell byte;
* ring bookkeeping: stage / fragment-set rotation of `ring_loop` for NS = 3 and 4 (a stage is never overwritten before its tile was
  consumed, never read before its DMA was waited for), the vmcnt immediates, the conv's compile-time tap schedule (U = lcm(unroll, 9)).
The constants are read from the sources so that the test follows the code."""
import math
import os
import re

import pytest

from test_pl_layout import B128_READ_GROUPS, B128_WRITE_GROUPS, CELL, lds_chunk, swz, worst_conflict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RING = open(os.path.join(ROOT, 'neural-motifs_amd', 'csrc', 'pl_ring.h')).read()
GEMM = open(os.path.join(ROOT, 'neural-motifs_amd', 'csrc', 'pl_gemm.hip')).read()
CONV = open(os.path.join(ROOT, 'neural-motifs_amd', 'csrc', 'pl_conv.hip')).read()

# (WM, WN, SM, SN, NS) of the typedefs that ship
SHAPES = [tuple(int(x) for x in m) for m in re.findall(r'typedef Ring<(\d+), (\d+), (\d+), (\d+), (\d+)> C?R\d+x\d+(?:s\d)?;', GEMM + CONV)]
B64_WRITE_GROUPS = [list(range(16 * g, 16 * g + 16)) for g in range(4)]


def test_sources_say_what_the_model_assumes():
    # round 6: the 64-channel ring tiles of conv1_2 (Ring<4, 1, 2, 2, 4> ships, <4, 1, 2, 2, 3> is its three-stage A/B arm)
    assert sorted(set(SHAPES)) == [(4, 1, 2, 2, 3), (4, 1, 2, 2, 4), (4, 1, 2, 4, 3), (4, 2, 2, 4, 4)]
    assert 'return 16 * (j * R::waves + wave) + (lane >> 2);' in RING                               # dma_row
    assert 'row * kCell + 16 * ((lane & 3) ^ swz(row))' in RING                                     # source offset of a lane
    assert 'stage + (j * R::waves + wave) * 1024' in RING and 'stage + R::a_bytes + (j * R::waves + wave) * 1024' in RING
    assert '__builtin_amdgcn_mfma_f32_32x32x16_f16(f.b[sn][kTermB[t]], f.a[sm][kTermA[t]]' in RING   # operand roles swapped
    assert 'wait_vmcnt<R::nd * (R::NS - 3)>();' in RING and 'wait_vmcnt<R::nd * (NS - 2)>();' in RING
    assert 'stg + j * 36 + 8 * q + 4 * g' in GEMM and 'stg + j * 36 + 8 * q + 4 * g' in CONV
    assert 'constexpr int kPitch = 80;' in CONV
    assert 'char *cell = stg + ((q >> 1) * 32 + j) * kPitch + 16 * (q & 1) + 8 * g;' in CONV


@pytest.mark.parametrize('wm,wn,sm,sn,ns', sorted(set(SHAPES)))
def test_dma_pieces_meet_the_fragment_reads(wm, wn, sm, sn, ns):
    waves = wm * wn
    bm, bn = 32 * sm * wm, 32 * sn * wn
    a_bytes = bm * CELL
    lds = {}
    for name, rows, base in (('A', bm, 0), ('B', bn, a_bytes)):
        pieces = rows // 16
        assert pieces % waves == 0
        for wave in range(waves):
            for j in range(pieces // waves):
                q = j * waves + wave
                for lane in range(64):
                    row = 16 * q + (lane >> 2)                            # dma_row
                    chunk = (lane & 3) ^ swz(row)                         # what plan_dma makes the lane FETCH (row * 64 + 16 * chunk)
                    dst = base + q * 1024 + 16 * lane                     # where the DMA deposits it (linear in the lane)
                    assert dst == base + lds_chunk(row, chunk)
                    assert dst not in lds
                    lds[dst] = (name, row, chunk)
    assert len(lds) == (bm + bn) * 4
    for wave in range(waves):
        wm0, wn0 = (wave // wn) * 32 * sm, (wave % wn) * 32 * sn          # rwave_origin
        for p in range(2):
            for s in range(sm):
                addr = lambda lane: lds_chunk(wm0 + (lane & 31), 2 * p + (lane >> 5)) + 2048 * s
                for lane in range(64):
                    assert lds[addr(lane)] == ('A', wm0 + 32 * s + (lane & 31), 2 * p + (lane >> 5))
                assert worst_conflict(B128_READ_GROUPS, addr, 16, 64) == 1
            for s in range(sn):
                addr = lambda lane: a_bytes + lds_chunk(wn0 + (lane & 31), 2 * p + (lane >> 5)) + 2048 * s
                for lane in range(64):
                    assert lds[addr(lane)] == ('B', wn0 + 32 * s + (lane & 31), 2 * p + (lane >> 5))
                assert worst_conflict(B128_READ_GROUPS, addr, 16, 64) == 1


@pytest.mark.parametrize('wm,wn,sm,sn,ns', sorted(set(SHAPES)))
def test_transposed_accumulators_cover_the_tile_once(wm, wn, sm, sn, ns):
    """v_mfma_f32_32x32x16: D[i][j] with i = row of the A operand, j = column of the B operand lives in lane (j, g) register r with
    i = (r & 3) + 8 (r >> 2) + 4 g.  With the weights as A and the pixels as B: i = channel, j = pixel."""
    seen = {}
    for wave in range(wm * wn):
        wm0, wn0 = (wave // wn) * 32 * sm, (wave % wn) * 32 * sn
        for lane in range(64):
            j, g = lane & 31, lane >> 5
            for a in range(sm):
                for b in range(sn):
                    for q in range(4):
                        cols = [wn0 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * g for r in range(4 * q, 4 * q + 4)]
                        assert cols == list(range(wn0 + 32 * b + 8 * q + 4 * g, wn0 + 32 * b + 8 * q + 4 * g + 4))   # racc_quads
                        for c in cols:
                            key = (wm0 + 32 * a + j, c)
                            assert key not in seen
                            seen[key] = (wave, lane, a, b, q)
    assert len(seen) == (32 * sm * wm) * (32 * sn * wn)


def test_fp32_patch_round_trip_and_banks():
    patch = {}
    for lane in range(64):
        j, g = lane & 31, lane >> 5
        for q in range(4):
            for k in range(4):
                patch[(j * 36 + 8 * q + 4 * g + k) * 4] = (j, 8 * q + 4 * g + k)
    assert len(patch) == 32 * 32
    for q in range(4):       # ds_write_b128: eight contiguous 8-lane groups over 32 banks
        assert worst_conflict(B128_WRITE_GROUPS, lambda lane: ((lane & 31) * 36 + 8 * q + 4 * (lane >> 5)) * 4, 16, 32) == 1
    out = set()
    for i in range(4):       # copy-out: lane -> row 8 i + lane / 8, columns 4 (lane % 8) .. + 3: whole 128-byte lines of the output
        for lane in range(64):
            rr, c4 = 8 * i + (lane >> 3), 4 * (lane & 7)
            for k in range(4):
                assert patch[(rr * 36 + c4 + k) * 4] == (rr, c4 + k)
                out.add((rr, c4 + k))
        assert worst_conflict(B128_READ_GROUPS, lambda lane: ((8 * i + (lane >> 3)) * 36 + 4 * (lane & 7)) * 4, 16, 64) <= 2
    assert len(out) == 32 * 32


def test_image_patch_builds_the_cells():
    """cell (chunk, pixel) = h1[16 channels] | h2[16 channels] (2 bytes each); lane (j, g) quad q holds channels 8 q + 4 g .. + 3 of the
    32-channel accumulator = chunk q >> 1, channels 8 (q & 1) + 4 g .. + 3 of that chunk"""
    pitch = 80
    cellmap = {}
    for lane in range(64):
        j, g = lane & 31, lane >> 5
        for q in range(4):
            base = ((q >> 1) * 32 + j) * pitch + 16 * (q & 1) + 8 * g
            for plane in range(2):
                for k in range(4):                                       # split2(v0, v1) -> dword 0, split2(v2, v3) -> dword 1 of the 8 bytes
                    byte = base + 32 * plane + 2 * k
                    ch = 8 * q + 4 * g + k
                    assert (byte - ((q >> 1) * 32 + j) * pitch) == 32 * plane + 2 * (ch % 16)
                    cellmap[byte] = (q >> 1, j, plane, ch % 16)
    assert len(cellmap) == 2 * 32 * 2 * 16
    for q in range(4):
        for plane in range(2):
            w = worst_conflict(B64_WRITE_GROUPS, lambda lane: ((q >> 1) * 32 + (lane & 31)) * pitch + 16 * (q & 1) + 8 * (lane >> 5) + 32 * plane, 8, 32)
            assert w <= 2
    for i in range(4):       # copy-out instruction i: chunk i / 2, pixels 16 (i % 2) .. + 15, 16-byte piece lane % 4: one contiguous KiB
        dst = [(((i >> 1) * 1000 + 16 * (i & 1) + (lane >> 2)) * CELL + 16 * (lane & 3)) for lane in range(64)]
        assert dst == list(range(dst[0], dst[0] + 1024, 16))


@pytest.mark.parametrize('ns,nd', [(3, 6), (4, 4)])
def test_ring_rotation_never_reads_early_or_overwrites_late(ns, nd):
    """replay ring_loop's schedule for one wave: issue order of the DMA tiles, the vmcnt immediates, which stage is read and which is
    overwritten at every step"""
    unroll = ns if ns % 2 == 0 else 2 * ns
    ntiles = 3 * unroll + 5
    steps = -(-ntiles // unroll) * unroll
    issued = []                                   # tiles in issue order; a tile's nd pieces count against vmcnt
    landed_upto = -1                              # tiles 0 .. landed_upto are known to be in LDS (this wave's pieces)
    stage_of = {}
    for s in range(ns - 1):
        issued.append(s); stage_of[s] = s
    # prologue wait: vmcnt(nd * (ns - 2)) leaves ns - 2 tiles in flight
    landed_upto = len(issued) - 1 - (ns - 2)
    assert landed_upto == 0
    read_for = {0: 0}                             # fragment set of step 0 read from stage 0 (tile 0)
    for kt in range(steps):
        u = kt % unroll
        landed_upto = len(issued) - 1 - (ns - 3)  # wait_vmcnt<nd * (NS - 3)> at the top of the step
        assert landed_upto >= kt + 1, 'tile kt + 1 must have landed before its fragments are read in this step'
        dst, rd = (u + ns - 1) % ns, (u + 1) % ns
        # the stage overwritten now held tile kt - 1, whose fragments were read during step kt - 2 and consumed in step kt - 1
        holder = [t for t, st in stage_of.items() if st == dst and t > kt - 1]
        assert not holder, 'stage %d still holds tile %s at step %d' % (dst, holder, kt)
        issued.append(kt + ns - 1); stage_of[kt + ns - 1] = dst
        assert stage_of[kt + 1] == rd             # the fragments of tile kt + 1 are read from the stage its DMA went to
        assert stage_of[kt] != dst and stage_of[kt + 1] != dst
    assert (nd * (ns - 3)) < 64 and (nd * (ns - 2)) < 64


@pytest.mark.parametrize('ns', [3, 4])
def test_conv_tap_schedule(ns):
    unroll = ns if ns % 2 == 0 else 2 * ns
    U = unroll if unroll % 9 == 0 else (3 * unroll if unroll % 3 == 0 else 9 * unroll)
    assert U == unroll * 9 // math.gcd(unroll, 9) and U % 9 == 0 and U % unroll == 0
    # the tap of the tile issued with index `ui` (prologue: ui = s; loop: ui = u + NS - 1) when K slices start at multiples of 9
    for kt_begin in (0, 9, 27):
        for s in range(ns - 1):
            assert (kt_begin + s) % 9 == s % 9
        for kt in range(kt_begin, kt_begin + 3 * U, U):
            for u in range(U):
                assert (kt + u + ns - 1) % 9 == (u + ns - 1) % 9
