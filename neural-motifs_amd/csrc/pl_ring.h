// pl_ring.h -- the round-4 K loop of the plane engine (gfx950 only): LDS-DMA staging into a ring of stages, register
// double-buffered fragments, 64x128 (or 128x64 / 128x128) wave tiles.
//
// What round 3's loop (pl_tile.h: k_step) was limited by, per 16-k tile of a 128x128 block:
//     4 waves x 8 ds_read_b128 (4 LDS cycles each)   = 128 LDS cycles
//     4 waves x 4 ds_write_b128 (13 cycles each)     = 208 LDS cycles      -> 336 LDS cycles against 384 MFMA cycles per SIMD,
// one s_barrier per 384 MFMA cycles, fragments read right after the barrier (matrix pipe idle while both waves of a SIMD read).
// Here:
//   * global -> LDS goes through `buffer_load_dwordx4 ... lds` (LDS-DMA): no staging VGPRs, no ds_write at all.  A wave
//     instruction deposits 64 lanes x 16 B = 1 KiB LINEARLY at M0 + offset (profiles/r02_dma_probe.jsonl: per-lane SOURCE
//     offsets are free, out-of-range lanes write zeros), i.e. 16 tile rows of 64 B.  The bank swizzle of the tile (16-byte slot
//     c ^ ((row >> 2) & 3), the layout pl_tile.h proved conflict-free for ds_read_b128) is therefore applied on the SOURCE
//     side: lane l of piece q fetches chunk (l & 3) ^ swz(row) of row 16 q + (l >> 2) -- still whole 64-byte cells per 4 lanes.
//   * a wave owns SM x SN = 2 x 4 (4 x 2, 4 x 4) accumulators: 24 (48) MFMAs per 12 (16) fragment reads instead of 12 per 8;
//   * ring of NS stages (one 16-k tile each); step kt:  wait for MY pieces of tile kt+1 (vmcnt), ONE barrier (everybody's
//     pieces of kt+1 are in LDS, everybody is done reading tile kt-1), issue the DMA of tile kt+NS-1 into the stage tile kt-1
//     occupied, read the fragments of tile kt+1 into the OTHER fragment set while the MFMAs of tile kt run from this one.
//     The matrix pipe never waits for an LDS read of its own step, and NS-2 tiles are in flight across the barrier
//     (LDS-DMA requests stay in flight across s_barrier, MI355X_MICROARCH.md "Two waves per SIMD" item 7).
//   * the loop is unrolled NS x (NS even) so stages and fragment sets are immediates; a K range that is not a multiple of NS
//     is padded (to a multiple of NS, or 2 NS when NS is odd) with steps whose A pieces are fetched out of range (zeros): acc += 0 * B.
// Shapes: Ring<WM, WN, SM, SN, NS>: WM x WN waves, block tile (32 SM WM) x (32 SN WN).
#pragma once
#include "pl_tile.h"

namespace mh {
namespace pl {

typedef __attribute__((address_space(3))) void *lds_ptr_t;

template <int WM, int WN, int SM, int SN, int NS_>
struct Ring {
    static constexpr int wm = WM, wn = WN, sm = SM, sn = SN, NS = NS_;
    static constexpr int waves = WM * WN, threads = 64 * waves;
    static constexpr int bm = 32 * SM * WM, bn = 32 * SN * WN;
    static constexpr int a_bytes = bm * kCell, b_bytes = bn * kCell, stage_bytes = a_bytes + b_bytes;
    static constexpr int lds_bytes = NS * stage_bytes;
    static constexpr int pa = bm / 16, pb = bn / 16;                  // 1 KiB DMA pieces of one stage
    static_assert(pa % waves == 0 && pb % waves == 0, "pieces must divide over the waves");
    static constexpr int na = pa / waves, nb = pb / waves, nd = na + nb;
    static constexpr int mfmas = 3 * SM * SN, nread = 2 * (SM + SN);
    static_assert(NS >= 3, "ring: at least three stages");
    static constexpr int unroll = (NS % 2 == 0) ? NS : 2 * NS;        // stages AND the two fragment sets are immediates
    static_assert(nd * (NS - 3) < 64, "vmcnt is a 6-bit counter");
};

template <class R>
struct RAcc {
    f32x16 v[R::sm][R::sn];
};
template <class R>
__device__ __forceinline__ void racc_zero(RAcc<R> &a)
{
#pragma unroll
    for (int i = 0; i < R::sm; ++i)
#pragma unroll
        for (int j = 0; j < R::sn; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) a.v[i][j][r] = 0.f;
}

// per-lane source offsets of this lane's DMA pieces: piece j of wave w is piece q = j * waves + w of the operand tile = rows
// 16 q .. 16 q + 15; the lane deposits LDS slot (lane & 3) of row 16 q + (lane >> 2), which holds chunk slot ^ swz(row)
template <class R>
struct DmaPlan {
    unsigned va[R::na], vb[R::nb];
};
template <class R>
__device__ __forceinline__ int dma_row(int j, int wave, int lane) { return 16 * (j * R::waves + wave) + (lane >> 2); }
template <class R, typename RowOkA, typename RowOkB>
__device__ __forceinline__ void plan_dma(DmaPlan<R> &p, RowOkA a_ok, RowOkB b_ok, int wave, int lane)
{
#pragma unroll
    for (int j = 0; j < R::na; ++j) {
        const int row = dma_row<R>(j, wave, lane);
        p.va[j] = a_ok(row) ? (unsigned)(row * kCell + 16 * ((lane & 3) ^ swz(row))) : kOob;
    }
#pragma unroll
    for (int j = 0; j < R::nb; ++j) {
        const int row = dma_row<R>(j, wave, lane);
        p.vb[j] = b_ok(row) ? (unsigned)(row * kCell + 16 * ((lane & 3) ^ swz(row))) : kOob;
    }
}
__device__ __forceinline__ void dma16(const Src &s, char *lds_dst, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rsrc, (lds_ptr_t)lds_dst, 16, (int)voff, (int)soff, 0, 0);
}
// the pieces of one stage: A rows from `sa` (+ scalar offset oa), B rows from `sb` (+ ob); `wave` must be wave-uniform (SGPR)
template <class R>
__device__ __forceinline__ void dma_stage(const Src &sa, const Src &sb, const unsigned (&va)[R::na], const unsigned (&vb)[R::nb], unsigned oa,
                                          unsigned ob, char *stage, int wave)
{
#pragma unroll
    for (int j = 0; j < R::na; ++j) dma16(sa, stage + (j * R::waves + wave) * 1024, va[j], oa);
#pragma unroll
    for (int j = 0; j < R::nb; ++j) dma16(sb, stage + R::a_bytes + (j * R::waves + wave) * 1024, vb[j], ob);
}

template <class R>
struct RFrags {
    f16x8 a[R::sm][2], b[R::sn][2];   // [sub-tile][plane]
};
template <class R>
__device__ __forceinline__ void rplan_frags(FragPlan &f, int wm0, int wn0, int lane)
{
    const int i = lane & 31, g = lane >> 5;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        f.a[p] = lds_chunk(wm0 + i, 2 * p + g);
        f.b[p] = (unsigned)R::a_bytes + lds_chunk(wn0 + i, 2 * p + g);
    }
}
template <class R>
__device__ __forceinline__ void rfetch(RFrags<R> &f, const FragPlan &fp, const char *stage)
{
#pragma unroll
    for (int s = 0; s < R::sm; ++s) f.a[s][1] = *reinterpret_cast<const f16x8 *>(stage + fp.a[1] + 2048 * s);
#pragma unroll
    for (int s = 0; s < R::sn; ++s) f.b[s][0] = *reinterpret_cast<const f16x8 *>(stage + fp.b[0] + 2048 * s);
#pragma unroll
    for (int s = 0; s < R::sn; ++s) f.b[s][1] = *reinterpret_cast<const f16x8 *>(stage + fp.b[1] + 2048 * s);
#pragma unroll
    for (int s = 0; s < R::sm; ++s) f.a[s][0] = *reinterpret_cast<const f16x8 *>(stage + fp.a[0] + 2048 * s);
}
// three terms per accumulator, smallest first (h2a h1b, h1a h2b, h1a h1b); consecutive MFMAs are independent.
// OPERAND ROLES ARE SWAPPED: the B fragment (columns of the product: output channels / N) is the matrix core's A operand and
// the A fragment (rows: pixels / M) its B operand, so the 32x32 result arrives TRANSPOSED: lane (j, g) register r holds
//     C[row = 32 sm + j][col = 32 sn + (r & 3) + 8 (r >> 2) + 4 g]
// i.e. every group of four registers is four CONSECUTIVE columns of one row -- 16 contiguous bytes of a row-major C (or of
// an NHWC pixel): what the epilogues' 16-byte stores want (racc_quads).
template <class R>
__device__ __forceinline__ void rmma(const RFrags<R> &f, RAcc<R> &acc)
{
    constexpr int kTermA[3] = {1, 0, 0}, kTermB[3] = {0, 1, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int sm = 0; sm < R::sm; ++sm)
#pragma unroll
            for (int sn = 0; sn < R::sn; ++sn)
                acc.v[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.b[sn][kTermB[t]], f.a[sm][kTermA[t]], acc.v[sm][sn], 0, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt");
    // gfx9 s_waitcnt: vmcnt = imm[3:0] | imm[15:14] << 4, expcnt imm[6:4], lgkmcnt imm[11:8]; leave the other two counters alone
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

// One step: see the header.  `issue(stage_ptr)` launches the DMA of tile kt + NS - 1 into the stage tile kt - 1 occupied;
// `cur` holds the fragments of tile kt (read one step earlier), `nxt` receives those of tile kt + 1 from `rd_stage`.
template <class R, typename Issue>
__device__ __forceinline__ void ring_step(Issue issue, char *dma_stage_ptr, const char *rd_stage, const FragPlan &fp, const RFrags<R> &cur,
                                          RFrags<R> &nxt, RAcc<R> &acc)
{
    wait_vmcnt<R::nd * (R::NS - 3)>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    issue(dma_stage_ptr);
    rfetch<R>(nxt, fp, rd_stage);
    rmma<R>(cur, acc);
    // pin: DMA issue first, then {2 MFMA : 1 LDS read} until the reads are out, then the remaining MFMAs
    __builtin_amdgcn_sched_group_barrier(0x020, R::nd, 0);            // VMEM reads (the DMA pieces)
#pragma unroll
    for (int i = 0; i < R::nread; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, R::mfmas / R::nread > 0 ? R::mfmas / R::nread : 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// wave -> origin of its sub-tile: waves walk N fastest
template <class R>
__device__ __forceinline__ void rwave_origin(int wave, int &wm0, int &wn0)
{
    wm0 = (wave / R::wn) * 32 * R::sm;
    wn0 = (wave % R::wn) * 32 * R::sn;
}

// visitor over the (transposed) accumulators: f(row, col0, sn, v0, v1, v2, v3) for every quad of four consecutive columns this
// lane holds (row / col0 relative to the block tile; col0 is a multiple of 4)
template <class R, typename F>
__device__ __forceinline__ void racc_quads(const RAcc<R> &acc, int wm0, int wn0, int lane, F f)
{
    const int j = lane & 31, g = lane >> 5;
#pragma unroll
    for (int sm = 0; sm < R::sm; ++sm)
#pragma unroll
        for (int sn = 0; sn < R::sn; ++sn)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                f(wm0 + 32 * sm + j, wn0 + 32 * sn + 8 * q + 4 * g, sn, acc.v[sm][sn][4 * q], acc.v[sm][sn][4 * q + 1], acc.v[sm][sn][4 * q + 2],
                  acc.v[sm][sn][4 * q + 3]);
}

// The whole K loop over tiles [kt_begin, kt_end): `issue(kt, ui, stage_ptr)` must launch the R::nd DMA pieces of tile kt (for
// kt >= kt_end: A pieces out of range -> zeros, B pieces any valid tile) into stage_ptr.  Tiles are issued strictly in order;
// `ui` is the tile's index relative to the start of the current unrolled body (kt - kt_begin modulo U for the first NS - 1
// tiles, then u + NS - 1): a COMPILE-TIME constant after unrolling, which the conv uses to know a tile's tap without any
// table (U a multiple of 9 there).  GRAN > 0: leave the unrolled body after any whole group of GRAN steps once the range is
// exhausted (bounds the padding of a short K range to GRAN - 1 steps instead of U - 1).
// Ends with every wave past its last fragment read and a barrier: the ring is reusable as epilogue scratch.
template <class R, int U = R::unroll, int GRAN = 0, typename Issue>
__device__ __forceinline__ void ring_loop(Issue issue, int kt_begin, int kt_end, char *lds, const FragPlan &fp, RAcc<R> &acc)
{
    constexpr int NS = R::NS;
    static_assert(U % R::unroll == 0, "the unrolled body must cover whole cycles of stages and fragment sets");
    static_assert(GRAN == 0 || (U % GRAN == 0 && GRAN >= NS), "exit granule");
    // prologue: tiles kt_begin .. kt_begin + NS - 2 into stages 0 .. NS - 2
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(kt_begin + s, s, lds + s * R::stage_bytes);
    wait_vmcnt<R::nd * (NS - 2)>();                      // my pieces of tile kt_begin
    __builtin_amdgcn_s_barrier();
    RFrags<R> f0, f1;
    rfetch<R>(f0, fp, lds);
    for (int kt = kt_begin; kt < kt_end; kt += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (GRAN > 0 && u > 0 && u % GRAN == 0 && kt + u >= kt_end) break;       // wave-uniform
            // step kt + u: consume set u % 2, read tile kt + u + 1 from stage (u + 1) % NS, DMA tile kt + u + NS - 1 into stage (u + NS - 1) % NS
            char *dst = lds + ((u + NS - 1) % NS) * R::stage_bytes;
            const char *rd = lds + ((u + 1) % NS) * R::stage_bytes;
            const int ktn = kt + u + NS - 1;
            if (u % 2 == 0) ring_step<R>([&](char *st) { issue(ktn, u + NS - 1, st); }, dst, rd, fp, f0, f1, acc);
            else ring_step<R>([&](char *st) { issue(ktn, u + NS - 1, st); }, dst, rd, fp, f1, f0, acc);
        }
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
}

}  // namespace pl
}  // namespace mh
