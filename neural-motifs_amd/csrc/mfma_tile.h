// mfma_tile.h -- the shared fp32 matrix-core block-tile engine (gfx950), block tile BM x BN x 16.
//
// One workgroup = 256 threads = 4 waves; each wave owns a 64x64 output tile held as 2x2 32x32 fp32 accumulators
// (16 VGPRs each).  Two block shapes are used:
//   128x128 (waves 2x2)  -- the default
//   256x64  (waves 4x1)  -- for N <= 64 (the 64-channel VGG layer), so no MFMA issues on padding columns
// Two interchangeable inner loops compute the same fp32 result (selected at build time, MH_MFMA_SPLIT):
//   6 (default): "bf16x6" -- each fp32 operand is split exactly into three bf16 terms in registers and six
//      v_mfma_f32_32x32x16_bf16 accumulate the cross terms in fp32 (see below): 6/16 of the matrix-core time of
//   0: v_mfma_f32_32x32x2_f32, the f32-input MFMA (an exact fp32 fma chain, 64 cycles per 2 k).
// Measured error against fp64 is the same for both (tools/gemm_accuracy.py).
//
// An operand tile lives in LDS as fp32 in the orientation its GLOBAL storage has, so that staging is always a
// straight 16-byte copy (global_load_dwordx4 -> ds_write_b128), never a transposing scatter:
//   * "KM" (k-major, [k][w], row stride w+4): operands stored with the tile's row/column dimension contiguous
//     (B of y = x*W when W is [K,N]; both operands of a weight gradient).  The wave's 64 rows are interleaved over
//     its two 32-row MFMA sub-tiles (tile row 2i+s -> sub-tile s), so one ds_read_b64 feeds both sub-tiles.
//   * "WM" (width-major, [w][k], row stride 16+4): operands stored K-contiguous (activations, nn.Linear weights,
//     NHWC pixels).  Sub-tile s owns tile rows i+32s; a lane reads 4 consecutive k of its row with one
//     ds_read_b128 (conflict-free at stride 20: the 16 lanes of a service group cover all 64 banks).
// The k index of a tile is permuted consistently for both operands: MFMA lane group g = lane>>5 takes
// k = 8g + step (step = 0..7): the f32 loop runs 8 steps of 2 k, the bf16 loop consumes the lane's 8 k at once
// (order of the fp32 summation over k changes, nothing else).
// In the f32 loop the LDS reads of step kk+1 are issued before the 4 MFMAs of step kk (order pinned with
// sched_group_barrier) so the MFMA issue covers the LDS latency in-wave.
// Accumulator (C/D) map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5), r in [0,16).
#pragma once
#include "common.h"

#ifndef MH_MINW
#define MH_MINW 2
#endif
#ifndef MH_MFMA_SPLIT
#define MH_MFMA_SPLIT 6   /* 6: bf16x6 split (fp32-accurate, default); 0: f32-input MFMA; 3: bf16x3 (2^-17, tests only) */
#endif

namespace mh {

constexpr int kBK = 16;
constexpr int kThreads = 256;
constexpr int kLdW = kBK + 4;   // WM row stride (floats)

template <int WD, bool WM>
struct TileGeom {
    static constexpr int ld = WM ? kLdW : WD + 4;        // LDS row stride (floats); keeps 16-B alignment
    static constexpr int floats = WM ? WD * kLdW : kBK * (WD + 4);
    static constexpr int nv = WD * kBK / 1024;            // float4 staged per thread (256 threads)
};

struct Acc {
    f32x16 v[2][2];
};

__device__ __forceinline__ void acc_zero(Acc &a)
{
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) a.v[i][j][r] = 0.f;
}

// Per-lane operand fragments of one k-tile.  WM: two float4 per phase of 4 k-steps (sub-tiles s = 0,1);
// KM: one float2 per k-step (x = sub-tile 0, y = sub-tile 1).
template <bool WM>
struct FragBuf;
template <>
struct FragBuf<true> {
    float4 v[2][2];   // [phase parity][sub-tile]
};
template <>
struct FragBuf<false> {
    float2 v[kBK / 2];   // one register pair per k-step (no reuse inside a k-tile: lets the waits be counted exactly)
};

// issue the LDS reads that provide k-step `kk`; returns the number of DS instructions issued (compile-time folded)
template <int WD>
__device__ __forceinline__ int frag_fetch(FragBuf<true> &f, const float *__restrict__ tile, int w0, int lane, int kk)
{
    if (kk % 4 != 0) return 0;
    const float *p = tile + (w0 + (lane & 31)) * kLdW + 8 * (lane >> 5) + kk;
    f.v[(kk / 4) & 1][0] = *reinterpret_cast<const float4 *>(p);
    f.v[(kk / 4) & 1][1] = *reinterpret_cast<const float4 *>(p + 32 * kLdW);
    return 2;
}
template <int WD>
__device__ __forceinline__ int frag_fetch(FragBuf<false> &f, const float *__restrict__ tile, int w0, int lane, int kk)
{
    constexpr int ld = TileGeom<WD, false>::ld;
    f.v[kk] = *reinterpret_cast<const float2 *>(tile + (8 * (lane >> 5) + kk) * ld + w0 + 2 * (lane & 31));
    return 1;
}
__device__ __forceinline__ float frag_get(const FragBuf<true> &f, int kk, int s)
{
    const float4 &x = f.v[(kk / 4) & 1][s];
    return (kk % 4 == 0) ? x.x : (kk % 4 == 1) ? x.y : (kk % 4 == 2) ? x.z : x.w;
}
__device__ __forceinline__ float frag_get(const FragBuf<false> &f, int kk, int s)
{
    return s == 0 ? f.v[kk].x : f.v[kk].y;
}

// All MFMAs of one k-tile for this wave.  wm/wn = the wave's row/col origin inside the block tile.
// The LDS reads for k-step kk+1 are issued before the 4 MFMAs of step kk and the order is pinned with
// sched_group_barrier, so the 4 x 64-cycle MFMA issue covers the LDS latency inside the SAME wave.
template <bool AWM, bool BWM, int BM, int BN>
__device__ __forceinline__ void mma_ktile_f32(const float *__restrict__ As, const float *__restrict__ Bs, int wm, int wn,
                                          int lane, Acc &acc)
{
    FragBuf<AWM> a;
    FragBuf<BWM> b;
    frag_fetch<BM>(a, As, wm, lane, 0);
    frag_fetch<BN>(b, Bs, wn, lane, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, (AWM ? 2 : 1) + (BWM ? 2 : 1), 0);   // the prologue reads form group 0
#pragma unroll
    for (int kk = 0; kk < kBK / 2; ++kk) {
        int nreads = 0;
        if (kk + 1 < kBK / 2) {
            nreads += frag_fetch<BM>(a, As, wm, lane, kk + 1);
            nreads += frag_fetch<BN>(b, Bs, wn, lane, kk + 1);
        }
        const float a0 = frag_get(a, kk, 0), a1 = frag_get(a, kk, 1);
        const float b0 = frag_get(b, kk, 0), b1 = frag_get(b, kk, 1);
        acc.v[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc.v[0][0], 0, 0, 0);
        acc.v[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc.v[0][1], 0, 0, 0);
        acc.v[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc.v[1][0], 0, 0, 0);
        acc.v[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc.v[1][1], 0, 0, 0);
        if (nreads == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (nreads == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        if (nreads == 3) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        if (nreads == 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// FP32-accurate products on the BF16 matrix cores ("bf16x6").
// gfx950's v_mfma_f32_32x32x16_bf16 retires 16 k per 32 cycles, the f32-input MFMA 2 k per 64 cycles: 16x the rate.
// An fp32 number splits EXACTLY into three bf16 terms by truncation, a = a1 + a2 + a3 with 8 mantissa bits each
// (a1 = top 16 bits of a; r = a - a1 is exact; a2 = top 16 bits of r; a3 = r - a2 has <= 8 significant bits), a
// bf16 x bf16 product is exact in fp32, and the MFMA accumulates in fp32.  Keeping the six cross terms down to
// 2^-24 relative magnitude,
//     a*b = a1b1 + (a1b2 + a2b1) + (a1b3 + a3b1 + a2b2) + O(2^-27 |ab|),
// gives products accurate to fp32 rounding at 6/16 of the f32-MFMA matrix-core time (MH_MFMA_SPLIT == 6); the first
// three terms alone (== 3) are accurate to 2^-17.  The LDS tiles, staging and epilogue are unchanged: a lane's
// operand for the K=16 instruction is 8 consecutive k of its row -- exactly the k = 8g + step permutation above.
// ---------------------------------------------------------------------------------------------------------------
typedef short bf16x8 __attribute__((ext_vector_type(8)));

struct SplitFrag {
    bf16x8 p[3];   // hi, mid, lo planes of 8 consecutive k
};

// 8 fp32 -> three bf16x8 planes (exact truncation split)
__device__ __forceinline__ void split8(const float (&x)[8], SplitFrag &f)
{
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned xb = __builtin_bit_cast(unsigned, x[i]);
        const unsigned hb = xb & 0xffff0000u;
        const float r1 = x[i] - __builtin_bit_cast(float, hb);
        const unsigned r1b = __builtin_bit_cast(unsigned, r1);
        const unsigned mb = r1b & 0xffff0000u;
        const float r2 = r1 - __builtin_bit_cast(float, mb);
        h[i] = hb; m[i] = mb; l[i] = __builtin_bit_cast(unsigned, r2);
    }
    unsigned ph[4], pm[4], pl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // element 2i in the low half, 2i+1 in the high half
        ph[i] = (h[2 * i] >> 16) | (h[2 * i + 1] & 0xffff0000u);
        pm[i] = (m[2 * i] >> 16) | (m[2 * i + 1] & 0xffff0000u);
        pl[i] = (l[2 * i] >> 16) | (l[2 * i + 1] & 0xffff0000u);
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    f.p[0] = __builtin_bit_cast(bf16x8, (u32x4){ph[0], ph[1], ph[2], ph[3]});
    f.p[1] = __builtin_bit_cast(bf16x8, (u32x4){pm[0], pm[1], pm[2], pm[3]});
    f.p[2] = __builtin_bit_cast(bf16x8, (u32x4){pl[0], pl[1], pl[2], pl[3]});
}

// fetch the 8 k-values (k = 8g .. 8g+7) of this lane's row for both sub-tiles and split them
template <bool WM, int WD>
__device__ __forceinline__ void fetch_split(SplitFrag (&f)[2], const float *__restrict__ tile, int w0, int lane)
{
    const int i = lane & 31, g = lane >> 5;
    float x[2][8];
    if (WM) {
        const float *p = tile + (w0 + i) * kLdW + 8 * g;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float4 lo = *reinterpret_cast<const float4 *>(p + 32 * s * kLdW);
            const float4 hi = *reinterpret_cast<const float4 *>(p + 32 * s * kLdW + 4);
            x[s][0] = lo.x; x[s][1] = lo.y; x[s][2] = lo.z; x[s][3] = lo.w;
            x[s][4] = hi.x; x[s][5] = hi.y; x[s][6] = hi.z; x[s][7] = hi.w;
        }
    } else {
        constexpr int ld = TileGeom<WD, false>::ld;
        const float *p = tile + (8 * g) * ld + w0 + 2 * i;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float2 v = *reinterpret_cast<const float2 *>(p + q * ld);
            x[0][q] = v.x;
            x[1][q] = v.y;
        }
    }
    split8(x[0], f[0]);
    split8(x[1], f[1]);
}

template <bool AWM, bool BWM, int BM, int BN>
__device__ __forceinline__ void mma_ktile_split(const float *__restrict__ As, const float *__restrict__ Bs, int wm,
                                                int wn, int lane, Acc &acc)
{
    SplitFrag a[2], b[2];
    fetch_split<AWM, BM>(a, As, wm, lane);
    fetch_split<BWM, BN>(b, Bs, wn, lane);
#pragma unroll
    for (int sm = 0; sm < 2; ++sm)
#pragma unroll
        for (int sn = 0; sn < 2; ++sn) {
            f32x16 c = acc.v[sm][sn];
#if MH_MFMA_SPLIT >= 6
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sm].p[2], b[sn].p[0], c, 0, 0, 0);   // smallest terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sm].p[0], b[sn].p[2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sm].p[1], b[sn].p[1], c, 0, 0, 0);
#endif
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sm].p[1], b[sn].p[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sm].p[0], b[sn].p[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sm].p[0], b[sn].p[0], c, 0, 0, 0);
            acc.v[sm][sn] = c;
        }
}

template <bool AWM, bool BWM, int BM, int BN>
__device__ __forceinline__ void mma_ktile(const float *__restrict__ As, const float *__restrict__ Bs, int wm, int wn,
                                          int lane, Acc &acc)
{
#if MH_MFMA_SPLIT
    mma_ktile_split<AWM, BWM, BM, BN>(As, Bs, wm, wn, lane, acc);
#else
    mma_ktile_f32<AWM, BWM, BM, BN>(As, Bs, wm, wn, lane, acc);
#endif
}

// Staging registers for one operand k-tile of width WD: WD/64 float4 per thread.
template <int WD>
struct Stage {
    float4 v[WD * kBK / 1024];
};

// FAST = operands are 16-B aligned with the contiguous extent a multiple of 4 (checked on the host): every float4
// is entirely inside or entirely outside the matrix, so the load is issued from a clamped address and zeroed by a
// select -- no per-element exec-masked branches, the k-tile's loads all stay in flight together.
template <bool FAST>
__device__ __forceinline__ float4 load4_guarded(const float *p, int c, int extent, bool vec, const float *safe)
{
    if (FAST) {
        const bool ok = (p != nullptr) && (c < extent);
        const float4 v = *reinterpret_cast<const float4 *>(ok ? p + c : safe);
        return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p == nullptr) return v;
    if (vec && c + 3 < extent) return *reinterpret_cast<const float4 *>(p + c);
    if (c + 0 < extent) v.x = p[c + 0];
    if (c + 1 < extent) v.y = p[c + 1];
    if (c + 2 < extent) v.z = p[c + 2];
    if (c + 3 < extent) v.w = p[c + 3];
    return v;
}

// ---- KM operand: global rows are k (contiguous along the tile's w dimension).
// tile = kBK rows x WD floats; float4 f = tid + 256*j sits at (row f / (WD/4), float4-column f % (WD/4)).
// row_ptr(k) returns the address of matrix element (k, 0) or nullptr when row k is all-zero / out of range;
// `col0` = first tile column in the matrix, `ncols` = matrix extent along the contiguous dimension.
template <int WD, bool FAST, typename RowPtr>
__device__ __forceinline__ void load_km(Stage<WD> &s, RowPtr row_ptr, int k0, int col0, int ncols, bool vec, int tid,
                                        const float *safe)
{
    constexpr int c4n = WD / 4;
#pragma unroll
    for (int j = 0; j < TileGeom<WD, false>::nv; ++j) {
        const int f = tid + kThreads * j;
        s.v[j] = load4_guarded<FAST>(row_ptr(k0 + f / c4n), col0 + 4 * (f % c4n), ncols, vec, safe);
    }
}

template <int WD>
__device__ __forceinline__ void store_km(const Stage<WD> &s, float *tile, int tid)
{
    constexpr int c4n = WD / 4;
#pragma unroll
    for (int j = 0; j < TileGeom<WD, false>::nv; ++j) {
        const int f = tid + kThreads * j;
        *reinterpret_cast<float4 *>(tile + (f / c4n) * TileGeom<WD, false>::ld + 4 * (f % c4n)) = s.v[j];
    }
}

// ---- WM operand: global rows are the tile's w dimension (k contiguous).  float4 f = tid + 256*j belongs to tile
// row f / 4, k-quad f % 4: four consecutive lanes read one row's 64 contiguous bytes.
// row_ptr(r) returns the address of element (r, k = 0) or nullptr when the row is out of range / zero;
// `kext` = matrix extent along k.
template <int WD, bool FAST, typename RowPtr>
__device__ __forceinline__ void load_wm(Stage<WD> &s, RowPtr row_ptr, int k0, int kext, bool vec, int tid,
                                        const float *safe)
{
#pragma unroll
    for (int j = 0; j < TileGeom<WD, true>::nv; ++j) {
        const int f = tid + kThreads * j;
        s.v[j] = load4_guarded<FAST>(row_ptr(f >> 2), k0 + 4 * (f & 3), kext, vec, safe);
    }
}

template <int WD>
__device__ __forceinline__ void store_wm(const Stage<WD> &s, float *tile, int tid)
{
#pragma unroll
    for (int j = 0; j < TileGeom<WD, true>::nv; ++j) {
        const int f = tid + kThreads * j;
        *reinterpret_cast<float4 *>(tile + (f >> 2) * kLdW + 4 * (f & 3)) = s.v[j];
    }
}

// tile row (or column) held by MFMA index idx (0..31) of sub-tile s, for the two operand layouts
template <bool WM>
__device__ __forceinline__ int tile_coord(int idx, int s)
{
    return WM ? idx + 32 * s : 2 * idx + s;
}

// Epilogue visitor: calls f(row, col0, col1, v0, v1) for the two outputs (sub-tile columns sn = 0,1) this lane holds
// in tile row `row` (all relative to the block tile).  With a KM B operand col1 == col0 + 1 (8-byte stores).
template <bool AWM, bool BWM, typename F>
__device__ __forceinline__ void acc_foreach_pair(const Acc &acc, int wm, int wn, int lane, F f)
{
    const int j = lane & 31, g = lane >> 5;
    const int c0 = wn + tile_coord<BWM>(j, 0), c1 = wn + tile_coord<BWM>(j, 1);
#pragma unroll
    for (int sm = 0; sm < 2; ++sm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ri = (r & 3) + 8 * (r >> 2) + 4 * g;
            f(wm + tile_coord<AWM>(ri, sm), c0, c1, acc.v[sm][0][r], acc.v[sm][1][r]);
        }
}

// wave -> 64x64 sub-tile origin for the two block shapes
template <int BM, int BN>
__device__ __forceinline__ void wave_origin(int wave, int &wm, int &wn)
{
    if (BN == 128) { wm = (wave >> 1) * 64; wn = (wave & 1) * 64; }
    else { wm = wave * 64; wn = 0; }
}

// XCD-aware remap of a linear block id: consecutive ids are dispatched round-robin over the 8 XCDs, so give
// each XCD a contiguous chunk of the tile space (neighbouring tiles share operand panels -> shared L2).
__device__ __forceinline__ int xcd_remap(int bid, int nblocks)
{
    constexpr int kXcd = 8;
    const int q = nblocks / kXcd, r = nblocks % kXcd;
    const int xcd = bid % kXcd, idx = bid / kXcd;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// defined in gemm.hip
int choose_splitk_tiles(long long tiles, int ktiles, double out_elems, double flops);
int launch_splitk_reduce(const float *partial, int splitk, long long M, int N, float *C, int ldc, const float *bias,
                         int epilogue, int accumulate, hipStream_t st);

}  // namespace mh
