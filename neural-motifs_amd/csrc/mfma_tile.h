// mfma_tile.h -- the shared FP32 MFMA block-tile engine (gfx950), block tile BM x BN x 16.
//
// One workgroup = 256 threads = 4 waves; each wave owns a 64x64 output tile held as 2x2 accumulators of
// v_mfma_f32_32x32x2_f32 (16 VGPRs each, exact fp32 fma chain, 64 cycles/SIMD).  Two block shapes are used:
//   128x128 (waves 2x2)  -- the default
//   256x64  (waves 4x1)  -- for N <= 64 (the 64-channel VGG layer), so no MFMA issues on padding columns
//
// LDS layout (both operands "k-major"):  As[k][m], Bs[k][n], row stride = tile width + 4 floats.
//   * operand register for one MFMA is ONE float per lane: lane l supplies A[i = l&31][k = l>>5].
//   * the wave's 64 rows are interleaved over its two 32-row MFMA sub-tiles: tile row 2*i+s belongs to sub-tile
//     s, so a lane fetches BOTH sub-tiles' operands with a single ds_read_b64 (a 32-lane group reads 256
//     contiguous bytes: conflict-free), and in the epilogue holds two horizontally adjacent outputs -> 8-byte
//     stores.  Same for B/columns.
//   * K-contiguous global operands (activations, nn.Linear weights) are transposed while staging: a lane owns
//     one tile row, loads float4s along k and issues conflict-free ds_write_b32 (consecutive lanes ->
//     consecutive LDS words).
// Accumulator (C/D) map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5), r in [0,16).
#pragma once
#include "common.h"

namespace mh {

#ifndef MH_BK
#define MH_BK 16
#endif
#ifndef MH_MINW
#define MH_MINW 2
#endif
constexpr int kBK = MH_BK;
constexpr int kThreads = 256;

template <int WD>
struct TileGeom {
    static constexpr int ld = WD + 4;             // padded LDS row (floats); keeps 16-B alignment
    static constexpr int floats = kBK * ld;       // one operand tile
    static constexpr int nv = WD * kBK / 1024;    // float4 staged per thread (256 threads)
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

// One k-tile (kBK deep) of MFMAs for this wave.  wm/wn = wave's row/col offset inside the block tile.
// MFMAs of k-steps [KK0, KK1) of the current k-tile (a k-step = 2 k values = one 32x32x2 MFMA per accumulator).
// The fragment reads of step kk+1 are issued before the 4 MFMAs of step kk and the order is pinned with
// sched_group_barrier, so the 4 x 64-cycle MFMA issue of one step covers the LDS latency of the next inside the SAME
// wave (the compiler otherwise batches [reads, wait, 8 MFMAs] and exposes the LDS latency four times per k-tile).
template <int LDA, int LDB, int KK0, int KK1>
__device__ __forceinline__ void mma_ksteps(const float *__restrict__ As, const float *__restrict__ Bs, int wm, int wn,
                                           int lane, Acc &acc)
{
    const int i = lane & 31, g = lane >> 5;
    const float *ap = As + g * LDA + wm + 2 * i;
    const float *bp = Bs + g * LDB + wn + 2 * i;
    float2 a = *reinterpret_cast<const float2 *>(ap + 2 * KK0 * LDA);
    float2 b = *reinterpret_cast<const float2 *>(bp + 2 * KK0 * LDB);
#pragma unroll
    for (int kk = KK0; kk < KK1; ++kk) {
        float2 an = a, bn = b;
        if (kk + 1 < KK1) {
            an = *reinterpret_cast<const float2 *>(ap + 2 * (kk + 1) * LDA);
            bn = *reinterpret_cast<const float2 *>(bp + 2 * (kk + 1) * LDB);
        }
        acc.v[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc.v[0][0], 0, 0, 0);
        acc.v[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.y, acc.v[0][1], 0, 0, 0);
        acc.v[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, acc.v[1][0], 0, 0, 0);
        acc.v[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc.v[1][1], 0, 0, 0);
        a = an;
        b = bn;
        if (kk + 1 < KK1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 LDS reads (step kk+1)
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                       // 4 MFMAs   (step kk)
    }
}

template <int LDA, int LDB>
__device__ __forceinline__ void mma_ktile(const float *__restrict__ As, const float *__restrict__ Bs, int wm, int wn,
                                          int lane, Acc &acc)
{
    mma_ksteps<LDA, LDB, 0, kBK / 2>(As, Bs, wm, wn, lane, acc);
}

// Staging registers for one operand k-tile of width WD: WD/64 float4 per thread.
template <int WD>
struct Stage {
    float4 v[TileGeom<WD>::nv];
};

// FAST = operands are 16-B aligned with the contiguous extent a multiple of 4 (checked on the host): every float4
// is entirely inside or entirely outside the matrix, so the load is issued UNCONDITIONALLY from a clamped
// address and zeroed by a select -- no exec-masked branches, the k-tile's loads all stay in flight together.
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

// ---- "MC" operand: stored k-major in global memory (row = k, contiguous along the tile's m/n dimension).
// tile = kBK rows x WD floats; float4 f = tid + 256*j sits at (row f / (WD/4), float4-column f % (WD/4)).
// row_ptr(k) returns the address of matrix element (k, 0) or nullptr when row k is all-zero / out of range;
// `col0` = first tile column in the matrix, `ncols` = matrix extent along the contiguous dimension.
template <int WD, bool FAST, typename RowPtr>
__device__ __forceinline__ void load_mc(Stage<WD> &s, RowPtr row_ptr, int k0, int col0, int ncols, bool vec, int tid,
                                        const float *safe)
{
    constexpr int c4n = WD / 4;
#pragma unroll
    for (int j = 0; j < TileGeom<WD>::nv; ++j) {
        const int f = tid + kThreads * j;
        s.v[j] = load4_guarded<FAST>(row_ptr(k0 + f / c4n), col0 + 4 * (f % c4n), ncols, vec, safe);
    }
}

template <int WD>
__device__ __forceinline__ void store_mc(const Stage<WD> &s, float *tile, int tid)
{
    constexpr int c4n = WD / 4;
#pragma unroll
    for (int j = 0; j < TileGeom<WD>::nv; ++j) {
        const int f = tid + kThreads * j;
        *reinterpret_cast<float4 *>(tile + (f / c4n) * TileGeom<WD>::ld + 4 * (f % c4n)) = s.v[j];
    }
}

// ---- "KC" operand: stored row-major with k contiguous (row = tile row m/n).  Thread t owns tile row t % WD and
// the 4*nv consecutive k values starting at 4*nv*(t / WD).  row_ptr(r) returns the address of element (r, k = 0)
// or nullptr when the row is out of range / zero; `kext` = matrix extent along k.
template <int WD, bool FAST, typename RowPtr>
__device__ __forceinline__ void load_kc(Stage<WD> &s, RowPtr row_ptr, int k0, int kext, bool vec, int tid,
                                        const float *safe)
{
    const float *p = row_ptr(tid % WD);
    const int k = k0 + 4 * TileGeom<WD>::nv * (tid / WD);
#pragma unroll
    for (int j = 0; j < TileGeom<WD>::nv; ++j) s.v[j] = load4_guarded<FAST>(p, k + 4 * j, kext, vec, safe);
}

template <int WD>
__device__ __forceinline__ void store_kc(const Stage<WD> &s, float *tile, int tid)
{
    constexpr int ld = TileGeom<WD>::ld;
    float *base = tile + (4 * TileGeom<WD>::nv * (tid / WD)) * ld + (tid % WD);
#pragma unroll
    for (int j = 0; j < TileGeom<WD>::nv; ++j) {
        base[(4 * j + 0) * ld] = s.v[j].x;
        base[(4 * j + 1) * ld] = s.v[j].y;
        base[(4 * j + 2) * ld] = s.v[j].z;
        base[(4 * j + 3) * ld] = s.v[j].w;
    }
}

// Epilogue visitor: calls f(row, col, v0, v1) for every pair of horizontally adjacent outputs this lane holds
// (row/col relative to the block tile; v0 at col, v1 at col+1).
template <typename F>
__device__ __forceinline__ void acc_foreach_pair(const Acc &acc, int wm, int wn, int lane, F f)
{
    const int j = lane & 31, g = lane >> 5;
#pragma unroll
    for (int sm = 0; sm < 2; ++sm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ri = (r & 3) + 8 * (r >> 2) + 4 * g;
            f(wm + 2 * ri + sm, wn + 2 * j, acc.v[sm][0][r], acc.v[sm][1][r]);
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

}  // namespace mh

namespace mh {
// defined in gemm.hip
int choose_splitk_tiles(long long tiles, int ktiles, double out_elems, double flops);
int launch_splitk_reduce(const float *partial, int splitk, long long M, int N, float *C, int ldc, const float *bias,
                         int epilogue, int accumulate, hipStream_t st);
}  // namespace mh
