// mfma_tile.h -- the shared fp32 matrix-core block-tile engine (gfx950), block tile BM x BN x 16.
//
// One workgroup = 256 threads = 4 waves; each wave owns a 64x64 output tile held as 2x2 32x32 fp32 accumulators
// (16 VGPRs each).  Two block shapes are used:
//   128x128 (waves 2x2)  -- the default
//   256x64  (waves 4x1)  -- for N <= 64 (the 64-channel VGG layer), so no MFMA issues on padding columns
// The inner loop is "f16x3": an fp32 operand row is scaled by a power of two and split into two f16 terms by the thread that
// stages it (once per block), the LDS image holds the (h1 | h2) planes, and three v_mfma_f32_32x32x16_f16 per accumulator
// and k-tile accumulate h2*h1', h1*h2', h1*h1' in fp32 (error against fp64: tools/split_check.cpp,
// profiles/r02_split_check.jsonl).  This is the round-2 engine: since round 3 the big products run on pre-split plane images
// with no arithmetic in the K loop (pl_tile.h); this header serves the small products, the mask tower's conv and the conv
// weight gradient.
//
// Operand orientations: "WM" (width-major, K-contiguous: activations, nn.Linear weights, NHWC pixels) and "KM" (k-major:
// the tile's row / column dimension contiguous -- B of y = x*W with W [K,N], both operands of a weight gradient, which are
// transposed in registers through k-pairs while being split).  The k index of a tile is permuted consistently for both
// operands: MFMA lane group g = lane>>5 takes k = 8g .. 8g+7 (the order of the fp32 summation over k changes, nothing else).
// Accumulator (C/D) map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5), r in [0,16).
#pragma once
#include "common.h"

#ifndef MH_MINW
#define MH_MINW 2
#endif
// The engine of this header is "f16x3": two-term f16 split with a power-of-two scale per operand row, three
// v_mfma_f32_32x32x16_f16 per product.  Round 1 / 2 carried two more builds of the same kernels (bf16x6: three-term bf16
// split, six MFMAs; f32-input MFMA) behind MH_SPLIT_F16 / MH_MFMA_SPLIT / MH_SPLIT_RN; their measurements are in
// profiles/r01_split_check.jsonl and r02_split_check.jsonl, their code was removed in round 3 (git history: c0ffe9e).
#define MH_MFMA_SPLIT 3      /* MFMAs per fp32 product (reported by mh_mfma_split()) */
#define MH_SPLIT_F16 1       /* reported by mh_split_f16() */

namespace mh {

constexpr int kBK = 16;
constexpr int kThreads = 256;
constexpr int kLdW = kBK + 4;   // WM row stride (floats)

constexpr int kRowDw = 24;      // bf16-plane layout: dwords per operand row (3 planes x 8 dwords = 96 B)
constexpr int kNumPlanes = 2;          // planes a row really holds (h1 | h2; the row keeps round 1's 96-byte stride, third slot pair unused)
constexpr int kPlaneChunks = 2 * kNumPlanes;              // 16-byte chunks per row of an operand stored as planes in HBM
constexpr int kPlaneRowBytes = 16 * kPlaneChunks;         // ... and its bytes per (row, k-tile): 96 (bf16x6) / 64 (f16x3)

template <int WD, bool WM>
struct TileGeom {
    static constexpr int ld = WM ? kLdW : WD + 4;        // fp32 layout: LDS row stride (floats); keeps 16-B alignment
    static constexpr int floats = WD * kRowDw;
    static constexpr int nv = WD * kBK / 1024;            // float4 staged per thread for a WM operand (256 threads)
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

// ---------------------------------------------------------------------------------------------------------------
// FP32-accurate products on the f16 matrix cores ("f16x3").
// gfx950's v_mfma_f32_32x32x16_f16 retires 16 k per 32 cycles, the f32-input MFMA 2 k per 64 cycles: 16x the rate.
// A row of an operand is scaled by 2^e (row_exponent: its largest magnitude lands in [2^14, 2^15)) and every element is
// split into two f16 terms, a*2^e = h1 + h2 + r with |r| <= 2^-24 |a*2^e|; f16 x f16 products are exact in fp32 and the
// MFMA accumulates in fp32, so three MFMAs (h2*h1', h1*h2', h1*h1') give the product to ~2^-22 and the scales come off
// exactly in the epilogue (measured against float64: DESIGN.md 3.1, profiles/r02_split_check.jsonl).  A lane's operand for
// the K=16 instruction is 8 consecutive k of its row (k = 8g .. 8g+7, g = lane >> 5).  The split is done ONCE per element
// when a k-tile is staged (below).  The vector type keeps round 1's name: 8 x 16 bits, here f16 bit patterns.
// ---------------------------------------------------------------------------------------------------------------
typedef short bf16x8 __attribute__((ext_vector_type(8)));


// Staging registers for one operand k-tile of width WD: WD/64 float4 per thread.
template <int WD>
struct Stage {
    float4 v[WD >= 128 ? WD / 64 : 2];   // WD = 64 staged k-major in bf16-plane mode needs a k-pair (2 float4) per thread
};

// One global operand as a block sees it: `base` is a wave-uniform origin and `rsrc` a raw buffer descriptor over
// [base, base + 1 GiB).  Loads through the descriptor are branch-free: a lane that must read zero passes an offset
// beyond the descriptor and the hardware returns 0 -- no exec-masked blocks around the loads, so every load of a
// k-tile is issued unconditionally and the compiler can count them (s_waitcnt vmcnt(N) with N > 0 is what keeps
// the loads of tile kt+2 in flight while tile kt+1 is consumed).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kBufBytes = 0x40000000u;    // operands are addressed within 1 GiB of their (block) origin
constexpr unsigned kOobOffset = 0x80000000u;   // per-lane offset of a lane that must read zero
constexpr unsigned kDeadTile = kBufBytes;      // scalar offset that pushes a whole tile out of range (no wrap with
                                               // kOobOffset: 0x80000000 + 0x40000000 < 2^32)
struct GSrc {
    const float *base;
    __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ GSrc make_gsrc(const float *base)
{
    GSrc g;
    g.base = base;
    g.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)kBufBytes, 0x00020000);
    return g;
}
__device__ __forceinline__ float4 buffer_load4(const GSrc &g, unsigned byte_off, unsigned sbyte_off = 0)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const auto raw = __builtin_amdgcn_raw_buffer_load_b128(g.rsrc, (int)byte_off, (int)sbyte_off, 0);
    static_assert(sizeof(raw) == 16, "b128");
    const f32x4 v = __builtin_bit_cast(f32x4, raw);   // (component access on the builtin's own vector type splats)
    return make_float4(v.x, v.y, v.z, v.w);
}

// FAST = operands are 16-B aligned with the contiguous extent a multiple of 4 and span < 1 GiB from g.base (checked
// on the host): every float4 is entirely inside or entirely outside the matrix -> one buffer load, see GSrc.
// Otherwise (ragged / unaligned operands) per-element guarded loads.
template <bool FAST>
__device__ __forceinline__ float4 load4_guarded(const float *p, int c, int extent, bool vec, const GSrc &g)
{
    if (FAST) {
        const bool ok = (p != nullptr) && (c < extent);
        const unsigned off = (unsigned)((p - g.base) + c) * 4u;
        return buffer_load4(g, ok ? off : kOobOffset);
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


// ---- WM operand: global rows are the tile's w dimension (k contiguous).  float4 f = tid + 256*j belongs to tile
// row f / 4, k-quad f % 4: four consecutive lanes read one row's 64 contiguous bytes.
// row_ptr(r) returns the address of element (r, k = 0) or nullptr when the row is out of range / zero;
// `kext` = matrix extent along k.
template <int WD, bool FAST, typename RowPtr>
__device__ __forceinline__ void load_wm(Stage<WD> &s, RowPtr row_ptr, int k0, int kext, bool vec, int tid,
                                        const GSrc &safe)
{
#pragma unroll
    for (int j = 0; j < TileGeom<WD, true>::nv; ++j) {
        const int f = tid + kThreads * j;
        s.v[j] = load4_guarded<FAST>(row_ptr(f >> 2), k0 + 4 * (f & 3), kext, vec, safe);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Plane LDS image: every element is split ONCE, by the thread that stages it, and LDS holds the planes k-contiguous per
// operand row, whatever the global orientation (the row keeps round 1's three-plane stride; f16x3 fills two):
//     row r (96 B = 24 dwords):  [ h1: k0..k15 | h2: k0..k15 | unused ]
// dword d of a plane holds k = 2d (low half) and 2d+1 (high half); the 16-B slot index (2*plane + k/8) is XORed
// with swz(r) = bit 3 of r.  A lane's MFMA operand (8 consecutive k of its row, one plane) is one ds_read_b128:
// conflict-free for the four 16-lane service groups of the instruction (rows r and r+8 would otherwise meet on
// the same banks: 8 * 96 B = 3 * 256 B).  A 128x128 block's double buffer is 48 KB: three blocks per CU.
//   * WM operand: the staging thread owns 4 consecutive k of one row -> 2 packed dwords per plane -> 3 ds_write_b64
//     (conflict-free).
//   * KM operand: the staging thread loads a k-PAIR (rows 2kp, 2kp+1) of 4 consecutive w -> one packed dword per
//     (w, plane) -> 12 ds_write_b32, i.e. the transposition costs nothing but narrow writes.  Lane l of a wave takes
//     w-quad l&7 and k-pair l>>3: the 32 lanes of a write group meet at most 2-way on a bank, and a global load
//     instruction reads 8 rows x 128 contiguous bytes.  Tile column w lives in row 64*(w/64) + 32*(w&1) + (w%64)/2
//     (the wave's two MFMA sub-tiles interleaved), so a lane's two outputs are adjacent columns (8-byte stores).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int plane_swz(int r) { return (r >> 3) & 1; }


// f16x3: (x0, x1) scaled by 2^e of their operand ROW (e0 / e1: the two elements may belong to different rows), then
// a*2^e = h1 + h2 + r with h1 = f16(a*2^e), h2 = f16(a*2^e - h1) (round to nearest even), |r| <= 2^-24 |a*2^e|.
// The row exponent puts the row's largest magnitude into [2^14, 2^15) (row_exponent), so nothing overflows f16.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_f16(float x0, float x1, int e0, int e1, unsigned &p1, unsigned &p2)
{
    const f32x2_t xs = {__builtin_ldexpf(x0, e0), __builtin_ldexpf(x1, e1)};
    const f16x2_t h1 = __builtin_convertvector(xs, f16x2_t);
    const f32x2_t r = xs - __builtin_convertvector(h1, f32x2_t);     // exact
    p1 = __builtin_bit_cast(unsigned, h1);
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
}
// ---- the second split of this header: "bf16x6" (round 5, small products: gemm.hip) ------------------------------------
// An fp32 number is the EXACT sum of three bf16 terms (8 + 8 + 8 significant bits), each rounded to nearest even
// (v_cvt_pk_bf16_f32): x = b1 + b2 + b3, |b2| <= 2^-8 |x|, |b3| <= 2^-16 |x|.  bf16 has fp32's exponent range, so NO scale and
// therefore NO pass over the operands for their row maxima is needed: a product is one launch.  Six MFMAs per accumulator and
// k-tile (b3*b1', b1*b3', b2*b2', b2*b1', b1*b2', b1*b1', smallest first); the three dropped cross terms are below 2^-24 |ab|.
// Measured against float64 on MI355X in round 1 (profiles/r01_split_check.jsonl, "split_rne": 1): rms 7.9e-7 of rms(C) on
// mixed-sign operands, 6.3e-7 on all-positive ones (mean signed error -1.9e-7), 24-bit operands copied exactly through a
// permutation matrix.  Twice the matrix-core work of f16x3 -- used where a product is latency- or bandwidth-bound anyway.
constexpr int kSplitF16x3 = 0, kSplitBf16x6 = 1;
typedef __bf16 bf16pair_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16_rne(float lo16, float hi16)
{
    const f32x2_t v = {lo16, hi16};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16pair_t));
}
// (x0, x1) = values at k even / k odd -> packed dwords of the three planes
__device__ __forceinline__ void split_pair_bf16(float x0, float x1, unsigned &p1, unsigned &p2, unsigned &p3)
{
    p1 = pack_bf16_rne(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, p1 << 16);               // exact
    const float r1 = x1 - __builtin_bit_cast(float, p1 & 0xffff0000u);
    p2 = pack_bf16_rne(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, p2 << 16);               // exact
    const float s1 = r1 - __builtin_bit_cast(float, p2 & 0xffff0000u);
    p3 = pack_bf16_rne(s0, s1);                                              // exact: at most 8 significant bits are left
}
template <int SP>
__device__ __forceinline__ void split_pair_any(float x0, float x1, int e0, int e1, unsigned (&pl)[3])
{
    if (SP == kSplitBf16x6) split_pair_bf16(x0, x1, pl[0], pl[1], pl[2]);
    else split_pair_f16(x0, x1, e0, e1, pl[0], pl[1]);
}
template <int SP>
struct SplitPlanes { static constexpr int n = (SP == kSplitBf16x6) ? 3 : kNumPlanes; };

// exponent e with max * 2^e in [2^14, 2^15) from the bit pattern of max = the largest |x| of a row (0 for an all-zero,
// inf or nan row: nothing to scale / nothing to save)
__host__ __device__ __forceinline__ int row_exponent(unsigned absmax_bits)
{
    const int biased = (int)(absmax_bits >> 23) & 0xff;
    if (biased == 0 || biased == 0xff) return 0;      // zero / denormal rows need no help: denormals are < 2^-126
    return 14 - (biased - 127);
}

// KM task t -> (w-quad q, k-pair kp); tasks = 2*WD (8 k-pairs x WD/4 quads), 64 per wave
template <int WD>
__device__ __forceinline__ bool km_task(int t, int &q, int &kp)
{
    q = 8 * (t >> 6) + (t & 7);
    kp = (t >> 3) & 7;
    return (2 * WD >= kThreads) || (t < 2 * WD);
}

template <int WD, bool FAST, typename RowPtr>
__device__ __forceinline__ void load_km(Stage<WD> &s, RowPtr row_ptr, int k0, int col0, int ncols, bool vec, int tid,
                                        const GSrc &safe)
{
    constexpr int ntask = (2 * WD + kThreads - 1) / kThreads;
#pragma unroll
    for (int jt = 0; jt < ntask; ++jt) {
        int q, kp;
        if (!km_task<WD>(tid + kThreads * jt, q, kp)) continue;   // wave-uniform (WD = 64: waves 2, 3 idle)
        s.v[2 * jt] = load4_guarded<FAST>(row_ptr(k0 + 2 * kp), col0 + 4 * q, ncols, vec, safe);
        s.v[2 * jt + 1] = load4_guarded<FAST>(row_ptr(k0 + 2 * kp + 1), col0 + 4 * q, ncols, vec, safe);
    }
}

// Row exponents of the f16x3 engine as the staging thread needs them: one per staged float4 of a WM operand (its
// tile row), four per k-pair task of a KM operand (its four tile columns).  Empty in the other builds.
template <int WD>
struct StageExp {
    int wm[WD >= 128 ? WD / 64 : 2];
    int km[(2 * WD + kThreads - 1) / kThreads][4];
};
// exps[i] = exponent of tile row / column i (i relative to the tile origin `o0`, `n` valid entries from there); with
// `bits` the array holds the rows' largest |x| as fp32 bit patterns (what the absmax pass writes) and the exponent is
// derived here -- no separate bits -> exponent launch
template <int WD>
__device__ __forceinline__ void load_stage_exp(StageExp<WD> &se, const int *__restrict__ exps, long long o0, long long n,
                                               bool wm, int tid, bool bits = false)
{
    auto ex = [&](long long i) { return bits ? row_exponent((unsigned)exps[i]) : exps[i]; };
    if (wm) {
#pragma unroll
        for (int j = 0; j < TileGeom<WD, true>::nv; ++j) {
            const int r = (tid + kThreads * j) >> 2;
            se.wm[j] = (o0 + r < n) ? ex(o0 + r) : 0;
        }
    } else {
        constexpr int ntask = (2 * WD + kThreads - 1) / kThreads;
#pragma unroll
        for (int jt = 0; jt < ntask; ++jt) {
            int q, kp;
            km_task<WD>(tid + kThreads * jt, q, kp);
#pragma unroll
            for (int j = 0; j < 4; ++j) se.km[jt][j] = (o0 + 4 * q + j < n) ? ex(o0 + 4 * q + j) : 0;
        }
    }
}

template <int WD, int SP = kSplitF16x3>
__device__ __forceinline__ void store_km(const Stage<WD> &s, float *tile, int tid, const StageExp<WD> &se = StageExp<WD>())
{
    constexpr int ntask = (2 * WD + kThreads - 1) / kThreads;
    unsigned *t32 = reinterpret_cast<unsigned *>(tile);
#pragma unroll
    for (int jt = 0; jt < ntask; ++jt) {
        int q, kp;
        if (!km_task<WD>(tid + kThreads * jt, q, kp)) continue;
        const float e[4] = {s.v[2 * jt].x, s.v[2 * jt].y, s.v[2 * jt].z, s.v[2 * jt].w};
        const float o[4] = {s.v[2 * jt + 1].x, s.v[2 * jt + 1].y, s.v[2 * jt + 1].z, s.v[2 * jt + 1].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int w = 4 * q + j;
            const int r = 64 * (w >> 6) + 32 * (w & 1) + ((w & 63) >> 1);
            const int sw = plane_swz(r);
            unsigned pl[3];
            split_pair_any<SP>(e[j], o[j], se.km[jt][j], se.km[jt][j], pl);     // k, k+1 of the same column
#pragma unroll
            for (int pidx = 0; pidx < SplitPlanes<SP>::n; ++pidx)
                t32[r * kRowDw + 4 * ((2 * pidx + (kp >> 2)) ^ sw) + (kp & 3)] = pl[pidx];
        }
    }
}

template <int WD, int SP = kSplitF16x3>
__device__ __forceinline__ void store_wm(const Stage<WD> &s, float *tile, int tid, const StageExp<WD> &se = StageExp<WD>())
{
    unsigned *t32 = reinterpret_cast<unsigned *>(tile);
#pragma unroll
    for (int j = 0; j < TileGeom<WD, true>::nv; ++j) {
        const int f = tid + kThreads * j;
        const int r = f >> 2, kq = f & 3, sw = plane_swz(r);
        unsigned a[3], b[3];
        split_pair_any<SP>(s.v[j].x, s.v[j].y, se.wm[j], se.wm[j], a);
        split_pair_any<SP>(s.v[j].z, s.v[j].w, se.wm[j], se.wm[j], b);
#pragma unroll
        for (int pidx = 0; pidx < SplitPlanes<SP>::n; ++pidx) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u32x2 *>(t32 + r * kRowDw + 4 * ((2 * pidx + (kq >> 1)) ^ sw) + 2 * (kq & 1)) =
                (u32x2){a[pidx], b[pidx]};
        }
    }
}

// The wave's operand fragments of one k-tile: [sub-tile][plane] for A and B, 12 ds_read_b128 in all.
struct PlaneFrags {
    bf16x8 a[2][3], b[2][3];
};
template <int BM, int BN, int SP = kSplitF16x3>
__device__ __forceinline__ void fetch_frags(PlaneFrags &f, const float *__restrict__ As, const float *__restrict__ Bs,
                                            int wm, int wn, int lane)
{
    const int i = lane & 31, g = lane >> 5;
    auto fetch = [&](const float *tile, int row, int pidx) -> bf16x8 {
        return *reinterpret_cast<const bf16x8 *>(tile + row * kRowDw + 4 * ((2 * pidx + g) ^ plane_swz(row)));
    };
    if (SP == kSplitBf16x6) {
        constexpr int kOrderA[3] = {2, 0, 1}, kOrderB[3] = {0, 2, 1};   // planes in order of first use: 12 ds_read_b128
#pragma unroll
        for (int o = 0; o < 3; ++o) {
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
                f.a[sidx][kOrderA[o]] = fetch(As, wm + 32 * sidx + i, kOrderA[o]);
                f.b[sidx][kOrderB[o]] = fetch(Bs, wn + 32 * sidx + i, kOrderB[o]);
            }
        }
        return;
    }
    constexpr int kOrderA[2] = {1, 0}, kOrderB[2] = {0, 1};         // planes in order of first use: 8 ds_read_b128
#pragma unroll
    for (int o = 0; o < 2; ++o) {
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
            f.a[sidx][kOrderA[o]] = fetch(As, wm + 32 * sidx + i, kOrderA[o]);
            f.b[sidx][kOrderB[o]] = fetch(Bs, wn + 32 * sidx + i, kOrderB[o]);
        }
    }
}
// All MFMAs of one k-tile; the four accumulators are interleaved so that consecutive MFMAs are independent.
// f16x3: h2*h1, h1*h2, h1*h1 (smallest first); h2*h2 <= 2^-24 |ab| is dropped
// bf16x6: b3*b1, b1*b3, b2*b2, b2*b1, b1*b2, b1*b1 (smallest first); b2*b3, b3*b2, b3*b3 <= 2^-24 |ab| are dropped
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <int SP = kSplitF16x3>
__device__ __forceinline__ void mma_frags(const PlaneFrags &f, Acc &acc)
{
    if (SP == kSplitBf16x6) {
        constexpr int kTermA[6] = {2, 0, 1, 1, 0, 0}, kTermB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int sm = 0; sm < 2; ++sm)
#pragma unroll
                for (int sn = 0; sn < 2; ++sn)
                    acc.v[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, f.a[sm][kTermA[t]]),
                                                                            __builtin_bit_cast(bf16x8_t, f.b[sn][kTermB[t]]),
                                                                            acc.v[sm][sn], 0, 0, 0);
        return;
    }
    constexpr int kTermA[3] = {1, 0, 0}, kTermB[3] = {0, 1, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int sm = 0; sm < 2; ++sm)
#pragma unroll
            for (int sn = 0; sn < 2; ++sn)
                acc.v[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.a[sm][kTermA[t]]),
                                                                       __builtin_bit_cast(f16x8, f.b[sn][kTermB[t]]),
                                                                       acc.v[sm][sn], 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// Planned loads (FAST operands in bf16-plane mode).  The per-lane byte offset of every staged float4 is fixed for
// the whole K loop -- only a wave-uniform amount is added per k-tile -- so it is computed ONCE (`plan`), invalid
// rows / columns get kOobOffset, and a k-tile's loads are nothing but buffer_load_dwordx4 v, voff, rsrc, soff with
// the tile's advance in the scalar offset: no address VALU in the main loop.  A tile beyond the end of the loop is
// loaded with soff = kDeadTile (zeros); the last, partial k-tile (K % 16 != 0) uses the second offset set `t`, in
// which the lanes whose k lies beyond K are out of range too (one v_cndmask on a scalar condition per load).
// ---------------------------------------------------------------------------------------------------------------
template <int WD>
struct Plan {
    static constexpr int n = WD >= 128 ? WD / 64 : 2;
    unsigned v[n], t[n];
};

// WM operand: float4 j of this thread = tile row f >> 2, k-quad f & 3 (f = tid + 256 j); ld in floats;
// ktail = number of valid k in the last k-tile (1..16)
template <int WD, typename RowOk>
__device__ __forceinline__ void plan_wm(Plan<WD> &pl, RowOk row_ok, int ld, int ktail, int tid)
{
#pragma unroll
    for (int j = 0; j < TileGeom<WD, true>::nv; ++j) {
        const int f = tid + kThreads * j, r = f >> 2, kq = f & 3;
        const unsigned off = ((unsigned)r * (unsigned)ld + 4u * kq) * 4u;
        pl.v[j] = row_ok(r) ? off : kOobOffset;
        pl.t[j] = (4 * kq < ktail) ? pl.v[j] : kOobOffset;
    }
}
template <int WD>
__device__ __forceinline__ void load_planned_wm(Stage<WD> &s, const Plan<WD> &pl, const GSrc &g, unsigned soff, bool tail)
{
#pragma unroll
    for (int j = 0; j < TileGeom<WD, true>::nv; ++j) s.v[j] = buffer_load4(g, tail ? pl.t[j] : pl.v[j], soff);
}

// KM operand (see km_task): float4 2jt + i = row k = 2 kp + i, columns 4q .. 4q+3 of the tile; ncols_left = number
// of valid columns counted from the tile's first column
template <int WD>
__device__ __forceinline__ void plan_km(Plan<WD> &pl, int ncols_left, int ld, int ktail, int tid)
{
    constexpr int ntask = (2 * WD + kThreads - 1) / kThreads;
#pragma unroll
    for (int jt = 0; jt < ntask; ++jt) {
        const int t = tid + kThreads * jt;
        const int q = 8 * (t >> 6) + (t & 7), kp = (t >> 3) & 7;
        const bool ok = (4 * q < ncols_left) && ((2 * WD >= kThreads) || (t < 2 * WD));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned off = ((unsigned)(2 * kp + i) * (unsigned)ld + 4u * q) * 4u;
            pl.v[2 * jt + i] = ok ? off : kOobOffset;
            pl.t[2 * jt + i] = (2 * kp + i < ktail) ? pl.v[2 * jt + i] : kOobOffset;
        }
    }
}
template <int WD>
__device__ __forceinline__ void load_planned_km(Stage<WD> &s, const Plan<WD> &pl, const GSrc &g, unsigned soff, bool tail)
{
    constexpr int ntask = (2 * WD + kThreads - 1) / kThreads;
#pragma unroll
    for (int j = 0; j < 2 * ntask; ++j) s.v[j] = buffer_load4(g, tail ? pl.t[j] : pl.v[j], soff);
}

// ---------------------------------------------------------------------------------------------------------------
// Operands that already ARE bf16 planes in global memory (packed weights): a tile row's k-tile is 96 contiguous
// bytes in exactly the LDS row format, so staging is 16-byte chunks copied global -> registers -> LDS with no VALU:
// chunk e = tid + 256 j is row e / 6, slot e % 6 (six consecutive lanes read one row's 96 bytes).
// ---------------------------------------------------------------------------------------------------------------
template <int WD>
struct PStage {
    static constexpr int n = (WD * kPlaneChunks + kThreads - 1) / kThreads;
    u32x4 v[n];
};
template <int WD>
struct PPlan {
    unsigned v[PStage<WD>::n];     // global byte offset of the chunk (kOobOffset: reads zeros)
    unsigned lds[PStage<WD>::n];   // LDS byte offset of the chunk (slot swizzled)
};
template <int WD, typename RowOk>
__device__ __forceinline__ void plan_planes(PPlan<WD> &pl, RowOk row_ok, unsigned row_stride_bytes, int tid)
{
#pragma unroll
    for (int j = 0; j < PStage<WD>::n; ++j) {
        const int e = tid + kThreads * j, c = e % kPlaneChunks;
        int r = e / kPlaneChunks;
        // f16x3 rows use 64 of their 96 bytes, so the four rows a 16-lane group of the ds_write_b128 covers must be chosen
        // such that their spans tile the 256 bytes of the 64 banks: rows {0,6,4,2} / {1,7,5,3} of every 8 sit at
        // 0,64,128,192 / 96,160,224,32 (mod 256).  Consecutive rows (0,96,192,288) put row 3 on row 0's banks: 2-way
        // conflicts on every write (PMC: SQ_LDS_BANK_CONFLICT = 20 % of the LDS-active cycles of the conv, r02_c14).
        r = (r & ~7) | ((0x35712460u >> (4 * (r & 7))) & 7);
        const bool ok = (e < WD * kPlaneChunks) && row_ok(r);
        pl.v[j] = ok ? (unsigned)r * row_stride_bytes + 16u * c : kOobOffset;
        pl.lds[j] = (unsigned)(r * kRowDw * 4 + 16 * (c ^ plane_swz(r)));
    }
}
template <int WD>
__device__ __forceinline__ void load_planes(PStage<WD> &s, const PPlan<WD> &pl, const GSrc &g, unsigned soff)
{
#pragma unroll
    for (int j = 0; j < PStage<WD>::n; ++j) {
        const auto raw = __builtin_amdgcn_raw_buffer_load_b128(g.rsrc, (int)pl.v[j], (int)soff, 0);
        s.v[j] = __builtin_bit_cast(u32x4, raw);
    }
}
template <int WD>
__device__ __forceinline__ void store_planes(const PStage<WD> &s, const PPlan<WD> &pl, float *tile, int tid)
{
    char *base = reinterpret_cast<char *>(tile);
#pragma unroll
    for (int j = 0; j < PStage<WD>::n; ++j)
        if ((WD * kPlaneChunks) % kThreads == 0 || tid + kThreads * j < WD * kPlaneChunks)
            *reinterpret_cast<u32x4 *>(base + pl.lds[j]) = s.v[j];
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

// Patch-major numbering of a tiles_m x tiles_n tile space: tiles are numbered patch after patch (ph x pw tiles, row-major
// inside a patch; bands of ph tile rows, the last band / the last patch of a band may be smaller).  Composed with
// xcd_remap, the ~64 blocks an XCD runs at any time (32 CUs x 2) form one or two compact patches: they walk K roughly in
// step, so each k-slice of a patch's ph row panels and pw column panels is fetched from the fabric ONCE per patch and
// served to the other blocks by that XCD's L2 -- the A operand crosses the fabric tiles_n/pw times and B tiles_m/ph times,
// instead of 1 and 8 times with the round-1 order (row bands x all columns per XCD).
__device__ __forceinline__ void patch_tile(int t, int tiles_m, int tiles_n, int ph, int pw, int &tm, int &tn)
{
    const int band = t / (ph * tiles_n);
    int rem = t - band * (ph * tiles_n);
    const int bh = min(ph, tiles_m - band * ph);
    const int npw = (tiles_n + pw - 1) / pw;
    const int j = min(rem / (bh * pw), npw - 1);
    rem -= j * (bh * pw);
    const int w = min(pw, tiles_n - j * pw);
    tm = band * ph + rem / w;
    tn = j * pw + rem % w;
}

// Launch a tile kernel with its LDS image as dynamic shared memory (sized by the layout in use: tile_lds_bytes).
template <auto Kern, typename Args>
inline void launch_tile_kernel(dim3 grid, size_t lds_bytes, hipStream_t st, const Args &p)
{
    static bool raised[64] = {};
    int dev = 0;
    hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !raised[dev]) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes);
        raised[dev] = true;
    }
    hipLaunchKernelGGL(Kern, grid, dim3(kThreads), lds_bytes, st, p);
}
template <int BM, int BN, bool AWM, bool BWM>
constexpr size_t tile_lds_bytes() { return 2 * (size_t)(TileGeom<BM, AWM>::floats + TileGeom<BN, BWM>::floats) * sizeof(float); }

// One branch-free half-step of the main loop (bf16-plane mode), shared by the GEMM and the conv kernel:
//   1. global loads of tile kt+2 (pinned at the top: two iterations of flight time),
//   2. the 12 fragment reads of tile kt (LDS buffer `cur`),
//   3. MFMAs of tile kt interleaved with the split + LDS write of tile kt+1 (into the other buffer), in ONE basic
//      block: the reads are issued before the writes in program order, so the MFMAs depend on registers only and
//      the scheduler may put the staging VALU / DS-write work into their shadow (recipe below),
//   4. barrier.
// Tiles beyond the last one are loaded as zeros (masked buffer loads), which makes the phantom half-step of an odd
// tile count harmless (acc += 0) and keeps the loop free of branches -- the load count per step is static, so the
// compiler waits with vmcnt(N > 0) and tile kt+2 stays in flight while tile kt+1 is consumed.
template <int BM, int BN, int SP = kSplitF16x3, typename LoadFn, typename StoreFn>
__device__ __forceinline__ void half_step(LoadFn load_far, StoreFn store_next, const float *As, const float *Bs, int wm,
                                          int wn, int lane, Acc &acc)
{
    load_far();
    __builtin_amdgcn_sched_barrier(0);
    PlaneFrags f;
    fetch_frags<BM, BN, SP>(f, As, Bs, wm, wn, lane);
    store_next();
    mma_frags<SP>(f, acc);
#ifndef MH_F16_VALU
#define MH_F16_VALU 7     /* split / address VALU instructions the scheduler may place behind each MFMA (build knob for A/B) */
#endif
    if (SP == kSplitBf16x6) {
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);       // fragment reads (three planes)
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);       // first split ops while the reads land
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);    // split / address VALU
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // LDS write of the next tile
        }
    } else {
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);        // fragment reads (two planes)
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, MH_F16_VALU, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
    }
    __syncthreads();
}

// defined in gemm.hip
int launch_row_exponents(const float *X, bool k_contiguous, long long n_rows, long long kext, long long ld, int *exps,
                         hipStream_t st, bool bits_only = false);
// largest |x| (fp32 bit patterns) of the rows of TWO operands in one launch (+ one memset when a k-major operand needs
// zeroed atomics): bitsA[a_rows], bitsB[b_rows] must be ADJACENT in memory (bitsA first, 256-byte aligned sizes)
int launch_operand_absmax(const float *A, bool a_kcontig, long long a_rows, long long a_kext, long long lda, int *bitsA,
                          const float *B, bool b_kcontig, long long b_rows, long long b_kext, long long ldb, int *bitsB,
                          hipStream_t st);
int choose_splitk_tiles(long long tiles, int ktiles, double out_elems, double flops);
// defined in conv.hip: the tile schedule of a 3x3 conv launch with bm x bn block tiles (ConvArgs explains the fields)
struct ConvTilePlan {
    int tiles_m, tiles_n, splitk, body_mtiles, tail_slices;
};
// blocks of a tile kernel the schedules assume resident at once: 2 per CU.  The f16x3 kernels fit 3 (154-156 VGPRs, 48 KB
// LDS), but schedules built for 768 slots measured SLOWER on MI355X (trunk 6.11 ms vs 5.82 ms at 512, gpurun r02_c3);
// MH_SLOTS overrides it for A/B runs.
int resident_slots();
void set_resident_slots_override(int slots);   // > 0: what resident_slots() returns on this thread until reset with 0
double makespan_units(long long blocks);   // time of `blocks` equal blocks in units of (one block alone on a full CU)
ConvTilePlan plan_conv_tiles(long long M, int Cin, int Cout, int bm, int bn);
int launch_splitk_reduce(const float *partial, int splitk, long long M, int N, float *C, int ldc, const float *bias,
                         int epilogue, int accumulate, hipStream_t st);

}  // namespace mh
