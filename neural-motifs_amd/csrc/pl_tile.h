// pl_tile.h -- the round-3 "plane" tile engine of libmotifs_hip.so (gfx950 only).
//
// fp32-accurate matrix products on the f16 matrix cores ("f16x3"): an fp32 operand element x of operand
// ROW r is stored as two f16 terms  h1 = f16(x 2^e_r),  h2 = f16(x 2^e_r - h1)  (round to nearest), and
//     sum_k a b  =  2^-(e_a + e_b) * sum_k (h2a h1b + h1a h2b + h1a h1b)         [h2a h2b <= 2^-22 |ab| dropped]
// with three v_mfma_f32_32x32x16_f16 per 16 k, accumulated in fp32.  Rounds 1-2 split the operands INSIDE the K loop
// (5.5 VALU instructions per MFMA, 1.3 GHz effective clock, profiles/r02_pmc_mfma_kernels.csv).  Here every operand
// already IS a plane image in HBM, written once by its producer (pl_gemm.hip: make_planes; the conv's packed weights;
// the activation converter), and the K loop is nothing but   buffer_load_dwordx4 -> ds_write_b128 -> ds_read_b128 -> MFMA.
//
// Plane image of an operand with R rows (the non-K index) and K columns ("PL16", k-chunk-major):
//     cell(kc, r) = 64 bytes at ((kc * R) + r) * 64 :   h1[k = 16 kc .. 16 kc + 15] | h2[same k]      (k >= K: zeros)
// so the 16-byte chunk c of a cell is  c = 2 * plane + (k / 8) % 2,  and the cells of consecutive rows of one k-chunk are
// contiguous: a BM-row tile of one k-tile is ONE contiguous BM * 64-byte run (full 128-byte lines, no half-line
// segments -- the fp32 row-major operands of rounds 1-2 were fetched as 64-byte row segments).
// Behind the cells live the rows' largest |x| as fp32 bit patterns (uint32 per row, "maxbits"); the power-of-two scale is
// row_exponent(maxbits[r]).
//
// Block = 256 threads = 4 waves.  Block tile BM x BN x 16; a wave owns (32 SM) x (32 SN) outputs as SM x SN 32x32
// accumulators.  Shapes in use: 256x128 (SM 4, SN 2: 24 MFMAs, 12 fragment reads, 6 chunk copies per thread and k-tile),
// 128x128 (2, 2), 256x64 (2, 2; waves stacked 4 x 1).
// LDS tile: row r = 64 bytes, chunk c at slot  c ^ ((r >> 2) & 3):
//   * ds_read_b128 of a fragment (lane (i, g) reads row w0 + i, chunk 2 p + g) is conflict-free for the instruction's four
//     16-lane service groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: inside a group the pairs (r & 3, (r >> 2) & 3)
//     are all distinct, i.e. the 16 lanes cover the 16 sixteen-byte slots of the 256-byte bank row exactly once;
//   * ds_write_b128 of the staging copy (chunk e = tid + 256 j -> row e / 4, chunk e % 4; eight consecutive lanes = two
//     rows = 128 contiguous bytes, permuted) is conflict-free too.
//   (tests/test_pl_layout.py replays both with the bank model of MI355X_MICROARCH.md.)
// Pipeline: prefetch distance 1 with ONE register stage.  Step kt:  issue the loads of tile kt+1 | read the fragments of
// tile kt | MFMAs, with the ds_writes of tile kt+1 (other LDS buffer) interleaved into the second half of the MFMA
// sequence | barrier.  24 MFMAs = 768 matrix-pipe cycles per wave and step give the loads their flight time; the second
// resident block of the CU fills what is left.
#pragma once
#include "common.h"

namespace mh {
namespace pl {

constexpr int kBK = 16;
constexpr int kThreads = 256;
constexpr int kCell = 64;                      // bytes of one (row, k-chunk) cell
constexpr unsigned kOob = 0x80000000u;         // per-lane offset of a lane that must read zeros
constexpr unsigned kRsrcBytes = 0x7ffffff0u;   // descriptor extent: every real offset (< 2 GiB, host-checked) is inside

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// exponent e with max * 2^e in [2^14, 2^15) from the bit pattern of max = the largest |x| of a row (0 for an all-zero, inf
// or nan row: nothing to scale / nothing to save)
__host__ __device__ __forceinline__ int row_exponent(unsigned absmax_bits)
{
    const int biased = (int)(absmax_bits >> 23) & 0xff;
    if (biased == 0 || biased == 0xff) return 0;
    return 14 - (biased - 127);
}

// (x0, x1) -> packed (h1(x0), h1(x1)), (h2(x0), h2(x1)) after scaling by 2^e
__device__ __forceinline__ void split2(float x0, float x1, int e, unsigned &p1, unsigned &p2)
{
    const f32x2 xs = {__builtin_ldexpf(x0, e), __builtin_ldexpf(x1, e)};
    const f16x2 h1 = __builtin_convertvector(xs, f16x2);
    const f32x2 r = xs - __builtin_convertvector(h1, f32x2);     // exact
    p1 = __builtin_bit_cast(unsigned, h1);
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

// the same for a pair that is already scaled
__device__ __forceinline__ void split2_scaled(float xs0, float xs1, unsigned &p1, unsigned &p2)
{
    const f32x2 xs = {xs0, xs1};
    const f16x2 h1 = __builtin_convertvector(xs, f16x2);
    const f32x2 r = xs - __builtin_convertvector(h1, f32x2);     // exact
    p1 = __builtin_bit_cast(unsigned, h1);
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

struct Src {
    __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ Src make_src(const void *base)
{
    Src s;
    s.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)kRsrcBytes, 0x00020000);
    return s;
}
__device__ __forceinline__ u32x4 load16(const Src &s, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, (int)voff, (int)soff, 0));
}

__device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }
// byte offset of chunk c of tile row `row` inside an LDS operand tile
__device__ __forceinline__ unsigned lds_chunk(int row, int c) { return (unsigned)(row * kCell + 16 * (c ^ swz(row))); }

template <int BM, int BN, int SM, int SN>
struct Shape {
    static constexpr int bm = BM, bn = BN, sm = SM, sn = SN;
    static constexpr int waves_m = BM / (32 * SM), waves_n = BN / (32 * SN);
    static_assert(waves_m * waves_n == 4, "four waves per block");
    static constexpr int na = BM / 64, nb = BN / 64;             // 16-byte chunks per thread and k-tile (A, B)
    static constexpr int a_bytes = BM * kCell, b_bytes = BN * kCell;
    static constexpr int buf_bytes = a_bytes + b_bytes;
    static constexpr int lds_bytes = 2 * buf_bytes;
    static constexpr int mfmas = 3 * SM * SN;
};

template <class S>
struct Acc {
    f32x16 v[S::sm][S::sn];
};
template <class S>
__device__ __forceinline__ void acc_zero(Acc<S> &a)
{
#pragma unroll
    for (int i = 0; i < S::sm; ++i)
#pragma unroll
        for (int j = 0; j < S::sn; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) a.v[i][j][r] = 0.f;
}

// staging registers of one k-tile
template <class S>
struct Stage {
    u32x4 a[S::na], b[S::nb];
};

// chunk j of this thread: tile row (tid >> 2) + 64 j, chunk tid & 3
template <class S>
struct CopyPlan {
    unsigned va[S::na], vb[S::nb];   // per-lane global byte offsets (kOob: the row is outside the operand -> zeros)
    unsigned lds_a, lds_b;           // LDS byte offset of chunk j = lds_x + 4096 j (64 rows further down: same swizzle)
};
template <class S, typename RowOkA, typename RowOkB>
__device__ __forceinline__ void plan_copy(CopyPlan<S> &p, RowOkA a_ok, RowOkB b_ok, int tid)
{
    const int row = tid >> 2, c = tid & 3;
#pragma unroll
    for (int j = 0; j < S::na; ++j) p.va[j] = a_ok(row + 64 * j) ? (unsigned)(tid * 16 + 4096 * j) : kOob;
#pragma unroll
    for (int j = 0; j < S::nb; ++j) p.vb[j] = b_ok(row + 64 * j) ? (unsigned)(tid * 16 + 4096 * j) : kOob;
    p.lds_a = lds_chunk(row, c);
    p.lds_b = (unsigned)S::a_bytes + lds_chunk(row, c);
}
template <class S>
__device__ __forceinline__ void store_stage(const Stage<S> &st, const CopyPlan<S> &p, char *buf)
{
#pragma unroll
    for (int j = 0; j < S::na; ++j) *reinterpret_cast<u32x4 *>(buf + p.lds_a + 4096 * j) = st.a[j];
#pragma unroll
    for (int j = 0; j < S::nb; ++j) *reinterpret_cast<u32x4 *>(buf + p.lds_b + 4096 * j) = st.b[j];
}

template <class S>
struct Frags {
    f16x8 a[S::sm][2], b[S::sn][2];   // [sub-tile][plane]
};
// per-lane LDS byte offsets of the fragment reads: row i = lane & 31 of a sub-tile, k-half g = lane >> 5; a sub-tile
// further down (32 rows = 2048 bytes) keeps the swizzle, so sub-tiles are immediates
struct FragPlan {
    unsigned a[2], b[2];   // [plane]
};
template <class S>
__device__ __forceinline__ void plan_frags(FragPlan &f, int wm, int wn, int lane)
{
    const int i = lane & 31, g = lane >> 5;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        f.a[p] = lds_chunk(wm + i, 2 * p + g);
        f.b[p] = (unsigned)S::a_bytes + lds_chunk(wn + i, 2 * p + g);
    }
}
// planes in order of first use by mma(): A h2, B h1 (term 0), then B h2, A h1
template <class S>
__device__ __forceinline__ void fetch_frags(Frags<S> &f, const FragPlan &fp, const char *buf)
{
#pragma unroll
    for (int s = 0; s < S::sm; ++s) f.a[s][1] = *reinterpret_cast<const f16x8 *>(buf + fp.a[1] + 2048 * s);
#pragma unroll
    for (int s = 0; s < S::sn; ++s) f.b[s][0] = *reinterpret_cast<const f16x8 *>(buf + fp.b[0] + 2048 * s);
#pragma unroll
    for (int s = 0; s < S::sn; ++s) f.b[s][1] = *reinterpret_cast<const f16x8 *>(buf + fp.b[1] + 2048 * s);
#pragma unroll
    for (int s = 0; s < S::sm; ++s) f.a[s][0] = *reinterpret_cast<const f16x8 *>(buf + fp.a[0] + 2048 * s);
}
// the three terms of every accumulator, smallest first: h2a h1b, h1a h2b, h1a h1b; consecutive MFMAs are independent
template <class S>
__device__ __forceinline__ void mma(const Frags<S> &f, Acc<S> &acc)
{
    constexpr int kTermA[3] = {1, 0, 0}, kTermB[3] = {0, 1, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int sm = 0; sm < S::sm; ++sm)
#pragma unroll
            for (int sn = 0; sn < S::sn; ++sn)
                acc.v[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[sm][kTermA[t]], f.b[sn][kTermB[t]], acc.v[sm][sn], 0, 0, 0);
}

// One step of the K loop: `issue(st)` issues the loads of the NEXT tile into st (pinned at the top), the fragments of the
// current tile are read from `cur`, the MFMAs run, and the stage is written to `nxt` interleaved into the tail of the MFMA
// sequence (the loads have had the head of the sequence to land), then one barrier.
template <class S, typename Issue>
__device__ __forceinline__ void k_step(Issue issue, Stage<S> &st, const CopyPlan<S> &cp, const FragPlan &fp, const char *cur,
                                       char *nxt, Acc<S> &acc)
{
    issue(st);
    __builtin_amdgcn_sched_barrier(0);
    Frags<S> f;
    fetch_frags<S>(f, fp, cur);
    store_stage<S>(st, cp, nxt);
    mma<S>(f, acc);
    constexpr int nread = 2 * (S::sm + S::sn), nwrite = S::na + S::nb;
    constexpr int head = S::mfmas - 2 * nwrite;                  // MFMAs before the first LDS write (>= 0 for all shapes)
    static_assert(head >= 0, "shape");
    __builtin_amdgcn_sched_group_barrier(0x100, nread, 0);       // fragment reads
    if (head > 0) __builtin_amdgcn_sched_group_barrier(0x008, head, 0);
#pragma unroll
    for (int i = 0; i < nwrite; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
    __syncthreads();
}

// wave -> origin of its (32 SM) x (32 SN) sub-tile: waves walk N fastest
template <class S>
__device__ __forceinline__ void wave_origin(int wave, int &wm, int &wn)
{
    wm = (wave / S::waves_n) * 32 * S::sm;
    wn = (wave % S::waves_n) * 32 * S::sn;
}

// Epilogue visitor: f(row, col, sn, v) for every output this lane holds (row / col relative to the block tile).  The 32 lanes
// of a half-wave hold 32 consecutive columns of one row: 128-byte stores.
// C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), r in [0, 16).
template <class S, typename F>
__device__ __forceinline__ void acc_foreach(const Acc<S> &acc, int wm, int wn, int lane, F f)
{
    const int j = lane & 31, g = lane >> 5;
#pragma unroll
    for (int sm = 0; sm < S::sm; ++sm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm + 32 * sm + (r & 3) + 8 * (r >> 2) + 4 * g;
#pragma unroll
            for (int sn = 0; sn < S::sn; ++sn) f(row, wn + 32 * sn + j, sn, acc.v[sm][sn][r]);
        }
}

// XCD-aware remap of a linear block id: consecutive ids go round-robin over the 8 XCDs, so give each XCD a contiguous
// chunk of the tile numbering (neighbouring tiles share operand panels -> that XCD's L2)
__host__ __device__ __forceinline__ int xcd_remap(int bid, int nblocks)
{
    constexpr int kXcd = 8;
    const int q = nblocks / kXcd, r = nblocks % kXcd;
    const int xcd = bid % kXcd, idx = bid / kXcd;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}
// patch-major numbering of a tiles_m x tiles_n tile space (ph x pw tile patches, row-major inside a patch)
__host__ __device__ __forceinline__ void patch_tile(int t, int tiles_m, int tiles_n, int ph, int pw, int &tm, int &tn)
{
    const int band = t / (ph * tiles_n);
    int rem = t - band * (ph * tiles_n);
    const int bh = ph < tiles_m - band * ph ? ph : tiles_m - band * ph;
    const int npw = (tiles_n + pw - 1) / pw;
    const int j = rem / (bh * pw) < npw - 1 ? rem / (bh * pw) : npw - 1;
    rem -= j * (bh * pw);
    const int w = pw < tiles_n - j * pw ? pw : tiles_n - j * pw;
    tm = band * ph + rem / w;
    tn = j * pw + rem % w;
}

// largest |x| (fp32 bits) of a wave's lanes -> ONE atomicMax when all lanes share `key` (the usual case); lanes with
// differing keys fall back to their own atomics.  Every lane of the wave must call it.
//
// Round 6: the atomic is issued only when it can raise the word.  A layer's tiles all aim at the same B words (one per image,
// and the tiles of one image run at the same time): conv1_2 sent 65 k atomicMax at six addresses per launch, conv2_x 16 k --
// same-address atomics are served one after the other by one L2 channel (~11-13 ns each, MI355X_MICROARCH.md "fanin"), and a
// wave does not retire before its own has been served.  A maximum converges after the first few tiles, so a relaxed
// agent-scope load (L2-served, never a stale L1 line) in front of the atomic removes nearly all of them; a load that is behind
// the word's latest value only costs an atomic that was not needed (the word never decreases).
__device__ __forceinline__ bool may_raise(const unsigned *word, unsigned v)
{
#ifdef MH_ATOMIC_ALWAYS       // A/B build knob (csrc/build.py): rounds 3-5, every wave sends its atomic
    return true;
#else
    return v > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void wave_atomic_max(unsigned *words, int key, unsigned v)
{
    const int key0 = __shfl(key, 0);
    if (__all(key == key0)) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
        if ((threadIdx.x & 63) == 0 && v && may_raise(words + key0, v)) atomicMax(words + key0, v);
    } else if (v && may_raise(words + key, v)) {
        atomicMax(words + key, v);
    }
}

template <auto Kern, typename Args>
inline void launch(dim3 grid, size_t lds_bytes, hipStream_t st, const Args &p, size_t lds_max = 0, int threads = kThreads)
{
    // lds_max: the largest dynamic LDS size ANY launch of this kernel may ask for (the attribute is set once per device)
    static bool raised[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !raised[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(lds_max > lds_bytes ? lds_max : lds_bytes));
        raised[dev] = true;
    }
    hipLaunchKernelGGL(Kern, grid, dim3((unsigned)threads), lds_bytes, st, p);
}

}  // namespace pl
}  // namespace mh
