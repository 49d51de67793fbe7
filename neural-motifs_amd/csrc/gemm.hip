// gemm.hip -- the SMALL-PRODUCT engine (round 5) and the helpers the other tile engines share.
//   * mh_gemm_small_f32 / mh_gemm_f32_v2: every nn.Linear-shaped product that is not worth plane images (pl_gemm.hip routes
//     products below 20 GFLOP or with K < 512 here): the ~40 small / skinny products of a step -- obj_embed, pos_embed, the LSTM
//     and decoder projections, post_lstm, rel_compress, the object RoI head's fc6 / fc7 on ~120 rows, and their input / weight
//     gradients.  fp32 operands are read ONCE and split into three bf16 terms by the threads that stage them ("bf16x6": six
//     v_mfma_f32_32x32x16_bf16 per accumulator and k-tile, fp32 accumulate; mfma_tile.h); no power-of-two scales, so no pass
//     over the operands in front of the product.  Block tile 128x128x16 (256x64 for N <= 64), LDS double-buffered, prefetch
//     distance 2.  These products are latency chains, not throughput problems: plan_small() cuts K into as many slices as pays,
//     and the last block of a tile to finish adds the slices in slice order inside the same launch (arrival counters): ONE launch
//     per product where round 4 issued three (row maxima, product, reduction).
//   * absmax_dual_kernel & co.: row maxima of operands for the f16x3 engines (plane images: pl_gemm.hip; conv.hip's weight
//     gradient), the work-distribution model shared with the conv schedules, splitk_reduce_kernel.
#include <algorithm>
#include <cstdlib>

#include <type_traits>
#include "mfma_tile.h"

namespace mh {

struct GemmArgs {
    int M, N, K;
    const float *A;
    int lda;
    const float *B;
    int ldb;
    float *C;
    int ldc;
    const float *bias;
    int epilogue, accumulate;
    int splitk, ktiles_per_split;
    float *partial;  // [splitk][M][N] when splitk > 1
    int vecA, vecB, vecC;
    int tiles_m, tiles_n;
    int patch_h, patch_w;     // patch-major tile order (mfma_tile.h: patch_tile)
    const int *expA, *expB;   // f16x3: power-of-two exponent per row of op(A) [M] and per column of op(B) [N]
    int *counters;            // split-K, fused reduction: one arrival counter per output tile (zero on entry, left zero), or NULL
    int vecC4;                // C rows 16-byte aligned (fused reduction: float4 stores)
};

__device__ __forceinline__ float apply_epi(float v, int epilogue)
{
    if (epilogue == MH_EPI_RELU) return fmaxf(v, 0.f);
    if (epilogue == MH_EPI_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
    return v;
}

// SP = kSplitF16x3: the operands' row maxima (expA / expB) must have been computed by a pass in front of this kernel;
// SP = kSplitBf16x6: no scales, the kernel is the whole product (mfma_tile.h).
// Split-K (blockIdx.y = K slice) with p.counters: the LAST block to finish a tile adds the slices' partial sums in slice
// order (bit-identical to splitk_reduce_kernel, whatever the arrival order) and applies bias / activation -- no reduce launch.
template <bool TA, bool TB, int BM, int BN, bool FAST, int SP = kSplitF16x3>
__global__ __launch_bounds__(kThreads, MH_MINW) void gemm_kernel(const GemmArgs p)
{
    constexpr bool AWM = !TA, BWM = TB;          // K-contiguous global storage -> width-major LDS tile
    constexpr int FA = TileGeom<BM, AWM>::floats, FB = TileGeom<BN, BWM>::floats;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x (FA + FB) floats, see launch_tile_kernel
    auto As = [&](int buf) -> float * { return lds + buf * (FA + FB); };
    auto Bs = [&](int buf) -> float * { return lds + buf * (FA + FB) + FA; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int wm, wn;
    wave_origin<BM, BN>(wave, wm, wn);
    const int ntiles = p.tiles_m * p.tiles_n;
    const int t = xcd_remap(blockIdx.x, ntiles);
    int tm, tn;
    patch_tile(t, p.tiles_m, p.tiles_n, p.patch_h, p.patch_w, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.y;
    const int total_kt = (p.K + kBK - 1) / kBK;
    const int kt_begin = z * p.ktiles_per_split;
    const int kt_end = min(total_kt, kt_begin + p.ktiles_per_split);

    auto a_row = [&](int r) -> const float * {  // WM: r = tile row ; KM: r = k
        if (TA) return (r < p.K) ? p.A + (size_t)r * p.lda : nullptr;
        return (m0 + r < p.M) ? p.A + (size_t)(m0 + r) * p.lda : nullptr;
    };
    auto b_row = [&](int r) -> const float * {
        if (TB) return (n0 + r < p.N) ? p.B + (size_t)(n0 + r) * p.ldb : nullptr;
        return (r < p.K) ? p.B + (size_t)r * p.ldb : nullptr;
    };
    const GSrc ga = make_gsrc(p.A), gb = make_gsrc(p.B);
    // FAST operands: per-lane offsets planned once, the tile's advance goes into the scalar offset (mfma_tile.h)
    Plan<BM> pa;
    Plan<BN> pb;
    if (FAST) {
        const int ktail = (p.K % kBK) ? (p.K % kBK) : kBK;
        if (TA) plan_km<BM>(pa, p.M - m0, p.lda, ktail, tid);
        else plan_wm<BM>(pa, [&](int r) { return m0 + r < p.M; }, p.lda, ktail, tid);
        if (TB) plan_wm<BN>(pb, [&](int r) { return n0 + r < p.N; }, p.ldb, ktail, tid);
        else plan_km<BN>(pb, p.N - n0, p.ldb, ktail, tid);
    }
    // `live` = false turns every load of the tile into a zero-returning out-of-range access (FAST) instead of
    // skipping it: the loads of a k-tile are issued unconditionally, so their number is known to the compiler
    auto load_tiles = [&](Stage<BM> &sa, Stage<BN> &sb, int kt, bool live) {
        const int k0 = kt * kBK;
        if (FAST) {
            const bool tail = k0 + kBK > p.K;
            const unsigned sa_off = TA ? ((unsigned)k0 * (unsigned)p.lda + (unsigned)m0) * 4u
                                       : ((unsigned)m0 * (unsigned)p.lda + (unsigned)k0) * 4u;
            const unsigned sb_off = TB ? ((unsigned)n0 * (unsigned)p.ldb + (unsigned)k0) * 4u
                                       : ((unsigned)k0 * (unsigned)p.ldb + (unsigned)n0) * 4u;
            if (TA) load_planned_km<BM>(sa, pa, ga, live ? sa_off : kDeadTile, tail);
            else load_planned_wm<BM>(sa, pa, ga, live ? sa_off : kDeadTile, tail);
            if (TB) load_planned_wm<BN>(sb, pb, gb, live ? sb_off : kDeadTile, tail);
            else load_planned_km<BN>(sb, pb, gb, live ? sb_off : kDeadTile, tail);
            return;
        }
        auto a_live = [&](int r) -> const float * { return live ? a_row(r) : nullptr; };
        auto b_live = [&](int r) -> const float * { return live ? b_row(r) : nullptr; };
        if (TA) load_km<BM, FAST>(sa, a_live, k0, m0, p.M, p.vecA != 0, tid, ga);
        else load_wm<BM, FAST>(sa, a_live, k0, p.K, p.vecA != 0, tid, ga);
        if (TB) load_wm<BN, FAST>(sb, b_live, k0, p.K, p.vecB != 0, tid, gb);
        else load_km<BN, FAST>(sb, b_live, k0, n0, p.N, p.vecB != 0, tid, gb);
    };
    StageExp<BM> ea;
    StageExp<BN> eb;
    if (SP == kSplitF16x3) {
        load_stage_exp<BM>(ea, p.expA, m0, p.M, AWM, tid, true);     // the arrays hold |x| maxima as bit patterns
        load_stage_exp<BN>(eb, p.expB, n0, p.N, BWM, tid, true);
    }
    auto store_tiles = [&](const Stage<BM> &sa, const Stage<BN> &sb, int buf) {
        if (TA) store_km<BM, SP>(sa, As(buf), tid, ea); else store_wm<BM, SP>(sa, As(buf), tid, ea);
        if (TB) store_wm<BN, SP>(sb, Bs(buf), tid, eb); else store_km<BN, SP>(sb, Bs(buf), tid, eb);
    };
    // f16x3: the accumulators hold sum (a 2^ea[row]) (b 2^eb[col]): the exact inverse power of two goes on before anything else
    auto unscale = [&](int row, int col, float v) {
        if (SP != kSplitF16x3) return v;
        return __builtin_ldexpf(v, -(row_exponent((unsigned)p.expA[row]) + row_exponent((unsigned)p.expB[col])));
    };

    Acc acc;
    acc_zero(acc);
    // Software pipeline, prefetch distance 2: while the MFMAs of k-tile kt run from LDS buffer `cur`, tile kt+1 sits
    // in one register stage (split + written to the other LDS buffer after the MFMAs) and the global loads of tile
    // kt+2 are in flight into the other stage -- a load has a whole iteration plus an MFMA phase to land.
    Stage<BM> sa0, sa1;
    Stage<BN> sb0, sb1;
    load_tiles(sa0, sb0, kt_begin, kt_begin < kt_end);
    store_tiles(sa0, sb0, 0);
    load_tiles(sa1, sb1, kt_begin + 1, kt_begin + 1 < kt_end);
    __syncthreads();
    // one branch-free half-step (see half_step in mfma_tile.h); kt == kt_end is the phantom half-step of an odd count
    auto step = [&](auto PAR, int kt) {
        constexpr int cur = decltype(PAR)::value;
        Stage<BM> &sa_next = cur ? sa0 : sa1, &sa_far = cur ? sa1 : sa0;   // tile kt+1 / tile kt+2
        Stage<BN> &sb_next = cur ? sb0 : sb1, &sb_far = cur ? sb1 : sb0;
        auto load_far = [&]() { load_tiles(sa_far, sb_far, kt + 2, kt + 2 < kt_end); };
        auto store_next = [&]() { store_tiles(sa_next, sb_next, cur ^ 1); };
        half_step<BM, BN, SP>(load_far, store_next, As(cur), Bs(cur), wm, wn, lane, acc);
    };
    for (int kt = kt_begin; kt < kt_end; kt += 2) {
        step(std::integral_constant<int, 0>{}, kt);
        step(std::integral_constant<int, 1>{}, kt + 1);
    }

    if (p.splitk > 1) {
        float *dst = p.partial + (size_t)z * p.M * p.N;
        const bool vec = !BWM && (p.N % 2) == 0;
        acc_foreach_pair<AWM, BWM>(acc, wm, wn, lane, [&](int r, int c0, int c1, float v0, float v1) {
            const int row = m0 + r, col0 = n0 + c0, col1 = n0 + c1;
            if (row >= p.M) return;
            if (col0 < p.N) v0 = unscale(row, col0, v0);
            if (col1 < p.N) v1 = unscale(row, col1, v1);
            float *q = dst + (size_t)row * p.N;
            if (vec && col1 < p.N) *reinterpret_cast<float2 *>(q + col0) = make_float2(v0, v1);
            else { if (col0 < p.N) q[col0] = v0; if (col1 < p.N) q[col1] = v1; }
        });
        if (p.counters == nullptr) return;             // the caller launches splitk_reduce_kernel
        // ---- fused reduction.  Release: every thread's partial-sum stores are ordered before the barrier, thread 0's
        // agent-scope release-increment publishes them (the grid barrier of lstm.hip uses the same idiom across XCDs);
        // the block that draws the last ticket acquires and owns the tile.
        __shared__ int s_last;
        // every wave drains its own partial-sum stores first: the s_barrier of __syncthreads() does not wait for the other waves'
        // global stores (no s_waitcnt vmcnt(0) in front of it at workgroup scope), so thread 0's release could overtake them
        // (found live in the conv kernel's copy of this idiom, gpurun r06_c2)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(p.counters + t, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old == p.splitk - 1) ? 1 : 0;
            if (s_last) p.counters[t] = 0;             // re-armed for the next product on this stream (kernel boundary orders it)
        }
        __syncthreads();
        if (!s_last) return;
        __atomic_thread_fence(__ATOMIC_ACQUIRE);       // agent scope: the other blocks' partial sums (any XCD) are visible
        const size_t plane = (size_t)p.M * p.N;
        const int rows = min(BM, p.M - m0), cols = min(BN, p.N - n0);
        auto finish = [&](float v, int row, int col) {
            if (p.bias) v += p.bias[col];
            v = apply_epi(v, p.epilogue);
            if (p.accumulate) v += p.C[(size_t)row * p.ldc + col];
            return v;
        };
        // One block pulls the s slices of its tile: what bounds it is the number of loads it keeps in flight (a serial
        // "one output, four slices at a time" loop ran at 15 GB/s: 190 us for the 40 slices of a 120 x 151 tile, gpurun
        // r05_c1).  Here a thread owns kU outputs per pass and issues kU x 4 unconditional 16-byte loads before the first add
        // (64 KB in flight per block); the slice order of every sum stays 0, 1, 2, ...
        constexpr int kU = 4;
        if ((p.N & 3) == 0) {                          // n0 and cols are multiples of 4 then
            const int c4 = cols >> 2, W = rows * c4;
            for (int base = tid; base < W; base += kThreads * kU) {
                const float *src[kU];
                float4 v[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int idx = min(base + u * kThreads, W - 1);       // the tail re-reads the last output (never stored)
                    const int r = idx / c4, c = (idx - r * c4) * 4;
                    src[u] = p.partial + (size_t)(m0 + r) * p.N + n0 + c;
                    v[u] = *reinterpret_cast<const float4 *>(src[u]);
                }
                int zz = 1;
                for (; zz + 3 < p.splitk; zz += 4) {
                    float4 w[kU][4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int u = 0; u < kU; ++u) w[u][q] = *reinterpret_cast<const float4 *>(src[u] + (size_t)(zz + q) * plane);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int u = 0; u < kU; ++u) { v[u].x += w[u][q].x; v[u].y += w[u][q].y; v[u].z += w[u][q].z; v[u].w += w[u][q].w; }
                }
                for (; zz < p.splitk; ++zz) {
                    float4 w[kU];
#pragma unroll
                    for (int u = 0; u < kU; ++u) w[u] = *reinterpret_cast<const float4 *>(src[u] + (size_t)zz * plane);
#pragma unroll
                    for (int u = 0; u < kU; ++u) { v[u].x += w[u].x; v[u].y += w[u].y; v[u].z += w[u].z; v[u].w += w[u].w; }
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int idx = base + u * kThreads;
                    if (idx >= W) break;
                    const int r = idx / c4, c = (idx - r * c4) * 4;
                    const int row = m0 + r, col = n0 + c;
                    float4 o = v[u];
                    o.x = finish(o.x, row, col); o.y = finish(o.y, row, col + 1);
                    o.z = finish(o.z, row, col + 2); o.w = finish(o.w, row, col + 3);
                    float *q = p.C + (size_t)row * p.ldc + col;
                    if (p.vecC4) *reinterpret_cast<float4 *>(q) = o;
                    else { q[0] = o.x; q[1] = o.y; q[2] = o.z; q[3] = o.w; }
                }
            }
        } else {
            const int W = rows * cols;
            for (int base = tid; base < W; base += kThreads * kU) {
                const float *src[kU];
                float v[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int idx = min(base + u * kThreads, W - 1);
                    const int r = idx / cols, c = idx - r * cols;
                    src[u] = p.partial + (size_t)(m0 + r) * p.N + n0 + c;
                    v[u] = *src[u];
                }
                int zz = 1;
                for (; zz + 3 < p.splitk; zz += 4) {
                    float w[kU][4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int u = 0; u < kU; ++u) w[u][q] = src[u][(size_t)(zz + q) * plane];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int u = 0; u < kU; ++u) v[u] += w[u][q];
                }
                for (; zz < p.splitk; ++zz) {
#pragma unroll
                    for (int u = 0; u < kU; ++u) v[u] += src[u][(size_t)zz * plane];
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int idx = base + u * kThreads;
                    if (idx >= W) break;
                    const int r = idx / cols, c = idx - r * cols;
                    p.C[(size_t)(m0 + r) * p.ldc + n0 + c] = finish(v[u], m0 + r, n0 + c);
                }
            }
        }
        return;
    }
    acc_foreach_pair<AWM, BWM>(acc, wm, wn, lane, [&](int r, int c0, int c1, float v0, float v1) {
        const int row = m0 + r, col0 = n0 + c0, col1 = n0 + c1;
        if (row >= p.M) return;
        const bool has0 = (col0 < p.N), has1 = (col1 < p.N);
        if (has0) v0 = unscale(row, col0, v0);
        if (has1) v1 = unscale(row, col1, v1);
        if (p.bias) { if (has0) v0 += p.bias[col0]; if (has1) v1 += p.bias[col1]; }
        v0 = apply_epi(v0, p.epilogue);
        v1 = apply_epi(v1, p.epilogue);
        float *q = p.C + (size_t)row * p.ldc;
        if (p.accumulate) { if (has0) v0 += q[col0]; if (has1) v1 += q[col1]; }
        if (!BWM && p.vecC && has1) *reinterpret_cast<float2 *>(q + col0) = make_float2(v0, v1);
        else { if (has0) q[col0] = v0; if (has1) q[col1] = v1; }
    });
}

}  // namespace mh
namespace mh {
// C = epi(sum_z partial[z] + bias) (+ C)
__global__ void splitk_reduce_kernel(const float *__restrict__ partial, int splitk, int M, int N, float *__restrict__ C,
                                     int ldc, const float *__restrict__ bias, int epilogue, int accumulate)
{
    const long long total = (long long)M * N;
    const size_t plane = (size_t)M * N;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int row = idx / N, col = idx % N;
        float v = 0.f;
        for (int z = 0; z < splitk; ++z) v += partial[z * plane + idx];
        if (bias) v += bias[col];
        v = apply_epi(v, epilogue);
        float *q = C + (size_t)row * ldc + col;
        if (accumulate) v += *q;
        *q = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// f16x3 row exponents.  An operand row (a row of op(A), a column of op(B)) is scaled by 2^e with its largest |x| in
// [2^14, 2^15); e is constant along K, so it factors out of the dot product and is removed exactly in the epilogue.
// Two passes over the operand: |x| maxima as float bit patterns (monotone as unsigned), then bits -> exponent in place.
// ---------------------------------------------------------------------------------------------------------------
__global__ void row_absmax_kernel(const float *__restrict__ X, long long rows, int cols, long long ld,
                                  unsigned *__restrict__ out)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *p = X + row * ld;
    unsigned m = 0;
    for (int c = lane; c < cols; c += 64) m = max(m, __float_as_uint(p[c]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if (lane == 0) out[row] = m;
}
// X stored [rows = k][cols = operand rows]: out[cols] (zero-initialised) collects the column maxima
__global__ void col_absmax_kernel(const float *__restrict__ X, long long rows, int cols, long long ld, int rows_per_block,
                                  unsigned *__restrict__ out)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const long long r0 = (long long)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    unsigned m = 0;
    for (long long r = r0; r < r1; ++r) m = max(m, __float_as_uint(X[r * ld + c]) & 0x7fffffffu);
    atomicMax(out + c, m);
}
// Vector forms (16-B aligned operand, cols % 4 == 0, ld % 4 == 0): float4 loads, several independent loads per lane.
__device__ __forceinline__ unsigned absmax4(const float4 v)
{
    return max(max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu),
               max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu));
}
// Short rows (pixel channel vectors, small GEMM operands): LPR lanes share a row, a wave covers 64 / LPR rows.
// to_exp: write the exponent instead of the bit pattern (saves the bits -> exponent launch).
template <int LPR>
__global__ __launch_bounds__(256) void row_absmax_vec_kernel(const float *__restrict__ X, long long rows, int cols, long long ld,
                                                             unsigned *__restrict__ out, int to_exp)
{
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPR, rw = lane / LPR;
    const int n4 = cols >> 2;
    for (long long row0 = ((long long)blockIdx.x * 4 + wave) * RPW; row0 < rows; row0 += (long long)gridDim.x * 4 * RPW) {
        const long long row = row0 + rw;
        unsigned m0 = 0, m1 = 0, m2 = 0, m3 = 0;
        if (row < rows) {
            const float4 *p = reinterpret_cast<const float4 *>(X + row * ld);
            int c = sub;
            for (; c + 3 * LPR < n4; c += 4 * LPR) {
                const float4 v0 = p[c], v1 = p[c + LPR], v2 = p[c + 2 * LPR], v3 = p[c + 3 * LPR];
                m0 = max(m0, absmax4(v0)); m1 = max(m1, absmax4(v1)); m2 = max(m2, absmax4(v2)); m3 = max(m3, absmax4(v3));
            }
            for (; c < n4; c += LPR) m0 = max(m0, absmax4(p[c]));
        }
        unsigned m = max(max(m0, m1), max(m2, m3));
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        if (sub == 0 && row < rows) out[row] = to_exp ? (unsigned)row_exponent(m) : m;
    }
}
// Long rows (fc6 operands: 25088 floats): one workgroup per row, 8 float4 in flight per thread.
__global__ __launch_bounds__(256) void row_absmax_long_kernel(const float *__restrict__ X, long long rows, int cols, long long ld,
                                                              unsigned *__restrict__ out, int to_exp)
{
    __shared__ unsigned red[4];
    const int tid = threadIdx.x, n4 = cols >> 2;
    for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
        const float4 *p = reinterpret_cast<const float4 *>(X + row * ld);
        unsigned m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int c = tid;
        for (; c + 7 * 256 < n4; c += 8 * 256) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p[c + 256 * k];
#pragma unroll
            for (int k = 0; k < 8; ++k) m[k] = max(m[k], absmax4(v[k]));
        }
        for (; c < n4; c += 256) m[0] = max(m[0], absmax4(p[c]));
        unsigned mm = max(max(max(m[0], m[1]), max(m[2], m[3])), max(max(m[4], m[5]), max(m[6], m[7])));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mm = max(mm, (unsigned)__shfl_xor((int)mm, o));
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = mm;
        __syncthreads();
        if (tid == 0) {
            const unsigned r = max(max(red[0], red[1]), max(red[2], red[3]));
            out[row] = to_exp ? (unsigned)row_exponent(r) : r;
        }
    }
}
// Columns: a workgroup owns a strip of 256 columns (64 float4 lanes) x `rows_per_block` rows, its four waves take rows
// r, r+4, ... with four loads in flight each; the strip's maxima are combined in LDS and ONE atomicMax per column leaves
// the block.
__global__ __launch_bounds__(256) void col_absmax_vec_kernel(const float *__restrict__ X, long long rows, int cols, long long ld,
                                                             int rows_per_block, unsigned *__restrict__ out)
{
    __shared__ unsigned red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 4;
    const long long r0 = (long long)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    unsigned m[4] = {0, 0, 0, 0};
    if (c < cols) {
        auto fold = [&](const float4 v) {
            m[0] = max(m[0], __float_as_uint(v.x) & 0x7fffffffu); m[1] = max(m[1], __float_as_uint(v.y) & 0x7fffffffu);
            m[2] = max(m[2], __float_as_uint(v.z) & 0x7fffffffu); m[3] = max(m[3], __float_as_uint(v.w) & 0x7fffffffu);
        };
        long long r = r0 + wave;
        for (; r + 12 < r1; r += 16) {
            const float4 v0 = *reinterpret_cast<const float4 *>(X + r * ld + c), v1 = *reinterpret_cast<const float4 *>(X + (r + 4) * ld + c),
                         v2 = *reinterpret_cast<const float4 *>(X + (r + 8) * ld + c), v3 = *reinterpret_cast<const float4 *>(X + (r + 12) * ld + c);
            fold(v0); fold(v1); fold(v2); fold(v3);
        }
        for (; r < r1; r += 4) fold(*reinterpret_cast<const float4 *>(X + r * ld + c));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[wave][4 * lane + k] = m[k];
    __syncthreads();
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col < cols) {
        const unsigned mm = max(max(red[0][threadIdx.x], red[1][threadIdx.x]), max(red[2][threadIdx.x], red[3][threadIdx.x]));
        atomicMax(out + col, mm);
    }
}
// ---- both operands of a GEMM in ONE launch --------------------------------------------------------------------------
// The step is launch-bound as much as bandwidth-bound here (a GEMM call used to issue 4-6 tiny launches for its two
// exponent arrays): blocks [0, a.nblocks) scan operand A, the rest operand B, each in the mode its storage calls for.
struct AbsmaxPart {
    const float *X;
    long long rows, kext, ld;     // rows = operand rows (the non-K index), kext = extent along K
    unsigned *bits;
    int mode;                     // 0: K-contiguous, long rows (workgroup per row); 1: K-contiguous, lpr lanes per row;
                                  // 2: k-major, 256-column strips + atomicMax; 3 / 4: unaligned fallbacks of 1 / 2
    int lpr, rows_per_block, col_blocks, nblocks;
};
__device__ __forceinline__ void absmax_part(const AbsmaxPart &p, int bid, unsigned (*red)[256])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (p.mode == 0) {
        const int n4 = (int)(p.kext >> 2);
        for (long long row = bid; row < p.rows; row += p.nblocks) {
            const float4 *q = reinterpret_cast<const float4 *>(p.X + row * p.ld);
            unsigned m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int c = tid;
            for (; c + 7 * 256 < n4; c += 8 * 256) {
                float4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = q[c + 256 * k];
#pragma unroll
                for (int k = 0; k < 8; ++k) m[k] = max(m[k], absmax4(v[k]));
            }
            for (; c < n4; c += 256) m[0] = max(m[0], absmax4(q[c]));
            unsigned mm = max(max(max(m[0], m[1]), max(m[2], m[3])), max(max(m[4], m[5]), max(m[6], m[7])));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mm = max(mm, (unsigned)__shfl_xor((int)mm, o));
            __syncthreads();
            if (lane == 0) red[0][wave] = mm;
            __syncthreads();
            if (tid == 0) p.bits[row] = max(max(red[0][0], red[0][1]), max(red[0][2], red[0][3]));
        }
    } else if (p.mode == 1 || p.mode == 3) {
        const int lpr = p.lpr, rpw = 64 / lpr, sub = lane & (lpr - 1), rw = lane / lpr;
        for (long long row0 = ((long long)bid * 4 + wave) * rpw; row0 < p.rows; row0 += (long long)p.nblocks * 4 * rpw) {
            const long long row = row0 + rw;
            unsigned m0 = 0, m1 = 0;
            if (row < p.rows) {
                if (p.mode == 1) {
                    const float4 *q = reinterpret_cast<const float4 *>(p.X + row * p.ld);
                    const int n4 = (int)(p.kext >> 2);
                    int c = sub;
                    for (; c + lpr < n4; c += 2 * lpr) {
                        const float4 v0 = q[c], v1 = q[c + lpr];
                        m0 = max(m0, absmax4(v0)); m1 = max(m1, absmax4(v1));
                    }
                    for (; c < n4; c += lpr) m0 = max(m0, absmax4(q[c]));
                } else {
                    const float *q = p.X + row * p.ld;
                    for (long long c = sub; c < p.kext; c += lpr) m0 = max(m0, __float_as_uint(q[c]) & 0x7fffffffu);
                }
            }
            unsigned m = max(m0, m1);
            for (int o = lpr >> 1; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
            if (sub == 0 && row < p.rows) p.bits[row] = m;
        }
    } else if (p.mode == 2) {
        const int cb = bid % p.col_blocks, chunk = bid / p.col_blocks;
        const long long c = ((long long)cb * 64 + lane) * 4;
        const long long r0 = (long long)chunk * p.rows_per_block, r1 = min(p.kext, r0 + p.rows_per_block);
        unsigned m[4] = {0, 0, 0, 0};
        if (c < p.rows) {
            auto fold = [&](const float4 v) {
                m[0] = max(m[0], __float_as_uint(v.x) & 0x7fffffffu); m[1] = max(m[1], __float_as_uint(v.y) & 0x7fffffffu);
                m[2] = max(m[2], __float_as_uint(v.z) & 0x7fffffffu); m[3] = max(m[3], __float_as_uint(v.w) & 0x7fffffffu);
            };
            long long r = r0 + wave;
            for (; r + 12 < r1; r += 16) {
                const float4 v0 = *reinterpret_cast<const float4 *>(p.X + r * p.ld + c), v1 = *reinterpret_cast<const float4 *>(p.X + (r + 4) * p.ld + c),
                             v2 = *reinterpret_cast<const float4 *>(p.X + (r + 8) * p.ld + c), v3 = *reinterpret_cast<const float4 *>(p.X + (r + 12) * p.ld + c);
                fold(v0); fold(v1); fold(v2); fold(v3);
            }
            for (; r < r1; r += 4) fold(*reinterpret_cast<const float4 *>(p.X + r * p.ld + c));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) red[wave][4 * lane + k] = m[k];
        __syncthreads();
        const long long col = (long long)cb * 256 + tid;
        if (col < p.rows) atomicMax(p.bits + col, max(max(red[0][tid], red[1][tid]), max(red[2][tid], red[3][tid])));
    } else {   // 4: k-major, unaligned: one column per thread
        const int cb = bid % p.col_blocks, chunk = bid / p.col_blocks;
        const long long col = (long long)cb * 256 + tid;
        if (col < p.rows) {
            const long long r0 = (long long)chunk * p.rows_per_block, r1 = min(p.kext, r0 + p.rows_per_block);
            unsigned m = 0;
            for (long long r = r0; r < r1; ++r) m = max(m, __float_as_uint(p.X[r * p.ld + col]) & 0x7fffffffu);
            atomicMax(p.bits + col, m);
        }
    }
}
__global__ __launch_bounds__(256) void absmax_dual_kernel(const AbsmaxPart a, const AbsmaxPart b)
{
    __shared__ unsigned red[4][256];
    if ((int)blockIdx.x < a.nblocks) absmax_part(a, (int)blockIdx.x, red);
    else absmax_part(b, (int)blockIdx.x - a.nblocks, red);
}
static AbsmaxPart make_absmax_part(const float *X, bool k_contiguous, long long rows, long long kext, long long ld, int *bits)
{
    AbsmaxPart p;
    p.X = X; p.rows = rows; p.kext = kext; p.ld = ld; p.bits = reinterpret_cast<unsigned *>(bits);
    p.lpr = 64; p.rows_per_block = 1; p.col_blocks = 1;
    const bool vec = (reinterpret_cast<uintptr_t>(X) & 15) == 0 && ld % 4 == 0;
    if (k_contiguous) {
        if (vec && kext % 4 == 0 && kext >= 4096) {
            p.mode = 0;
            p.nblocks = (int)std::min<long long>(rows, 256 * 32);
        } else {
            p.mode = (vec && kext % 4 == 0) ? 1 : 3;
            const long long per_lane = (p.mode == 1) ? (kext >> 2) : kext;
            p.lpr = per_lane >= 64 ? 64 : per_lane >= 32 ? 32 : per_lane >= 16 ? 16 : per_lane >= 8 ? 8 : 4;
            p.nblocks = (int)std::min<long long>(ceil_div(ceil_div(rows, (long long)(64 / p.lpr)), 4LL), 256 * 16);
        }
    } else {
        p.mode = (vec && rows % 4 == 0) ? 2 : 4;
        p.col_blocks = (int)ceil_div(rows, 256LL);
        p.rows_per_block = (int)std::max<long long>(64, ceil_div(kext, std::max<long long>(1, 1024 / p.col_blocks)));
        p.nblocks = p.col_blocks * (int)ceil_div(kext, (long long)p.rows_per_block);
    }
    p.nblocks = std::max(p.nblocks, 1);
    return p;
}
int launch_operand_absmax(const float *A, bool a_kcontig, long long a_rows, long long a_kext, long long lda, int *bitsA,
                          const float *B, bool b_kcontig, long long b_rows, long long b_kext, long long ldb, int *bitsB,
                          hipStream_t st)
{
    const AbsmaxPart a = make_absmax_part(A, a_kcontig, a_rows, a_kext, lda, bitsA);
    const AbsmaxPart b = make_absmax_part(B, b_kcontig, b_rows, b_kext, ldb, bitsB);
    if (!a_kcontig || !b_kcontig) {       // the column passes fold into zero-initialised words: one memset over both arrays
        char *lo = reinterpret_cast<char *>(std::min(bitsA, bitsB)), *hi = reinterpret_cast<char *>(bitsA > bitsB ? bitsA + a_rows : bitsB + b_rows);
        hipError_t e = hipMemsetAsync(lo, 0, (size_t)(hi - lo), st);
        if (e != hipSuccess) { set_last_error("hipMemsetAsync(operand absmax)", e); return (int)e; }
    }
    hipLaunchKernelGGL(absmax_dual_kernel, dim3((unsigned)(a.nblocks + b.nblocks)), dim3(256), 0, st, a, b);
    return check_launch("absmax_dual_kernel");
}

__global__ void bits_to_exp_kernel(unsigned *__restrict__ io, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) io[i] = (unsigned)row_exponent(io[i]);
}

// exps[n_rows] for an operand whose `n_rows` rows (the non-K index) have `kext` elements along K; k_contiguous = the
// storage is [n_rows][kext] (ld), otherwise [kext][n_rows] (ld)
int launch_row_exponents(const float *X, bool k_contiguous, long long n_rows, long long kext, long long ld, int *exps,
                         hipStream_t st, bool bits_only)
{
    unsigned *bits = reinterpret_cast<unsigned *>(exps);
    const bool vec = (reinterpret_cast<uintptr_t>(X) & 15) == 0 && ld % 4 == 0;
    if (k_contiguous) {
        const int to_exp = bits_only ? 0 : 1;       // one owner per row: the exponent is written directly
        if (vec && kext % 4 == 0 && kext >= 4096) {
            hipLaunchKernelGGL(row_absmax_long_kernel, dim3((unsigned)std::min<long long>(n_rows, 256 * 32)), dim3(256), 0, st, X,
                               n_rows, (int)kext, ld, bits, to_exp);
            return check_launch("row_absmax_long_kernel");
        }
        if (vec && kext % 4 == 0) {
            const int n4 = (int)(kext >> 2);
            const int lpr = n4 >= 64 ? 64 : n4 >= 32 ? 32 : n4 >= 16 ? 16 : n4 >= 8 ? 8 : 4;
            const long long waves = ceil_div(n_rows, (long long)(64 / lpr));
            const dim3 grid((unsigned)std::min<long long>(ceil_div(waves, 4LL), 256 * 16));
            if (lpr == 64) hipLaunchKernelGGL(row_absmax_vec_kernel<64>, grid, dim3(256), 0, st, X, n_rows, (int)kext, ld, bits, to_exp);
            else if (lpr == 32) hipLaunchKernelGGL(row_absmax_vec_kernel<32>, grid, dim3(256), 0, st, X, n_rows, (int)kext, ld, bits, to_exp);
            else if (lpr == 16) hipLaunchKernelGGL(row_absmax_vec_kernel<16>, grid, dim3(256), 0, st, X, n_rows, (int)kext, ld, bits, to_exp);
            else if (lpr == 8) hipLaunchKernelGGL(row_absmax_vec_kernel<8>, grid, dim3(256), 0, st, X, n_rows, (int)kext, ld, bits, to_exp);
            else hipLaunchKernelGGL(row_absmax_vec_kernel<4>, grid, dim3(256), 0, st, X, n_rows, (int)kext, ld, bits, to_exp);
            return check_launch("row_absmax_vec_kernel");
        }
        hipLaunchKernelGGL(row_absmax_kernel, dim3((unsigned)ceil_div(n_rows, 4LL)), dim3(256), 0, st, X, n_rows, (int)kext, ld, bits);
    } else {
        hipError_t e = hipMemsetAsync(bits, 0, (size_t)n_rows * sizeof(unsigned), st);
        if (e != hipSuccess) { set_last_error("hipMemsetAsync(row exponents)", e); return (int)e; }
        if (vec && n_rows % 4 == 0) {
            // 256-column strips; the K extent is cut so that ~1024 blocks stream (>= 64 rows each): one atomic per column and chunk
            const long long col_blocks = ceil_div(n_rows, 256LL);
            const int rows_per_block = (int)std::max<long long>(64, ceil_div(kext, std::max<long long>(1, 1024 / col_blocks)));
            hipLaunchKernelGGL(col_absmax_vec_kernel, dim3((unsigned)col_blocks, (unsigned)ceil_div(kext, (long long)rows_per_block)),
                               dim3(256), 0, st, X, kext, (int)n_rows, ld, rows_per_block, bits);
        } else {
            const int rows_per_block = (int)std::max<long long>(64, ceil_div(kext, 2048LL));     // up to 2048 row chunks x cols/256 blocks
            hipLaunchKernelGGL(col_absmax_kernel, dim3((unsigned)ceil_div(n_rows, 256LL), (unsigned)ceil_div(kext, (long long)rows_per_block)),
                               dim3(256), 0, st, X, kext, (int)n_rows, ld, rows_per_block, bits);
        }
    }
    int rc = check_launch("absmax_kernel");
    if (rc || bits_only) return rc;        // bits_only: the caller combines maxima before taking exponents (conv)
    hipLaunchKernelGGL(bits_to_exp_kernel, dim3((unsigned)ceil_div(n_rows, 256LL)), dim3(256), 0, st, bits, n_rows);
    return check_launch("bits_to_exp_kernel");
}

// Work-distribution model shared by GEMM and conv: `tiles` output tiles of `ktiles` k-steps each are cut into
// S k-slices.  A CU holds k = resident_slots() / 256 blocks of the tile engine (LDS / VGPR budget) and runs them at
// 1/k of its speed each; a CU left with fewer blocks only reaches part of its throughput (nothing hides its barrier and
// LDS latency: ~0.6 with a single block, measured on the bf16x6 engine).  Equal blocks finish in lockstep, so with
// t1 = one tile on one fully occupied CU the makespan in units of t1/S is
//     floor(blocks / slots) * k  +  j / f(j),   j = ceil(remainder / 256) blocks per CU in the last round, f(k) = 1
// (PMC: the 384-tile fc6 forward at S = 2 ran 1.5 rounds with an average of 1.2 waves per SIMD).  The partial-sum round
// trip costs S*M*N*8 bytes of HBM traffic.
static thread_local int g_slots_override = 0;
void set_resident_slots_override(int slots) { g_slots_override = slots; }

int resident_slots()
{
    if (g_slots_override > 0) return g_slots_override;      // a shape with another residency is being planned (pl_conv.hip: ring shapes)
    static const int slots = [] {
        const char *e = getenv("MH_SLOTS");
        const int v = e ? atoi(e) : 0;
        return (v >= 256 && v % 256 == 0) ? v : 512;
    }();
    return slots;
}

double makespan_units(long long blocks)
{
    const int slots = resident_slots(), k = slots / 256;
    const long long rem = blocks % slots;
    double units = (double)(blocks / slots) * k;
    if (rem) {
        const int j = (int)((rem + 255) / 256);
        const double f = (k > 1) ? 0.6 + 0.4 * (double)(j - 1) / (double)(k - 1) : 1.0;
        units += j / f;
    }
    return units;
}

int choose_splitk_tiles(long long tiles, int ktiles, double out_elems, double flops)
{
    if (tiles >= 4096 || ktiles < 16) return 1;
    static const int cand[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64, 80, 96, 128, 160, 192, 256};
    const double t1 = flops / (double)tiles / (170e12 / 256.0);   // seconds per tile on a fully occupied CU
    double best = 1e30;
    int best_s = 1;
    for (int s : cand) {
        if (s > 1 && ktiles / s < 6) break;
        const double t_partial = (s > 1) ? (out_elems * 8.0 * s) / 4.0e12 + 4e-6 : 0.0;
        const double cost = makespan_units(tiles * s) * t1 / s + t_partial;
        if (cost < best * 0.97) { best = cost; best_s = s; }
    }
    return best_s;
}

// ---- the schedule of the small-product engine (round 5).  choose_splitk_tiles prices a tile by its flops, which is right for
// chip-filling products and wrong here: a 120 x 151 x 4096 product is two tiles whose 256 k-steps each cost what a k-step
// of the in-loop engine costs whatever the tile holds -- ~0.75 us alone on a CU, ~0.9 us next to a second block (measured:
// 194 us for 256 k-steps of one block, gpurun r05_c1) -- so the products of a step are latency chains that only more K
// slices shorten, until the reduction of the slices costs more than the loop saves.  The reduction is either FUSED (the last
// block of a tile adds its s slices: one block sustains ~50 GB/s, fine up to ~1 MB per tile) or a second launch
// (splitk_reduce_kernel: chip-wide, ~2 TB/s on these sizes, + ~10 us for the launch and its gap).
struct SmallPlan {
    int splitk;
    bool fused;
};
constexpr double kFusedBytesMax = 1.5e6;       // slices x tile bytes one block is asked to add up
static SmallPlan plan_small(int M, int N, int K, int want_splitk, bool counters_ok)
{
    const int bm = (N <= 64) ? 256 : 128, bn = (N <= 64) ? 64 : 128;
    const long long tiles = (long long)ceil_div(M, bm) * ceil_div(N, bn);
    const int ktiles = ceil_div(K, kBK);
    const double tile_bytes = 4.0 * std::min(M, bm) * std::min(N, bn);
    auto fused_ok = [&](int sk) { return counters_ok && sk > 1 && sk * tile_bytes <= kFusedBytesMax && tiles <= kGemmCounters; };
    if (want_splitk > 0) {
        const int sk = std::max(1, std::min(std::min(want_splitk, ktiles), 64));
        return {sk, fused_ok(sk)};
    }
    if (tiles >= 4096 || ktiles < 4) return {1, false};
    static const int cand[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64};
    SmallPlan best = {1, false};
    double best_cost = 1e30;
    for (int sk : cand) {
        if (sk > 1 && ktiles / sk < 2) break;
        const long long blocks = tiles * sk;
        const double kt = (double)ceil_div(ktiles, sk);
        const double rounds = (double)ceil_div(blocks, 512LL);
        const double loop = rounds * kt * (blocks <= 256 ? 0.75e-6 : 0.9e-6);
        const double fixed = 4e-6;
        for (int f = 0; f < 2; ++f) {
            if (sk == 1 && f) continue;
            if (f && !fused_ok(sk)) continue;
            const double red = (sk == 1) ? 0.0 : f ? (sk * tile_bytes / 50e9 + 1e-6) : ((double)sk * M * N * 8.0 / 2e12 + 10e-6);
            const double cost = loop + fixed + red;
            if (cost < best_cost * 0.97) { best_cost = cost; best = {sk, f != 0}; }
        }
    }
    return best;
}

int launch_splitk_reduce(const float *partial, int splitk, long long M, int N, float *C, int ldc, const float *bias,
                         int epilogue, int accumulate, hipStream_t st)
{
    const long long total = M * N;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, partial, splitk, (int)M, N, C, ldc, bias,
                       epilogue, accumulate);
    return check_launch("splitk_reduce_kernel");
}

}  // namespace mh

using namespace mh;

extern "C" {

int mh_mfma_split(void) { return MH_MFMA_SPLIT; }
int mh_split_f16(void) { return MH_SPLIT_F16; }
int mh_split_rne(void) { return 0; }      /* the bf16 round-to-nearest split variant is gone (round 3); kept for ABI stability */

int mh_gemm_auto_splitk_v2(int M, int N, int K) { return (M > 0 && N > 0 && K > 0) ? plan_small(M, N, K, 0, true).splitk : 1; }

size_t mh_gemm_ws_bytes_v2(int M, int N, int K, int splitk)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (splitk <= 0) splitk = plan_small(M, N, K, 0, true).splitk;
    const size_t exps = align_up((size_t)M * sizeof(int), 256) + align_up((size_t)N * sizeof(int), 256);
    if (splitk <= 1) return exps;
    return exps + align_up((size_t)splitk * M * N * sizeof(float), 256);
}

}  // extern "C"

namespace mh {

// The small-product engine runs bf16x6 only.  Round 2's f16x3 evaluation of the same kernel family (a pass over both operands
// for their row maxima in front of every product, then three f16 MFMAs) was kept as an A/B arm for one measurement and removed:
// same box, cfg2 step, 330.7 img/s against 348.3 / 351.3 with bf16x6 (profiles/r05_bench_c1_*.json).
static int small_gemm_split() { return kSplitBf16x6; }

// The in-loop-split product: fp32 operands are read once and split by the threads that stage them.
//   sp           kSplitBf16x6
//   counters     split-K arrival counters (>= one int per output tile, ZERO on entry, left zero): the last block of a tile
//                reduces its slices inside the GEMM launch; NULL (or too few): a separate splitk_reduce_kernel launch
static int gemm_inloop_impl(int sp, int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
                            float *C, int ldc, const float *bias, int epilogue, int accumulate, int splitk, void *workspace,
                            size_t ws_bytes, int *counters, int n_counters, void *stream)
{
    MH_REQUIRE(M >= 0 && N >= 0 && K >= 0);
    if (M == 0 || N == 0) return MH_OK;
    MH_REQUIRE(A && B && C && K > 0);
    MH_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N);
    MH_REQUIRE(epilogue >= MH_EPI_NONE && epilogue <= MH_EPI_RELU6);
    const SmallPlan pln = plan_small(M, N, K, splitk, counters != nullptr && n_counters > 0);
    splitk = pln.splitk;
    const int total_kt = ceil_div(K, kBK);
    hipStream_t st = as_stream(stream);
    GemmArgs p;
    p.expA = p.expB = nullptr;
    MH_REQUIRE(sp == kSplitBf16x6);       // (the f16x3 form of this kernel needs row exponents: conv.hip still uses that half of mfma_tile.h)
    if (splitk > 1 && (workspace == nullptr || ws_bytes < (size_t)splitk * M * N * sizeof(float))) splitk = 1;
    p.M = M; p.N = N; p.K = K;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
    p.bias = bias; p.epilogue = epilogue; p.accumulate = accumulate;
    p.ktiles_per_split = ceil_div(total_kt, splitk);
    splitk = ceil_div(total_kt, p.ktiles_per_split);  // drop empty tail splits
    p.splitk = splitk;
    p.partial = reinterpret_cast<float *>(workspace);
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    p.vecA = al16(A) && (lda % 4 == 0);
    p.vecB = al16(B) && (ldb % 4 == 0);
    p.vecC = ((reinterpret_cast<uintptr_t>(C) & 7) == 0) && (ldc % 2 == 0);
    p.vecC4 = al16(C) && (ldc % 4 == 0);
    const bool narrow = (N <= 64);
    p.tiles_m = ceil_div(M, narrow ? 256 : 128);
    p.tiles_n = ceil_div(N, narrow ? 64 : 128);
    const long long ntiles = (long long)p.tiles_m * p.tiles_n;
    MH_REQUIRE(ntiles < (1LL << 31) && splitk <= 65535);
    p.counters = (splitk > 1 && pln.fused && ntiles <= (long long)n_counters) ? counters : nullptr;
    {   // patch-major tile order: ~64 tiles per patch (what one XCD runs at a time); the operand whose panel is the
        // more expensive to re-fetch gets the longer patch side.  MH_GEMM_PATCH=rows restores the round-1 order (A/B runs).
        static const bool rows_order = [] { const char *e = getenv("MH_GEMM_PATCH"); return e && e[0] == 'r'; }();
        int ph = 8, pw = 8;
        if (rows_order) { ph = 1; pw = p.tiles_n; }
        else {
            if (p.tiles_m < 8) { ph = p.tiles_m; pw = std::min(p.tiles_n, 64 / ph); }
            else if (p.tiles_n < 8) { pw = p.tiles_n; ph = std::min(p.tiles_m, 64 / pw); }
        }
        p.patch_h = std::max(ph, 1);
        p.patch_w = std::max(pw, 1);
    }
    dim3 grid((unsigned)ntiles, (unsigned)splitk);
    // FAST: both operands 16-B aligned and their contiguous extents multiples of 4 (see load4_guarded)
    // ... and each operand spans < 1 GiB (32-bit buffer offsets inside a 1 GiB descriptor, see GSrc)
    const unsigned long long spanA = (unsigned long long)(transA ? K : M) * lda * sizeof(float);
    const unsigned long long spanB = (unsigned long long)(transB ? N : K) * ldb * sizeof(float);
    const bool fast = p.vecA && p.vecB && ((transA ? M : K) % 4 == 0) && ((transB ? K : N) % 4 == 0) &&
                      spanA < (1ull << 30) && spanB < (1ull << 30);
#define MH_LAUNCH_GEMM3(TA_, TB_, F_, SP_)                                                                      \
    do {                                                                                                        \
        if (narrow) launch_tile_kernel<gemm_kernel<TA_, TB_, 256, 64, F_, SP_>>(                                \
                grid, tile_lds_bytes<256, 64, !TA_, TB_>(), st, p);                                             \
        else launch_tile_kernel<gemm_kernel<TA_, TB_, 128, 128, F_, SP_>>(                                      \
                grid, tile_lds_bytes<128, 128, !TA_, TB_>(), st, p);                                            \
    } while (0)
#define MH_LAUNCH_GEMM2(TA_, TB_, F_) MH_LAUNCH_GEMM3(TA_, TB_, F_, kSplitBf16x6)
#define MH_LAUNCH_GEMM(TA_, TB_)                          \
    do {                                                  \
        if (fast) MH_LAUNCH_GEMM2(TA_, TB_, true);        \
        else MH_LAUNCH_GEMM2(TA_, TB_, false);            \
    } while (0)
    if (!transA && !transB) MH_LAUNCH_GEMM(false, false);
    else if (!transA && transB) MH_LAUNCH_GEMM(false, true);
    else if (transA && !transB) MH_LAUNCH_GEMM(true, false);
    else MH_LAUNCH_GEMM(true, true);
#undef MH_LAUNCH_GEMM
#undef MH_LAUNCH_GEMM2
#undef MH_LAUNCH_GEMM3
    int rc = check_launch("gemm_kernel");
    if (rc) return rc;
    if (splitk > 1 && p.counters == nullptr)
        rc = launch_splitk_reduce(p.partial, splitk, M, N, C, ldc, bias, epilogue, accumulate, st);
    return rc;
}

// for the C++ callers inside the library (lstm.hip through gemm_f32_ctr)
int gemm_small(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
               const float *bias, int epilogue, int accumulate, void *workspace, size_t ws_bytes, int *counters, int n_counters,
               void *stream)
{
    return gemm_inloop_impl(small_gemm_split(), transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, epilogue, accumulate, 0,
                            workspace, ws_bytes, counters, n_counters, stream);
}

}  // namespace mh

extern "C" {

// the round-2 entry point (fp32 operands split inside the K loop), now the small-product engine without arrival counters:
// the split-K reduction is a separate launch
int mh_gemm_f32_v2(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
                float *C, int ldc, const float *bias, int epilogue, int accumulate, int splitk, void *workspace,
                size_t ws_bytes, void *stream)
{
    return gemm_inloop_impl(small_gemm_split(), transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, epilogue, accumulate,
                            splitk, workspace, ws_bytes, nullptr, 0, stream);
}

int mh_gemm_small_max_counters(void) { return kGemmCounters; }
/* the plan of an auto-split small product: K slices | (1 << 16 when their reduction is fused into the launch) */
int mh_debug_small_plan(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0) return 1;
    const SmallPlan pl = plan_small(M, N, K, 0, true);
    return pl.splitk | (pl.fused ? (1 << 16) : 0);
}      /* choose_splitk_tiles never splits a product of >= 4096 tiles */

int mh_gemm_small_f32(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
                      float *C, int ldc, const float *bias, int epilogue, int accumulate, int splitk, void *workspace,
                      size_t ws_bytes, int *counters, int n_counters, void *stream)
{
    MH_REQUIRE(n_counters >= 0 && (counters != nullptr || n_counters == 0));
    return gemm_inloop_impl(small_gemm_split(), transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, epilogue, accumulate,
                            splitk, workspace, ws_bytes, counters, n_counters, stream);
}

}  // extern "C"
