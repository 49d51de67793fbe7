// pl_gemm.hip -- fp32-accurate GEMM on the plane engine (pl_tile.h): C[M,N] = epi(A[M,K] . B[N,K]^T + bias), both operands
// read as pre-split f16 plane images (no VALU in the K loop), plus the operand-preparation pass that writes such images.
//
//   mh_make_planes      fp32 matrix (either storage orientation) -> plane image + row maxima       (one pass per operand use)
//   mh_gemm_planes      the product of two plane images
//   mh_gemm_f32         the round-1/2 entry point, now = make_planes(A), make_planes(B) into the workspace + mh_gemm_planes
// Every nn.Linear-shaped contraction of the path goes through here (fc6/fc7 x3 with cached weight images, score/bbox heads,
// LSTM input projections, post_lstm, rel_compress and their dgrad/wgrad): reference lib/rel_model.py:366-373,403-414,
// lib/object_detector.py:129-138 (cuBLAS there).
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "mfma_tile.h"     // launch_row_exponents / launch_splitk_reduce / makespan_units (gemm.hip)
#include "pl_ring.h"
#include "pl_tile.h"

namespace mh {
namespace pl {

// ------------------------------------------------------------------------------------------------- operand preparation
struct PlaneJob {
    const float *X;
    long long ld;
    long long rows, K;          // operand rows (non-K index) and K extent
    char *cells;                // [Kc][rows][64]
    const unsigned *maxbits;    // |x| maxima (fp32 bits), one per `exp_div` consecutive rows
    int exp_div;
    int kmajor;                 // 0: X is [rows][K] (K contiguous); 1: X is [K][rows]
    int vec;                    // 16-byte aligned base, ld % 4 == 0
    int tiles_k, nblocks;       // 64 x 64 tiles along K; blocks of this job
};

// one 64 (operand rows) x 64 (k) tile: coalesced fp32 reads in the operand's own orientation, split, transpose through
// LDS into cell order, then four contiguous 4 KB runs (one per k-chunk) leave the block as 16-byte stores
__device__ __forceinline__ void planes_tile(const PlaneJob &p, int bid, char *lds)
{
    const int tid = threadIdx.x;
    const long long tk = bid % p.tiles_k, tr = bid / p.tiles_k;
    const long long r0 = tr * 64, k0 = tk * 64;
    const long long Kc = (p.K + kBK - 1) / kBK;
    auto rexp = [&](long long r) { return row_exponent(p.maxbits[r / p.exp_div]); };
    if (!p.kmajor) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 256 * j, row = f >> 4, q = f & 15;
            const long long r = r0 + row, k = k0 + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            int e = 0;
            if (r < p.rows) {
                e = rexp(r);
                const float *src = p.X + r * p.ld + k;
                if (p.vec && k + 3 < p.K) v = *reinterpret_cast<const float4 *>(src);
                else {
                    if (k + 0 < p.K) v.x = src[0];
                    if (k + 1 < p.K) v.y = src[1];
                    if (k + 2 < p.K) v.z = src[2];
                    if (k + 3 < p.K) v.w = src[3];
                }
            }
            unsigned a1, a2, b1, b2;
            split2(v.x, v.y, e, a1, a2);
            split2(v.z, v.w, e, b1, b2);
            char *cell = lds + ((q >> 2) * 64 + row) * kCell + 8 * (q & 3);
            *reinterpret_cast<u32x2 *>(cell) = (u32x2){a1, b1};
            *reinterpret_cast<u32x2 *>(cell + 32) = (u32x2){a2, b2};
        }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = tid + 256 * j, quad = t & 15, kp = t >> 4;          // 4 operand rows x one k pair
            const long long r = r0 + 4 * quad, k = k0 + 2 * kp;
            float ev[4] = {0.f, 0.f, 0.f, 0.f}, od[4] = {0.f, 0.f, 0.f, 0.f};
            auto fetch = [&](long long kk, float *dst) {
                if (kk >= p.K) return;
                const float *src = p.X + kk * p.ld + r;
                if (p.vec && r + 3 < p.rows) {
                    const float4 v = *reinterpret_cast<const float4 *>(src);
                    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (r + i < p.rows) dst[i] = src[i];
                }
            };
            fetch(k, ev);
            fetch(k + 1, od);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = (r + i < p.rows) ? rexp(r + i) : 0;
                unsigned p1, p2;
                split2(ev[i], od[i], e, p1, p2);
                char *cell = lds + ((kp >> 3) * 64 + 4 * quad + i) * kCell + 4 * (kp & 7);
                *reinterpret_cast<unsigned *>(cell) = p1;
                *reinterpret_cast<unsigned *>(cell + 32) = p2;
            }
        }
    }
    __syncthreads();
    const int row = tid >> 2, c = tid & 3;
    if (r0 + row < p.rows) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long kc = k0 / kBK + j;
            if (kc < Kc)
                *reinterpret_cast<u32x4 *>(p.cells + ((size_t)kc * p.rows + r0 + row) * kCell + 16 * c) =
                    *reinterpret_cast<const u32x4 *>(lds + (j * 64 + row) * kCell + 16 * c);
        }
    }
}

__global__ __launch_bounds__(256) void planes_dual_kernel(const PlaneJob a, const PlaneJob b)
{
    __shared__ __attribute__((aligned(16))) char lds[4 * 64 * kCell];
    if ((int)blockIdx.x < a.nblocks) planes_tile(a, (int)blockIdx.x, lds);
    else planes_tile(b, (int)blockIdx.x - a.nblocks, lds);
}

// both images of ONE matrix from one HBM read: block (i, j) runs the K-contiguous job on X's tile (row tile i, column tile j)
// and then the k-major job on the same 16 KB tile (its second read comes out of L1 / L2)
__global__ __launch_bounds__(256) void planes_both_kernel(const PlaneJob rows_job, const PlaneJob cols_job)
{
    __shared__ __attribute__((aligned(16))) char lds[4 * 64 * kCell];
    const int tj = (int)(blockIdx.x % rows_job.tiles_k), ti = (int)(blockIdx.x / rows_job.tiles_k);
    planes_tile(rows_job, ti * rows_job.tiles_k + tj, lds);
    __syncthreads();
    planes_tile(cols_job, tj * cols_job.tiles_k + ti, lds);
}

__global__ __launch_bounds__(256) void zero_two_kernel(unsigned *__restrict__ a, long long na, unsigned *__restrict__ b, long long nb)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < na || i < nb; i += (long long)gridDim.x * 256) {
        if (i < na) a[i] = 0u;
        if (i < nb) b[i] = 0u;
    }
}

// row AND column |x| maxima of X [R][C] in one pass (both arrays zero on entry): block = 64 rows x 256 columns; a wave owns
// rows w, w + 4, ... (one shuffle reduction + one atomicMax per row), column maxima are combined over the four waves in LDS
__global__ __launch_bounds__(256) void absmax_both_kernel(const float *__restrict__ X, long long R, long long C, long long ld, int vec,
                                                          unsigned *__restrict__ rowbits, unsigned *__restrict__ colbits)
{
    __shared__ unsigned red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long c0 = (long long)blockIdx.x * 256 + 4 * lane, r0 = (long long)blockIdx.y * 64;
    unsigned cm[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16; ++i) {
        const long long r = r0 + wave + 4 * i;
        unsigned v[4] = {0, 0, 0, 0};
        if (r < R && c0 < C) {
            const float *src = X + r * ld + c0;
            if (vec && c0 + 3 < C) {
                const float4 q = *reinterpret_cast<const float4 *>(src);
                v[0] = __float_as_uint(q.x) & 0x7fffffffu; v[1] = __float_as_uint(q.y) & 0x7fffffffu;
                v[2] = __float_as_uint(q.z) & 0x7fffffffu; v[3] = __float_as_uint(q.w) & 0x7fffffffu;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (c0 + k < C) v[k] = __float_as_uint(src[k]) & 0x7fffffffu;
            }
        }
        unsigned rm = max(max(v[0], v[1]), max(v[2], v[3]));
#pragma unroll
        for (int k = 0; k < 4; ++k) cm[k] = max(cm[k], v[k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rm = max(rm, (unsigned)__shfl_xor((int)rm, o));
        if (lane == 0 && r < R && rm) atomicMax(rowbits + r, rm);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[wave][4 * lane + k] = cm[k];
    __syncthreads();
    const long long col = (long long)blockIdx.x * 256 + threadIdx.x;
    if (col < C) {
        const unsigned m = max(max(red[0][threadIdx.x], red[1][threadIdx.x]), max(red[2][threadIdx.x], red[3][threadIdx.x]));
        if (m) atomicMax(colbits + col, m);
    }
}

static PlaneJob make_job(const float *X, bool k_contiguous, long long rows, long long K, long long ld, void *cells,
                         const unsigned *maxbits, int exp_div)
{
    PlaneJob j;
    j.X = X; j.ld = ld; j.rows = rows; j.K = K; j.cells = reinterpret_cast<char *>(cells); j.maxbits = maxbits;
    j.exp_div = exp_div; j.kmajor = k_contiguous ? 0 : 1;
    j.vec = ((reinterpret_cast<uintptr_t>(X) & 15) == 0 && ld % 4 == 0) ? 1 : 0;
    j.tiles_k = (int)((K + 63) / 64);
    j.nblocks = (int)(j.tiles_k * ((rows + 63) / 64));
    return j;
}

// ------------------------------------------------------------------------------------------------- the GEMM kernel
struct GemmArgs {
    const char *A, *B;             // cells of A [Kc][M][64], B [Kc][N][64]
    const unsigned *mbA, *mbB;     // row maxima (fp32 bits)
    int M, N, Kc;
    float *C;
    int ldc;
    const float *bias;
    int epilogue, accumulate;
    int splitk, kc_per_split;
    float *partial;                // [splitk][M][N] when splitk > 1
    int tiles_m, tiles_n, patch_h, patch_w;
    // round 6, XCD-banded order (1-D grid): 0 = the round-3 numbering on a (tiles, splitk) grid; 1 = every XCD owns whole K slices
    // (slice z on XCD z % 8); 2 = 8 / splitk XCDs share a slice and each owns a band of its tile space (band_n: along N)
    int order, bands, band_n;
};

// Which (tile, K slice) a block works on.  Blocks are handed to the XCDs round-robin by their linear id (block b on XCD b % 8,
// MI355X_MICROARCH.md) and every XCD has its own L2, so what the blocks an XCD runs AT THE SAME TIME have in common decides
// how often an operand panel crosses the fabric.  Round 3-5 gave XCD x a contiguous chunk of the patch-major tile numbering
// in EVERY K slice: ~12 tiles x 2-3 slices at a time, i.e. 12 panels per 12 tiles and slice -- the fc6 forward pulled 4.1x its
// operands through the fabric, its input gradient 5.7x (profiles/r05_gemm_traffic_summary.json).  Now the slices are spread
// over the XCDs first (an XCD streams its own K range of both operands: nothing is fetched twice across XCDs), and what is
// left of the XCDs per slice divides the tile space into bands along its longer side; inside its region an XCD walks
// patch_h x patch_w patches sized to what it runs at once (near-square in bytes: patch_h * bm ~ patch_w * bn).
__host__ __device__ __forceinline__ bool gemm_item(const GemmArgs &p, int bx, int by, int &tm, int &tn, int &z)
{
    if (p.order == 0) {
        const int t = xcd_remap(bx, p.tiles_m * p.tiles_n);
        patch_tile(t, p.tiles_m, p.tiles_n, p.patch_h, p.patch_w, tm, tn);
        z = by;
        return true;
    }
    const int xcd = bx & 7, idx = bx >> 3;
    if (p.order == 1) {
        const int tiles = p.tiles_m * p.tiles_n;
        const int zl = idx / tiles;
        z = xcd + 8 * zl;
        if (z >= p.splitk) return false;
        patch_tile(idx - zl * tiles, p.tiles_m, p.tiles_n, p.patch_h, p.patch_w, tm, tn);
        return true;
    }
    const int b = xcd % p.bands;
    z = xcd / p.bands;
    const int along = p.band_n ? p.tiles_n : p.tiles_m;
    const int lo = (int)((long long)b * along / p.bands), hi = (int)((long long)(b + 1) * along / p.bands);
    const int rm = p.band_n ? p.tiles_m : hi - lo, rn = p.band_n ? hi - lo : p.tiles_n;
    if (z >= p.splitk || idx >= rm * rn) return false;
    patch_tile(idx, rm, rn, p.patch_h < rm ? p.patch_h : rm, p.patch_w < rn ? p.patch_w : rn, tm, tn);
    if (p.band_n) tn += lo; else tm += lo;
    return true;
}

__device__ __forceinline__ float epi(float v, int epilogue)
{
    if (epilogue == MH_EPI_RELU) return fmaxf(v, 0.f);
    if (epilogue == MH_EPI_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
    return v;
}

template <class S>
__global__ __launch_bounds__(kThreads, 2) void gemm_kernel(const GemmArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int wm, wn;
    wave_origin<S>(wave, wm, wn);
    int tm, tn, z;
    if (!gemm_item(p, (int)blockIdx.x, (int)blockIdx.y, tm, tn, z)) return;
    const int m0 = tm * S::bm, n0 = tn * S::bn;
    const int kt_begin = z * p.kc_per_split, kt_end = min(p.Kc, kt_begin + p.kc_per_split);

    const Src sa = make_src(p.A + (size_t)m0 * kCell), sb = make_src(p.B + (size_t)n0 * kCell);
    const unsigned strideA = (unsigned)p.M * kCell, strideB = (unsigned)p.N * kCell;
    CopyPlan<S> cp;
    plan_copy<S>(cp, [&](int r) { return m0 + r < p.M; }, [&](int r) { return n0 + r < p.N; }, tid);
    FragPlan fp;
    plan_frags<S>(fp, wm, wn, lane);
    auto issue = [&](Stage<S> &st, int kt) {
        const unsigned oa = (unsigned)kt * strideA, ob = (unsigned)kt * strideB;
#pragma unroll
        for (int j = 0; j < S::na; ++j) st.a[j] = load16(sa, cp.va[j], oa);
#pragma unroll
        for (int j = 0; j < S::nb; ++j) st.b[j] = load16(sb, cp.vb[j], ob);
    };

    Acc<S> acc;
    acc_zero<S>(acc);
    Stage<S> st;
    char *b0 = lds, *b1 = lds + S::buf_bytes;
    issue(st, kt_begin);
    store_stage<S>(st, cp, b0);
    __syncthreads();
    int kt = kt_begin;
    for (; kt + 1 < kt_end; kt += 2) {
        k_step<S>([&](Stage<S> &s) { issue(s, kt + 1); }, st, cp, fp, b0, b1, acc);
        k_step<S>([&](Stage<S> &s) { issue(s, min(kt + 2, kt_end - 1)); }, st, cp, fp, b1, b0, acc);
    }
    if (kt < kt_end) k_step<S>([&](Stage<S> &s) { issue(s, kt); }, st, cp, fp, b0, b1, acc);   // odd count: harmless reload

    // the tile buffers are free after the last barrier: this tile's row / column exponents go there
    int *ex = reinterpret_cast<int *>(lds);
    for (int i = tid; i < S::bm + S::bn; i += kThreads) {
        const bool is_a = i < S::bm;
        const int idx = is_a ? m0 + i : n0 + (i - S::bm);
        const bool ok = is_a ? idx < p.M : idx < p.N;
        ex[i] = ok ? row_exponent(is_a ? p.mbA[idx] : p.mbB[idx]) : 0;
    }
    __syncthreads();
    int ecol[S::sn];
    float bcol[S::sn];
#pragma unroll
    for (int sn = 0; sn < S::sn; ++sn) {
        const int c = wn + 32 * sn + (lane & 31);
        ecol[sn] = ex[S::bm + c];
        bcol[sn] = (p.bias && p.splitk == 1 && n0 + c < p.N) ? p.bias[n0 + c] : 0.f;
    }
    if (p.splitk > 1) {
        float *dst = p.partial + (size_t)z * p.M * p.N;
        acc_foreach<S>(acc, wm, wn, lane, [&](int r, int c, int sn, float v) {
            const int row = m0 + r, col = n0 + c;
            if (row < p.M && col < p.N) dst[(size_t)row * p.N + col] = __builtin_ldexpf(v, -(ex[r] + ecol[sn]));
        });
        return;
    }
    acc_foreach<S>(acc, wm, wn, lane, [&](int r, int c, int sn, float v) {
        const int row = m0 + r, col = n0 + c;
        if (row >= p.M || col >= p.N) return;
        v = epi(__builtin_ldexpf(v, -(ex[r] + ecol[sn])) + bcol[sn], p.epilogue);
        float *q = p.C + (size_t)row * p.ldc + col;
        if (p.accumulate) v += *q;
        *q = v;
    });
}


// ------------------------------------------------------------------------------------------------- the ring GEMM kernel
// Round 4: the same product on the ring loop of pl_ring.h (LDS-DMA staging, register double-buffered fragments, 64x128 wave
// tiles).  Operand images, scales, epilogue and tile numbering are those of gemm_kernel above.
template <class R>
__global__ __launch_bounds__(R::threads, (R::threads == 512 || R::sm * R::sn <= 8) ? 2 : 1) void gemm_ring_kernel(const GemmArgs p)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int wm0, wn0;
    rwave_origin<R>(wave, wm0, wn0);
    int tm, tn, z;
    if (!gemm_item(p, (int)blockIdx.x, (int)blockIdx.y, tm, tn, z)) return;
    const int m0 = tm * R::bm, n0 = tn * R::bn;
    const int kt_begin = z * p.kc_per_split, kt_end = min(p.Kc, kt_begin + p.kc_per_split);

    const Src sa = make_src(p.A + (size_t)m0 * kCell), sb = make_src(p.B + (size_t)n0 * kCell);
    const unsigned strideA = (unsigned)p.M * kCell, strideB = (unsigned)p.N * kCell;
    DmaPlan<R> dp;
    plan_dma<R>(dp, [&](int r) { return m0 + r < p.M; }, [&](int r) { return n0 + r < p.N; }, wave, lane);
    FragPlan fp;
    rplan_frags<R>(fp, wm0, wn0, lane);
    auto issue = [&](int kt, int, char *stage) {
        const bool live = kt < kt_end;                        // padding steps: A out of range (zeros), B = the last tile again
        const int ktc = live ? kt : kt_end - 1;
        unsigned va[R::na];
#pragma unroll
        for (int j = 0; j < R::na; ++j) va[j] = live ? dp.va[j] : kOob;
        dma_stage<R>(sa, sb, va, dp.vb, (unsigned)ktc * strideA, (unsigned)ktc * strideB, stage, wave);
    };
    RAcc<R> acc;
    racc_zero<R>(acc);
    ring_loop<R>(issue, kt_begin, kt_end, lds, fp, acc);

    // The ring is free after the loop's last barrier.  Epilogue (see pl_ring.h: rmma): the accumulators are TRANSPOSED -- a lane
    // holds four consecutive columns of one row per register quad -- so every 32 x 32 accumulator is scaled, staged in a
    // wave-private LDS patch with ds_write_b128 and leaves as 16-byte stores that cover whole 128-byte lines of C.
    int *ex = reinterpret_cast<int *>(lds);                       // row exponents [bm], column exponents [bn]
    float *bs = reinterpret_cast<float *>(lds) + R::bm + R::bn;   // bias [bn]
    for (int i = tid; i < R::bm + R::bn; i += R::threads) {
        const bool is_a = i < R::bm;
        const int idx = is_a ? m0 + i : n0 + (i - R::bm);
        const bool ok = is_a ? idx < p.M : idx < p.N;
        ex[i] = ok ? row_exponent(is_a ? p.mbA[idx] : p.mbB[idx]) : 0;
        if (!is_a) bs[i - R::bm] = (ok && p.bias && p.splitk == 1) ? p.bias[idx] : 0.f;
    }
    __syncthreads();
    static_assert((R::bm + 2 * R::bn) * 4 <= 8192, "tables in front of the staging patches");
    const int j = lane & 31, g = lane >> 5;
    float *stg = reinterpret_cast<float *>(lds + 8192) + wave * (32 * 36);
    float *dst = (p.splitk > 1) ? p.partial + (size_t)z * p.M * p.N : p.C;
    const size_t ldd = (p.splitk > 1) ? (size_t)p.N : (size_t)p.ldc;
    const bool vec_ok = (ldd % 4 == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
#pragma unroll
    for (int sm = 0; sm < R::sm; ++sm) {
        const int ea = ex[wm0 + 32 * sm + j];
#pragma unroll
        for (int sn = 0; sn < R::sn; ++sn) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = wn0 + 32 * sn + 8 * q + 4 * g;
                const int4 eb = *reinterpret_cast<const int4 *>(ex + R::bm + c);
                const float4 bb = *reinterpret_cast<const float4 *>(bs + c);
                float4 v;
                v.x = __builtin_ldexpf(acc.v[sm][sn][4 * q + 0], -(ea + eb.x)) + bb.x;
                v.y = __builtin_ldexpf(acc.v[sm][sn][4 * q + 1], -(ea + eb.y)) + bb.y;
                v.z = __builtin_ldexpf(acc.v[sm][sn][4 * q + 2], -(ea + eb.z)) + bb.z;
                v.w = __builtin_ldexpf(acc.v[sm][sn][4 * q + 3], -(ea + eb.w)) + bb.w;
                if (p.splitk == 1) { v.x = epi(v.x, p.epilogue); v.y = epi(v.y, p.epilogue); v.z = epi(v.z, p.epilogue); v.w = epi(v.w, p.epilogue); }
                *reinterpret_cast<float4 *>(stg + j * 36 + 8 * q + 4 * g) = v;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = 8 * i + (lane >> 3), c4 = 4 * (lane & 7);
                const int row = m0 + wm0 + 32 * sm + rr, col = n0 + wn0 + 32 * sn + c4;
                float4 v = *reinterpret_cast<const float4 *>(stg + rr * 36 + c4);
                if (row >= p.M || col >= p.N) continue;
                float *q = dst + (size_t)row * ldd + col;
                if (vec_ok && col + 3 < p.N) {
                    if (p.accumulate && p.splitk == 1) { const float4 o = *reinterpret_cast<const float4 *>(q); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                    *reinterpret_cast<float4 *>(q) = v;
                } else {
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (col + k < p.N) q[k] = (p.accumulate && p.splitk == 1) ? q[k] + vv[k] : vv[k];
                }
            }
        }
    }
}

template <class R>
constexpr size_t ring_gemm_lds() { return (size_t)(R::lds_bytes > 8192 + R::waves * 32 * 36 * 4 ? R::lds_bytes : 8192 + R::waves * 32 * 36 * 4); }

typedef Ring<4, 2, 2, 4, 4> R256x256;        // 8 waves, 64x128 wave tiles, 4 stages x 32 KB: one block per CU
typedef Ring<4, 1, 2, 4, 3> R256x128;        // 4 waves, 3 stages x 24 KB: two blocks per CU

typedef Shape<256, 128, 4, 2> S256x128;
typedef Shape<128, 128, 2, 2> S128x128;
typedef Shape<256, 64, 2, 2> S256x64;

// ------------------------------------------------------------------------------------------------- host side
struct Plan {
    int shape;     // round-3 loop: 0: 256x128, 1: 128x128, 2: 256x64; ring loop: 3: 256x256 (8 waves), 4: 256x128 (4 waves, 2 / CU)
    int bm, bn, splitk;
};
static int g_force_shape = -1;      // mh_debug_pl_shape: A/B runs
// tile order (gemm_item): 1 = XCD-banded (round 6), 0 = round-3 patch numbering; MH_GEMM_ORDER / mh_debug_pl_order for A/B runs
static int g_order = [] { const char *e = getenv("MH_GEMM_ORDER"); return (e && e[0] == '0') ? 0 : 1; }();

struct Order {
    int order, bands, band_n, per_xcd;      // per_xcd: work items of the busiest XCD (grid = 8 * per_xcd blocks)
};
// how the (tiles_m x tiles_n x splitk) work items are dealt to the 8 XCDs (gemm_item)
static Order plan_order(int tiles_m, int tiles_n, int bm, int bn, int splitk)
{
    Order o = {0, 1, 1, 0};
    if (!g_order) return o;
    const int tiles = tiles_m * tiles_n;
    if (splitk >= 8) { o.order = 1; o.per_xcd = ceil_div(splitk, 8) * tiles; return o; }
    if (8 % splitk) return o;                                 // 3, 5, 6, 7 slices: the planner does not pick them when banding
    const int bands = 8 / splitk;
    const bool n_longer = (long long)tiles_n * bn >= (long long)tiles_m * bm;
    int band_n = n_longer ? 1 : 0;
    if ((band_n ? tiles_n : tiles_m) < bands) band_n ^= 1;
    const int along = band_n ? tiles_n : tiles_m;
    if (along < bands) return o;                              // too few tiles to give every XCD a band
    o.order = 2; o.bands = bands; o.band_n = band_n;
    o.per_xcd = ceil_div(along, bands) * (band_n ? tiles_m : tiles_n);
    return o;
}

static Plan plan_gemm(int M, int N, int K, int want_splitk)
{
    static const int bms[5] = {256, 128, 256, 256, 256}, bns[5] = {128, 128, 64, 256, 128};
    // fp32-equivalent FLOP/s one CU sustains with a full complement of blocks of the shape (measured: DESIGN.md 5)
    // (round 6, gpurun r06_c1: the fc6 input gradient 1536 x 25088 x 4096 ran at 350-357 TF/s on the 128x128 round-3 tiles this
    // table used to pick and at 394-402 on the 256x128 ring tiles: the round-3 rates were optimistic by ~7 %)
    static const double rate[5] = {340e12 / 256, 345e12 / 256, 300e12 / 256, 450e12 / 256, 430e12 / 256};
    static const int per_cu[5] = {2, 2, 2, 1, 2};          // resident blocks per CU the makespan model assumes
    const int ktiles = ceil_div(K, kBK);
    Plan best = {1, 128, 128, 1};
    double best_cost = 1e30;
    for (int s = 0; s < 5; ++s) {
        if (g_force_shape >= 0 && s != g_force_shape) continue;
        if (g_force_shape < 0) {
            // ring shapes (gpurun r04_c6, TFLOP/s on ready images, ring vs the round-3 loop: 4096^3 436 vs 389, fc6 forward 389 vs
            // 373, fc6 input gradient 386 vs 353, fc6 weight gradient 381 vs 333, fc7 forward 337 vs 302): 256x256 on eight waves
            // for wide products, 256x128 two per CU otherwise (a 4-wave variant with 128x128 wave tiles was measured and removed: never won)
            static const bool ring_off = [] { const char *e = getenv("MH_PL_RING"); return e && e[0] == '0'; }();      // A/B: MH_PL_RING=0
            if (ring_off && s >= 3) continue;
            if (s == 3 && (N <= 128 || M <= 128)) continue;
            if (s == 4 && (N <= 64 || M <= 128)) continue;
            if (s == 2 && N > 64) continue;
            if (s != 2 && N <= 64) continue;
            if (s == 0 && M <= 128) continue;
        }
        const long long tiles = (long long)ceil_div(M, bms[s]) * ceil_div(N, bns[s]);
        const double t1 = 2.0 * bms[s] * bns[s] * (double)K / rate[s];      // one tile on a fully occupied CU
        static const int cand[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64};
        set_resident_slots_override(256 * per_cu[s]);
        auto consider = [&](int sk) {
            const double t_partial = (sk > 1) ? ((double)M * N * 8.0 * sk) / 4.0e12 + 4e-6 : 0.0;
            // banded orders: the XCD with the most work items sets the pace (8 * per_xcd blocks' worth of rounds)
            const Order o = plan_order(ceil_div(M, bms[s]), ceil_div(N, bns[s]), bms[s], bns[s], sk);
            const long long blocks = o.order ? 8LL * o.per_xcd : tiles * sk;
            const double cost = makespan_units(blocks) * t1 / sk + t_partial;
            if (cost < best_cost * 0.97) { best_cost = cost; best = {s, bms[s], bns[s], sk}; }
        };
        if (want_splitk > 0) { consider(std::max(1, std::min(std::min(want_splitk, 64), ktiles))); set_resident_slots_override(0); continue; }
        for (int sk : cand) {
            if (sk > 1 && ktiles / sk < 8) break;
            if (g_order && sk < 8 && 8 % sk) continue;          // banded order: 1, 2, 4 slices share the XCDs evenly, >= 8 own them
            consider(sk);
        }
    }
    set_resident_slots_override(0);
    return best;
}

// the dispatch rule of mh_gemm_f32: below 20 GFLOP, or with a shallow K, a one-shot product is not worth two plane images
static inline bool small_product(int M, int N, int K) { return 2.0 * M * N * (double)K < 20e9 || K < 512; }

static inline size_t cells_bytes(long long rows, long long K) { return (size_t)ceil_div(K, (long long)kBK) * rows * kCell; }
static inline size_t image_bytes(long long rows, long long K) { return align_up(cells_bytes(rows, K), 256) + align_up((size_t)rows * 4, 256); }
static inline unsigned *image_maxbits(void *image, long long rows, long long K)
{
    return reinterpret_cast<unsigned *>(reinterpret_cast<char *>(image) + align_up(cells_bytes(rows, K), 256));
}

// tile counts, order and patch of a launch; returns the grid
static dim3 plan_launch(GemmArgs &p, const Plan &pl)
{
    p.tiles_m = ceil_div(p.M, pl.bm);
    p.tiles_n = ceil_div(p.N, pl.bn);
    int ph = 8, pw = 8;       // ~64 tiles per patch = what one XCD runs at a time (patch_tile)
    if (p.tiles_m < 8) { ph = p.tiles_m; pw = std::min(p.tiles_n, std::max(1, 64 / ph)); }
    else if (p.tiles_n < 8) { pw = p.tiles_n; ph = std::min(p.tiles_m, std::max(1, 64 / pw)); }
    p.patch_h = std::max(ph, 1);
    p.patch_w = std::max(pw, 1);
    const Order o = plan_order(p.tiles_m, p.tiles_n, pl.bm, pl.bn, p.splitk);
    p.order = o.order; p.bands = o.bands; p.band_n = o.band_n;
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)p.splitk);
    if (o.order) {
        // the region an XCD walks, and the patch of tiles it runs at once: conc blocks, near-square in operand bytes
        const int along = o.band_n ? p.tiles_n : p.tiles_m;
        const int rm = (o.order == 2 && !o.band_n) ? ceil_div(along, o.bands) : p.tiles_m;
        const int rn = (o.order == 2 && o.band_n) ? ceil_div(along, o.bands) : p.tiles_n;
        const int conc = (pl.shape == 3) ? 32 : 64;
        int h = std::max(1, (int)std::lround(std::sqrt((double)conc * pl.bn / pl.bm)));
        h = std::min(h, rm);
        int w = std::min(rn, std::max(1, conc / h));
        if (w == rn) h = std::min(rm, std::max(h, conc / w));      // a narrow region: taller patches
        p.patch_h = h; p.patch_w = w;
        grid = dim3((unsigned)(8 * o.per_xcd), 1);
    }
    return grid;
}

static int launch_gemm(const GemmArgs &p0, const Plan &pl, hipStream_t st)
{
    GemmArgs p = p0;
    const dim3 grid = plan_launch(p, pl);
    if (pl.shape == 3) launch<gemm_ring_kernel<R256x256>>(grid, ring_gemm_lds<R256x256>(), st, p, 0, R256x256::threads);
    else if (pl.shape == 4) launch<gemm_ring_kernel<R256x128>>(grid, ring_gemm_lds<R256x128>(), st, p, 0, R256x128::threads);
    else if (pl.shape == 0) launch<gemm_kernel<S256x128>>(grid, S256x128::lds_bytes, st, p);
    else if (pl.shape == 1) launch<gemm_kernel<S128x128>>(grid, S128x128::lds_bytes, st, p);
    else launch<gemm_kernel<S256x64>>(grid, S256x64::lds_bytes, st, p);
    return check_launch("pl::gemm_kernel");
}

}  // namespace pl
}  // namespace mh

namespace mh {
int gemm_f32_ctr(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
                 const float *bias, int epilogue, int accumulate, void *workspace, size_t ws_bytes, int *counters, int n_counters,
                 void *stream)
{
    if (pl::g_force_shape < 0 && M > 0 && N > 0 && K > 0 && pl::small_product(M, N, K))
        return gemm_small(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, epilogue, accumulate, workspace, ws_bytes, counters,
                          n_counters, stream);
    return mh_gemm_f32(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, epilogue, accumulate, 0, workspace, ws_bytes, stream);
}
}  // namespace mh

using namespace mh;

extern "C" {

void mh_debug_pl_shape(int shape) { pl::g_force_shape = shape; }
void mh_debug_pl_order(int order) { pl::g_order = order ? 1 : 0; }

// the work item of block `block` of the launch mh_gemm_planes(M, N, K, splitk) would make, computed on the host with the
// kernels' own mapping (tests/test_gemm_order.py: every (tile, slice) exactly once).  out = {grid_x, grid_y, tm, tn, z, valid,
// tiles_m, tiles_n, splitk, order}; returns MH_EINVAL for a block outside the grid.
int mh_debug_pl_item(int M, int N, int K, int splitk, long long block, int *out)
{
    MH_REQUIRE(M > 0 && N > 0 && K > 0 && out && block >= 0);
    const pl::Plan pln = pl::plan_gemm(M, N, K, splitk);
    pl::GemmArgs p = {};
    p.M = M; p.N = N; p.Kc = ceil_div(K, pl::kBK);
    p.kc_per_split = ceil_div(p.Kc, pln.splitk);
    p.splitk = ceil_div(p.Kc, p.kc_per_split);
    const dim3 grid = pl::plan_launch(p, pln);
    MH_REQUIRE(block < (long long)grid.x * grid.y);
    int tm = -1, tn = -1, z = -1;
    const bool ok = pl::gemm_item(p, (int)(block % grid.x), (int)(block / grid.x), tm, tn, z);
    out[0] = (int)grid.x; out[1] = (int)grid.y; out[2] = tm; out[3] = tn; out[4] = z; out[5] = ok ? 1 : 0;
    out[6] = p.tiles_m; out[7] = p.tiles_n; out[8] = p.splitk; out[9] = p.order;
    return MH_OK;
}

size_t mh_planes_bytes(long long rows, long long K)
{
    if (rows <= 0 || K <= 0) return 0;
    return pl::image_bytes(rows, K);
}

// plane image of the operand whose `rows` rows (the non-K index) have K elements: k_contiguous = X is [rows][K] (ld),
// otherwise X is [K][rows] (ld).  Two launches (+ one memset for k-major storage): row maxima, split.
int mh_make_planes(const float *X, int k_contiguous, long long rows, long long K, long long ld, void *image, void *stream)
{
    MH_REQUIRE(X && image && rows > 0 && K > 0);
    MH_REQUIRE(ld >= (k_contiguous ? K : rows));
    MH_REQUIRE((reinterpret_cast<uintptr_t>(image) & 255) == 0);
    MH_REQUIRE(pl::cells_bytes(rows, K) < (size_t)0x7ff00000u);      // 32-bit offsets inside the kernels' descriptors
    hipStream_t st = as_stream(stream);
    unsigned *mb = pl::image_maxbits(image, rows, K);
    int rc = launch_row_exponents(X, k_contiguous != 0, rows, K, ld, reinterpret_cast<int *>(mb), st, /*bits_only=*/true);
    if (rc) return rc;
    const pl::PlaneJob a = pl::make_job(X, k_contiguous != 0, rows, K, ld, image, mb, 1);
    pl::PlaneJob none = a;
    none.nblocks = 0;
    hipLaunchKernelGGL(pl::planes_dual_kernel, dim3((unsigned)a.nblocks), dim3(256), 0, st, a, none);
    return check_launch("pl::planes_dual_kernel");
}

// BOTH images of the matrix X [R][C] (ld): img_rows = operand rows are X's rows (K = C), img_cols = operand rows are X's
// columns (K = R).  One pass for both sets of maxima, one pass (one HBM read) for both images: what a Linear layer needs of
// its weight (forward / input gradient), its input (forward / weight gradient) and its output gradient.
int mh_make_planes_both(const float *X, long long R, long long C, long long ld, void *img_rows, void *img_cols, void *stream)
{
    MH_REQUIRE(X && img_rows && img_cols && R > 0 && C > 0 && ld >= C);
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(img_rows) | reinterpret_cast<uintptr_t>(img_cols)) & 255) == 0);
    MH_REQUIRE(pl::cells_bytes(R, C) < (size_t)0x7ff00000u && pl::cells_bytes(C, R) < (size_t)0x7ff00000u);
    hipStream_t st = as_stream(stream);
    unsigned *rb = pl::image_maxbits(img_rows, R, C), *cb = pl::image_maxbits(img_cols, C, R);
    // both sets of maxima cleared by ONE launch (they live in two buffers: two memsets were two launches in front of every pair of images)
    hipLaunchKernelGGL(pl::zero_two_kernel, dim3((unsigned)std::min<long long>(ceil_div(std::max(R, C), 256LL), 1024)), dim3(256), 0, st, rb, R, cb, C);
    {
        int rc0 = check_launch("pl::zero_two_kernel");
        if (rc0) return rc0;
    }
    const int vec = ((reinterpret_cast<uintptr_t>(X) & 15) == 0 && ld % 4 == 0) ? 1 : 0;
    MH_REQUIRE(ceil_div(R, 64LL) <= 65535);
    hipLaunchKernelGGL(pl::absmax_both_kernel, dim3((unsigned)ceil_div(C, 256LL), (unsigned)ceil_div(R, 64LL)), dim3(256), 0, st, X, R, C,
                       ld, vec, rb, cb);
    int rc = check_launch("pl::absmax_both_kernel");
    if (rc) return rc;
    const pl::PlaneJob jr = pl::make_job(X, true, R, C, ld, img_rows, rb, 1);
    const pl::PlaneJob jc = pl::make_job(X, false, C, R, ld, img_cols, cb, 1);
    hipLaunchKernelGGL(pl::planes_both_kernel, dim3((unsigned)jr.nblocks), dim3(256), 0, st, jr, jc);
    return check_launch("pl::planes_both_kernel");
}

int mh_gemm_planes_auto_splitk(int M, int N, int K) { return (M > 0 && N > 0 && K > 0) ? pl::plan_gemm(M, N, K, 0).splitk : 1; }

size_t mh_gemm_planes_ws_bytes(int M, int N, int K, int splitk)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const pl::Plan pln = pl::plan_gemm(M, N, K, splitk);
    return pln.splitk > 1 ? align_up((size_t)pln.splitk * M * N * sizeof(float), 256) : 0;
}

static int gemm_planes_impl(int M, int N, int K, const void *cellsA, const unsigned *mbA, const void *cellsB, const unsigned *mbB,
                            float *C, int ldc, const float *bias, int epilogue, int accumulate, int splitk, void *workspace,
                            size_t ws_bytes, void *stream)
{
    MH_REQUIRE(M >= 0 && N >= 0 && K >= 0);
    if (M == 0 || N == 0) return MH_OK;
    MH_REQUIRE(cellsA && cellsB && mbA && mbB && C && K > 0 && ldc >= N);
    MH_REQUIRE(epilogue >= MH_EPI_NONE && epilogue <= MH_EPI_RELU6);
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(cellsA) | reinterpret_cast<uintptr_t>(cellsB)) & 15) == 0);
    MH_REQUIRE(pl::cells_bytes(M, K) < (size_t)0x7ff00000u && pl::cells_bytes(N, K) < (size_t)0x7ff00000u);
    pl::Plan pln = pl::plan_gemm(M, N, K, splitk);
    if (pln.splitk > 1 && (workspace == nullptr || ws_bytes < (size_t)pln.splitk * M * N * sizeof(float))) pln.splitk = 1;
    pl::GemmArgs p;
    p.A = reinterpret_cast<const char *>(cellsA);
    p.B = reinterpret_cast<const char *>(cellsB);
    p.mbA = mbA;
    p.mbB = mbB;
    p.M = M; p.N = N; p.Kc = ceil_div(K, pl::kBK);
    p.C = C; p.ldc = ldc; p.bias = bias; p.epilogue = epilogue; p.accumulate = accumulate;
    p.kc_per_split = ceil_div(p.Kc, pln.splitk);
    p.splitk = ceil_div(p.Kc, p.kc_per_split);
    p.partial = reinterpret_cast<float *>(workspace);
    hipStream_t st = as_stream(stream);
    int rc = pl::launch_gemm(p, pln, st);
    if (rc || p.splitk == 1) return rc;
    return launch_splitk_reduce(p.partial, p.splitk, M, N, C, ldc, bias, epilogue, accumulate, st);
}

int mh_gemm_planes(int M, int N, int K, const void *A_image, const void *B_image, float *C, int ldc, const float *bias,
                   int epilogue, int accumulate, int splitk, void *workspace, size_t ws_bytes, void *stream)
{
    if (M > 0 && N > 0) MH_REQUIRE(A_image && B_image && K > 0);
    else return MH_OK;
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(A_image) | reinterpret_cast<uintptr_t>(B_image)) & 255) == 0);
    return gemm_planes_impl(M, N, K, A_image, pl::image_maxbits(const_cast<void *>(A_image), M, K), B_image,
                            pl::image_maxbits(const_cast<void *>(B_image), N, K), C, ldc, bias, epilogue, accumulate, splitk,
                            workspace, ws_bytes, stream);
}

// ---- the fp32-operand entry point (round-1 signature): images of both operands in the workspace, then the plane GEMM
size_t mh_gemm_ws_bytes(int M, int N, int K, int splitk)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const size_t v3 = align_up((size_t)M * 4, 256) + align_up((size_t)N * 4, 256) + align_up(pl::cells_bytes(M, K), 256) +
                      align_up(pl::cells_bytes(N, K), 256) + mh_gemm_planes_ws_bytes(M, N, K, splitk);
    return std::max(v3, mh_gemm_ws_bytes_v2(M, N, K, splitk));
}

int mh_gemm_auto_splitk(int M, int N, int K) { return mh_gemm_planes_auto_splitk(M, N, K); }

int mh_gemm_f32(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
                float *C, int ldc, const float *bias, int epilogue, int accumulate, int splitk, void *workspace,
                size_t ws_bytes, void *stream)
{
    MH_REQUIRE(M >= 0 && N >= 0 && K >= 0);
    if (M == 0 || N == 0) return MH_OK;
    MH_REQUIRE(A && B && C && K > 0);
    MH_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N);
    MH_REQUIRE(epilogue >= MH_EPI_NONE && epilogue <= MH_EPI_RELU6);
    // One-shot operands of a SMALL or thin product are not worth an image each (two extra passes + launches per operand,
    // gpurun r03_c6: 50 such calls per step cost 1.0 ms of preparation for 0.3 ms of products): those go to the in-loop-split
    // kernel, which reads the fp32 operands once.  Images pay where the product is big (>= 20 GFLOP) and K is deep.
    if (pl::g_force_shape < 0 && pl::small_product(M, N, K))
        return mh_gemm_f32_v2(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, epilogue, accumulate, splitk, workspace, ws_bytes, stream);
    // workspace: maxbits A | maxbits B (adjacent: the k-major pass zeroes them with one memset) | cells A | cells B | partials
    const size_t ma = align_up((size_t)M * 4, 256), mb = align_up((size_t)N * 4, 256);
    const size_t ca = align_up(pl::cells_bytes(M, K), 256), cb = align_up(pl::cells_bytes(N, K), 256);
    MH_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 && ws_bytes >= ma + mb + ca + cb);
    MH_REQUIRE(pl::cells_bytes(M, K) < (size_t)0x7ff00000u && pl::cells_bytes(N, K) < (size_t)0x7ff00000u);
    hipStream_t st = as_stream(stream);
    char *w = reinterpret_cast<char *>(workspace);
    unsigned *mbA = reinterpret_cast<unsigned *>(w), *mbB = reinterpret_cast<unsigned *>(w + ma);
    char *cellsA = w + ma + mb, *cellsB = cellsA + ca;
    int rc = launch_operand_absmax(A, !transA, M, K, lda, reinterpret_cast<int *>(mbA), B, transB != 0, N, K, ldb,
                                   reinterpret_cast<int *>(mbB), st);
    if (rc) return rc;
    const pl::PlaneJob ja = pl::make_job(A, !transA, M, K, lda, cellsA, mbA, 1);
    const pl::PlaneJob jb = pl::make_job(B, transB != 0, N, K, ldb, cellsB, mbB, 1);
    hipLaunchKernelGGL(pl::planes_dual_kernel, dim3((unsigned)(ja.nblocks + jb.nblocks)), dim3(256), 0, st, ja, jb);
    rc = check_launch("pl::planes_dual_kernel");
    if (rc) return rc;
    return gemm_planes_impl(M, N, K, cellsA, mbA, cellsB, mbB, C, ldc, bias, epilogue, accumulate, splitk, cellsB + cb,
                            ws_bytes - (ma + mb + ca + cb), stream);
}

}  // extern "C"
