// lstm.hip -- stacked alternating-direction highway LSTM (lib/lstm/highway_lstm_cuda) for gfx950.
//
// Restructuring vs the reference host loop (highway_lstm_kernel.cu:377-496 / :162-375):
//   * the input projection x_t*Wx does not depend on the recurrence -> ONE MFMA GEMM per layer over all T*B rows
//     (mh_gemm_f32) instead of T tiny cuBLAS calls; likewise dX, dWx, dWh in the backward are single GEMMs over
//     the stored per-step gate gradients.
//   * the serial part of a step is a small-batch GEMV (n <= B rows against Wh) fused with the gate math: one
//     launch per (layer, t), no device synchronisation, no side streams.  A wave owns 4 hidden units, 16 lanes
//     split K for each; the n right-hand sides sit in LDS; partial sums are reduced with 4 xor-shuffles.
//   * sigmoid / cell expressions keep the reference's operation order (elementWise_fp :108-160), including the
//     double-precision `(1. - r)` term.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace mh {

constexpr int kNB = 8;        // right-hand sides (batch rows) processed per pass
constexpr int kChunk = 512;   // floats of K consumed per block per pipeline stage (4 waves x 16 lanes x 2 float4)
constexpr int kGemvThreads = 256;

__device__ __forceinline__ float sigmoidf_ref(float x) { return 1.f / (1.f + expf(-x)); }

// ---------------------------------------------------------------------------------------------------
// Small-batch GEMV block engine.  A 256-thread block computes, for 4 "units" (one per 16-lane group index
// grp = lane>>4, identical in all 4 waves) and NR weight rows per unit, the dot products with up to kNB vectors
//       acc[r][b] = sum_k w_r[k] * v[b][k].
// K is consumed in chunks of 512 floats: wave w owns floats [128w, 128w+128) of the chunk, lane kq owns the two
// float4 at 128w + 4kq and 128w + 64 + 4kq.  The n vectors are staged through LDS by all 256 threads (float4),
// and the NEXT chunk's weight float4s and staging registers are loaded before the current chunk's FMAs, so each
// pipeline stage costs one (overlapped) memory round trip instead of one per k-step.
// After the K loop the per-lane partials are reduced over the 16 lanes of a group with xor-shuffles and over the
// 4 waves through LDS; the block-level totals are returned in `red` ([4 units][NR][kNB], valid after the
// trailing __syncthreads()).
// ---------------------------------------------------------------------------------------------------
template <int NR>
struct WRegs {
    float4 v[NR][2];
};

template <int NR>
__device__ __forceinline__ void load_w(WRegs<NR> &w, const float *const (&wrow)[NR], bool ok_rows, bool vec, int k0,
                                       int K, int wave, int kq)
{
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = k0 + 128 * wave + 64 * h + 4 * kq;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok_rows) {
                if (vec && k + 3 < K) {
                    x = *reinterpret_cast<const float4 *>(wrow[r] + k);
                } else {
                    if (k + 0 < K) x.x = wrow[r][k + 0];
                    if (k + 1 < K) x.y = wrow[r][k + 1];
                    if (k + 2 < K) x.z = wrow[r][k + 2];
                    if (k + 3 < K) x.w = wrow[r][k + 3];
                }
            }
            w.v[r][h] = x;
        }
    }
}

struct VStage {
    float4 v[4];
};

// chunk of v: kNB rows x 128 float4; thread t takes f = t + 256 i  ->  (b = f / 128, q = f % 128)
__device__ __forceinline__ void load_vstage(VStage &s, const float *v, int ldv, bool vec, int n, int b0, int k0, int K,
                                            int tid)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + kGemvThreads * i;
        const int b = b0 + (f >> 7), k = k0 + 4 * (f & 127);
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < n) {
            const float *p = v + (size_t)b * ldv;
            if (vec && k + 3 < K) {
                x = *reinterpret_cast<const float4 *>(p + k);
            } else {
                if (k + 0 < K) x.x = p[k + 0];
                if (k + 1 < K) x.y = p[k + 1];
                if (k + 2 < K) x.z = p[k + 2];
                if (k + 3 < K) x.w = p[k + 3];
            }
        }
        s.v[i] = x;
    }
}

__device__ __forceinline__ void store_vstage(const VStage &s, float *vs, int tid)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(vs + 4 * (tid + kGemvThreads * i)) = s.v[i];
}

template <int NR>
__device__ __forceinline__ void block_gemv(const float *const (&wrow)[NR], bool ok_rows, bool wvec, const float *v,
                                           int ldv, bool vvec, int n, int b0, int K, float *vs /*[kNB*kChunk]*/,
                                           float *red /*[4 waves][4 units][NR][kNB]*/, float (&tot)[NR])
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = lane >> 4, kq = lane & 15;
    float acc[NR][kNB];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int b = 0; b < kNB; ++b) acc[r][b] = 0.f;

    WRegs<NR> wc, wn;
    VStage st;
    load_vstage(st, v, ldv, vvec, n, b0, 0, K, tid);
    load_w<NR>(wc, wrow, ok_rows, wvec, 0, K, wave, kq);
    for (int k0 = 0; k0 < K; k0 += kChunk) {
        __syncthreads();               // previous chunk's LDS reads are done
        store_vstage(st, vs, tid);
        __syncthreads();
        const bool more = (k0 + kChunk < K);
        if (more) {
            load_vstage(st, v, ldv, vvec, n, b0, k0 + kChunk, K, tid);
            load_w<NR>(wn, wrow, ok_rows, wvec, k0 + kChunk, K, wave, kq);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float *vp = vs + 128 * wave + 64 * h + 4 * kq;
#pragma unroll
            for (int b = 0; b < kNB; ++b) {
                const float4 x = *reinterpret_cast<const float4 *>(vp + b * kChunk);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    acc[r][b] = fmaf(wc.v[r][h].x, x.x, acc[r][b]);
                    acc[r][b] = fmaf(wc.v[r][h].y, x.y, acc[r][b]);
                    acc[r][b] = fmaf(wc.v[r][h].z, x.z, acc[r][b]);
                    acc[r][b] = fmaf(wc.v[r][h].w, x.w, acc[r][b]);
                }
            }
        }
        if (more) wc = wn;
    }
    // reduce over the 16 lanes of the group, then over the 4 waves (LDS)
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int b = 0; b < kNB; ++b) {
            float x = acc[r][b];
            x += __shfl_xor(x, 1);
            x += __shfl_xor(x, 2);
            x += __shfl_xor(x, 4);
            x += __shfl_xor(x, 8);
            acc[r][b] = x;
        }
    if (kq < kNB) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float x = 0.f;
#pragma unroll
            for (int b = 0; b < kNB; ++b)
                if (kq == b) x = acc[r][b];
            red[((wave * 4 + grp) * NR + r) * kNB + kq] = x;
        }
    }
    __syncthreads();
    // thread (grp = t>>3 & 3, b = t&7) of wave 0 gathers the totals for (unit grp, batch b)
#pragma unroll
    for (int r = 0; r < NR; ++r) tot[r] = 0.f;
    if (tid < 4 * kNB) {
        const int g2 = tid >> 3, b = tid & 7;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float x = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) x += red[((w * 4 + g2) * NR + r) * kNB + b];
            tot[r] = x;
        }
    }
}

// Same engine with the weights RESIDENT in registers (persistent layer kernels below): a lane always multiplies the
// same NCH x NR x 8 weight floats, so they are loaded once per layer instead of once per timestep.  K <= NCH * 512.
// The arithmetic (FMA order, shuffle tree, cross-wave sum) is that of block_gemv.
template <int NR, int NCH>
struct WResident {
    WRegs<NR> c[NCH];
};
template <int NR, int NCH>
__device__ __forceinline__ void load_resident(WResident<NR, NCH> &w, const float *const (&wrow)[NR], bool ok_rows,
                                              bool vec, int K)
{
    const int tid = threadIdx.x, wave = tid >> 6, kq = tid & 15;
#pragma unroll
    for (int c = 0; c < NCH; ++c) load_w<NR>(w.c[c], wrow, ok_rows, vec, c * kChunk, K, wave, kq);
}
template <int NR, int NCH, typename Loader>
__device__ __forceinline__ void block_gemv_resident_from(const WResident<NR, NCH> &w, Loader load_chunk, int K, float *vs, float *red,
                                                         float (&tot)[NR]);
template <int NR, int NCH>
__device__ __forceinline__ void block_gemv_resident(const WResident<NR, NCH> &w, const float *v, int ldv, bool vvec,
                                                    int n, int b0, int K, float *vs, float *red, float (&tot)[NR])
{
    const int tid = threadIdx.x;
    block_gemv_resident_from<NR, NCH>(w, [&](VStage &st, int k0) { load_vstage(st, v, ldv, vvec, n, b0, k0, K, tid); }, K, vs, red, tot);
}
// the same engine with the right-hand sides fetched by `load_chunk(stage, k0)` (plain rows of a state buffer, or the tagged
// granules of the exchange buffer: hw_layer_fwd_kernel)
template <int NR, int NCH, typename Loader>
__device__ __forceinline__ void block_gemv_resident_from(const WResident<NR, NCH> &w, Loader load_chunk, int K, float *vs, float *red,
                                                         float (&tot)[NR])
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = lane >> 4, kq = lane & 15;
    float acc[NR][kNB];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int b = 0; b < kNB; ++b) acc[r][b] = 0.f;
    VStage st;
    load_chunk(st, 0);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int k0 = c * kChunk;
        if (k0 < K) {                      // uniform
            __syncthreads();               // previous chunk's LDS reads are done
            store_vstage(st, vs, tid);
            __syncthreads();
            if (k0 + kChunk < K) load_chunk(st, k0 + kChunk);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float *vp = vs + 128 * wave + 64 * h + 4 * kq;
#pragma unroll
                for (int b = 0; b < kNB; ++b) {
                    const float4 x = *reinterpret_cast<const float4 *>(vp + b * kChunk);
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        acc[r][b] = fmaf(w.c[c].v[r][h].x, x.x, acc[r][b]);
                        acc[r][b] = fmaf(w.c[c].v[r][h].y, x.y, acc[r][b]);
                        acc[r][b] = fmaf(w.c[c].v[r][h].z, x.z, acc[r][b]);
                        acc[r][b] = fmaf(w.c[c].v[r][h].w, x.w, acc[r][b]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int b = 0; b < kNB; ++b) {
            float x = acc[r][b];
            x += __shfl_xor(x, 1);
            x += __shfl_xor(x, 2);
            x += __shfl_xor(x, 4);
            x += __shfl_xor(x, 8);
            acc[r][b] = x;
        }
    if (kq < kNB) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float x = 0.f;
#pragma unroll
            for (int b = 0; b < kNB; ++b)
                if (kq == b) x = acc[r][b];
            red[((wave * 4 + grp) * NR + r) * kNB + kq] = x;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NR; ++r) tot[r] = 0.f;
    if (tid < 4 * kNB) {
        const int g2 = tid >> 3, b = tid & 7;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float x = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) x += red[((wv * 4 + g2) * NR + r) * kNB + b];
            tot[r] = x;
        }
    }
}

// Sequence schedule passed by value to the persistent kernels (lengths sorted descending, lengths[0] == T).
constexpr int kSeqMaxB = 32;
struct SeqSched {
    int T, B, forward;
    int packed;   // 0: padded [T][B] rows, state slots [T+1][B] (slot 0 = initial state);  1: time-major PACKED rows
                  // (PackedSequence order, step t = rows start(t) .. start(t)+n_t), state buffers [B + N] rows whose
                  // first B rows are the (zero) initial state; forward direction only
    int lengths[kSeqMaxB];
};
__device__ __forceinline__ int covered_at(const SeqSched &s, int t)
{
    int n = 0;
    for (int b = 0; b < s.B; ++b) n += (s.lengths[b] > t) ? 1 : 0;
    return n;
}
// row offsets of timestep t: `io` = rows of per-step inputs/outputs (pre_i, gates, grad_in, d_gates),
// `state` = rows of the state written at t, `before` / `after` = rows of the state of time t-1 / t+1 in walking
// order of the FORWARD pass (for a reverse-direction layer "before" is time t+1), n_after = valid rows there.
struct StepRows {
    size_t io, state, before, after;
    int n, n_after;
};
__device__ __forceinline__ StepRows step_rows(const SeqSched &s, int t)
{
    StepRows r;
    r.n = covered_at(s, t);
    if (!s.packed) {
        const int tb = s.forward ? t : (t + 2) % (s.T + 1), ta = s.forward ? (t + 2) % (s.T + 1) : t;
        r.io = (size_t)t * s.B;
        r.state = (size_t)(t + 1) * s.B;
        r.before = (size_t)tb * s.B;
        r.after = (size_t)ta * s.B;
        r.n_after = s.B;                     // rows beyond the covered ones are zero in the padded buffers
        return r;
    }
    int start = 0, n_prev = 0;
    for (int q = 0; q < t; ++q) { n_prev = covered_at(s, q); start += n_prev; }
    r.io = (size_t)start;
    r.state = (size_t)s.B + start;
    r.before = (t == 0) ? 0 : (size_t)s.B + start - n_prev;
    r.after = (t + 1 < s.T) ? (size_t)s.B + start + r.n : 0;
    r.n_after = (t + 1 < s.T) ? covered_at(s, t + 1) : 0;
    return r;
}


// ---------------------------------------------------------------------------------------------------
// Tagged-granule exchange (round 4): how the hidden state travels between the workgroups of a persistent layer WITHOUT a grid
// barrier.  A granule is one naturally aligned 8-byte word {fp32 value, 32-bit tag} written by ONE agent-scope store
// (`global_store_dwordx2 ... sc1`: write-through, single-copy atomic) and read by agent-scope loads that bypass the reader's L1 --
// the "8-B agent atomics on both sides" form of MI355X_MICROARCH.md (workgroup dispatch / price list rows handoff-1to1, allgather),
// valid for any placement of the workgroups.  A consumer sweeps the n x H granules of the previous step into its LDS stage and simply
// re-reads a granule until its tag is the step's tag: data and "ready" arrive together, there is no counter, no release fence
// (L2 write-back), no acquire (L1 invalidate), and nobody waits for workgroups whose data it already has.
// Tags are unique per (call, layer, step) -- the buffer is zeroed once per call -- and two slots alternate by step parity: a producer
// that writes slot s for step i + 2 has consumed every granule of step i + 1, which each workgroup publishes only after it is done
// reading step i from slot s.  The sweep is bounded in TIME (400 ms of the device's wall clock) and raises the same fault word as the
// barrier's spin when it expires.
// ---------------------------------------------------------------------------------------------------
constexpr unsigned long long kSweepBudgetTicks = 40000000ull;           // wall_clock64() runs at 100 MHz on gfx950: 400 ms
constexpr size_t kXchFwdBytes = (size_t)2 * kSeqMaxB * kChunk * 8;      // 2 slots x <= 32 rows x <= 512 units x 8 B = 256 KiB
constexpr size_t kXchBwdBytes = 5 * kXchFwdBytes;                       // the backward pass exchanges the 5 H gate gradients of every row

__device__ __forceinline__ void gran_store(unsigned long long *p, float v, unsigned tag)
{
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// rows b0 .. b0 + 7 x floats k0 .. k0 + 511 of the granule array g [rows][K] into a VStage (same element order as load_vstage);
// rows >= n_pub were not published by the previous step (initial state / sequences that start here): zeros.  Returns false on a time-out.
__device__ __forceinline__ bool load_vstage_gran(VStage &s, const unsigned long long *g, int n_pub, int b0, int k0, int K, int tid, unsigned tag,
                                                 const unsigned *broken)
{
    unsigned long long w[4][4];
    bool mine[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + kGemvThreads * i, b = b0 + (f >> 7), k = k0 + 4 * (f & 127);
        mine[i] = b < n_pub && k < K;                                     // K % 4 == 0 (persistent_ok)
        s.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    bool all_ok = false;
    // Bounded by TIME, not by a spin count: a workgroup whose producers are not co-resident yet -- chip-filling conv /
    // GEMM blocks of the other stream, a profiler -- may legitimately wait for many dispatch slots; 2^16 polls were 65-130 ms.
    // kSweepBudgetTicks of the 100 MHz wall clock = 400 ms, checked (with the sticky `broken` word) every 64 polls.
    const unsigned long long t_start = wall_clock64();
    for (unsigned spin = 0; !all_ok; ++spin) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!mine[i]) continue;
            const int f = tid + kGemvThreads * i, b = b0 + (f >> 7), k = k0 + 4 * (f & 127);
            const unsigned long long *q = g + (size_t)b * K + k;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[i][e] = __hip_atomic_load(q + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        all_ok = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!mine[i]) continue;
            bool ok = true;
#pragma unroll
            for (int e = 0; e < 4; ++e) ok = ok && (unsigned)(w[i][e] >> 32) == tag;
            if (ok) {
                s.v[i] = make_float4(__uint_as_float((unsigned)w[i][0]), __uint_as_float((unsigned)w[i][1]), __uint_as_float((unsigned)w[i][2]),
                                     __uint_as_float((unsigned)w[i][3]));
                mine[i] = false;
            } else {
                all_ok = false;
            }
        }
        if (!all_ok) {
            if ((spin & 63) == 63) {
                if (__hip_atomic_load(broken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
                if (wall_clock64() - t_start > kSweepBudgetTicks) return false;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return all_ok;
}

#ifndef MH_BAR_SLEEP
#define MH_BAR_SLEEP 8
#endif
constexpr int kBrokenWord = 48;   // counters: words 0..47 = one per layer, word 48 = "a barrier timed out"
// Where a timed-out grid barrier is reported: `host_word` is this device's word of a host-pinned, device-mapped array
// (fault_ctl below) -- the kernel stores 1 there and the NEXT mh_hwlstm_* / mh_hwcell_seq_* call (and mh_fault_pending)
// reads it on the host without synchronising anything.  `inflate` is the test hook of mh_debug_lstm_barrier_fault: it is
// added to every barrier target, so the barrier can never complete and the time-out path runs.
struct FaultCtl {
    unsigned *host_word;
    unsigned inflate;
};
// Grid-wide barrier for a kernel whose blocks are all resident or will become resident without waiting on this
// kernel (one block per 4 hidden units: <= 128 blocks).  Release / acquire at agent scope so that the plain stores
// before the barrier are visible to every XCD after it.  The spin is bounded: on a time-out the block raises the
// sticky `broken` word (every later barrier of the launch is skipped, so the device never hangs) AND the host-visible
// fault word; the kernel then poisons its outputs with NaN (poison_* below) and every later LSTM entry point returns
// MH_EFAULT until mh_fault_clear().  Nothing downstream can mistake the launch for a good one.
__device__ __forceinline__ void grid_barrier(unsigned *counters, int slot, unsigned target, const FaultCtl &fc)
{
    unsigned *counter = counters + slot;
    unsigned *broken = counters + kBrokenWord;    // sticky: set by the first block that times out
    target += fc.inflate;
    // every wave drains its own global stores first: the s_barrier of __syncthreads() does not wait for the other waves' stores
    // (round 6: the same idiom without this wait added up stale partial sums in the conv kernel, gpurun r06_c2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        // release: the block's stores (all waves: drained and ordered before by the barrier) become visible at agent scope
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        bool done = false;
        for (int spin = 0; spin < (1 << 18) && !done; ++spin) {
            done = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target ||
                   __hip_atomic_load(broken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (!done) __builtin_amdgcn_s_sleep(MH_BAR_SLEEP);
        }
        if (!done) {
            __hip_atomic_store(broken, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(fc.host_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);   // acquire (agent scope is HIP's default for this builtin)
    }
    __syncthreads();
}
// true (block-uniform) when any barrier of this launch timed out
__device__ __forceinline__ bool launch_broken(unsigned *counters)
{
    __shared__ unsigned flag;
    __syncthreads();
    if (threadIdx.x == 0) flag = __hip_atomic_load(counters + kBrokenWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return flag != 0;
}

// ---------------------------------------------------------------------------------------------------
// out[b][r] = sum_k v[b][k] * wt[r][k] (+ bias[r]);  4 output rows per block.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kGemvThreads) void gemv_rows_kernel(int n, int R, int K, const float *__restrict__ v,
                                                                int ldv, const float *__restrict__ wt, int ldw,
                                                                const float *__restrict__ bias,
                                                                float *__restrict__ out, int ldo, int wvec, int vvec)
{
    __shared__ __attribute__((aligned(16))) float vs[kNB * kChunk];
    __shared__ float red[4 * 4 * 1 * kNB];
    const int lane = threadIdx.x & 63, grp = lane >> 4;
    const int row = blockIdx.x * 4 + grp;
    const bool row_ok = row < R;
    const float *const wrow[1] = {wt + (size_t)(row_ok ? row : 0) * ldw};
    for (int b0 = 0; b0 < n; b0 += kNB) {
        float tot[1];
        block_gemv<1>(wrow, row_ok, wvec != 0, v, ldv, vvec != 0, n, b0, K, vs, red, tot);
        if (threadIdx.x < 4 * kNB) {
            const int r2 = blockIdx.x * 4 + (threadIdx.x >> 3), b = b0 + (threadIdx.x & 7);
            if (r2 < R && b < n) out[(size_t)b * ldo + r2] = tot[0] + (bias ? bias[r2] : 0.f);
        }
    }
}

// gate math of one (row, unit): elementWise_fp, highway_lstm_kernel.cu:108-160 (th = the 5 recurrent dot products)
__device__ __forceinline__ void cell_fwd_math(int H, int u, const float *pi, const float (&th)[5], const float *bias,
                                              const float *c_prev, const float *dropout, float *h_out, float *c_out,
                                              float *go, size_t o)
{
    float g[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        g[k] = pi[(size_t)k * H] + th[k];               // tmp_i + tmp_h
        if (bias) g[k] += bias[k * H + u];              // += bias
    }
    const float in_gate = sigmoidf_ref(g[0]);
    const float forget_gate = sigmoidf_ref(g[1]);
    const float act_gate = tanhf(g[2]);
    const float out_gate = sigmoidf_ref(g[3]);
    const float r_gate = sigmoidf_ref(g[4]);
    const float lin_gate = pi[(size_t)5 * H];
    float val = (forget_gate * c_prev[o]) + (in_gate * act_gate);
    c_out[o] = val;
    val = out_gate * tanhf(val);
    val = (float)((double)(val * r_gate) + (1. - (double)r_gate) * (double)lin_gate);
    if (dropout) val = val * dropout[o];
    h_out[o] = val;
    if (go) {
        go[0] = in_gate;
        go[(size_t)1 * H] = forget_gate;
        go[(size_t)2 * H] = act_gate;
        go[(size_t)3 * H] = out_gate;
        go[(size_t)4 * H] = r_gate;
        go[(size_t)5 * H] = lin_gate;
    }
}

// ---------------------------------------------------------------------------------------------------
// Fused highway-LSTM cell step (elementWise_fp + the recurrent Sgemm of highway_lstm_kernel.cu:453-485).
//   pre_i [n,6H] (ld_i)  : x_t*Wx (+ input bias for the decoder)
//   wh_t  [5H,H]         : recurrent weights, K(=H)-contiguous rows, row = gate*H + unit
//   bias  [5H] or NULL
// 4 hidden units per block, 5 gate rows each.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kGemvThreads) void hw_cell_fwd_kernel(int n, int H, const float *__restrict__ pre_i,
                                                                  int ld_i, const float *__restrict__ h_prev,
                                                                  const float *__restrict__ c_prev,
                                                                  const float *__restrict__ wh_t,
                                                                  const float *__restrict__ bias,
                                                                  const float *__restrict__ dropout,
                                                                  float *__restrict__ h_out,
                                                                  float *__restrict__ c_out,
                                                                  float *__restrict__ gates_out, int wvec, int vvec)
{
    __shared__ __attribute__((aligned(16))) float vs[kNB * kChunk];
    __shared__ float red[4 * 4 * 5 * kNB];
    const int lane = threadIdx.x & 63, grp = lane >> 4;
    const int ug = blockIdx.x * 4 + grp;
    const bool u_ok = ug < H;
    const int us = u_ok ? ug : 0;
    const float *const wrow[5] = {wh_t + ((size_t)0 * H + us) * H, wh_t + ((size_t)1 * H + us) * H,
                                  wh_t + ((size_t)2 * H + us) * H, wh_t + ((size_t)3 * H + us) * H,
                                  wh_t + ((size_t)4 * H + us) * H};
    for (int b0 = 0; b0 < n; b0 += kNB) {
        float th[5];
        block_gemv<5>(wrow, u_ok, wvec != 0, h_prev, H, vvec != 0, n, b0, H, vs, red, th);
        const int u = blockIdx.x * 4 + (threadIdx.x >> 3), row = b0 + (threadIdx.x & 7);
        if (threadIdx.x < 4 * kNB && u < H && row < n) {
            const size_t o = (size_t)row * H + u;
            cell_fwd_math(H, u, pre_i + (size_t)row * ld_i + u, th, bias, c_prev, dropout, h_out, c_out,
                          gates_out ? gates_out + (size_t)row * 6 * H + u : nullptr, o);
        }
    }
}

// elementWise_bp (highway_lstm_kernel.cu:46-104): d_h = (out_grad + h_rec_grad) * dropout ...
// d_gates [n,6H]: gates 0..4 are both the input- and the state-projection gradients, gate 5 = d_lin.
__global__ void hw_cell_bwd_kernel(int n, int H, const float *__restrict__ d_out, const float *__restrict__ d_h_rec,
                                   const float *__restrict__ d_c_out, const float *__restrict__ c_prev,
                                   const float *__restrict__ c_out, const float *__restrict__ gates,
                                   const float *__restrict__ dropout, float *__restrict__ d_gates,
                                   float *__restrict__ d_c_in)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * H) return;
    const int b = idx / H, u = idx % H;
    const float *gp = gates + (size_t)b * 6 * H + u;
    float d_h = d_out[idx] + (d_h_rec ? d_h_rec[idx] : 0.f);
    if (dropout) d_h = d_h * dropout[idx];
    const float in_gate = gp[0], forget_gate = gp[(size_t)H], act_gate = gp[(size_t)2 * H];
    const float out_gate = gp[(size_t)3 * H], r_gate = gp[(size_t)4 * H], lin_gate = gp[(size_t)5 * H];
    const float tc = tanhf(c_out[idx]);
    const float d_o = d_h * r_gate;
    const float d_c = d_o * out_gate * (1.f - tc * tc) + (d_c_out ? d_c_out[idx] : 0.f);
    const float h_prime = out_gate * tc;
    float *dg = d_gates + (size_t)b * 6 * H + u;
    dg[0] = d_c * act_gate * in_gate * (1.f - in_gate);
    dg[(size_t)H] = d_c * c_prev[idx] * forget_gate * (1.f - forget_gate);
    dg[(size_t)2 * H] = d_c * in_gate * (1.f - act_gate * act_gate);
    dg[(size_t)3 * H] = d_o * tc * out_gate * (1.f - out_gate);
    dg[(size_t)4 * H] = d_h * (h_prime - lin_gate) * r_gate * (1.f - r_gate);
    dg[(size_t)5 * H] = d_h * (1 - r_gate);
    d_c_in[idx] = forget_gate * d_c;
}

// ---------------------------------------------------------------------------------------------------
// Persistent layer kernels: ALL timesteps of one layer in one launch (instead of one launch per step).  A block owns
// 4 hidden units for the whole sequence, its slice of the recurrent weights stays in registers (40 floats per lane),
// and the steps are separated by a grid barrier -- the only cross-block dependency is h_t (forward) / the gate
// gradients of step t (backward).  Used when H <= 512, H % 4 == 0 and B <= 32; other shapes take the per-step path.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kGemvThreads) void hw_layer_fwd_kernel(SeqSched s, int H, const float *__restrict__ pre_i,
                                                                   float *hl, float *cl,
                                                                   const float *__restrict__ wh_t,
                                                                   const float *__restrict__ bias,
                                                                   const float *__restrict__ dropout, float *gates,
                                                                   unsigned *counters, int slot, FaultCtl fc,
                                                                   unsigned long long *xch, unsigned tag_base)
{
    // xch != nullptr: the hidden state travels as tagged granules (see gran_store / load_vstage_gran), no grid barrier;
    // xch == nullptr (MH_LSTM_GRAN=0, or no room in the workspace): every step ends with the grid barrier and reads h from `hl`
    __shared__ __attribute__((aligned(16))) float vs[kNB * kChunk];
    __shared__ float red[4 * 4 * 5 * kNB];
    __shared__ int timed_out;
    if (threadIdx.x == 0) timed_out = 0;
    const int lane = threadIdx.x & 63, grp = lane >> 4;
    const int ug = blockIdx.x * 4 + grp;
    const bool u_ok = ug < H;
    const int us = u_ok ? ug : 0;
    const float *const wrow[5] = {wh_t + ((size_t)0 * H + us) * H, wh_t + ((size_t)1 * H + us) * H,
                                  wh_t + ((size_t)2 * H + us) * H, wh_t + ((size_t)3 * H + us) * H,
                                  wh_t + ((size_t)4 * H + us) * H};
    WResident<5, 1> w;
    load_resident<5, 1>(w, wrow, u_ok, true, H);
    float bias_u[5] = {0.f, 0.f, 0.f, 0.f, 0.f};   // of the unit this thread does the gate math for
    {
        const int u = blockIdx.x * 4 + (threadIdx.x >> 3);
        if (bias && threadIdx.x < 4 * kNB && u < H)
#pragma unroll
            for (int k = 0; k < 5; ++k) bias_u[k] = bias[k * H + u];
    }
    unsigned epoch = 0;
    int n_pub = 0;                                   // rows the previous step published (0: the initial state is zero)
    for (int i = 0; i < s.T; ++i) {
        const int t = s.forward ? i : s.T - 1 - i;
        const StepRows rw = step_rows(s, t);
        const int n = rw.n;
        const float *h_prev = hl + rw.before * H, *c_prev = cl + rw.before * H;
        const unsigned long long *g_prev = xch ? xch + (size_t)((i + 1) & 1) * kSeqMaxB * H : nullptr;     // slot of step i - 1
        unsigned long long *g_out = xch ? xch + (size_t)(i & 1) * kSeqMaxB * H : nullptr;
        float *h_out = hl + rw.state * H, *c_out = cl + rw.state * H;
        const float *pi_t = pre_i + rw.io * 6 * H;
        float *g_t = gates ? gates + rw.io * 6 * H : nullptr;
        for (int b0 = 0; b0 < n; b0 += kNB) {
            // the gate-math threads fetch everything that does not depend on h_{t-1} BEFORE the recurrent product,
            // so those global-memory round trips overlap it
            const int u = blockIdx.x * 4 + (threadIdx.x >> 3), row = b0 + (threadIdx.x & 7);
            const bool mine = threadIdx.x < 4 * kNB && u < H && row < n;
            const size_t o = (size_t)row * H + u;
            float pv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, cp = 0.f, dm = 1.f;
            if (mine) {
                const float *pi = pi_t + (size_t)row * 6 * H + u;
#pragma unroll
                for (int k = 0; k < 6; ++k) pv[k] = pi[(size_t)k * H];
                cp = c_prev[o];
                if (dropout) dm = dropout[o];
            }
            float th[5];
            if (xch) {
                const unsigned tag = tag_base + (unsigned)i + fc.inflate;    // the tag step i - 1 published with (inflate: the test hook that makes it unreachable)
                block_gemv_resident_from<5, 1>(w, [&](VStage &st, int k0) {
                    if (!load_vstage_gran(st, g_prev, n_pub, b0, k0, H, threadIdx.x, tag, counters + kBrokenWord)) timed_out = 1;
                }, H, vs, red, th);
            } else {
                block_gemv_resident<5, 1>(w, h_prev, H, true, n, b0, H, vs, red, th);
            }
            if (mine) {
                float g[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) g[k] = pv[k] + th[k] + bias_u[k];   // (tmp_i + tmp_h) + bias
                const float in_gate = sigmoidf_ref(g[0]);
                const float forget_gate = sigmoidf_ref(g[1]);
                const float act_gate = tanhf(g[2]);
                const float out_gate = sigmoidf_ref(g[3]);
                const float r_gate = sigmoidf_ref(g[4]);
                const float lin_gate = pv[5];
                float val = (forget_gate * cp) + (in_gate * act_gate);
                c_out[o] = val;
                val = out_gate * tanhf(val);
                val = (float)((double)(val * r_gate) + (1. - (double)r_gate) * (double)lin_gate);
                if (dropout) val = val * dm;
                h_out[o] = val;
                if (g_out) gran_store(g_out + o, val, tag_base + (unsigned)i + 1u);
                if (g_t) {
                    float *go = g_t + (size_t)row * 6 * H + u;
                    go[0] = in_gate;
                    go[(size_t)1 * H] = forget_gate;
                    go[(size_t)2 * H] = act_gate;
                    go[(size_t)3 * H] = out_gate;
                    go[(size_t)4 * H] = r_gate;
                    go[(size_t)5 * H] = lin_gate;
                }
            }
        }
        n_pub = n;
        if (xch) {
            __syncthreads();                        // vs / red are reused by the next step; `timed_out` is complete
            if (timed_out) {                        // bounded sweep ran out: raise the sticky word + the host-visible fault, stop
                if (threadIdx.x == 0) {
                    __hip_atomic_store(counters + kBrokenWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(fc.host_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                break;
            }
        } else if (i + 1 < s.T) {
            grid_barrier(counters, slot, ++epoch * gridDim.x, fc);
        }
    }
    __syncthreads();
    // a timed-out barrier / sweep means some step read a stale h_{t-1}: make the whole layer output unmistakably invalid
    if (timed_out || launch_broken(counters)) {
        const float nan = __builtin_nanf("");
        for (int i = 0; i < s.T; ++i) {
            const StepRows rw = step_rows(s, i);
            for (int e = threadIdx.x; e < 4 * rw.n; e += blockDim.x) {
                const int u = blockIdx.x * 4 + (e & 3);
                if (u < H) hl[rw.state * H + (size_t)(e >> 2) * H + u] = nan;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Greedy label decoder (lib/lstm/decoder_rnn.py:205-227; eval arg-max feedback, and training steps whose GT label is
// background): the WHOLE decode in one persistent launch.  Per step:
//   (A) gate math of the block's 4 hidden units; the input projection is enc_proj[row] + emb_proj[prev label + 1] (the
//       embedding-row gather of the label chosen at the previous step), the state projection is the resident-weight GEMV
//   -- grid barrier --  (h_t complete)
//   (B) the class logits: block j owns classes j, j + gridDim.x, ...; one wave per (class, row) dot product of length H;
//       every non-background logit is folded into the row's 64-bit arg-max key with atomicMax
//       (order-preserving fp32 image in the high word, 2^32-1-class in the low word: ties go to the LOWER class, which is
//       what the reference's max(1)[1] returns)
//   -- grid barrier --  (keys complete) -> every block decodes the winner for its next step's gather.
// Outputs: the states h/c, the logits of every row (= self.out(h), no second pass in eval), the label fed at each row
// (`fed`, embedding index = label + 1, 0 = 'start') and the committed label of each row.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long logit_key(float v, int cls)
{
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)cls);
}
__device__ __forceinline__ int key_class(unsigned long long k) { return (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull)); }

__global__ __launch_bounds__(kGemvThreads) void hw_decoder_greedy_kernel(
    SeqSched s, int H, int C, const float *__restrict__ enc_proj, const float *__restrict__ emb_proj,
    const float *__restrict__ wh_t, const float *__restrict__ bias, const float *__restrict__ dropout,
    const float *__restrict__ w_out, const float *__restrict__ b_out, const long long *__restrict__ labels, float *hl,
    float *cl, float *logits, long long *fed, long long *commits, unsigned long long *keys, unsigned *counters, FaultCtl fc)
{
    __shared__ __attribute__((aligned(16))) float vs[kNB * kChunk];
    __shared__ float red[4 * 4 * 5 * kNB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane >> 4;
    const int ug = blockIdx.x * 4 + grp;
    const bool u_ok = ug < H;
    const int us = u_ok ? ug : 0;
    const float *const wrow[5] = {wh_t + ((size_t)0 * H + us) * H, wh_t + ((size_t)1 * H + us) * H,
                                  wh_t + ((size_t)2 * H + us) * H, wh_t + ((size_t)3 * H + us) * H,
                                  wh_t + ((size_t)4 * H + us) * H};
    WResident<5, 1> w;
    load_resident<5, 1>(w, wrow, u_ok, true, H);
    float bias_u[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    {
        const int u = blockIdx.x * 4 + (threadIdx.x >> 3);
        if (bias && threadIdx.x < 4 * kNB && u < H)
#pragma unroll
            for (int k = 0; k < 5; ++k) bias_u[k] = bias[k * H + u];
    }
    // label committed at row `r` of the previous step (rows r < n of step t belong to the same sequences as rows r of
    // step t-1: sequences are sorted by decreasing length)
    auto committed = [&](size_t row) -> long long {
        const unsigned long long k = __hip_atomic_load(keys + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long best = key_class(k);
        if (labels) {
            const long long l = labels[row];
            return l != 0 ? l : best;
        }
        return best;
    };
    unsigned epoch = 0;
    StepRows prev_rw = step_rows(s, 0);
    for (int i = 0; i < s.T; ++i) {
        const StepRows rw = step_rows(s, i);
        const int n = rw.n;
        const float *h_prev = hl + rw.before * H, *c_prev = cl + rw.before * H;
        float *h_out = hl + rw.state * H, *c_out = cl + rw.state * H;
        for (int b0 = 0; b0 < n; b0 += kNB) {
            const int u = blockIdx.x * 4 + (threadIdx.x >> 3), row = b0 + (threadIdx.x & 7);
            const bool mine = threadIdx.x < 4 * kNB && u < H && row < n;
            const size_t o = (size_t)row * H + u;
            float pv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, cp = 0.f, dm = 1.f;
            if (mine) {
                const long long p = (i == 0) ? 0 : committed(prev_rw.io + row) + 1;
                if (blockIdx.x == 0 && (threadIdx.x >> 3) == 0) fed[rw.io + row] = p;
                const float *pi = enc_proj + (rw.io + row) * 6 * H + u;
                const float *pe = emb_proj + (size_t)p * 6 * H + u;
#pragma unroll
                for (int k = 0; k < 6; ++k) pv[k] = pi[(size_t)k * H] + pe[(size_t)k * H];
                cp = c_prev[o];
                if (dropout) dm = dropout[o];
            }
            float th[5];
            block_gemv_resident<5, 1>(w, h_prev, H, true, n, b0, H, vs, red, th);
            if (mine) {
                float g[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) g[k] = pv[k] + th[k] + bias_u[k];
                const float in_gate = sigmoidf_ref(g[0]);
                const float forget_gate = sigmoidf_ref(g[1]);
                const float act_gate = tanhf(g[2]);
                const float out_gate = sigmoidf_ref(g[3]);
                const float r_gate = sigmoidf_ref(g[4]);
                const float lin_gate = pv[5];
                float val = (forget_gate * cp) + (in_gate * act_gate);
                c_out[o] = val;
                val = out_gate * tanhf(val);
                val = (float)((double)(val * r_gate) + (1. - (double)r_gate) * (double)lin_gate);
                if (dropout) val = val * dm;
                h_out[o] = val;
            }
        }
        grid_barrier(counters, 0, ++epoch * gridDim.x, fc);
        // (B) logits of this step: (class, row) pairs of this block round-robin over its 4 waves
        const int ncls = (C > (int)blockIdx.x) ? (C - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
        for (int q = wave; q < ncls * n; q += 4) {
            const int c = (int)blockIdx.x + (q / n) * (int)gridDim.x, b = q % n;
            const float *wr = w_out + (size_t)c * H, *hv = h_out + (size_t)b * H;
            float acc = 0.f;
            for (int k = 4 * lane; k < H; k += 256) {
                const float4 a = *reinterpret_cast<const float4 *>(wr + k), x = *reinterpret_cast<const float4 *>(hv + k);
                acc = fmaf(a.x, x.x, acc); acc = fmaf(a.y, x.y, acc); acc = fmaf(a.z, x.z, acc); acc = fmaf(a.w, x.w, acc);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
            if (lane == 0) {
                const float v = acc + (b_out ? b_out[c] : 0.f);
                logits[(rw.io + b) * C + c] = v;
                if (c >= 1) atomicMax(keys + rw.io + b, logit_key(v, c));
            }
        }
        grid_barrier(counters, 0, ++epoch * gridDim.x, fc);
        if (blockIdx.x == 0 && (int)threadIdx.x < n) commits[rw.io + threadIdx.x] = committed(rw.io + threadIdx.x);
        prev_rw = rw;
    }
    if (launch_broken(counters)) {
        const float nan = __builtin_nanf("");
        for (int i = 0; i < s.T; ++i) {
            const StepRows rw = step_rows(s, i);
            for (int e = threadIdx.x; e < 4 * rw.n; e += blockDim.x) {
                const int u = blockIdx.x * 4 + (e & 3);
                if (u < H) hl[rw.state * H + (size_t)(e >> 2) * H + u] = nan;
            }
            if (blockIdx.x == 0)
                for (int e = threadIdx.x; e < rw.n * C; e += blockDim.x) logits[rw.io * C + e] = nan;
        }
    }
}

// backward of one layer: per step  (A) elementWise_bp for the block's own units -> d_gates[t], c_grad[t+1];
// grid barrier;  (B) h_grad[t+1][:, own units] = d_gates[t][:, :5H] * Wh[own units, :]^T  (K = 5H, weights resident).
// (A) of the next step only needs the block's own h_grad / c_grad entries, so one barrier per step suffices.
__global__ __launch_bounds__(kGemvThreads) void hw_layer_bwd_kernel(SeqSched s, int H, const float *__restrict__ grad_in,
                                                                   float *h_grad, float *c_grad,
                                                                   const float *__restrict__ cl,
                                                                   const float *__restrict__ gates,
                                                                   const float *__restrict__ dropout, float *dg_all,
                                                                   const float *__restrict__ wh, unsigned *counters, int slot,
                                                                   FaultCtl fc, unsigned long long *xch, unsigned tag_base)
{
    // xch != nullptr: the gate gradients of a step travel as tagged granules ([slot][row][5 H], see hw_layer_fwd_kernel), no grid barrier
    __shared__ __attribute__((aligned(16))) float vs[kNB * kChunk];
    __shared__ float red[4 * 4 * 1 * kNB];
    __shared__ int timed_out;
    if (threadIdx.x == 0) timed_out = 0;
    const int lane = threadIdx.x & 63, grp = lane >> 4;
    const int ug = blockIdx.x * 4 + grp;
    const bool u_ok = ug < H;
    const float *const wrow[1] = {wh + (size_t)(u_ok ? ug : 0) * 5 * H};
    WResident<1, 5> w;
    load_resident<1, 5>(w, wrow, u_ok, true, 5 * H);
    unsigned epoch = 0;
    // inputs of phase (A) that do not depend on the recurrence, fetched one step ahead (they overlap phase (B))
    struct StepIn {
        float g[6], d_out, c_o, c_p, drop;
    };
    auto time_of = [&](int i) { return s.forward ? s.T - 1 - i : i; };
    auto fetch = [&](int i, StepIn &x) {
        const int t = time_of(i);
        const StepRows rw = step_rows(s, t);
        if (threadIdx.x >= 4 * rw.n) return;
        const int b = threadIdx.x >> 2, u = blockIdx.x * 4 + (threadIdx.x & 3);
        if (u >= H) return;
        const size_t idx = (size_t)b * H + u;
        const float *gp = gates + (rw.io + b) * 6 * H + u;
#pragma unroll
        for (int k = 0; k < 6; ++k) x.g[k] = gp[(size_t)k * H];
        x.d_out = grad_in[rw.io * H + idx];
        x.c_o = cl[rw.state * H + idx];
        x.c_p = cl[rw.before * H + idx];
        x.drop = dropout ? dropout[idx] : 1.f;
    };
    StepIn in;
    fetch(0, in);
    for (int i = 0; i < s.T; ++i) {
        // the backward pass walks time in the opposite order of the layer's forward pass: the gradients arriving
        // from the step processed just before sit in the rows of the state "after" t
        const int t = time_of(i);
        const StepRows rw = step_rows(s, t);
        const int n = rw.n;
        float *dg = dg_all + rw.io * 6 * H;
        if (threadIdx.x < 4 * n) {
            const int b = threadIdx.x >> 2, u = blockIdx.x * 4 + (threadIdx.x & 3);
            if (u < H) {
                const size_t idx = (size_t)b * H + u;
                const bool has_rec = b < rw.n_after;
                float d_h = in.d_out + (has_rec ? h_grad[rw.after * H + idx] : 0.f);
                if (dropout) d_h = d_h * in.drop;
                const float in_gate = in.g[0], forget_gate = in.g[1], act_gate = in.g[2];
                const float out_gate = in.g[3], r_gate = in.g[4], lin_gate = in.g[5];
                const float tc = tanhf(in.c_o);
                const float d_o = d_h * r_gate;
                const float d_c = d_o * out_gate * (1.f - tc * tc) + (has_rec ? c_grad[rw.after * H + idx] : 0.f);
                const float h_prime = out_gate * tc;
                float *dgp = dg + (size_t)b * 6 * H + u;
                const float d0 = d_c * act_gate * in_gate * (1.f - in_gate);
                const float d1 = d_c * in.c_p * forget_gate * (1.f - forget_gate);
                const float d2 = d_c * in_gate * (1.f - act_gate * act_gate);
                const float d3 = d_o * tc * out_gate * (1.f - out_gate);
                const float d4 = d_h * (h_prime - lin_gate) * r_gate * (1.f - r_gate);
                dgp[0] = d0;
                dgp[(size_t)H] = d1;
                dgp[(size_t)2 * H] = d2;
                dgp[(size_t)3 * H] = d3;
                dgp[(size_t)4 * H] = d4;
                dgp[(size_t)5 * H] = d_h * (1 - r_gate);
                c_grad[rw.state * H + idx] = forget_gate * d_c;
                if (xch) {
                    unsigned long long *gp = xch + ((size_t)(i & 1) * kSeqMaxB + b) * 5 * H + u;
                    const unsigned tg = tag_base + (unsigned)i + 1u;
                    gran_store(gp, d0, tg);
                    gran_store(gp + (size_t)H, d1, tg);
                    gran_store(gp + (size_t)2 * H, d2, tg);
                    gran_store(gp + (size_t)3 * H, d3, tg);
                    gran_store(gp + (size_t)4 * H, d4, tg);
                }
            }
        }
        if (!xch) grid_barrier(counters, slot, ++epoch * gridDim.x, fc);
        if (i + 1 < s.T) fetch(i + 1, in);
        for (int b0 = 0; b0 < n; b0 += kNB) {
            float tot[1];
            if (xch) {
                const unsigned long long *g_cur = xch + (size_t)(i & 1) * kSeqMaxB * 5 * H;
                const unsigned tag = tag_base + (unsigned)i + 1u + fc.inflate;
                block_gemv_resident_from<1, 5>(w, [&](VStage &st, int k0) {
                    if (!load_vstage_gran(st, g_cur, n, b0, k0, 5 * H, threadIdx.x, tag, counters + kBrokenWord)) timed_out = 1;
                }, 5 * H, vs, red, tot);
            } else {
                block_gemv_resident<1, 5>(w, dg, 6 * H, true, n, b0, 5 * H, vs, red, tot);
            }
            const int u = blockIdx.x * 4 + (threadIdx.x >> 3), row = b0 + (threadIdx.x & 7);
            if (threadIdx.x < 4 * kNB && u < H && row < n) h_grad[rw.state * H + (size_t)row * H + u] = tot[0];
        }
        if (xch) {
            __syncthreads();                        // this step's h_grad (other threads of the block read it next step); vs / red free
            if (timed_out) {
                if (threadIdx.x == 0) {
                    __hip_atomic_store(counters + kBrokenWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(fc.host_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                break;
            }
        }
    }
    __syncthreads();
    // see hw_layer_fwd_kernel: after a timed-out barrier the gate gradients (what the caller's dgrad / wgrad GEMMs read)
    // and the state gradients of this block's units become NaN
    if (timed_out || launch_broken(counters)) {
        const float nan = __builtin_nanf("");
        for (int i = 0; i < s.T; ++i) {
            const StepRows rw = step_rows(s, i);
            for (int e = threadIdx.x; e < 4 * rw.n; e += blockDim.x) {
                const int u = blockIdx.x * 4 + (e & 3);
                if (u >= H) continue;
                float *dgp = dg_all + (rw.io + (e >> 2)) * 6 * H + u;
#pragma unroll
                for (int k = 0; k < 6; ++k) dgp[(size_t)k * H] = nan;
                h_grad[rw.state * H + (size_t)(e >> 2) * H + u] = nan;
            }
        }
    }
}

// out[c] += sum_r in[r][c]   (bias gradient; the reference uses Sgemv with a ones vector, :342-355)
__global__ __launch_bounds__(256) void colsum_accum_kernel(const float *__restrict__ in, int rows, int cols, int ld,
                                                           float *__restrict__ out)
{
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rp = threadIdx.x >> 6;
    float s = 0.f;
    if (c < cols)
        for (int r = rp; r < rows; r += 4) s += in[(size_t)r * ld + c];
    part[rp][threadIdx.x & 63] = s;
    __syncthreads();
    if (rp == 0 && c < cols) out[c] += (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// [rows, cols] -> [cols, rows]
__global__ __launch_bounds__(256) void transpose2d_kernel(const float *__restrict__ src, int rows, int cols,
                                                          float *__restrict__ dst)
{
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < rows && c0 + tx < cols) tile[j][tx] = src[(size_t)(r0 + j) * cols + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < cols && r0 + tx < rows) dst[(size_t)(c0 + j) * rows + r0 + tx] = tile[tx][j];
}

struct LayerOffsets {
    size_t wx, wh;
    int in_size;
};

static LayerOffsets layer_offsets(int in_size, int H, int layer)
{
    // alternating_highway_lstm.py:213-229: per layer Wx[in_l,6H] then Wh[H,5H]
    LayerOffsets o;
    size_t w = 0;
    int ins = in_size;
    for (int l = 0; l <= layer; ++l) {
        ins = (l == 0) ? in_size : H;
        o.wx = w;
        o.wh = w + (size_t)6 * H * ins;
        w = o.wh + (size_t)5 * H * H;
    }
    o.in_size = ins;
    return o;
}

static bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- device-side fault reporting (see FaultCtl) -------------------------------------------------------------------
// One host-pinned, device-mapped word per device, allocated once per process on first use (64 words = one 256-B line).
constexpr int kMaxFaultDevices = 64;
static unsigned *g_fault_words = nullptr;   // host address == device address (hipHostMallocMapped | Portable, coherent)
static int g_force_barrier_fault = 0;       // mh_debug_lstm_barrier_fault
static unsigned *fault_words()
{
    if (!g_fault_words) {
        void *p = nullptr;
        if (hipHostMalloc(&p, kMaxFaultDevices * sizeof(unsigned), hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess)
            return nullptr;
        for (int i = 0; i < kMaxFaultDevices; ++i) reinterpret_cast<unsigned *>(p)[i] = 0u;
        g_fault_words = reinterpret_cast<unsigned *>(p);
    }
    return g_fault_words;
}
// rc != MH_OK: the allocation failed, or an earlier persistent launch on this device timed out (MH_EFAULT)
static int fault_ctl(FaultCtl &fc)
{
    unsigned *w = fault_words();
    int dev = 0;
    if (!w || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxFaultDevices) {
        set_last_error("fault word (hipHostMalloc / hipGetDevice)", hipErrorOutOfMemory);
        return (int)hipErrorOutOfMemory;
    }
    if (__atomic_load_n(&w[dev], __ATOMIC_RELAXED) != 0u) {
        set_last_error("an earlier persistent LSTM launch on this device timed out in its grid barrier: every result since "
                       "then is invalid (mh_fault_clear() re-arms the entry points)", hipErrorLaunchFailure);
        return MH_EFAULT;
    }
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, &w[dev], 0) != hipSuccess) {
        set_last_error("hipHostGetDevicePointer(fault word)", hipErrorInvalidValue);
        return (int)hipErrorInvalidValue;
    }
    fc.host_word = reinterpret_cast<unsigned *>(dp);
    fc.inflate = g_force_barrier_fault ? 1u : 0u;
    return MH_OK;
}

// Tags of the granule exchange are unique per (layer, step) of a call: layer l uses tag_base = 1 + kTagStride * l and step i the
// tag tag_base + i (+ the fault hook's inflate, 1 at most), so the stride must exceed the longest sequence (+ 2)
constexpr int kSeqMaxT = 4096;
constexpr unsigned kTagStride = 2u * kSeqMaxT;
static_assert(kTagStride > (unsigned)kSeqMaxT + 2u, "granule tags of neighbouring layers would alias");
constexpr size_t kCounterBytes = 256;   // grid-barrier counters of the persistent layer kernels (one per layer)
constexpr size_t kGemmCtrBytes = (size_t)kGemmCounters * sizeof(int);   // split-K arrival counters of the projections' small products
                                                                        // (gemm.hip: zero on entry, left zero), behind the barrier counters
// MH_LSTM_GRAN=0: the forward layers exchange h through `hl` + a grid barrier per step (rounds 2-3) instead of tagged granules (A/B)
static bool gran_enabled()
{
    static const bool on = [] { const char *e = getenv("MH_LSTM_GRAN"); return !(e && e[0] == '0'); }();
    return on;
}
static bool persistent_ok(int H, int B, int L)
{
    return H <= kChunk && H % 4 == 0 && B <= kSeqMaxB && L <= kBrokenWord;
}
static SeqSched make_sched(const int *lengths, int T, int B, bool forward_dir)
{
    SeqSched s;
    s.T = T;
    s.B = B;
    s.forward = forward_dir ? 1 : 0;
    s.packed = 0;
    for (int b = 0; b < kSeqMaxB; ++b) s.lengths[b] = (b < B) ? lengths[b] : 0;
    return s;
}

static int launch_cell_fwd(int n, int H, const float *pre_i, int ld_i, const float *h_prev, const float *c_prev,
                           const float *wh_t, const float *bias, const float *dropout, float *h_out, float *c_out,
                           float *gates_out, hipStream_t st)
{
    const int wvec = al16(wh_t) && (H % 4 == 0);
    const int vvec = al16(h_prev) && (H % 4 == 0);
    hipLaunchKernelGGL(hw_cell_fwd_kernel, dim3(ceil_div(H, 4)), dim3(kGemvThreads), 0, st, n, H, pre_i, ld_i, h_prev,
                       c_prev, wh_t, bias, dropout, h_out, c_out, gates_out, wvec, vvec);
    return check_launch("hw_cell_fwd_kernel");
}

static int launch_gemv_rows(int n, int R, int K, const float *v, int ldv, const float *wt, int ldw, const float *bias,
                            float *out, int ldo, hipStream_t st)
{
    const int wvec = al16(wt) && (ldw % 4 == 0);
    const int vvec = al16(v) && (ldv % 4 == 0);
    hipLaunchKernelGGL(gemv_rows_kernel, dim3(ceil_div(R, 4)), dim3(kGemvThreads), 0, st, n, R, K, v, ldv, wt, ldw, bias,
                       out, ldo, wvec, vvec);
    return check_launch("gemv_rows_kernel");
}

// (t, numCovered) in the order the reference visits the timesteps (highway_lstm_kernel.cu:410-424)
static void covered_schedule(const int *lengths, int T, int B, bool forward_dir, int *ts, int *ns)
{
    if (forward_dir) {
        int n = B;
        for (int t = 0; t < T; ++t) {
            while (n > 0 && lengths[n - 1] <= t) --n;
            ts[t] = t;
            ns[t] = n;
        }
    } else {
        int n = 0;
        for (int i = 0, t = T - 1; t >= 0; --t, ++i) {
            while (n < B && lengths[n] > t) ++n;
            ts[i] = t;
            ns[i] = n;
        }
    }
}

}  // namespace mh

using namespace mh;

#define MH_TRY(expr)          \
    do {                      \
        int rc__ = (expr);    \
        if (rc__) return rc__; \
    } while (0)

extern "C" {

int mh_gemv_rows(int n, int R, int K, const float *v, int ldv, const float *wt, int ldw, const float *bias, float *out,
                 int ldo, void *stream)
{
    MH_REQUIRE(n >= 0 && R >= 0 && K > 0);
    if (n == 0 || R == 0) return MH_OK;
    MH_REQUIRE(v && wt && out && ldv >= K && ldw >= K && ldo >= R);
    return launch_gemv_rows(n, R, K, v, ldv, wt, ldw, bias, out, ldo, as_stream(stream));
}

int mh_hwlstm_cell_fwd(int n, int H, const float *pre_i, const float *h_prev, const float *c_prev, const float *wh_t,
                       const float *bias_h, const float *dropout, float *h_out, float *c_out, float *gates_out,
                       void *stream)
{
    MH_REQUIRE(n >= 0 && H > 0);
    if (n == 0) return MH_OK;
    MH_REQUIRE(pre_i && h_prev && c_prev && wh_t && h_out && c_out);
    return launch_cell_fwd(n, H, pre_i, 6 * H, h_prev, c_prev, wh_t, bias_h, dropout, h_out, c_out, gates_out,
                           as_stream(stream));
}

int mh_hwlstm_cell_bwd(int n, int H, const float *d_h, const float *d_c_out, const float *c_prev, const float *c_out,
                       const float *gates, const float *dropout, float *d_gates, float *d_c_in, void *stream)
{
    MH_REQUIRE(n >= 0 && H > 0);
    if (n == 0) return MH_OK;
    MH_REQUIRE(d_h && c_prev && c_out && gates && d_gates && d_c_in);
    hipLaunchKernelGGL(hw_cell_bwd_kernel, dim3(ceil_div(n * H, 256)), dim3(256), 0, as_stream(stream), n, H, d_h,
                       (const float *)nullptr, d_c_out, c_prev, c_out, gates, dropout, d_gates, d_c_in);
    return check_launch("hw_cell_bwd_kernel");
}

// workspace layout (forward): tmp_i_all [T*B,6H] | wh_t [5H,H] | gemm split-K scratch
size_t mh_hwlstm_fwd_ws_bytes(int in_size, int H, int B, int L, int T)
{
    (void)L;
    size_t s = align_up((size_t)T * B * 6 * H * sizeof(float), 256) + align_up((size_t)5 * H * H * sizeof(float), 256);
    s += std::max(mh_gemm_ws_bytes(T * B, 6 * H, in_size, 0), mh_gemm_ws_bytes(T * B, 6 * H, H, 0));
    return s + kCounterBytes + kGemmCtrBytes + kXchFwdBytes + 256;
}

int mh_hwlstm_fwd(int in_size, int H, int B, int L, int T, const float *x, const int *lengths_host, float *h_data,
                  float *c_data, const float *weight, const float *bias, const float *dropout, float *gates,
                  int is_training, void *workspace, size_t ws_bytes, void *stream)
{
    MH_REQUIRE(in_size > 0 && H > 0 && B > 0 && L > 0 && T > 0);
    MH_REQUIRE(x && lengths_host && h_data && c_data && weight && bias && dropout && workspace);
    MH_REQUIRE(!is_training || gates);
    MH_REQUIRE(ws_bytes >= mh_hwlstm_fwd_ws_bytes(in_size, H, B, L, T));
    MH_REQUIRE(T <= kSeqMaxT);
    for (int b = 0; b < B; ++b) {
        MH_REQUIRE(lengths_host[b] >= 1 && lengths_host[b] <= T);
        MH_REQUIRE(b == 0 || lengths_host[b] <= lengths_host[b - 1]);
    }
    MH_REQUIRE(lengths_host[0] == T);
    FaultCtl fc;
    MH_TRY(fault_ctl(fc));
    hipStream_t st = as_stream(stream);
    char *ws = reinterpret_cast<char *>(workspace);
    float *tmp_i = reinterpret_cast<float *>(ws);
    ws += align_up((size_t)T * B * 6 * H * sizeof(float), 256);
    float *wh_t = reinterpret_cast<float *>(ws);
    ws += align_up((size_t)5 * H * H * sizeof(float), 256);
    const size_t numEl = (size_t)B * H;
    int ts[kSeqMaxT], ns[kSeqMaxT];
    unsigned *counters = reinterpret_cast<unsigned *>(ws);   // one barrier counter per layer
    ws += kCounterBytes;
    int *gemm_ctr = reinterpret_cast<int *>(ws);
    ws += kGemmCtrBytes;
    unsigned long long *xch = reinterpret_cast<unsigned long long *>(ws);     // granule exchange of the forward layers (behind the counters)
    ws += kXchFwdBytes;
    const bool persistent = persistent_ok(H, B, L) && al16(h_data) && al16(wh_t);
    const bool gran = persistent && gran_enabled();
    {   // ONE memset: barrier counters | GEMM arrival counters | (granule exchange)
        hipError_t e = hipMemsetAsync(counters, 0, kCounterBytes + kGemmCtrBytes + (gran ? kXchFwdBytes : 0), st);
        if (e != hipSuccess) return (int)e;
    }
    void *gws = ws;
    const size_t gws_bytes = ws_bytes - (size_t)(ws - reinterpret_cast<char *>(workspace));

    for (int layer = 0; layer < L; ++layer) {
        const LayerOffsets o = layer_offsets(in_size, H, layer);
        const bool fwd_dir = (layer % 2 == 0);
        const float *inp = (layer == 0) ? x : h_data + ((size_t)(layer - 1) * (T + 1) + 1) * numEl;
        // tmp_i[T*B, 6H] = inp[T*B, in] * Wx[in, 6H]
        MH_TRY(gemm_f32_ctr(0, 0, T * B, 6 * H, o.in_size, inp, o.in_size, weight + o.wx, 6 * H, tmp_i, 6 * H, nullptr,
                            MH_EPI_NONE, 0, gws, gws_bytes, gemm_ctr, kGemmCounters, stream));
        // wh_t[5H,H] = Wh[H,5H]^T so that every output column's K weights are contiguous
        hipLaunchKernelGGL(transpose2d_kernel, dim3(ceil_div(5 * H, 32), ceil_div(H, 32)), dim3(256), 0, st,
                           weight + o.wh, H, 5 * H, wh_t);
        MH_TRY(check_launch("transpose2d_kernel"));
        covered_schedule(lengths_host, T, B, fwd_dir, ts, ns);
        float *hl = h_data + (size_t)layer * (T + 1) * numEl;
        float *cl = c_data + (size_t)layer * (T + 1) * numEl;
        if (persistent) {
            SeqSched sched = make_sched(lengths_host, T, B, fwd_dir);
            hipLaunchKernelGGL(hw_layer_fwd_kernel, dim3(ceil_div(H, 4)), dim3(kGemvThreads), 0, st, sched, H, tmp_i, hl, cl,
                               wh_t, bias + (size_t)5 * H * layer, dropout + (size_t)layer * numEl,
                               is_training ? gates + (size_t)layer * T * 6 * numEl : nullptr, counters, layer, fc,
                               gran ? xch : nullptr, 1u + kTagStride * (unsigned)layer);
            MH_TRY(check_launch("hw_layer_fwd_kernel"));
            continue;
        }
        for (int i = 0; i < T; ++i) {
            const int t = ts[i], n = ns[i];
            if (n == 0) continue;
            const int prev = fwd_dir ? t : (t + 2) % (T + 1);
            MH_TRY(launch_cell_fwd(n, H, tmp_i + (size_t)t * B * 6 * H, 6 * H, hl + (size_t)prev * numEl,
                                   cl + (size_t)prev * numEl, wh_t, bias + (size_t)5 * H * layer,
                                   dropout + (size_t)layer * numEl, hl + (size_t)(t + 1) * numEl,
                                   cl + (size_t)(t + 1) * numEl,
                                   is_training ? gates + ((size_t)layer * T + t) * 6 * numEl : nullptr, st));
        }
    }
    return MH_OK;
}

// ---------------------------------------------------------------------------------------------------
// One highway-LSTM layer over a PACKED (PackedSequence, time-major) batch in one launch: the decoder recurrence of
// lib/lstm/decoder_rnn.py:151-215 under teacher forcing (inputs known up front).  batch_sizes_host[t] = rows of step
// t (non-increasing), N = their sum.  h_buf / c_buf have B + N rows: the first B rows are the initial state (the
// caller zeroes them), row B + r is the state of packed row r.  Same arithmetic as mh_hwlstm_cell_fwd step by step.
// Supported shapes: H <= 512, H % 4 == 0, B <= 32, 16-byte aligned buffers (MH_EINVAL otherwise: use the cell calls).
// ---------------------------------------------------------------------------------------------------
static int sched_from_batch_sizes(const int *batch_sizes, int T, int B, SeqSched &s)
{
    MH_REQUIRE(batch_sizes && T > 0 && B > 0 && B <= kSeqMaxB && batch_sizes[0] == B);
    for (int t = 1; t < T; ++t) MH_REQUIRE(batch_sizes[t] >= 1 && batch_sizes[t] <= batch_sizes[t - 1]);
    s.T = T;
    s.B = B;
    s.forward = 1;
    s.packed = 1;
    for (int b = 0; b < kSeqMaxB; ++b) {
        int len = 0;
        for (int t = 0; t < T; ++t) len += (batch_sizes[t] > b) ? 1 : 0;
        s.lengths[b] = len;
    }
    return MH_OK;
}

size_t mh_hwcell_seq_ws_bytes(void) { return kCounterBytes + kXchBwdBytes; }      // forward needs a fifth of it

int mh_hwcell_seq_fwd(int H, int B, int T, const int *batch_sizes_host, const float *pre_i, const float *w_state,
                      const float *b_state, const float *dropout, float *h_buf, float *c_buf, float *gates,
                      void *workspace, size_t ws_bytes, void *stream)
{
    MH_REQUIRE(H > 0 && pre_i && w_state && h_buf && c_buf && workspace && ws_bytes >= kCounterBytes);
    MH_REQUIRE(persistent_ok(H, B, 1) && al16(w_state) && al16(h_buf) && al16(workspace));
    SeqSched sched;
    MH_TRY(sched_from_batch_sizes(batch_sizes_host, T, B, sched));
    FaultCtl fc;
    MH_TRY(fault_ctl(fc));
    hipStream_t st = as_stream(stream);
    const bool gran = gran_enabled() && ws_bytes >= kCounterBytes + kXchFwdBytes;
    hipError_t e = hipMemsetAsync(workspace, 0, kCounterBytes + (gran ? kXchFwdBytes : 0), st);
    if (e != hipSuccess) return (int)e;
    unsigned long long *xch = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(workspace) + kCounterBytes);
    hipLaunchKernelGGL(hw_layer_fwd_kernel, dim3(ceil_div(H, 4)), dim3(kGemvThreads), 0, st, sched, H, pre_i, h_buf, c_buf,
                       w_state, b_state, dropout, gates, reinterpret_cast<unsigned *>(workspace), 0, fc, gran ? xch : nullptr, 1u);
    return check_launch("hw_layer_fwd_kernel");
}

// backward of mh_hwcell_seq_fwd: d_pre [N,6H] (gates 0..4: gradient of input AND state projections, gate 5: d_lin)
// from dh_all [N,H].  w_state_t = the state weight transposed to [H,5H]; hgrad_buf / cgrad_buf: scratch, B + N rows.
int mh_hwcell_seq_bwd(int H, int B, int T, const int *batch_sizes_host, const float *dh_all, const float *c_buf,
                      const float *gates, const float *dropout, const float *w_state_t, float *d_pre, float *hgrad_buf,
                      float *cgrad_buf, void *workspace, size_t ws_bytes, void *stream)
{
    MH_REQUIRE(H > 0 && dh_all && c_buf && gates && w_state_t && d_pre && hgrad_buf && cgrad_buf && workspace);
    MH_REQUIRE(ws_bytes >= kCounterBytes && persistent_ok(H, B, 1) && al16(w_state_t) && al16(d_pre) && al16(workspace));
    SeqSched sched;
    MH_TRY(sched_from_batch_sizes(batch_sizes_host, T, B, sched));
    FaultCtl fc;
    MH_TRY(fault_ctl(fc));
    hipStream_t st = as_stream(stream);
    const bool gran = gran_enabled() && ws_bytes >= kCounterBytes + kXchBwdBytes;
    hipError_t e = hipMemsetAsync(workspace, 0, kCounterBytes + (gran ? kXchBwdBytes : 0), st);
    if (e != hipSuccess) return (int)e;
    unsigned long long *xch = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(workspace) + kCounterBytes);
    hipLaunchKernelGGL(hw_layer_bwd_kernel, dim3(ceil_div(H, 4)), dim3(kGemvThreads), 0, st, sched, H, dh_all, hgrad_buf,
                       cgrad_buf, c_buf, gates, dropout, d_pre, w_state_t, reinterpret_cast<unsigned *>(workspace), 0, fc,
                       gran ? xch : nullptr, 1u);
    return check_launch("hw_layer_bwd_kernel");
}

// The greedy decoder in one launch (hw_decoder_greedy_kernel).  Workspace: barrier counters + one 64-bit arg-max key per row.
size_t mh_decoder_greedy_ws_bytes(int N) { return kCounterBytes + align_up((size_t)std::max(N, 1) * sizeof(unsigned long long), 256); }

int mh_decoder_greedy(int H, int B, int T, const int *batch_sizes_host, int C, const float *enc_proj, const float *emb_proj,
                      const float *w_state, const float *b_state, const float *dropout, const float *w_out, const float *b_out,
                      const long long *labels, float *h_buf, float *c_buf, float *logits, long long *fed, long long *commits,
                      void *workspace, size_t ws_bytes, void *stream)
{
    MH_REQUIRE(H > 0 && C > 1 && enc_proj && emb_proj && w_state && w_out && h_buf && c_buf && logits && fed && commits && workspace);
    MH_REQUIRE(persistent_ok(H, B, 1) && al16(w_state) && al16(h_buf) && al16(workspace) && al16(w_out));
    SeqSched sched;
    MH_TRY(sched_from_batch_sizes(batch_sizes_host, T, B, sched));
    long long N = 0;
    for (int t = 0; t < T; ++t) N += batch_sizes_host[t];
    MH_REQUIRE(ws_bytes >= mh_decoder_greedy_ws_bytes((int)N));
    FaultCtl fc;
    MH_TRY(fault_ctl(fc));
    hipStream_t st = as_stream(stream);
    hipError_t e = hipMemsetAsync(workspace, 0, mh_decoder_greedy_ws_bytes((int)N), st);
    if (e != hipSuccess) return (int)e;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(workspace) + kCounterBytes);
    hipLaunchKernelGGL(hw_decoder_greedy_kernel, dim3(ceil_div(H, 4)), dim3(kGemvThreads), 0, st, sched, H, C, enc_proj, emb_proj,
                       w_state, b_state, dropout, w_out, b_out, labels, h_buf, c_buf, logits, fed, commits, keys,
                       reinterpret_cast<unsigned *>(workspace), fc);
    return check_launch("hw_decoder_greedy_kernel");
}

// workspace layout (backward): d_gates_all [T*B,6H] | h_grad [T+1,B,H] | c_grad [T+1,B,H] | below_grad x2 [T,B,H]
//                              | gemm split-K scratch
size_t mh_hwlstm_bwd_ws_bytes(int in_size, int H, int B, int L, int T)
{
    (void)L;
    const size_t numEl = (size_t)B * H;
    size_t s = align_up((size_t)T * B * 6 * H * sizeof(float), 256);
    s += 2 * align_up((size_t)(T + 1) * numEl * sizeof(float), 256);
    s += 2 * align_up((size_t)T * numEl * sizeof(float), 256);
    s += kCounterBytes + kGemmCtrBytes + kXchBwdBytes;
    size_t g = 0;
    const int ins[2] = {in_size, H};
    for (int i = 0; i < 2; ++i) {
        g = std::max(g, mh_gemm_ws_bytes(T * B, ins[i], 6 * H, 0));
        g = std::max(g, mh_gemm_ws_bytes(ins[i], 6 * H, T * B, 0));
    }
    g = std::max(g, mh_gemm_ws_bytes(H, 5 * H, T * B, 0));
    return s + g + 256;
}

int mh_hwlstm_bwd(int in_size, int H, int B, int L, int T, const float *out_grad, const int *lengths_host,
                  const float *x, const float *h_data, const float *c_data, const float *weight, const float *gates,
                  const float *dropout, float *x_grad, float *weight_grad, float *bias_grad, int do_weight_grad,
                  void *workspace, size_t ws_bytes, void *stream)
{
    MH_REQUIRE(in_size > 0 && H > 0 && B > 0 && L > 0 && T > 0 && T <= kSeqMaxT);
    MH_REQUIRE(out_grad && lengths_host && x && h_data && c_data && weight && gates && dropout && x_grad && workspace);
    MH_REQUIRE(!do_weight_grad || (weight_grad && bias_grad));
    MH_REQUIRE(ws_bytes >= mh_hwlstm_bwd_ws_bytes(in_size, H, B, L, T));
    for (int b = 0; b < B; ++b) {
        MH_REQUIRE(lengths_host[b] >= 1 && lengths_host[b] <= T);
        MH_REQUIRE(b == 0 || lengths_host[b] <= lengths_host[b - 1]);
    }
    FaultCtl fc;
    MH_TRY(fault_ctl(fc));
    hipStream_t st = as_stream(stream);
    const size_t numEl = (size_t)B * H;
    char *ws = reinterpret_cast<char *>(workspace);
    float *dg_all = reinterpret_cast<float *>(ws);
    ws += align_up((size_t)T * B * 6 * H * sizeof(float), 256);
    float *h_grad = reinterpret_cast<float *>(ws);
    ws += align_up((size_t)(T + 1) * numEl * sizeof(float), 256);
    float *c_grad = reinterpret_cast<float *>(ws);
    ws += align_up((size_t)(T + 1) * numEl * sizeof(float), 256);
    float *below[2];
    below[0] = reinterpret_cast<float *>(ws);
    ws += align_up((size_t)T * numEl * sizeof(float), 256);
    below[1] = reinterpret_cast<float *>(ws);
    ws += align_up((size_t)T * numEl * sizeof(float), 256);
    unsigned *counters = reinterpret_cast<unsigned *>(ws);
    ws += kCounterBytes;
    int *gemm_ctr = reinterpret_cast<int *>(ws);
    ws += kGemmCtrBytes;
    unsigned long long *xch = reinterpret_cast<unsigned long long *>(ws);     // granule exchange of the backward layers
    ws += kXchBwdBytes;
    const bool persistent = persistent_ok(H, B, L) && al16(weight) && al16(dg_all);
    const bool gran = persistent && gran_enabled();
    {   // ONE memset: barrier counters | GEMM arrival counters | (granule exchange)
        hipError_t e = hipMemsetAsync(counters, 0, kCounterBytes + kGemmCtrBytes + (gran ? kXchBwdBytes : 0), st);
        if (e != hipSuccess) return (int)e;
    }
    void *gws = ws;
    const size_t gws_bytes = ws_bytes - (size_t)(ws - reinterpret_cast<char *>(workspace));
    int ts[kSeqMaxT], ns[kSeqMaxT];

    const float *grad_in = out_grad;  // gradient arriving at the current layer's outputs [T,B,H]
    for (int layer = L - 1; layer >= 0; --layer) {
        const LayerOffsets o = layer_offsets(in_size, H, layer);
        const bool fwd_dir = (layer % 2 == 0);
        // dg_all | h_grad | c_grad are adjacent in the workspace: one memset instead of three
        hipError_t e = hipMemsetAsync(dg_all, 0, (size_t)(reinterpret_cast<char *>(below[0]) - reinterpret_cast<char *>(dg_all)), st);
        if (e != hipSuccess) return (int)e;
        const float *hl = h_data + (size_t)layer * (T + 1) * numEl;
        const float *cl = c_data + (size_t)layer * (T + 1) * numEl;
        // the backward pass walks time in the opposite order of the forward pass (:199-231)
        covered_schedule(lengths_host, T, B, !fwd_dir ? true : false, ts, ns);
        // (for a forward-direction layer the backward visits t = T-1..0 with n growing, which is exactly the
        //  schedule of a backward-direction forward pass, and vice versa)
        if (persistent && (o.wh % 4 == 0)) {
            SeqSched sched = make_sched(lengths_host, T, B, fwd_dir);
            hipLaunchKernelGGL(hw_layer_bwd_kernel, dim3(ceil_div(H, 4)), dim3(kGemvThreads), 0, st, sched, H, grad_in,
                               h_grad, c_grad, cl, gates + (size_t)layer * T * 6 * numEl,
                               dropout + (size_t)layer * numEl, dg_all, weight + o.wh, counters, layer, fc, gran ? xch : nullptr,
                               1u + kTagStride * (unsigned)layer);
            MH_TRY(check_launch("hw_layer_bwd_kernel"));
        } else
        for (int i = 0; i < T; ++i) {
            const int t = ts[i], n = ns[i];
            if (n == 0) continue;
            const int prev_grad = fwd_dir ? (t + 2) % (T + 1) : t;
            const int prev = fwd_dir ? t : (t + 2) % (T + 1);
            float *dg = dg_all + (size_t)t * B * 6 * H;
            hipLaunchKernelGGL(hw_cell_bwd_kernel, dim3(ceil_div(n * H, 256)), dim3(256), 0, st, n, H,
                               grad_in + (size_t)t * numEl, h_grad + (size_t)prev_grad * numEl,
                               c_grad + (size_t)prev_grad * numEl, cl + (size_t)prev * numEl,
                               cl + (size_t)(t + 1) * numEl, gates + ((size_t)layer * T + t) * 6 * numEl,
                               dropout + (size_t)layer * numEl, dg, c_grad + (size_t)(t + 1) * numEl);
            MH_TRY(check_launch("hw_cell_bwd_kernel"));
            // h_grad[t+1][:n] = dg[:n,:5H] * Wh^T      (Wh [H,5H]: row k is contiguous over the 5H gate columns)
            MH_TRY(launch_gemv_rows(n, H, 5 * H, dg, 6 * H, weight + o.wh, 5 * H, nullptr,
                                    h_grad + (size_t)(t + 1) * numEl, H, st));
        }
        const float *inp = (layer == 0) ? x : h_data + ((size_t)(layer - 1) * (T + 1) + 1) * numEl;
        float *inp_grad = (layer == 0) ? x_grad : below[layer & 1];
        // d(inp)[T*B, in] = dg_all[T*B, 6H] * Wx[in, 6H]^T
        MH_TRY(gemm_f32_ctr(0, 1, T * B, o.in_size, 6 * H, dg_all, 6 * H, weight + o.wx, 6 * H, inp_grad, o.in_size,
                            nullptr, MH_EPI_NONE, 0, gws, gws_bytes, gemm_ctr, kGemmCounters, stream));
        if (do_weight_grad) {
            // dWx[in, 6H] += inp[T*B, in]^T * dg_all[T*B, 6H]
            // (every region of weight_grad is written exactly once per call: plain stores, the caller need not zero 64 MB first)
            MH_TRY(gemm_f32_ctr(1, 0, o.in_size, 6 * H, T * B, inp, o.in_size, dg_all, 6 * H, weight_grad + o.wx, 6 * H,
                                nullptr, MH_EPI_NONE, 0, gws, gws_bytes, gemm_ctr, kGemmCounters, stream));
            // dWh[H, 5H] += h_prev^T * dg_all[:, :5H];  h_prev(t) = slot t (forward layers) or slot t+2
            // (backward layers; t = T-1 reads the all-zero slot 0 and contributes nothing)
            if (fwd_dir) {
                MH_TRY(gemm_f32_ctr(1, 0, H, 5 * H, T * B, hl, H, dg_all, 6 * H, weight_grad + o.wh, 5 * H, nullptr,
                                    MH_EPI_NONE, 0, gws, gws_bytes, gemm_ctr, kGemmCounters, stream));
            } else if (T > 1) {
                MH_TRY(gemm_f32_ctr(1, 0, H, 5 * H, (T - 1) * B, hl + 2 * numEl, H, dg_all, 6 * H, weight_grad + o.wh,
                                    5 * H, nullptr, MH_EPI_NONE, 0, gws, gws_bytes, gemm_ctr, kGemmCounters, stream));
            } else {                    // a one-step backward-direction layer has no previous state: dWh = 0
                hipError_t ez = hipMemsetAsync(weight_grad + o.wh, 0, (size_t)H * 5 * H * sizeof(float), st);
                if (ez != hipSuccess) return (int)ez;
            }
            hipLaunchKernelGGL(colsum_accum_kernel, dim3(ceil_div(5 * H, 64)), dim3(256), 0, st, dg_all, T * B, 5 * H,
                               6 * H, bias_grad + (size_t)5 * H * layer);
            MH_TRY(check_launch("colsum_accum_kernel"));
        }
        grad_in = inp_grad;
    }
    return MH_OK;
}


// ---- fault state of the persistent (grid-barrier) kernels --------------------------------------------------------
int mh_fault_pending(void)
{
    const unsigned *w = g_fault_words;
    if (!w) return 0;
    int n = 0;
    for (int i = 0; i < kMaxFaultDevices; ++i) n += (__atomic_load_n(&w[i], __ATOMIC_RELAXED) != 0u) ? 1 : 0;
    return n;
}

int mh_fault_clear(void)
{
    unsigned *w = g_fault_words;
    if (w)
        for (int i = 0; i < kMaxFaultDevices; ++i) __atomic_store_n(&w[i], 0u, __ATOMIC_RELAXED);
    return MH_OK;
}

int mh_debug_lstm_barrier_fault(int enable)
{
    g_force_barrier_fault = enable ? 1 : 0;
    return MH_OK;
}

}  // extern "C"
