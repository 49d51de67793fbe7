// exact_ops.hip -- the operators whose results must be BIT-EXACT against the CPU oracle:
// NMS keep lists, RoIAlign (border tests decide zero fill), union-box masks, fp32 IoU.
// This file is compiled with -ffp-contract=off (see build.py): every fp32 expression is evaluated
// exactly as written, one IEEE rounding per operation, like oracle/native_ops.c.
//
// gfx950 notes: wavefront = 64, so one wave covers a whole 64-box NMS tile and the per-row
// suppression words are native 64-bit lane values; cross-lane traffic uses readlane/ballot, not LDS.
#include <algorithm>
#include <cstdlib>
#include <atomic>

#include "common.h"

namespace mh {

static thread_local char g_err[256] = "";
void set_last_error(const char *what, hipError_t e)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}

// =====================================================================================
// NMS
// =====================================================================================
// IoU with the +1 pixel convention, same operation order as the reference's devIoU
// (lib/fpn/nms/src/cuda/nms_kernel.cu:23-31).
__device__ __forceinline__ float iou_p1(const float4 a, const float4 b)
{
    float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
    float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
    float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
    float interS = width * height;
    float Sa = (a.z - a.x + 1) * (a.w - a.y + 1);
    float Sb = (b.z - b.x + 1) * (b.w - b.y + 1);
    return interS / (Sa + Sb - interS);
}

// One wave per 64x64 tile of the (row box, column box) IoU matrix; only tiles on or above the
// diagonal are needed by the sweep.  Lane i owns row box i and produces its 64-bit word.
// grid = (col_blocks, row_blocks, nseg).  mask row stride = cb_stride words.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4 *__restrict__ boxes,
                                                      const int *__restrict__ seg_offsets, int n_single,
                                                      float thresh, unsigned long long *__restrict__ mask,
                                                      int cb_stride)
{
    const int seg = blockIdx.z;
    const int base = seg_offsets ? seg_offsets[seg] : 0;
    const int n = seg_offsets ? seg_offsets[seg + 1] - base : n_single;
    const int row_blk = blockIdx.y, col_blk = blockIdx.x;
    if (col_blk < row_blk) return;
    if (row_blk * 64 >= n || col_blk * 64 >= n) return;
    const int lane = threadIdx.x;
    const int col_size = min(n - col_blk * 64, 64);
    const int row = row_blk * 64 + lane;

    // the column boxes are read through the scalar cache (wave-uniform address: s_load_dwordx4, eight in flight), the row box
    // lives in the lane.  Round 3 kept the column boxes one per lane and broadcast them with four v_readlane per column.
    float4 rbox = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < n) rbox = boxes[base + row];
    const float4 *__restrict__ cbox = boxes + base + col_blk * 64;

    unsigned long long t = 0;
    const int start = (row_blk == col_blk) ? lane + 1 : 0;
#pragma unroll 8
    for (int j = 0; j < col_size; ++j) {
        const float4 cb = cbox[j];
        const unsigned long long hit = iou_p1(rbox, cb) > thresh;      // evaluated for every column: no branch around the read
        t |= (hit & (unsigned long long)(j >= start)) << j;
    }
    if (row < n) mask[(size_t)(base + row) * cb_stride + col_blk] = t;
}

// Sequential greedy sweep, one 256-thread block per segment (the chain over block-rows is serial by definition: whether box i
// survives depends on every kept box before it; /root/reference lib/fpn/nms/src/cuda/nms_kernel.cu:96-131 does it on the host).
// What bounds it is the latency of the mask reads, so none of them may depend on the chain:
//   * the diagonal words of kSweepStage block-rows are staged in LDS at a time;
//   * the words of block-row r right of the diagonal are fetched by ALL 256 threads into registers one block-row ahead
//     (thread = row group tid>>4 x column lane tid&15: rows rg + 16p, columns jb + cl + 16t; 128 contiguous bytes per 16
//     lanes), whether or not the row is going to be kept -- the whole upper triangle streams through once, 2.3 MB at 6000 boxes;
//   * wave 0 resolves the 64 boxes of the block-row from the diagonal words by walking only the boxes still alive
//     (s_ff1 over a scalar word + v_readlane), the block ORs the kept rows' registers into `removed` (ds_or_b64).
// MH_NMS_SWEEP=chain (read once per process) selects round 3's kernel below, one dependent global read per kept row and
// column: the A/B of tools/r04/nms_time.py.  Index model of this kernel without a GPU: tests/test_nms_sweep_model.py.
constexpr int kSweepStage = 32;     // block-rows whose diagonal words are in LDS at a time (16 KiB)
constexpr int kSweepP = 4;          // row passes: 4 x 16 row groups = the 64 rows of a block-row
constexpr int kSweepT = 6;          // column steps: 6 x 16 lanes = 96 columns per chunk

struct SweepRegs {
    unsigned long long w[kSweepP][kSweepT];
};

__device__ __forceinline__ void sweep_fetch(SweepRegs &s, const unsigned long long *__restrict__ mask, int base, int n,
                                            int cb_stride, int col_blocks, int r, int jb, int rg, int cl)
{
    // every read is issued unconditionally (24 in flight per thread); a row or column beyond the segment reads the last valid
    // one instead (written by nms_mask_kernel: the last column is on or right of every row's diagonal).  Such a row is never
    // kept and such a column is never folded (sweep_fold), so the values need no zeroing here -- a select on the loaded value
    // makes the compiler branch around each read and wait for it inside the branch.
#pragma unroll
    for (int p = 0; p < kSweepP; ++p) {
        const int row = min(r * 64 + rg + 16 * p, n - 1);
        const unsigned long long *src = mask + (size_t)(base + row) * cb_stride;
#pragma unroll
        for (int t = 0; t < kSweepT; ++t) s.w[p][t] = src[min(jb + cl + 16 * t, col_blocks - 1)];
    }
}

__device__ __forceinline__ void sweep_fold(const SweepRegs &s, unsigned long long kept, unsigned long long *removed, int jb,
                                           int col_blocks, int rg, int cl)
{
#pragma unroll
    for (int t = 0; t < kSweepT; ++t) {
        unsigned long long acc = 0ULL;
#pragma unroll
        for (int p = 0; p < kSweepP; ++p) acc |= ((kept >> (rg + 16 * p)) & 1ULL) ? s.w[p][t] : 0ULL;
        const int j = jb + cl + 16 * t;
        if (acc != 0ULL && j < col_blocks) atomicOr(removed + j, acc);
    }
}

__global__ __launch_bounds__(256) void nms_sweep_kernel(const unsigned long long *__restrict__ mask,
                                                        const int *__restrict__ seg_offsets, int n_single,
                                                        int cb_stride, int *__restrict__ keep,
                                                        int *__restrict__ num_keep)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long smem[];  // removed[cb_stride] | kept word | diag stage
    const int seg = blockIdx.x;
    const int base = seg_offsets ? seg_offsets[seg] : 0;
    const int n = seg_offsets ? seg_offsets[seg + 1] - base : n_single;
    const int col_blocks = (n + 63) / 64;
    unsigned long long *removed = smem;
    unsigned long long *kept_word = smem + cb_stride;
    unsigned long long *diag = smem + cb_stride + 2;              // [kSweepStage][64]
    const int tid = threadIdx.x, rg = tid >> 4, cl = tid & 15;
    for (int j = tid; j < col_blocks; j += 256) removed[j] = 0ULL;

    SweepRegs regs;
    if (col_blocks > 0) sweep_fetch(regs, mask, base, n, cb_stride, col_blocks, 0, 1, rg, cl);
    int total = 0;  // meaningful in wave 0 only (uniform)
    for (int r = 0; r < col_blocks; ++r) {
        if (r % kSweepStage == 0) {
            unsigned long long dv[kSweepStage / 4];
#pragma unroll
            for (int s = 0; s < kSweepStage / 4; ++s) {           // all reads first, then the LDS writes
                const int row = min(r * 64 + tid + 256 * s, n - 1);
                dv[s] = mask[(size_t)(base + row) * cb_stride + (row >> 6)];
            }
#pragma unroll
            for (int s = 0; s < kSweepStage / 4; ++s) diag[tid + 256 * s] = (r * 64 + tid + 256 * s < n) ? dv[s] : 0ULL;
            __syncthreads();
        }
        if (tid < 64) {
            const int lane = tid;
            const int rows = min(n - r * 64, 64);
            const unsigned long long d = diag[(r % kSweepStage) * 64 + lane];
            const unsigned dlo = (unsigned)d, dhi = (unsigned)(d >> 32);
            unsigned long long alive_v = ~removed[r];
            if (rows < 64) alive_v &= (1ULL << rows) - 1ULL;
            unsigned long long alive = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(alive_v >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)alive_v);
            unsigned long long kept = 0ULL;
            while (alive) {                                       // uniform: one trip per KEPT box of this block-row
                const int i = __builtin_ctzll(alive);
                kept |= 1ULL << i;
                const unsigned lo = __builtin_amdgcn_readlane(dlo, i), hi = __builtin_amdgcn_readlane(dhi, i);
                alive &= ~((((unsigned long long)hi << 32) | lo) | (1ULL << i));     // row i's word holds bits j > i only
            }
            if ((kept >> lane) & 1ULL) {
                const int rank = __popcll(kept & ((1ULL << lane) - 1ULL));
                keep[base + total + rank] = r * 64 + lane;
            }
            total += __popcll(kept);
            if (lane == 0) kept_word[0] = kept;
        }
        __syncthreads();
        const unsigned long long kept = kept_word[0];
        // `regs` holds the chunk (r, r + 1); the chunk fetched after each fold is the next one in the order the sweep needs
        // them: the next 96 columns of this block-row (segments beyond 6208 boxes), else the first chunk of block-row r + 1
        int jb = r + 1;
        do {
            if (kept) sweep_fold(regs, kept, removed, jb, col_blocks, rg, cl);
            jb += 16 * kSweepT;
            if (kept && jb < col_blocks) sweep_fetch(regs, mask, base, n, cb_stride, col_blocks, r, jb, rg, cl);
            else if (r + 1 < col_blocks) sweep_fetch(regs, mask, base, n, cb_stride, col_blocks, r + 1, r + 2, rg, cl);
        } while (kept && jb < col_blocks);
        __syncthreads();                                          // removed[r + 1] is final; kept_word may be rewritten
    }
    if (tid == 0) num_keep[seg] = total;
}
static size_t sweep_lds_bytes(int cb) { return (size_t)(cb + 2 + kSweepStage * 64) * 8; }

// Round 3's sweep (MH_NMS_SWEEP=chain): wave 0 resolves a block-row with a 64-trip loop, then each thread ORs the kept rows'
// words of its columns one dependent global read after the other.
__global__ __launch_bounds__(256) void nms_sweep_chain_kernel(const unsigned long long *__restrict__ mask,
                                                              const int *__restrict__ seg_offsets, int n_single,
                                                              int cb_stride, int *__restrict__ keep,
                                                              int *__restrict__ num_keep)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long smem[];  // [cb_stride] removed + 2 words
    const int seg = blockIdx.x;
    const int base = seg_offsets ? seg_offsets[seg] : 0;
    const int n = seg_offsets ? seg_offsets[seg + 1] - base : n_single;
    const int col_blocks = (n + 63) / 64;
    unsigned long long *removed = smem;
    unsigned long long *kept_word = smem + cb_stride;  // [0] = kept bits of the current block-row
    const int tid = threadIdx.x;
    for (int j = tid; j < col_blocks; j += blockDim.x) removed[j] = 0ULL;
    __syncthreads();

    int total = 0;  // meaningful in wave 0 only (uniform)
    for (int r = 0; r < col_blocks; ++r) {
        const int rows = min(n - r * 64, 64);
        if (tid < 64) {
            const int lane = tid;
            unsigned long long diag = 0ULL;
            if (lane < rows) diag = mask[(size_t)(base + r * 64 + lane) * cb_stride + r];
            unsigned long long alive = ~removed[r];
            if (rows < 64) alive &= (1ULL << rows) - 1ULL;
            unsigned long long kept = 0ULL;
            const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
            for (int i = 0; i < rows; ++i) {  // uniform trip count
                if ((alive >> i) & 1ULL) {
                    kept |= 1ULL << i;
                    unsigned lo = __builtin_amdgcn_readlane(dlo, i);
                    unsigned hi = __builtin_amdgcn_readlane(dhi, i);
                    alive &= ~(((unsigned long long)hi << 32) | lo);
                }
            }
            if ((kept >> lane) & 1ULL) {
                int rank = __popcll(kept & ((1ULL << lane) - 1ULL));
                keep[base + total + rank] = r * 64 + lane;
            }
            total += __popcll(kept);
            if (lane == 0) kept_word[0] = kept;
        }
        __syncthreads();
        const unsigned long long kept = kept_word[0];
        // removed[j] |= OR_{i kept} mask[r*64+i][j]   for j > r  (j == r is already final)
        for (int j = r + 1 + tid; j < col_blocks; j += blockDim.x) {
            unsigned long long acc = 0ULL;
            unsigned long long bits = kept;
            while (bits) {
                int i = __ffsll((long long)bits) - 1;
                bits &= bits - 1;
                acc |= mask[(size_t)(base + r * 64 + i) * cb_stride + j];
            }
            removed[j] |= acc;
        }
        __syncthreads();
    }
    if (tid == 0) num_keep[seg] = total;
}

static bool sweep_chain() { static const bool v = [] { const char *e = getenv("MH_NMS_SWEEP"); return e && e[0] == 'c'; }(); return v; }
static int launch_sweep(const unsigned long long *mask, const int *seg_offsets, int nseg, int n_single, int cb, int *keep,
                        int *num_keep, hipStream_t st)
{
    if (sweep_chain()) {
        hipLaunchKernelGGL(nms_sweep_chain_kernel, dim3(nseg), dim3(256), (size_t)(cb + 2) * 8, st, mask, seg_offsets, n_single, cb,
                           keep, num_keep);
        return check_launch("nms_sweep_chain_kernel");
    }
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(nseg), dim3(256), sweep_lds_bytes(cb), st, mask, seg_offsets, n_single, cb, keep,
                       num_keep);
    return check_launch("nms_sweep_kernel");
}

// =====================================================================================
// RoIAlign
// =====================================================================================
struct RoiGeom {
    int b_in;
    float x1, y1, x2, y2;  // normalised
};

// The Python wrapper's normalisation (lib/fpn/roi_align/functions/roi_align.py:25-32): the divisor
// (H-1)/spatial_scale is computed in double on the host and rounded to fp32 (passed in as inv args).
__device__ __forceinline__ RoiGeom load_roi(const float *rois, int n, float width, float height)
{
    RoiGeom g;
    const float *r = rois + 5 * n;
    g.b_in = (int)r[0];
    g.x1 = r[1] / width;
    g.y1 = r[2] / height;
    g.x2 = r[3] / width;
    g.y2 = r[4] / height;
    return g;
}

struct Sample {
    int top, bottom, left, right;  // -1 in `top` marks "outside -> extrapolation value 0"
    float y_lerp, x_lerp;
};

// lib/fpn/roi_align/src/cuda/roi_align_kernel.cu:36-66, same expression order.
__device__ __forceinline__ Sample make_sample(const RoiGeom &g, int y, int x, int H, int W, int ph, int pw)
{
    Sample s;
    s.top = -1;
    s.bottom = s.left = s.right = 0;
    s.y_lerp = s.x_lerp = 0.f;
    const float height_scale = (ph > 1) ? (g.y2 - g.y1) * (H - 1) / (ph - 1) : 0;
    const float width_scale = (pw > 1) ? (g.x2 - g.x1) * (W - 1) / (pw - 1) : 0;
    const float in_y = (ph > 1) ? g.y1 * (H - 1) + y * height_scale
                                : (float)(0.5 * (g.y1 + g.y2) * (H - 1));
    if (in_y < 0 || in_y > H - 1) return s;
    const float in_x = (pw > 1) ? g.x1 * (W - 1) + x * width_scale
                                : (float)(0.5 * (g.x1 + g.x2) * (W - 1));
    if (in_x < 0 || in_x > W - 1) return s;
    s.top = (int)floorf(in_y);
    s.bottom = (int)ceilf(in_y);
    s.y_lerp = in_y - s.top;
    s.left = (int)floorf(in_x);
    s.right = (int)ceilf(in_x);
    s.x_lerp = in_x - s.left;
    return s;
}

__device__ __forceinline__ float bilerp(float tl, float tr, float bl, float br, const Sample &s)
{
    const float top = tl + (tr - tl) * s.x_lerp;
    const float bottom = bl + (br - bl) * s.x_lerp;
    return top + (bottom - top) * s.y_lerp;
}

// NCHW features: one thread per output element (x fastest), the reference's own decomposition.
__global__ void roi_align_fwd_nchw(const float *__restrict__ feat, const float *__restrict__ rois, int n_rois,
                                   int B, int C, int H, int W, int ph, int pw, float width, float height,
                                   float *__restrict__ out)
{
    const long long total = (long long)n_rois * C * ph * pw;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        long long t = idx;
        const int x = t % pw; t /= pw;
        const int y = t % ph; t /= ph;
        const int d = t % C;
        const int n = t / C;
        const RoiGeom g = load_roi(rois, n, width, height);
        float v = 0.f;
        if (g.b_in >= 0 && g.b_in < B) {
            const Sample s = make_sample(g, y, x, H, W, ph, pw);
            if (s.top >= 0) {
                const float *plane = feat + ((size_t)g.b_in * C + d) * H * W;
                v = bilerp(plane[s.top * W + s.left], plane[s.top * W + s.right],
                           plane[s.bottom * W + s.left], plane[s.bottom * W + s.right], s);
            }
        }
        out[idx] = v;
    }
}

// NHWC features: block = (roi, 64-channel chunk).  A coalesced HBM gather with an LDS transposition:
//   * thread = (channel quad cq = tid & 15, bin group bg = tid >> 4): 16 lanes read one corner's 64 channels as 16-byte
//     vectors (256 contiguous bytes), and every thread ISSUES ALL ITS LOADS FIRST -- up to 4 bins x 4 corners = 16
//     independent 16-byte loads in flight -- before the first bilinear blend (round 2 issued four dependent-latency scalar
//     loads per bin and ran at 0.83 TB/s of output, latency-bound: profiles/r02_bench_n1_kernel_stats.csv);
//   * the [64][bins] result tile is transposed through LDS so that the [n][c][y][x] output run (64 * bins floats,
//     contiguous and 16-byte aligned) leaves the block as 16-byte stores.
// Arithmetic per element is unchanged (bilerp, same expression order): outputs are bit-identical to the reference kernel.
constexpr int kRoiCh = 64;
constexpr int kRoiBinsPerThread = 4;      // bins <= 16 * 4 take the fast path (7x7 = 49 does); larger grids loop
template <bool VEC>
__global__ __launch_bounds__(256) void roi_align_fwd_nhwc(const float *__restrict__ feat,
                                                          const float *__restrict__ rois, int B, int C, int H,
                                                          int W, int ph, int pw, float width, float height,
                                                          float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [bins] samples (6 words) + tile
    // 1-D grid, CHANNEL CHUNK FASTEST: consecutive block ids go round-robin over the 8 XCDs, so with C = 512 (8 chunks) every
    // block of chunk c runs on XCD c, whose 4 MB L2 then holds that chunk of the whole feature map (6 x 37 x 37 x 64 x 4 B =
    // 2.1 MB) -- the 4x over-read of the bilinear gather is served by L2 instead of the fabric (r03_c6: 0.96 TB/s of output
    // with (roi, chunk) blocks spread over all XCDs, each re-fetching the map from the Infinity Cache)
    const int nchunks = (C + kRoiCh - 1) / kRoiCh;
    const int n = blockIdx.x / nchunks;
    const int c0 = (blockIdx.x % nchunks) * kRoiCh;
    const int bins = ph * pw;
    Sample *samp = reinterpret_cast<Sample *>(lds);
    // staging tile, BIN-major [bins][kRoiCh + 1]: a thread's four channels of one bin are four consecutive words -- banks
    // (bg + 4 cq + k) % 64 on the way in, consecutive bins one bank apart on the way out: conflict-free both ways.
    // NOTE (round 3): this file is compiled WITHOUT packed FP32 VALU instructions (csrc/build.py).  With them, the
    // SLP-vectorised interpolation below returned wrong low halves in lanes 48-63 whenever MFMA waves of another HIP stream
    // shared the SIMD (profiles/r03_packed_f32_coresidency.txt); neither the load pattern nor this staging had a part in it.
    float *tile = lds + bins * (sizeof(Sample) / sizeof(float));
    constexpr int kTileLd = kRoiCh + 1;
    const RoiGeom g = load_roi(rois, n, width, height);
    const bool valid_im = (g.b_in >= 0 && g.b_in < B);
    for (int b = threadIdx.x; b < bins; b += blockDim.x) samp[b] = make_sample(g, b / pw, b % pw, H, W, ph, pw);
    __syncthreads();
    const int cq = threadIdx.x & 15, bg = threadIdx.x >> 4;
    float *const t0 = tile + bg * kTileLd + 4 * cq;
    // VEC (C % 4 == 0 and the chunk lies inside C: the launcher decides): every corner is ONE unconditional 16-byte load --
    // a bin outside the map (top < 0) or beyond `bins` reads pixel (0, 0) and its result is discarded, so the 16 loads of a
    // thread are straight-line code the compiler keeps in flight together.
    const float *img = feat + (size_t)(valid_im ? g.b_in : 0) * H * W * C + c0 + 4 * cq;
    for (int b0 = 0; b0 < bins; b0 += 16 * kRoiBinsPerThread) {
        float4 tl[kRoiBinsPerThread], tr[kRoiBinsPerThread], bl[kRoiBinsPerThread], br[kRoiBinsPerThread];
        Sample sm[kRoiBinsPerThread];
#pragma unroll
        for (int i = 0; i < kRoiBinsPerThread; ++i) {
            const int b = b0 + bg + 16 * i;
            sm[i] = samp[min(b, bins - 1)];
            if (b >= bins) sm[i].top = -1;
            const bool in = sm[i].top >= 0;
            const int y0 = in ? sm[i].top : 0, y1 = in ? sm[i].bottom : 0, x0 = in ? sm[i].left : 0, x1 = in ? sm[i].right : 0;
            auto fetch = [&](int yy, int xx) -> float4 {
                const float *q = img + ((size_t)yy * W + xx) * C;
                if (VEC) return *reinterpret_cast<const float4 *>(q);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c0 + 4 * cq + 0 < C) v.x = q[0];
                if (c0 + 4 * cq + 1 < C) v.y = q[1];
                if (c0 + 4 * cq + 2 < C) v.z = q[2];
                if (c0 + 4 * cq + 3 < C) v.w = q[3];
                return v;
            };
            tl[i] = fetch(y0, x0);
            tr[i] = fetch(y0, x1);
            bl[i] = fetch(y1, x0);
            br[i] = fetch(y1, x1);
        }
#pragma unroll
        for (int i = 0; i < kRoiBinsPerThread; ++i) {
            const int b = b0 + bg + 16 * i;
            if (b >= bins) continue;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid_im && sm[i].top >= 0) {
                v.x = bilerp(tl[i].x, tr[i].x, bl[i].x, br[i].x, sm[i]);
                v.y = bilerp(tl[i].y, tr[i].y, bl[i].y, br[i].y, sm[i]);
                v.z = bilerp(tl[i].z, tr[i].z, bl[i].z, br[i].z, sm[i]);
                v.w = bilerp(tl[i].w, tr[i].w, bl[i].w, br[i].w, sm[i]);
            }
            float *t = t0 + (b0 + 16 * i) * kTileLd;
            t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
        }
    }
    __syncthreads();
    const int nch = min(kRoiCh, C - c0);
    float *dst = out + ((size_t)n * C + c0) * bins;
    const int total = nch * bins;
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        for (int i = 4 * threadIdx.x; i < total; i += 4 * blockDim.x) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int j = min(i + k, total - 1); v[k] = tile[(j % bins) * kTileLd + (j / bins)]; }
            if (i + 3 < total) *reinterpret_cast<float4 *>(dst + i) = make_float4(v[0], v[1], v[2], v[3]);
            else for (int k = 0; i + k < total; ++k) dst[i + k] = v[k];
        }
    } else {
        for (int i = threadIdx.x; i < total; i += blockDim.x) dst[i] = tile[(i % bins) * kTileLd + (i / bins)];
    }
}

// Backward: the reference scatters with atomicAdd (roi_align_kernel.cu:103-170); same here.
// layout 0 = NCHW, 1 = NHWC for grad_feat.
__global__ void roi_align_bwd_kernel(const float *__restrict__ grad_out, const float *__restrict__ rois,
                                     int n_rois, int B, int C, int H, int W, int ph, int pw, float width,
                                     float height, int layout, float *__restrict__ grad_feat)
{
    const long long total = (long long)n_rois * C * ph * pw;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        long long t = idx;
        const int x = t % pw; t /= pw;
        const int y = t % ph; t /= ph;
        const int d = t % C;
        const int n = t / C;
        const RoiGeom g = load_roi(rois, n, width, height);
        if (g.b_in < 0 || g.b_in >= B) continue;
        const Sample s = make_sample(g, y, x, H, W, ph, pw);
        if (s.top < 0) continue;
        const float go = grad_out[idx];
        const float dtop = (1 - s.y_lerp) * go;
        const float dbottom = s.y_lerp * go;
        auto at = [&](int yy, int xx) -> float * {
            return layout == 0 ? grad_feat + (((size_t)g.b_in * C + d) * H + yy) * W + xx
                               : grad_feat + (((size_t)g.b_in * H + yy) * W + xx) * C + d;
        };
        atomicAdd(at(s.top, s.left), (1 - s.x_lerp) * dtop);
        atomicAdd(at(s.top, s.right), s.x_lerp * dtop);
        atomicAdd(at(s.bottom, s.left), (1 - s.x_lerp) * dbottom);
        atomicAdd(at(s.bottom, s.right), s.x_lerp * dbottom);
    }
}

// Deterministic backward (the detector pre-training path back-propagates through RoIAlign into
// the trunk): a GATHER instead of the reference's atomicAdd scatter.  One workgroup per input pixel (b, Y, X); it walks the
// RoIs of image b in order and, the sampling grid being separable, only the <= 2 bin rows and <= 2 bin columns whose
// bilinear footprint touches the pixel; every matching corner adds the same product the scatter kernel would have added
// (cx * (cy * grad)), in a fixed order (roi, bin row, bin column, corner) -- the result is bit-reproducible run to run.
__global__ __launch_bounds__(256) void roi_align_bwd_gather_kernel(const float *__restrict__ grad_out, const float *__restrict__ rois,
                                                                  int n_rois, int B, int C, int H, int W, int ph, int pw,
                                                                  float width, float height, int layout,
                                                                  float *__restrict__ grad_feat)
{
    const int pix = blockIdx.x;
    const int X = pix % W, Y = (pix / W) % H, b = pix / (W * H);
    constexpr int kMaxPer = 4;                              // C <= 1024 with 256 threads
    float acc[kMaxPer] = {0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < n_rois; ++n) {
        const RoiGeom g = load_roi(rois, n, width, height);
        if (g.b_in != b) continue;                          // block-uniform
        const float height_scale = (ph > 1) ? (g.y2 - g.y1) * (H - 1) / (ph - 1) : 0;
        const float width_scale = (pw > 1) ? (g.x2 - g.x1) * (W - 1) / (pw - 1) : 0;
        for (int py = 0; py < ph; ++py) {
            const float in_y = (ph > 1) ? g.y1 * (H - 1) + py * height_scale : (float)(0.5 * (g.y1 + g.y2) * (H - 1));
            if (in_y < 0 || in_y > H - 1) continue;
            const int top = (int)floorf(in_y), bottom = (int)ceilf(in_y);
            if (top != Y && bottom != Y) continue;
            const float y_lerp = in_y - top;
            for (int px = 0; px < pw; ++px) {
                const float in_x = (pw > 1) ? g.x1 * (W - 1) + px * width_scale : (float)(0.5 * (g.x1 + g.x2) * (W - 1));
                if (in_x < 0 || in_x > W - 1) continue;
                const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
                if (left != X && right != X) continue;
                const float x_lerp = in_x - left;
#pragma unroll
                for (int k = 0; k < kMaxPer; ++k) {
                    const int d = threadIdx.x + 256 * k;
                    if (d >= C) break;
                    const float go = grad_out[(((size_t)n * C + d) * ph + py) * pw + px];
                    const float dtop = (1 - y_lerp) * go, dbottom = y_lerp * go;
                    float a = acc[k];
                    if (top == Y && left == X) a += (1 - x_lerp) * dtop;
                    if (top == Y && right == X) a += x_lerp * dtop;
                    if (bottom == Y && left == X) a += (1 - x_lerp) * dbottom;
                    if (bottom == Y && right == X) a += x_lerp * dbottom;
                    acc[k] = a;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kMaxPer; ++k) {
        const int d = threadIdx.x + 256 * k;
        if (d >= C) break;
        if (layout == 0) grad_feat[(((size_t)b * C + d) * H + Y) * W + X] = acc[k];
        else grad_feat[(((size_t)b * H + Y) * W + X) * C + d] = acc[k];
    }
}

// =====================================================================================
// union-box mask rasteriser (lib/draw_rectangles/draw_rectangles.pyx:41-66)
// =====================================================================================
__device__ __forceinline__ float clamp01(float x)
{
    float t = (0.f > x) ? 0.f : x;   // same comparisons as the generated C (no NaN handling)
    return (1.f < t) ? 1.f : t;
}

// one block per box pair; separable: 2*P x- and 2*P y-contributions in LDS, then P*P*2 outputs
__global__ __launch_bounds__(128) void draw_union_boxes_kernel(const float *__restrict__ pairs, int P,
                                                               float offset, int channels_last,
                                                               float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float contrib[];  // xc[2][P], yc[2][P]
    const int n = blockIdx.x;
    const float *bp = pairs + 8 * (size_t)n;
    const float x1_union = fminf(bp[0], bp[4]);
    const float y1_union = fminf(bp[1], bp[5]);
    const float x2_union = fmaxf(bp[2], bp[6]);
    const float y2_union = fmaxf(bp[3], bp[7]);
    const float w = x2_union - x1_union;
    const float h = y2_union - y1_union;
    const float Pf = (float)(unsigned)P;
    for (int t = threadIdx.x; t < 4 * P; t += blockDim.x) {
        const int which = t / (2 * P);  // 0: x, 1: y
        const int i = (t / P) & 1;
        const int k = t % P;
        float lo, hi;
        if (which == 0) {
            lo = (bp[0 + 4 * i] - x1_union) * Pf / w;
            hi = (bp[2 + 4 * i] - x1_union) * Pf / w;
        } else {
            lo = (bp[1 + 4 * i] - y1_union) * Pf / h;
            hi = (bp[3 + 4 * i] - y1_union) * Pf / h;
        }
        contrib[t] = clamp01((float)(unsigned)(k + 1) - lo) * clamp01(hi - (float)(unsigned)k);
    }
    __syncthreads();
    const float *xc = contrib, *yc = contrib + 2 * P;
    const int per = 2 * P * P;
    float *dst = out + (size_t)n * per;
    for (int t = threadIdx.x; t < per; t += blockDim.x) {
        int i, j, k;
        if (channels_last) { i = t % 2; k = (t / 2) % P; j = t / (2 * P); }
        else { k = t % P; j = (t / P) % P; i = t / (P * P); }
        dst[t] = xc[i * P + k] * yc[i * P + j] + offset;
    }
}

// =====================================================================================
// fp32 pairwise IoU (torch semantics of lib/fpn/box_utils.py:85-131)
// =====================================================================================
// ---------------------------------------------------------------------------------------------------
// Triplet matching of the Recall@K evaluator on the device (lib/evaluation/sg_eval.py:243-284,
// _compute_pred_matches): prediction p matches ground-truth relation g when the (subject, predicate, object)
// labels are equal and both boxes overlap with IoU >= thresh.  IoU is the reference's float64 Cython formula
// (bbox.pyx:21-61: +1 pixel convention, zero unless both extents are positive) on the float32 boxes promoted exactly.
// Outputs: first_match[g] = smallest matching p (INT_MAX if none) -> recall@K = #{g : first_match[g] < K} / G;
//          nmatch[p] = number of ground-truth relations prediction p matches.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double iou_f64(const float *b, const float *q)
{
    const double bx1 = b[0], by1 = b[1], bx2 = b[2], by2 = b[3], qx1 = q[0], qy1 = q[1], qx2 = q[2], qy2 = q[3];
    const double box_area = (qx2 - qx1 + 1.0) * (qy2 - qy1 + 1.0);
    const double iw = fmin(bx2, qx2) - fmax(bx1, qx1) + 1.0;
    if (!(iw > 0.0)) return 0.0;
    const double ih = fmin(by2, qy2) - fmax(by1, qy1) + 1.0;
    if (!(ih > 0.0)) return 0.0;
    const double ua = (bx2 - bx1 + 1.0) * (by2 - by1 + 1.0) + box_area - iw * ih;
    return iw * ih / ua;
}

__global__ void triplet_match_kernel(const int *__restrict__ gt_trip, const float *__restrict__ gt_boxes, int G,
                                     const int *__restrict__ pred_trip, const float *__restrict__ pred_boxes, int P,
                                     double thresh, int *__restrict__ first_match, int *__restrict__ nmatch)
{
    const long long total = (long long)G * P;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int g = (int)(idx / P), p = (int)(idx % P);
        const int *gt = gt_trip + 3 * g, *pt = pred_trip + 3 * p;
        if (gt[0] != pt[0] || gt[1] != pt[1] || gt[2] != pt[2]) continue;
        const float *gb = gt_boxes + 8 * g, *pb = pred_boxes + 8 * p;
        if (iou_f64(pb, gb) >= thresh && iou_f64(pb + 4, gb + 4) >= thresh) {
            atomicMin(first_match + g, p);
            atomicAdd(nmatch + p, 1);
        }
    }
}

__global__ void fill_int_kernel(int *p, int n, int v)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) p[i] = v;
}

__global__ void bbox_overlaps_kernel(const float4 *__restrict__ a, int na, const float4 *__restrict__ b, int nb,
                                     float *__restrict__ out)
{
    const long long total = (long long)na * nb;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const float4 p = a[idx / nb], q = b[idx % nb];
        const float iw = fmaxf(fminf(p.z, q.z) - fmaxf(p.x, q.x) + 1.0f, 0.f);
        const float ih = fmaxf(fminf(p.w, q.w) - fmaxf(p.y, q.y) + 1.0f, 0.f);
        const float inter = iw * ih;
        const float area_a = (p.z - p.x + 1.0f) * (p.w - p.y + 1.0f);
        const float area_b = (q.z - q.x + 1.0f) * (q.w - q.y + 1.0f);
        out[idx] = inter / (area_a + area_b - inter);
    }
}

// ---------------------------------------------------------------------------------------------------
// Class-wise greedy suppression of the decoder's label commitments in SGDet evaluation
// (lib/lstm/decoder_rnn.py:230-247; the reference copies the [N,N,C] IoU tensor and the [N,C] probabilities to the
// host and loops there).  One workgroup: the probability table lives in LDS; N rounds of
//   (box, cls) = first arg-max of the table (row-major, like np.argmax);  commit;  zero column `cls` for every box
//   whose class-`cls` box overlaps this one (IoU >= thresh, computed with the operation order of
//   lib/fpn/box_utils.nms_overlaps);  retire the row (-1).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decoder_nms_commit_kernel(const float *__restrict__ probs, const float *__restrict__ boxes,
                                                                 int N, int C, float thresh, long long *__restrict__ commits)
{
    extern __shared__ float tab[];                     // [N*C]
    __shared__ unsigned long long red[4];
    __shared__ int win[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = N * C;
    for (int i = tid; i < total; i += 256) tab[i] = (i % C == 0) ? 0.f : probs[i];
    __syncthreads();
    for (int round = 0; round < N; ++round) {
        // arg-max with the smallest flat index among equal values: key = (ordered value, ~index)
        unsigned long long best = 0ull;
        for (int i = tid; i < total; i += 256) {
            unsigned u = __float_as_uint(tab[i]);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            const unsigned long long k = ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
            best = k > best ? k : best;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(best, off);
            best = o > best ? o : best;
        }
        if (lane == 0) red[wave] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long b = red[0];
            for (int w = 1; w < 4; ++w) b = red[w] > b ? red[w] : b;
            const int flat = (int)(0xFFFFFFFFu - (unsigned)(b & 0xFFFFFFFFull));
            win[0] = flat / C;
            win[1] = flat % C;
            commits[flat / C] = flat % C;
        }
        __syncthreads();
        const int bi = win[0], ci = win[1];
        const float4 a = *reinterpret_cast<const float4 *>(boxes + ((size_t)bi * C + ci) * 4);
        const float area_i = (a.z - a.x + 1.0f) * (a.w - a.y + 1.0f);
        for (int j = tid; j < N; j += 256) {
            const float4 b = *reinterpret_cast<const float4 *>(boxes + ((size_t)j * C + ci) * 4);
            const float iw = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x) + 1.0f, 0.f);
            const float ih = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y) + 1.0f, 0.f);
            const float inters = iw * ih;
            const float area_j = (b.z - b.x + 1.0f) * (b.w - b.y + 1.0f);
            const float uni = (-inters + area_j) + area_i;
            if (inters / uni >= thresh) tab[j * C + ci] = 0.0f;
        }
        __syncthreads();
        for (int c = tid; c < C; c += 256) tab[bi * C + c] = -1.0f;
        __syncthreads();
    }
}

// ---- relation tail: prod[r] = subj[i1[r]] * obj[i2[r]] (* vis[r]) and its backward (lib/rel_model.py:500-512 of the reference:
// `subj_rep[rel_inds[:, 1]] * obj_rep[rel_inds[:, 2]]`, then `* vr`) -----------------------------------------------------------
// The framework evaluates this as two row gathers and two multiplies, and its backward as four multiplies, two sort-based
// index-add chains (~10 launches each) and two select-backward fills: ~30 launches of 2-20 us on the main stream between the end
// of the forward pass and the first product of the backward pass (profiles/r06_step_launches_c13.txt).  Here: one launch forward,
// two backward; the sums over the rows that share a subject / an object run over a list the HOST made from the pair indices it
// sampled itself (stable order: deterministic, no atomics).
// edge [n][2][D]: subject / object representation of every box; i1 / i2 [R]; out [R][D].
__global__ __launch_bounds__(256) void pair_product_fwd_kernel(const float4 *__restrict__ edge, const long long *__restrict__ i1,
                                                               const long long *__restrict__ i2, const float4 *__restrict__ vis,
                                                               int D4, float4 *__restrict__ out)
{
    const int r = blockIdx.x;
    const float4 *s = edge + (size_t)i1[r] * 2 * D4, *o = edge + ((size_t)i2[r] * 2 + 1) * D4;
    for (int c = threadIdx.x; c < D4; c += 256) {
        const float4 a = s[c], b = o[c];
        float4 p = make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);           // (subj * obj) * vis: the reference's order
        if (vis) {
            const float4 v = vis[(size_t)r * D4 + c];
            p = make_float4(p.x * v.x, p.y * v.y, p.z * v.z, p.w * v.w);
        }
        out[(size_t)r * D4 + c] = p;
    }
}
// d_vis[r] = g[r] * (subj[i1[r]] * obj[i2[r]])
__global__ __launch_bounds__(256) void pair_product_dvis_kernel(const float4 *__restrict__ edge, const long long *__restrict__ i1,
                                                                const long long *__restrict__ i2, const float4 *__restrict__ g,
                                                                int D4, float4 *__restrict__ dvis)
{
    const int r = blockIdx.x;
    const float4 *s = edge + (size_t)i1[r] * 2 * D4, *o = edge + ((size_t)i2[r] * 2 + 1) * D4;
    for (int c = threadIdx.x; c < D4; c += 256) {
        const float4 a = s[c], b = o[c], gv = g[(size_t)r * D4 + c];
        dvis[(size_t)r * D4 + c] = make_float4(gv.x * (a.x * b.x), gv.y * (a.y * b.y), gv.z * (a.z * b.z), gv.w * (a.w * b.w));
    }
}
// d_edge[i][side] = sum over the rows r of segment (side, i), in list order, of (g[r] * vis[r]) * partner(r):
// partner = obj[i2[r]] for side 0 (the rows whose subject is i), subj[i1[r]] for side 1.  grid (n, 2, column chunks)
__global__ __launch_bounds__(256) void pair_product_dedge_kernel(const float4 *__restrict__ edge, const long long *__restrict__ i1,
                                                                 const long long *__restrict__ i2, const float4 *__restrict__ vis,
                                                                 const float4 *__restrict__ g, const int *__restrict__ order,
                                                                 const int *__restrict__ ptr, int n, int D4, float4 *__restrict__ dedge)
{
    const int i = blockIdx.x, side = blockIdx.y, c = blockIdx.z * 256 + threadIdx.x;
    if (c >= D4) return;
    const int k0 = ptr[side * (n + 1) + i], k1 = ptr[side * (n + 1) + i + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = k0; k < k1; ++k) {
        const int r = order[k];
        const float4 *partner = side == 0 ? edge + ((size_t)i2[r] * 2 + 1) * D4 : edge + (size_t)i1[r] * 2 * D4;
        float4 gv = g[(size_t)r * D4 + c];
        if (vis) {
            const float4 v = vis[(size_t)r * D4 + c];
            gv = make_float4(gv.x * v.x, gv.y * v.y, gv.z * v.z, gv.w * v.w);
        }
        const float4 pv = partner[c];
        acc.x += gv.x * pv.x; acc.y += gv.y * pv.y; acc.z += gv.z * pv.z; acc.w += gv.w * pv.w;
    }
    dedge[((size_t)i * 2 + side) * D4 + c] = acc;
}

// ---- relation tail: rel_dists + FrequencyBias[subject class, object class] (reference lib/rel_model.py:528-531,
// lib/sparse_targets.py:39-45) ---------------------------------------------------------------------------------------------------
// Forward: key[r] = labels[i1[r]] * num_objs + labels[i2[r]]; out[r] = logits[r] + table[key[r]] -- one launch instead of two label
// gathers, a stack, the key arithmetic, the embedding gather and the add.  Backward of the table: the framework sorts the keys on the
// device and segments the rows (8 launches, ~85 us at 1536 rows).  Here: one block per row l; it is the LEADER of its key when no
// earlier row has the same key, and then sums the gradient rows of its key in ascending row order -- deterministic, no sort, no
// atomics.  d_table must be zero on entry (the entry point clears it).
constexpr int kFbMaxRows = 8192;       // rows per call the leader kernel handles (keys staged through LDS in pieces of 1024)
__global__ __launch_bounds__(256) void freq_bias_add_kernel(const float *__restrict__ logits, const float *__restrict__ table,
                                                            const long long *__restrict__ labels, const long long *__restrict__ i1,
                                                            const long long *__restrict__ i2, int R, int P, int num_objs,
                                                            float *__restrict__ out, long long *__restrict__ keys)
{
    const long long total = (long long)R * P;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int r = (int)(idx / P), p = (int)(idx - (long long)r * P);
        const long long key = labels[i1[r]] * num_objs + labels[i2[r]];
        if (p == 0) keys[r] = key;
        out[idx] = logits[idx] + table[key * P + p];
    }
}
__global__ __launch_bounds__(64) void freq_bias_bwd_kernel(const float *__restrict__ grad, const long long *__restrict__ keys, int R, int P,
                                                           float *__restrict__ d_table)
{
    __shared__ int rows[kFbMaxRows];
    __shared__ int nrows;
    const int l = blockIdx.x, lane = threadIdx.x;
    const long long key = keys[l];
    // leader test: does an earlier row carry the same key?
    int earlier = 0;
    for (int r = lane; r < l; r += 64) earlier |= (keys[r] == key) ? 1 : 0;
    if (__any(earlier)) return;
    // the rows of this key at or after l, ascending (one wave: ballot + prefix keeps the order)
    if (lane == 0) nrows = 0;
    __syncthreads();
    for (int r0 = l; r0 < R; r0 += 64) {
        const int r = r0 + lane;
        const bool hit = r < R && keys[r] == key;
        const unsigned long long m = __ballot(hit);
        const int base = nrows;
        if (hit) rows[base + __popcll(m & ((1ull << lane) - 1ull))] = r;
        __syncthreads();
        if (lane == 0) nrows = base + __popcll(m);
        __syncthreads();
    }
    const int n = nrows;
    for (int p = lane; p < P; p += 64) {
        float acc = 0.f;
        for (int k = 0; k < n; ++k) acc += grad[(size_t)rows[k] * P + p];
        d_table[(size_t)key * P + p] = acc;
    }
}

// ---- the training script's two losses (reference models/train_rels.py:140-141: F.cross_entropy(result.rm_obj_dists, result.rm_obj_labels)
// and F.cross_entropy(result.rel_dists, result.rel_labels[:, -1])) as ONE node ---------------------------------------------------------
// The framework evaluates each as log_softmax + nll_loss (+ their two backward kernels and fills): ~25 launches of a few microseconds on
// the main stream between the relation tail and the first product of the backward pass.  Here: one wave per row (row maximum,
// sum of exponentials, log-sum-exp kept for the backward pass, row loss), a one-block ordered sum for the two means, and one backward
// launch: grad[r][c] = (exp(x - lse_r) - [c == label_r]) * upstream / rows.
__device__ __forceinline__ float wave_max_f(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float wave_sum_f(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
struct CeSide {
    const float *logits;        // [R][C]
    const long long *labels;    // element r at labels[r * label_stride]
    long long label_stride;
    int R, C;
};
__global__ __launch_bounds__(256) void ce_pair_rows_kernel(CeSide a, CeSide b, float *__restrict__ lse, float *__restrict__ rowloss)
{
    const int row = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= a.R + b.R) return;
    const CeSide &s = row < a.R ? a : b;
    const int r = row < a.R ? row : row - a.R;
    const float *x = s.logits + (size_t)r * s.C;
    float m = -__builtin_inff();
    for (int c = lane; c < s.C; c += 64) m = fmaxf(m, x[c]);
    m = wave_max_f(m);
    float sum = 0.f;
    for (int c = lane; c < s.C; c += 64) sum += expf(x[c] - m);
    sum = wave_sum_f(sum);
    if (lane == 0) {
        const float l = m + logf(sum);
        lse[row] = l;
        const long long lab = s.labels[(size_t)r * s.label_stride];
        // a label outside [0, C) is a caller's error (the framework's kernel asserts on the device): no read outside the row, the loss says so
        rowloss[row] = (lab >= 0 && lab < s.C) ? l - x[lab] : __builtin_nanf("");
    }
}
// losses[0] = mean of the first Ra row losses, losses[1] = mean of the next Rb: one block, fixed order (deterministic)
__global__ __launch_bounds__(256) void ce_pair_mean_kernel(const float *__restrict__ rowloss, int Ra, int Rb, float *__restrict__ losses)
{
    __shared__ float red[256];
    for (int side = 0; side < 2; ++side) {
        const float *p = rowloss + (side ? Ra : 0);
        const int n = side ? Rb : Ra;
        float acc = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) acc += p[i];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) losses[side] = n > 0 ? red[0] / (float)n : 0.f;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void ce_pair_bwd_kernel(CeSide a, CeSide b, const float *__restrict__ lse, const float *__restrict__ upstream,
                                                          float *__restrict__ grad_a, float *__restrict__ grad_b)
{
    const int row = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= a.R + b.R) return;
    const bool first = row < a.R;
    const CeSide &s = first ? a : b;
    const int r = first ? row : row - a.R;
    float *g = (first ? grad_a : grad_b);
    if (!g) return;
    const float scale = upstream[first ? 0 : 1] / (float)s.R;
    const float *x = s.logits + (size_t)r * s.C;
    const float l = lse[row];
    const int label = (int)s.labels[(size_t)r * s.label_stride];
    for (int c = lane; c < s.C; c += 64) g[(size_t)r * s.C + c] = (expf(x[c] - l) - (c == label ? 1.f : 0.f)) * scale;
}

}  // namespace mh

using namespace mh;

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" {

int mh_version(void) { return 100; }
const char *mh_last_error(void) { return mh::g_err; }

size_t mh_nms_ws_bytes(int n)
{
    const size_t cb = (size_t)(n + 63) / 64;
    return align_up((size_t)(n > 0 ? n : 1) * (cb ? cb : 1) * sizeof(unsigned long long), 256);
}

int mh_nms(const float *boxes_sorted, int n, float thresh, int *keep, int *num_keep, void *workspace,
           size_t ws_bytes, void *stream)
{
    MH_REQUIRE(n >= 0 && keep && num_keep);
    hipStream_t st = as_stream(stream);
    if (n == 0) return (int)hipMemsetAsync(num_keep, 0, sizeof(int), st);
    MH_REQUIRE(boxes_sorted && workspace && ws_bytes >= mh_nms_ws_bytes(n));
    MH_REQUIRE((reinterpret_cast<uintptr_t>(boxes_sorted) & 15) == 0);
    const int cb = (n + 63) / 64;
    auto *mask = reinterpret_cast<unsigned long long *>(workspace);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb, 1), dim3(64), 0, st,
                       reinterpret_cast<const float4 *>(boxes_sorted), (const int *)nullptr, n, thresh, mask, cb);
    int rc = check_launch("nms_mask_kernel");
    if (rc) return rc;
    return launch_sweep(mask, nullptr, 1, n, cb, keep, num_keep, st);
}

size_t mh_nms_batched_ws_bytes(int total_boxes, int nseg, int max_seg)
{
    (void)nseg;
    const size_t cb = (size_t)(max_seg + 63) / 64;
    return align_up((size_t)(total_boxes > 0 ? total_boxes : 1) * (cb ? cb : 1) * sizeof(unsigned long long), 256);
}

int mh_nms_batched(const float *boxes_sorted, const int *seg_offsets, int nseg, int total_boxes, int max_seg,
                   float thresh, int *keep, int *num_keep, void *workspace, size_t ws_bytes, void *stream)
{
    MH_REQUIRE(nseg >= 0 && total_boxes >= 0 && max_seg >= 0);
    if (nseg == 0) return MH_OK;
    hipStream_t st = as_stream(stream);
    MH_REQUIRE(seg_offsets && keep && num_keep);
    if (max_seg == 0 || total_boxes == 0) return (int)hipMemsetAsync(num_keep, 0, sizeof(int) * nseg, st);
    MH_REQUIRE(boxes_sorted && workspace && ws_bytes >= mh_nms_batched_ws_bytes(total_boxes, nseg, max_seg));
    MH_REQUIRE((reinterpret_cast<uintptr_t>(boxes_sorted) & 15) == 0);
    MH_REQUIRE(nseg <= 65535);
    const int cb = (max_seg + 63) / 64;
    auto *mask = reinterpret_cast<unsigned long long *>(workspace);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb, nseg), dim3(64), 0, st,
                       reinterpret_cast<const float4 *>(boxes_sorted), seg_offsets, 0, thresh, mask, cb);
    int rc = check_launch("nms_mask_kernel(batched)");
    if (rc) return rc;
    return launch_sweep(mask, seg_offsets, nseg, 0, cb, keep, num_keep, st);
}

static void roi_norm(int H, int W, float spatial_scale, float *width, float *height)
{
    // Python: height = (data_height - 1) / self.spatial_scale  in double, then fp32 tensor /= scalar
    *height = (float)((double)(H - 1) / (double)spatial_scale);
    *width = (float)((double)(W - 1) / (double)spatial_scale);
}

int mh_roi_align_fwd(const float *feat, int B, int C, int H, int W, int feat_layout, const float *rois, int n,
                     int ph, int pw, float spatial_scale, float *out, void *stream)
{
    MH_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && ph > 0 && pw > 0 && n >= 0);
    MH_REQUIRE(feat_layout == 0 || feat_layout == 1);
    if (n == 0) return MH_OK;
    MH_REQUIRE(feat && rois && out);
    float width, height;
    roi_norm(H, W, spatial_scale, &width, &height);
    hipStream_t st = as_stream(stream);
    if (feat_layout == 0) {
        const long long total = (long long)n * C * ph * pw;
        const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 32);
        hipLaunchKernelGGL(roi_align_fwd_nchw, dim3(blocks), dim3(256), 0, st, feat, rois, n, B, C, H, W, ph, pw,
                           width, height, out);
        return check_launch("roi_align_fwd_nchw");
    }
    const int bins = ph * pw;
    const size_t lds = (size_t)bins * sizeof(Sample) + (size_t)bins * (kRoiCh + 1) * sizeof(float);
    MH_REQUIRE(lds <= 64 * 1024);
    MH_REQUIRE((long long)n * ceil_div(C, kRoiCh) < (1LL << 31));
    const dim3 grid((unsigned)(n * ceil_div(C, kRoiCh)));
    if (C % kRoiCh == 0 && (reinterpret_cast<uintptr_t>(feat) & 15) == 0)
        hipLaunchKernelGGL(roi_align_fwd_nhwc<true>, grid, dim3(256), lds, st, feat, rois, B, C, H, W, ph, pw, width, height, out);
    else
        hipLaunchKernelGGL(roi_align_fwd_nhwc<false>, grid, dim3(256), lds, st, feat, rois, B, C, H, W, ph, pw, width, height, out);
    return check_launch("roi_align_fwd_nhwc");
}

int mh_roi_align_bwd(const float *grad_out, int B, int C, int H, int W, int feat_layout, const float *rois, int n,
                     int ph, int pw, float spatial_scale, float *grad_feat, void *stream)
{
    MH_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && ph > 0 && pw > 0 && n >= 0 && grad_feat);
    MH_REQUIRE(feat_layout == 0 || feat_layout == 1);
    hipStream_t st = as_stream(stream);
    hipError_t e = hipMemsetAsync(grad_feat, 0, sizeof(float) * (size_t)B * C * H * W, st);
    if (e != hipSuccess) return (int)e;
    if (n == 0) return MH_OK;
    MH_REQUIRE(grad_out && rois);
    float width, height;
    roi_norm(H, W, spatial_scale, &width, &height);
    const long long total = (long long)n * C * ph * pw;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(roi_align_bwd_kernel, dim3(blocks), dim3(256), 0, st, grad_out, rois, n, B, C, H, W, ph, pw,
                       width, height, feat_layout, grad_feat);
    return check_launch("roi_align_bwd_kernel");
}

int mh_roi_align_bwd_det(const float *grad_out, int B, int C, int H, int W, int feat_layout, const float *rois, int n,
                         int ph, int pw, float spatial_scale, float *grad_feat, void *stream)
{
    MH_REQUIRE(B > 0 && C > 0 && C <= 1024 && H > 0 && W > 0 && ph > 0 && pw > 0 && n >= 0 && grad_feat);
    MH_REQUIRE(feat_layout == 0 || feat_layout == 1);
    MH_REQUIRE(n == 0 || (grad_out && rois));
    MH_REQUIRE((long long)B * H * W < (1LL << 31));
    float width, height;
    roi_norm(H, W, spatial_scale, &width, &height);
    hipLaunchKernelGGL(roi_align_bwd_gather_kernel, dim3((unsigned)(B * H * W)), dim3(256), 0, as_stream(stream), grad_out, rois,
                       n, B, C, H, W, ph, pw, width, height, feat_layout, grad_feat);
    return check_launch("roi_align_bwd_gather_kernel");
}

int mh_draw_union_boxes(const float *box_pairs, int n, int P, float offset, int channels_last, float *out,
                        void *stream)
{
    MH_REQUIRE(n >= 0 && P > 0 && P <= 1024);
    if (n == 0) return MH_OK;
    MH_REQUIRE(box_pairs && out);
    hipLaunchKernelGGL(draw_union_boxes_kernel, dim3(n), dim3(128), (size_t)4 * P * sizeof(float), as_stream(stream),
                       box_pairs, P, offset, channels_last, out);
    return check_launch("draw_union_boxes_kernel");
}

int mh_triplet_match(const int *gt_triplets, const float *gt_boxes, int G, const int *pred_triplets,
                     const float *pred_boxes, int P, double iou_thresh, int *first_match, int *nmatch, void *stream)
{
    MH_REQUIRE(G >= 0 && P >= 0 && first_match && nmatch);
    hipStream_t st = as_stream(stream);
    if (G > 0) hipLaunchKernelGGL(fill_int_kernel, dim3(ceil_div(G, 256)), dim3(256), 0, st, first_match, G, 0x7fffffff);
    if (P > 0) hipLaunchKernelGGL(fill_int_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, st, nmatch, P, 0);
    int rc = check_launch("fill_int_kernel");
    if (rc || G == 0 || P == 0) return rc;
    MH_REQUIRE(gt_triplets && gt_boxes && pred_triplets && pred_boxes);
    const long long total = (long long)G * P;
    hipLaunchKernelGGL(triplet_match_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, st,
                       gt_triplets, gt_boxes, G, pred_triplets, pred_boxes, P, iou_thresh, first_match, nmatch);
    return check_launch("triplet_match_kernel");
}

// largest score table (N * C * 4 bytes) the single-workgroup kernel can hold in LDS on the current device: what the device
// reports as its per-block limit (160 KB on gfx950) minus 10 KB for the kernel's static arrays; 0 if the query fails
size_t mh_decoder_nms_commit_max_bytes()
{
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 10 * 1024) return 0;
    return (size_t)std::min(v - 10 * 1024, 150 * 1024);
}

int mh_decoder_nms_commit(const float *probs, const float *boxes, int N, int C, float thresh, long long *commits, void *stream)
{
    MH_REQUIRE(N >= 0 && C > 1);
    if (N == 0) return MH_OK;
    MH_REQUIRE(probs && boxes && commits && (reinterpret_cast<uintptr_t>(boxes) & 15) == 0);
    const size_t lds = (size_t)N * C * sizeof(float);
    const size_t cap = mh_decoder_nms_commit_max_bytes();
    MH_REQUIRE(cap > 0 && lds <= cap);
    static std::atomic<unsigned long long> raised{0};          // bit d: the dynamic-LDS limit of the kernel was raised on device d
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev < 0 || !((raised.load(std::memory_order_acquire) >> dev) & 1ULL)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(decoder_nms_commit_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);
        if (e != hipSuccess) { set_last_error("hipFuncSetAttribute(decoder_nms_commit_kernel, dynamic LDS)", e); return (int)e; }
        if (dev >= 0) raised.fetch_or(1ULL << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL(decoder_nms_commit_kernel, dim3(1), dim3(256), lds, as_stream(stream), probs, boxes, N, C, thresh, commits);
    return check_launch("decoder_nms_commit_kernel");
}

int mh_bbox_overlaps(const float *boxes_a, int na, const float *boxes_b, int nb, float *out, void *stream)
{
    MH_REQUIRE(na >= 0 && nb >= 0);
    if (na == 0 || nb == 0) return MH_OK;
    MH_REQUIRE(boxes_a && boxes_b && out);
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(boxes_a) | reinterpret_cast<uintptr_t>(boxes_b)) & 15) == 0);
    const long long total = (long long)na * nb;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(bbox_overlaps_kernel, dim3(blocks), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(boxes_a), na, reinterpret_cast<const float4 *>(boxes_b), nb,
                       out);
    return check_launch("bbox_overlaps_kernel");
}

int mh_pair_product_fwd(const float *edge, int n, int D, const long long *i1, const long long *i2, int R, const float *vis, float *out,
                        void *stream)
{
    MH_REQUIRE(n >= 0 && R >= 0 && D > 0 && D % 4 == 0);
    if (R == 0) return MH_OK;
    MH_REQUIRE(edge && i1 && i2 && out && n > 0);
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(edge) | reinterpret_cast<uintptr_t>(vis) | reinterpret_cast<uintptr_t>(out)) & 15) == 0);
    hipLaunchKernelGGL(pair_product_fwd_kernel, dim3((unsigned)R), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4 *>(edge),
                       i1, i2, reinterpret_cast<const float4 *>(vis), D / 4, reinterpret_cast<float4 *>(out));
    return check_launch("pair_product_fwd_kernel");
}

int mh_pair_product_bwd(const float *edge, int n, int D, const long long *i1, const long long *i2, int R, const float *vis,
                        const float *grad_out, const int *order, const int *ptr, float *d_edge, float *d_vis, void *stream)
{
    MH_REQUIRE(n > 0 && R >= 0 && D > 0 && D % 4 == 0 && edge && d_edge && ptr && (vis == nullptr) == (d_vis == nullptr));
    MH_REQUIRE(R == 0 || (i1 && i2 && grad_out && order));
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(edge) | reinterpret_cast<uintptr_t>(vis) | reinterpret_cast<uintptr_t>(grad_out) |
                 reinterpret_cast<uintptr_t>(d_edge) | reinterpret_cast<uintptr_t>(d_vis)) & 15) == 0);
    MH_REQUIRE(n <= 0x7fffffff / 2 && (D / 4 + 255) / 256 <= 65535);
    hipStream_t st = as_stream(stream);
    const int D4 = D / 4;
    if (d_vis && R > 0) {
        hipLaunchKernelGGL(pair_product_dvis_kernel, dim3((unsigned)R), dim3(256), 0, st, reinterpret_cast<const float4 *>(edge), i1, i2,
                           reinterpret_cast<const float4 *>(grad_out), D4, reinterpret_cast<float4 *>(d_vis));
        int rc = check_launch("pair_product_dvis_kernel");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(pair_product_dedge_kernel, dim3((unsigned)n, 2, (unsigned)((D4 + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const float4 *>(edge), i1, i2, reinterpret_cast<const float4 *>(vis),
                       reinterpret_cast<const float4 *>(grad_out), order, ptr, n, D4, reinterpret_cast<float4 *>(d_edge));
    return check_launch("pair_product_dedge_kernel");
}

int mh_freq_bias_add(const float *logits, const float *table, const long long *labels, const long long *i1, const long long *i2, int R,
                     int P, int num_objs, float *out, long long *keys, void *stream)
{
    MH_REQUIRE(R >= 0 && P > 0 && num_objs > 0);
    if (R == 0) return MH_OK;
    MH_REQUIRE(logits && table && labels && i1 && i2 && out && keys);
    const long long total = (long long)R * P;
    hipLaunchKernelGGL(freq_bias_add_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 256 * 8)), dim3(256), 0, as_stream(stream),
                       logits, table, labels, i1, i2, R, P, num_objs, out, keys);
    return check_launch("freq_bias_add_kernel");
}

int mh_freq_bias_bwd(const float *grad_out, const long long *keys, int R, int P, long long table_rows, float *d_table, void *stream)
{
    MH_REQUIRE(R >= 0 && R <= kFbMaxRows && P > 0 && table_rows > 0 && d_table);
    hipStream_t st = as_stream(stream);
    hipError_t e = hipMemsetAsync(d_table, 0, (size_t)table_rows * P * sizeof(float), st);
    if (e != hipSuccess) { set_last_error("hipMemsetAsync(frequency-bias gradient)", e); return (int)e; }
    if (R == 0) return MH_OK;
    MH_REQUIRE(grad_out && keys);
    hipLaunchKernelGGL(freq_bias_bwd_kernel, dim3((unsigned)R), dim3(64), 0, st, grad_out, keys, R, P, d_table);
    return check_launch("freq_bias_bwd_kernel");
}

int mh_ce_pair_fwd(const float *logits_a, const long long *labels_a, long long stride_a, int Ra, int Ca, const float *logits_b,
                   const long long *labels_b, long long stride_b, int Rb, int Cb, float *lse, float *rowloss, float *losses, void *stream)
{
    MH_REQUIRE(Ra > 0 && Rb > 0 && Ca > 0 && Cb > 0 && stride_a > 0 && stride_b > 0);
    MH_REQUIRE(logits_a && labels_a && logits_b && labels_b && lse && rowloss && losses);
    const CeSide a{logits_a, labels_a, stride_a, Ra, Ca}, b{logits_b, labels_b, stride_b, Rb, Cb};
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(ce_pair_rows_kernel, dim3((unsigned)((Ra + Rb + 3) / 4)), dim3(256), 0, st, a, b, lse, rowloss);
    int rc = check_launch("ce_pair_rows_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(ce_pair_mean_kernel, dim3(1), dim3(256), 0, st, rowloss, Ra, Rb, losses);
    return check_launch("ce_pair_mean_kernel");
}

int mh_ce_pair_bwd(const float *logits_a, const long long *labels_a, long long stride_a, int Ra, int Ca, const float *logits_b,
                   const long long *labels_b, long long stride_b, int Rb, int Cb, const float *lse, const float *upstream, float *grad_a,
                   float *grad_b, void *stream)
{
    MH_REQUIRE(Ra > 0 && Rb > 0 && Ca > 0 && Cb > 0 && stride_a > 0 && stride_b > 0);
    MH_REQUIRE(logits_a && labels_a && logits_b && labels_b && lse && upstream && (grad_a || grad_b));
    const CeSide a{logits_a, labels_a, stride_a, Ra, Ca}, b{logits_b, labels_b, stride_b, Rb, Cb};
    hipLaunchKernelGGL(ce_pair_bwd_kernel, dim3((unsigned)((Ra + Rb + 3) / 4)), dim3(256), 0, as_stream(stream), a, b, lse, upstream, grad_a, grad_b);
    return check_launch("ce_pair_bwd_kernel");
}

}  // extern "C"
