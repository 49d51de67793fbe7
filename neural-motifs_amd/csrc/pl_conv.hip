// pl_conv.hip -- the VGG trunk's 3x3 convolutions on the plane engine (pl_tile.h): implicit GEMM whose BOTH operands are
// pre-split f16 plane images, so the K loop is copies + MFMAs only (rounds 1-2 split every activation element inside the
// loop: 4.5 VALU + 3.5 SALU per MFMA, profiles/r02_pmc_mfma_kernels.csv).  Reference: lib/object_detector.py:110-118
// (vgg16.features, cuDNN there), :623-633 (load_vgg).
//
//   activations   image [Cin/16][B*H*W pixels][64 B]  ("NC16HW": the cells of consecutive pixels of one 16-channel chunk are
//                 contiguous, so a tile's A rows for any tap are ONE contiguous run, shifted by the tap), scaled per IMAGE:
//                 exponent = row_exponent(maxbits[b]), maxbits[b] = the largest |x| of image b (fp32 bits).  An output pixel
//                 gathers nine input pixels, so the scale has to be shared by everything a row of the implicit GEMM reads;
//                 one scale per image keeps results independent of what else is in the batch.
//   weights       image [tap][Cin/16][Cout][64 B], exponent per output channel (+ maxbits[Cout] behind the cells)
//   output        fp32 NHWC (bias + ReLU fused) AND the per-image maxima of what was written (one atomicMax per wave), which
//                 is all the next layer's converter needs:
//   mh_act_planes fp32 NHWC -> activation image in ONE pass, optionally through the 2x2/2 max-pool that follows the layer
//                 (replaces the pool launches and the per-pixel exponent passes of round 2).
// M = B*H*W, N = Cout, K = 9 taps x Cin, walked 16-channel chunk outer / tap inner (the nine k-tiles of a chunk re-read the
// same three pixel rows: L1 / L2 hits).  Tile schedule = plan_conv_tiles (conv.hip): whole tiles for the full rounds of
// resident blocks + the leftover tiles cut into K slices.
#include <algorithm>
#include <cstdlib>

#include "mfma_tile.h"     // plan_conv_tiles, ConvTilePlan
#include "pl_ring.h"
#include "pl_tile.h"

namespace mh {
namespace pl {

__device__ __forceinline__ float conv_epi(float v, int epilogue)
{
    if (epilogue == MH_EPI_RELU) return fmaxf(v, 0.f);
    if (epilogue == MH_EPI_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
    return v;
}

// ------------------------------------------------------------------------------------------------- fp32 NHWC -> image
struct ActArgs {
    const float *x;            // [B][H][W][C] fp32
    const unsigned *maxbits;   // [B] largest |x| per image of the INPUT tensor (bounds the pooled tensor too)
    int B, H, W, C, pool;      // pool: 2x2/2 max-pool first (output Ho = H/2, Wo = W/2)
    char *cells;               // [C/16][B*Ho*Wo][64]
    unsigned *maxbits_out;     // copy of maxbits behind the cells (the image is self-contained)
};

// thread = one output pixel x one 16-channel chunk.  Lanes run over the chunks of a pixel first (Gb = min(C/16, 16) chunks,
// then the next pixel): a pixel's Gb * 64 input bytes are read as one contiguous run (4 runs when pooling), and the cells of
// one chunk written by a wave are 64 / Gb consecutive pixels = a 256 B .. 1 KB run.  (The first version walked pixels first:
// every lane of a load touched a different 64-byte segment, 2.2-3.9 TB/s; profiles/r03_pl_conv_check.jsonl.)
__global__ __launch_bounds__(256) void act_planes_kernel(const ActArgs p)
{
    const int Ho = p.pool ? p.H / 2 : p.H, Wo = p.pool ? p.W / 2 : p.W;
    const long long Mo = (long long)p.B * Ho * Wo;
    const int G = p.C / kBK, Gb = G < 16 ? G : 16, ngrp = (G + Gb - 1) / Gb;
    const int ppb = 256 / Gb;                                             // pixels per block step (threads beyond ppb * Gb idle)
    const long long nblk_m = (Mo + ppb - 1) / ppb;
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < p.B; i += 256) p.maxbits_out[i] = p.maxbits[i];
    for (long long blk = blockIdx.x; blk < nblk_m * ngrp; blk += gridDim.x) {
        const int g = (int)(blk % ngrp) * Gb + (int)(threadIdx.x % Gb);
        const long long m = (blk / ngrp) * ppb + threadIdx.x / Gb;
        if (m >= Mo || g >= G || (int)threadIdx.x >= ppb * Gb) continue;
        const int b = (int)(m / ((long long)Ho * Wo));
        const int rem = (int)(m % ((long long)Ho * Wo)), yo = rem / Wo, xo = rem % Wo;
        const int e = row_exponent(p.maxbits[b]);
        float v[16];
        if (p.pool) {
            const float *q = p.x + ((((size_t)b * p.H + 2 * yo) * p.W + 2 * xo) * p.C + g * kBK);
            const size_t dx = p.C, dy = (size_t)p.W * p.C;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 a = *reinterpret_cast<const float4 *>(q + 4 * i), bq = *reinterpret_cast<const float4 *>(q + dx + 4 * i),
                             c = *reinterpret_cast<const float4 *>(q + dy + 4 * i), d = *reinterpret_cast<const float4 *>(q + dy + dx + 4 * i);
                v[4 * i + 0] = fmaxf(fmaxf(a.x, bq.x), fmaxf(c.x, d.x));
                v[4 * i + 1] = fmaxf(fmaxf(a.y, bq.y), fmaxf(c.y, d.y));
                v[4 * i + 2] = fmaxf(fmaxf(a.z, bq.z), fmaxf(c.z, d.z));
                v[4 * i + 3] = fmaxf(fmaxf(a.w, bq.w), fmaxf(c.w, d.w));
            }
        } else {
            const float *q = p.x + ((size_t)m * p.C + g * kBK);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 a = *reinterpret_cast<const float4 *>(q + 4 * i);
                v[4 * i + 0] = a.x; v[4 * i + 1] = a.y; v[4 * i + 2] = a.z; v[4 * i + 3] = a.w;
            }
        }
        unsigned h1[8], h2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) split2(v[2 * i], v[2 * i + 1], e, h1[i], h2[i]);
        u32x4 *dst = reinterpret_cast<u32x4 *>(p.cells + ((size_t)g * Mo + m) * kCell);
        dst[0] = (u32x4){h1[0], h1[1], h1[2], h1[3]};
        dst[1] = (u32x4){h1[4], h1[5], h1[6], h1[7]};
        dst[2] = (u32x4){h2[0], h2[1], h2[2], h2[3]};
        dst[3] = (u32x4){h2[4], h2[5], h2[6], h2[7]};
    }
}

// ------------------------------------------------------------------------------------------------- packed weights
// cells [tap][k / 16][n][64 B] of element (tap, n, k) = w[n][k][tap], or w[k][n][8 - tap] when flip_transpose (the dgrad
// conv: channel roles swapped, taps mirrored); maxbits[N] of output channel n over its 9*K weights behind the cells
__global__ __launch_bounds__(256) void weight_maxbits_kernel(const float *__restrict__ w, int N, int K, int flip_transpose, int src_cin,
                                                             unsigned *__restrict__ bits)
{
    __shared__ unsigned red[256];
    const int n = blockIdx.x;
    unsigned m = 0;
    for (int i = threadIdx.x; i < 9 * K; i += 256) {
        const int k = i / 9, tap = i % 9;
        const float v = flip_transpose ? w[((size_t)k * src_cin + n) * 9 + tap] : w[((size_t)n * src_cin + k) * 9 + tap];
        m = max(m, __float_as_uint(v) & 0x7fffffffu);
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) bits[n] = red[0];
}
// wnorm = max over output channels n of sum |w_n| (one block per n, float bits are monotone for non-negative values)
__global__ __launch_bounds__(256) void weight_norm_kernel(const float *__restrict__ w, int N, int K, int flip_transpose, int src_cin,
                                                          unsigned *__restrict__ wnorm_bits)
{
    __shared__ float red[256];
    const int n = blockIdx.x;
    float a = 0.f;
    for (int i = threadIdx.x; i < 9 * K; i += 256) {
        const int k = i / 9, tap = i % 9;
        a += fabsf(flip_transpose ? w[((size_t)k * src_cin + n) * 9 + tap] : w[((size_t)n * src_cin + k) * 9 + tap]);
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(wnorm_bits, __float_as_uint(red[0] * 1.0001f));      // a hair above the rounded sum
}
__global__ void pack_weight_kernel(const float *__restrict__ w, int N, int K, int flip_transpose, int src_cin,
                                   const unsigned *__restrict__ bits, unsigned *__restrict__ cells)
{
    const int G = K / kBK;
    const long long total = 9LL * G * N * 16;        // dwords
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)blockDim.x * gridDim.x) {
        const int d = (int)(idx % 16), plane = d / 8, kp = d % 8;
        long long t = idx / 16;
        const int n = (int)(t % N); t /= N;
        const int g = (int)(t % G);
        const int tap = (int)(t / G);
        auto src = [&](int k) -> float {
            return flip_transpose ? w[((size_t)k * src_cin + n) * 9 + (8 - tap)] : w[((size_t)n * src_cin + k) * 9 + tap];
        };
        unsigned p1, p2;
        split2(src(g * kBK + 2 * kp), src(g * kBK + 2 * kp + 1), row_exponent(bits[n]), p1, p2);
        cells[idx] = plane ? p2 : p1;
    }
}

// ------------------------------------------------------------------------------------------------- the conv kernel
struct ConvArgs {
    const char *in;            // activation cells [G][Mtot][64]
    const unsigned *in_bits;   // [B]
    int B, H, W, Cin;
    const char *wt;            // weight cells [9][G][Cout][64]
    const unsigned *wt_bits;   // [Cout]
    int Cout;
    const float *bias;
    int epilogue;
    float *out;                // [Mtot][Cout] fp32 NHWC, or nullptr when the output is written as an image:
    char *out_cells;           // [Cout/16][Mtot][64] activation image of the output (epilogue splits it itself), or nullptr
    unsigned *out_scale_bits;  // [B] behind out_cells: the bound whose exponent scales image b (written by block 0)
    const unsigned *in_true_bits;   // [B] TRUE largest |x| per input image (== in_bits when the input image came from mh_act_planes)
    const float *wnorm;        // max over output channels of sum |w| (packed weights' tail): |y| <= max|x| * wnorm + max|bias|
    unsigned *out_bits;        // [B] TRUE largest |y| per image, zero before the launch (may be nullptr)
    int tiles_m, tiles_n;
    // tile schedule (conv.hip: ConvArgs has the long explanation): blocks [0, tail_tiles * tail_slices) = the leftover tiles
    // cut into K slices; then body_tiles * splitk blocks of whole (or uniformly split) tiles
    int body_tiles, splitk, ktiles_per_split;
    int tail_tiles, tail_slices, tail_ktiles;
    long long tail_row0;
    float *partial, *partial_tail;
    int *counters;             // ring kernels: arrival counters of the sliced tiles [body_tiles + tail_tiles], zero on entry, left zero
    int debug_flags;           // measurement only (mh_debug_plconv_flags): bit 1 = the ring kernel returns without its epilogue (no output):
                               // what the K loop alone costs (gpurun r04_c5)
};

constexpr int kStageOff = 4096;       // LDS offset of the image epilogue's staging (behind the exponent tables)
template <class S>
constexpr int conv_lds_bytes() { return S::lds_bytes > kStageOff + 4 * 4096 * S::sn ? S::lds_bytes : kStageOff + 4 * 4096 * S::sn; }

template <class S, bool IMG>      // IMG: the output is written as an activation image (p.out_cells), else fp32 NHWC (p.out)
__global__ __launch_bounds__(kThreads, (S::bm * S::bn <= 128 * 128) ? 3 : 2) void conv3x3_kernel(const ConvArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int wm, wn;
    wave_origin<S>(wave, wm, wn);
    const int tail_blocks = p.tail_tiles * p.tail_slices;
    const bool is_tail = (int)blockIdx.x < tail_blocks;
    int t, slice, kt_per_slice, nslices;
    if (is_tail) {
        t = p.body_tiles + (int)blockIdx.x / p.tail_slices;
        slice = (int)blockIdx.x % p.tail_slices;
        kt_per_slice = p.tail_ktiles;
        nslices = p.tail_slices;
    } else {
        const int bb = (int)blockIdx.x - tail_blocks;
        t = xcd_remap(bb % p.body_tiles, p.body_tiles);
        slice = bb / p.body_tiles;
        kt_per_slice = p.ktiles_per_split;
        nslices = p.splitk;
    }
    // consecutive tiles walk over Cout first: they share the same input pixels in L2
    const long long m0 = (long long)(t / p.tiles_n) * S::bm;
    const int n0 = (t % p.tiles_n) * S::bn;
    const long long HW = (long long)p.H * p.W, Mtot = (long long)p.B * HW;
    const int G = p.Cin / kBK;
    const int total_kt = 9 * G;
    const int kt_begin = slice * kt_per_slice, kt_end = min(total_kt, kt_begin + kt_per_slice);

    // A is addressed relative to one halo (W + 1 pixels) before the tile's first pixel: every tap of every valid pixel has a
    // non-negative offset; which of the nine taps stay inside the image is a 9-bit mask per staged row
    const int halo = p.W + 1;
    const Src sa = make_src(p.in + (m0 - halo) * (long long)kCell), sb = make_src(p.wt + (size_t)n0 * kCell);
    CopyPlan<S> cp;
    plan_copy<S>(cp, [&](int) { return true; }, [&](int r) { return n0 + r < p.Cout; }, tid);
    unsigned a_taps[S::na];
#pragma unroll
    for (int j = 0; j < S::na; ++j) {
        const long long pix = m0 + (tid >> 2) + 64 * j;
        const bool ok = pix < Mtot;
        const int rem = (int)((ok ? pix : 0) % HW), py = rem / p.W, px = rem % p.W;
        unsigned mask = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            if (ok && (unsigned)(py + dy) < (unsigned)p.H && (unsigned)(px + dx) < (unsigned)p.W) mask |= 1u << tap;
        }
        a_taps[j] = mask;
    }
    FragPlan fp;
    plan_frags<S>(fp, wm, wn, lane);
    const unsigned strideA = (unsigned)(Mtot * kCell), strideB = (unsigned)p.Cout * kCell;     // bytes per 16-channel chunk
    // Per-k-tile operand offsets come from a table in LDS (behind the tile buffers / the image staging), built once per block:
    // entry kt = (A scalar offset, B scalar offset, tap bit) of (chunk kt / 9, tap kt % 9).  The first version advanced
    // (chunk, tap) counters and recomputed both offsets every k-tile: 27 scalar instructions, six of them s_mul_i32, in
    // front of the next tile's loads (profiles/r03_pmc_plane_conv_set3.csv: 2.6 SALU per MFMA on the 288-k-tile layers).
    // issue() reads the entry of the tile AFTER the one it loads, so the LDS latency hides behind a step's MFMAs.
    uint4 *ktab = reinterpret_cast<uint4 *>(lds + (IMG ? conv_lds_bytes<S>() : S::lds_bytes));
    for (int i = tid; i < total_kt; i += kThreads) {
        const int g = i / 9, tap = i - 9 * g, dy = tap / 3, dx = tap - 3 * dy;
        ktab[i] = make_uint4((unsigned)g * strideA + (unsigned)(halo + (dy - 1) * p.W + (dx - 1)) * kCell,
                             (unsigned)(tap * G + g) * strideB, 1u << tap, 0u);
    }
    __syncthreads();
    int it_kt = kt_begin;
    uint4 ent = ktab[it_kt];
    auto issue = [&](Stage<S> &st) {
        const unsigned oa = (unsigned)__builtin_amdgcn_readfirstlane((int)ent.x), ob = (unsigned)__builtin_amdgcn_readfirstlane((int)ent.y),
                       bit = (unsigned)__builtin_amdgcn_readfirstlane((int)ent.z);
#pragma unroll
        for (int j = 0; j < S::na; ++j) st.a[j] = load16(sa, (a_taps[j] & bit) ? cp.va[j] : kOob, oa);
#pragma unroll
        for (int j = 0; j < S::nb; ++j) st.b[j] = load16(sb, cp.vb[j], ob);
        if (it_kt + 1 < kt_end) ++it_kt;        // after the last tile of the slice the state stays (the harmless reload of the final step)
        ent = ktab[it_kt];
    };

    Acc<S> acc;
    acc_zero<S>(acc);
    Stage<S> st;
    char *b0 = lds, *b1 = lds + S::buf_bytes;
    issue(st);
    store_stage<S>(st, cp, b0);
    __syncthreads();
    int kt = kt_begin;
    for (; kt + 1 < kt_end; kt += 2) {
        k_step<S>(issue, st, cp, fp, b0, b1, acc);
        k_step<S>(issue, st, cp, fp, b1, b0, acc);
    }
    if (kt < kt_end) k_step<S>(issue, st, cp, fp, b0, b1, acc);

    // exponents of this tile's rows (their image's) and columns; with an image output also the OUTPUT exponent of every
    // row's image, from the bound |y| <= max|x_b| * wnorm + max|bias| (the true maximum of y is not known before it is
    // written; the bound costs a few bits of the two-term split's 2^18 dynamic window, never an overflow)
    int *ex = reinterpret_cast<int *>(lds);
    unsigned *red = reinterpret_cast<unsigned *>(lds) + 2 * S::bm + S::bn;
    float bmax = 0.f;
    if (IMG) {
        unsigned m = 0;
        if (p.bias)
            for (int n = tid; n < p.Cout; n += kThreads) m = max(m, __float_as_uint(p.bias[n]) & 0x7fffffffu);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        if (lane == 0) red[wave] = m;
        __syncthreads();
        bmax = __uint_as_float(max(max(red[0], red[1]), max(red[2], red[3])));
    }
    auto bound_bits = [&](int b) { return __float_as_uint(__uint_as_float(p.in_true_bits[b]) * p.wnorm[0] + bmax); };
    for (int i = tid; i < S::bm + S::bn; i += kThreads) {
        int e = 0;
        if (i < S::bm) {
            if (m0 + i < Mtot) {
                e = row_exponent(p.in_bits[(m0 + i) / HW]);
                if (IMG) ex[S::bm + S::bn + i] = row_exponent(bound_bits((int)((m0 + i) / HW)));
            } else if (IMG) ex[S::bm + S::bn + i] = 0;
        } else if (n0 + (i - S::bm) < p.Cout) e = row_exponent(p.wt_bits[n0 + (i - S::bm)]);
        ex[i] = e;
    }
    if (IMG && blockIdx.x == 0 && tid < p.B) p.out_scale_bits[tid] = bound_bits(tid);
    __syncthreads();
    int ecol[S::sn];
    float bcol[S::sn];
#pragma unroll
    for (int sn = 0; sn < S::sn; ++sn) {
        const int c = wn + 32 * sn + (lane & 31);
        ecol[sn] = ex[S::bm + c];
        bcol[sn] = (p.bias && n0 + c < p.Cout) ? p.bias[n0 + c] : 0.f;
    }
    if (nslices > 1) {
        // partial sums of this K slice: rows numbered from the first row of the block's region (body / tail)
        const long long region_row0 = is_tail ? p.tail_row0 : 0, region_rows = is_tail ? Mtot - p.tail_row0 : p.tail_row0;
        float *dst = (is_tail ? p.partial_tail : p.partial) + (size_t)slice * region_rows * p.Cout;
        acc_foreach<S>(acc, wm, wn, lane, [&](int r, int c, int sn, float v) {
            const long long row = m0 + r;
            if (row < Mtot && n0 + c < p.Cout) dst[(size_t)(row - region_row0) * p.Cout + n0 + c] = __builtin_ldexpf(v, -(ex[r] + ecol[sn]));
        });
        return;
    }
    // per-image maxima of what this tile writes: a tile covers one image or straddles two (two running maxima per lane, one
    // wave reduction + atomic each); tiles over more than two images (maps smaller than the tile) reduce row by row
    const long long last_row = min(m0 + S::bm, Mtot) - 1;
    const int b_lo = (int)(m0 / HW), b_hi = (int)(last_row / HW);
    const int split = (int)min((long long)S::bm, (long long)(b_lo + 1) * HW - m0);     // first tile row of the next image
    unsigned vlo = 0, vhi = 0;
    auto track = [&](int r, long long row, float v) {
        const unsigned bits = __float_as_uint(v) & 0x7fffffffu;
        if (b_hi - b_lo <= 1) { if (r < split) vlo = max(vlo, bits); else vhi = max(vhi, bits); }
        else if (p.out_bits) {
            unsigned m = bits;                       // one row = the 32 lanes of a half-wave: reduce, one atomic per row
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
            if ((lane & 31) == 0 && m && may_raise(p.out_bits + row / HW, m)) atomicMax(p.out_bits + row / HW, m);
        }
    };
    if (!IMG) {
        acc_foreach<S>(acc, wm, wn, lane, [&](int r, int c, int sn, float v) {
            const long long row = m0 + r;
            if (row >= Mtot || n0 + c >= p.Cout) return;
            v = conv_epi(__builtin_ldexpf(v, -(ex[r] + ecol[sn])) + bcol[sn], p.epilogue);
            p.out[(size_t)row * p.Cout + n0 + c] = v;
            track(r, row, v);
        });
    } else {
        // The output leaves the block AS the next layer's activation image: no fp32 round trip through HBM, no converter pass.
        // Per 32-row sub-tile a wave turns its values into (h1, h2) halves scaled by the image's exponent, pairs neighbouring
        // columns into dwords through one lane shuffle, lays them out as cells in its private LDS region and copies the cells
        // out with 16-byte stores (1 KB contiguous per instruction: the 64-byte cells of consecutive pixels are adjacent).
        char *stg = lds + kStageOff + wave * (4096 * S::sn);
        const int j = lane & 31, g = lane >> 5;
        const int *eo = ex + S::bm + S::bn;
        const long long chunk0 = (n0 + wn) / kBK;
#pragma unroll
        for (int sm = 0; sm < S::sm; ++sm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * g, tr = wm + 32 * sm + rr;
                const long long row = m0 + tr;
#pragma unroll
                for (int sn = 0; sn < S::sn; ++sn) {
                    float v = conv_epi(__builtin_ldexpf(acc.v[sm][sn][r], -(ex[tr] + ecol[sn])) + bcol[sn], p.epilogue);
                    if (row >= Mtot || n0 + wn + 32 * sn + j >= p.Cout) v = 0.f;
                    track(tr, row, v);
                    const float ys = __builtin_ldexpf(v, eo[tr]);
                    const _Float16 h1 = (_Float16)ys;
                    const _Float16 h2 = (_Float16)(ys - (float)h1);
                    const unsigned mine = (unsigned)__builtin_bit_cast(unsigned short, h1) | ((unsigned)__builtin_bit_cast(unsigned short, h2) << 16);
                    const unsigned other = (unsigned)__shfl_xor((int)mine, 1);
                    // even column lane: h1 of (me, right neighbour); odd column lane: h2 of (left neighbour, me)
                    const unsigned dw = (j & 1) ? ((other >> 16) | (mine & 0xffff0000u)) : ((mine & 0xffffu) | (other << 16));
                    char *cell = stg + ((2 * sn + (j >> 4)) * 32 + rr) * kCell;
                    *reinterpret_cast<unsigned *>(cell + ((j & 1) ? 32 : 0) + 4 * ((j & 15) >> 1)) = dw;
                }
                __builtin_amdgcn_sched_barrier(0);      // one row at a time: keeps the epilogue inside the main loop's register budget
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4 * S::sn; ++i) {
                const int q = lane + 64 * i, cc = q >> 7, rr = (q & 127) >> 2, c16 = q & 3;
                const long long row = m0 + wm + 32 * sm + rr, chunk = chunk0 + cc;
                if (row < Mtot && chunk * kBK < p.Cout)
                    *reinterpret_cast<u32x4 *>(p.out_cells + ((size_t)chunk * Mtot + row) * kCell + 16 * c16) =
                        *reinterpret_cast<const u32x4 *>(stg + (cc * 32 + rr) * kCell + 16 * c16);
            }
            __syncthreads();
        }
    }
    if (p.out_bits && b_hi - b_lo <= 1) {
        wave_atomic_max(p.out_bits, b_lo, vlo);
        if (b_hi > b_lo) wave_atomic_max(p.out_bits, b_hi, vhi);
    }
}


// ------------------------------------------------------------------------------------------------- the ring conv kernel
// Round 4: the same implicit GEMM on the ring loop of pl_ring.h (LDS-DMA staging: no staging registers, no ds_write;
// register double-buffered fragments; one barrier per k-tile with NS - 2 tiles in flight).  Images and scales are those of
// conv3x3_kernel above; what changes is how a k-tile reaches LDS:
//   * a lane owns R::na rows of the A tile (one per DMA piece) and keeps their 9-bit tap masks; for the k-tile of tap t a
//     row whose tap leaves the image is fetched out of range (the DMA writes zeros), selected per piece with one v_cndmask;
//   * padding steps of the unrolled ring (and steps past a K slice's end) carry tap bit 0: A = zeros, acc += 0 * B.
// Round 6:
//   * MODE 2 -- the 2x2 / 2 max-pool that follows the layer happens in the epilogue and the POOLED output leaves as the next
//     layer's activation image (conv1_2, conv2_2, conv3_3, conv4_3 wrote 1 GB of fp32 per step for a converter pass to read
//     back).  The rows of a tile are the pixels in POOL ORDER: row q = 4 * (pooled pixel) + 2 * dx + dy, so the four pixels of
//     a pooling window are four consecutive rows = four neighbouring lanes of one accumulator register (two DPP row maxima),
//     no window straddles a tile, and the pooled pixels of a tile are consecutive in the pooled image.  The order costs
//     nothing in the K loop: a lane's DMA source offset is free (dma_probe), a tap is still one scalar offset for the whole
//     tile.  max commutes with the (monotone) epilogue, so the maximum is taken on the raw accumulators and a lane then
//     finishes a QUARTER of its registers: bit-identical to pooling afterwards.
//   * K slices are added up INSIDE the launch: every slice leaves its raw accumulators in the workspace in register order
//     (1 KiB per store instruction), takes a ticket on the tile's arrival counter (agent-scope release in front of it), and the
//     block that draws the last ticket acquires, adds the slices in slice order (the result does not depend on the arrival
//     order) and runs the tile's ordinary epilogue.  No reduce launch, no second epilogue code path, and the schedule may cut
//     a launch that fills half the chip (conv5: 132 tiles) into slices without paying for a second pass.
constexpr int kRingStageOff = 8192;   // LDS offset of the ring kernel's image staging (exponent tables of up to 512 + 256 rows in front)
template <class R>
constexpr int ring_conv_lds_bytes(int mode)
{
    const int epi = kRingStageOff + R::waves * (mode == 1 ? 2 * 32 * 80 : (mode == 2 ? 2 * 8 * 80 : 32 * 36 * 4));      // wave-private staging patches
    return epi > R::lds_bytes ? epi : R::lds_bytes;
}
// floats one (tile, slice) leaves in the workspace: the block's accumulators in register order
template <class R>
constexpr size_t ring_partial_floats() { return (size_t)R::bm * R::bn; }

__device__ __forceinline__ float quad_max(float v)
{
    // maximum over the four lanes of a quad (DPP quad_perm [1,0,3,2], then [2,3,0,1]): every lane of the quad gets it
    float t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v = fmaxf(v, t);
    t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    return fmaxf(v, t);
}

template <class R, int MODE>      // MODE 0: fp32 NHWC output, 1: activation image, 2: 2x2-pooled activation image
__global__ __launch_bounds__(R::threads, (R::threads == 512 || R::sm * R::sn <= 8) ? 2 : 1) void conv3x3_ring_kernel(const ConvArgs p)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    constexpr bool IMG = MODE != 0, POOL = MODE == 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int wm, wn;
    rwave_origin<R>(wave, wm, wn);
    const int tail_blocks = p.tail_tiles * p.tail_slices;
    const bool is_tail = (int)blockIdx.x < tail_blocks;
    int t, slice, kt_per_slice, nslices;
    if (is_tail) {
        t = p.body_tiles + (int)blockIdx.x / p.tail_slices;
        slice = (int)blockIdx.x % p.tail_slices;
        kt_per_slice = p.tail_ktiles;
        nslices = p.tail_slices;
    } else {
        const int bb = (int)blockIdx.x - tail_blocks;
        t = xcd_remap(bb % p.body_tiles, p.body_tiles);
        slice = bb / p.body_tiles;
        kt_per_slice = p.ktiles_per_split;
        nslices = p.splitk;
    }
    const long long m0 = (long long)(t / p.tiles_n) * R::bm;
    const int n0 = (t % p.tiles_n) * R::bn;
    const long long HW = (long long)p.H * p.W, Mtot = (long long)p.B * HW;
    const unsigned uW = (unsigned)p.W, uHW = (unsigned)HW;                 // B*H*W < 2^25 (host-checked image size)
    const unsigned uWo = uW >> 1, uHoWo = uHW >> 2;
    const int G = p.Cin / kBK;
    const int total_kt = 9 * G;
    const int kt_begin = slice * kt_per_slice, kt_end = min(total_kt, kt_begin + kt_per_slice);

    // tile row -> (image, y, x) and the linear pixel index its cells are addressed by
    auto locate = [&](unsigned q, unsigned &py, unsigned &px) -> unsigned {
        if (!POOL) {
            const unsigned rem = q % uHW;
            py = rem / uW; px = rem - py * uW;
            return q;
        }
        const unsigned quad = q >> 2, b = quad / uHoWo, rem = quad - b * uHoWo, yo = rem / uWo, xo = rem - yo * uWo;
        py = 2 * yo + (q & 1); px = 2 * xo + ((q >> 1) & 1);
        return (b * (unsigned)p.H + py) * uW + px;
    };
    // A is addressed relative to one halo (W + 1 pixels) before the tile's first pixel (the smallest pixel index of the tile in
    // either order): every tap of every valid pixel has a non-negative offset
    const int halo = p.W + 1;
    unsigned py0, px0;
    const unsigned pix0 = locate((unsigned)min(m0, Mtot - 1), py0, px0);
    const Src sa = make_src(p.in + ((long long)pix0 - halo) * (long long)kCell), sb = make_src(p.wt + (size_t)n0 * kCell);
    DmaPlan<R> dp;
    plan_dma<R>(dp, [&](int) { return true; }, [&](int r) { return n0 + r < p.Cout; }, wave, lane);
    unsigned a_taps[R::na];
#pragma unroll
    for (int j = 0; j < R::na; ++j) {
        const int row = dma_row<R>(j, wave, lane);
        const bool ok = m0 + row < Mtot;
        unsigned py, px;
        const unsigned pix = locate(ok ? (unsigned)(m0 + row) : (unsigned)m0, py, px);
        unsigned mask = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            if (ok && (unsigned)((int)py + dy) < (unsigned)p.H && (unsigned)((int)px + dx) < (unsigned)p.W) mask |= 1u << tap;
        }
        a_taps[j] = mask;
        if (POOL) dp.va[j] = (pix - pix0) * (unsigned)kCell + 16u * (unsigned)((lane & 3) ^ swz(row));
    }
    FragPlan fp;
    rplan_frags<R>(fp, wm, wn, lane);
    const unsigned strideA = (unsigned)(Mtot * kCell), strideB = (unsigned)p.Cout * kCell;
    // Operand offsets of a k-tile WITHOUT a table: k-tiles run chunk-outer / tap-inner, K slices start at whole chunks
    // (host: multiples of 9 k-tiles) and the loop body is unrolled over a multiple of 9 steps, so a tile's tap is a
    // compile-time constant (`ui % 9`): the nine tap shifts and the nine per-tap weight offsets live in SGPRs, the chunk
    // offsets advance by one scalar add per chunk.  (A table in LDS, as in the round-3 kernel, cannot be used here: hipcc
    // orders any LDS read it cannot prove disjoint from an earlier `buffer_load ... lds` behind s_waitcnt vmcnt(0), so every
    // step waited for the DMA it had just issued -- gpurun r04_c2: conv2_2 251 TF/s against 310 for the round-3 loop.)
    unsigned tapoff[9], tapwt[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        tapoff[tap] = (unsigned)(halo + (tap / 3 - 1) * p.W + (tap % 3 - 1)) * kCell;
        tapwt[tap] = (unsigned)(tap * G) * strideB;
    }
    const int g0 = kt_begin / 9;                   // kt_begin is a multiple of 9
    unsigned ga = (unsigned)g0 * strideA - strideA, gb = (unsigned)g0 * strideB - strideB;      // advanced when a chunk's tap 0 is issued
    auto issue = [&](int kt, int ui, char *stage) {
        const int tap = ui % 9;                    // compile-time after unrolling (U % 9 == 0)
        if (tap == 0) { ga += strideA; gb += strideB; }
        const bool live = kt < kt_end;
        const unsigned bit = live ? (1u << tap) : 0u;                 // padding steps: A out of range (zeros) ...
        const unsigned ob = live ? gb + tapwt[tap] : 0u;              // ... and B = the first weight tile (finite values)
        unsigned va[R::na];
#pragma unroll
        for (int j = 0; j < R::na; ++j) va[j] = (a_taps[j] & bit) ? dp.va[j] : kOob;
        dma_stage<R>(sa, sb, va, dp.vb, ga + tapoff[tap], ob, stage, wave);
    };
    RAcc<R> acc;
    racc_zero<R>(acc);
    constexpr int U = (R::unroll % 9 == 0) ? R::unroll : ((R::unroll % 3 == 0) ? 3 * R::unroll : 9 * R::unroll);      // lcm(unroll, 9)
    ring_loop<R, U, 9>(issue, kt_begin, kt_end, lds, fp, acc);
    if (p.debug_flags & 2) {                    // measurement only: no epilogue (one store keeps the accumulators alive)
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < R::sm; ++i)
#pragma unroll
            for (int j = 0; j < R::sn; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc.v[i][j][r];
        if (s == 123.456f && p.out) p.out[0] = s;
        return;
    }

    // ---- K slices: leave the raw accumulators, take a ticket; the last block of the tile adds all slices (in slice order) and goes on
    if (nslices > 1) {
        const int t_local = is_tail ? t - p.body_tiles : t, region_tiles = is_tail ? p.tail_tiles : p.body_tiles;
        float *region = is_tail ? p.partial_tail : p.partial;
        const size_t tile_floats = ring_partial_floats<R>(), wave_floats = tile_floats / R::waves;
        float *mine = region + ((size_t)slice * region_tiles + t_local) * tile_floats + (size_t)wave * wave_floats;
#pragma unroll
        for (int sm = 0; sm < R::sm; ++sm)
#pragma unroll
            for (int sn = 0; sn < R::sn; ++sn)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4 *>(mine + ((size_t)((sm * R::sn + sn) * 4 + q) * 64 + lane) * 4) =
                        make_float4(acc.v[sm][sn][4 * q], acc.v[sm][sn][4 * q + 1], acc.v[sm][sn][4 * q + 2], acc.v[sm][sn][4 * q + 3]);
        int *ticket = reinterpret_cast<int *>(lds);
        int *counter = p.counters + (is_tail ? p.body_tiles : 0) + t_local;
        // every wave drains ITS stores before the barrier: hipcc puts no s_waitcnt vmcnt(0) in front of the s_barrier of
        // __syncthreads() (workgroup scope, non-tgsplit), so without this the other three waves' stores are still in flight
        // when wave 0 publishes the ticket (gpurun r06_c2: the last block added up stale partial sums)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            // drained stores -> barrier -> one agent-scope release -> drained -> relaxed ticket (MI355X_MICROARCH.md, valid producer form)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int got = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got == nslices - 1) {
                __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            *ticket = got;
        }
        __syncthreads();
        if (*ticket != nslices - 1) return;
        racc_zero<R>(acc);
        for (int z = 0; z < nslices; ++z) {
            const float *src = region + ((size_t)z * region_tiles + t_local) * tile_floats + (size_t)wave * wave_floats;
#pragma unroll
            for (int sm = 0; sm < R::sm; ++sm)
#pragma unroll
                for (int sn = 0; sn < R::sn; ++sn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4 *>(src + ((size_t)((sm * R::sn + sn) * 4 + q) * 64 + lane) * 4);
                        acc.v[sm][sn][4 * q] += v.x; acc.v[sm][sn][4 * q + 1] += v.y; acc.v[sm][sn][4 * q + 2] += v.z; acc.v[sm][sn][4 * q + 3] += v.w;
                    }
        }
        __syncthreads();                        // the ticket word is part of the tables below
    }

    // ---- epilogue.  The MFMAs ran with the operand roles SWAPPED (rmma: weights as the matrix-core A operand, pixels as B), so
    // accumulator (sm, sn) of lane (j = lane & 31, g = lane >> 5) holds, in registers 4 q .. 4 q + 3, FOUR CONSECUTIVE CHANNELS
    //     channel = wn + 32 sn + 8 q + 4 g + (0..3)     of tile row     wm + 32 sm + j
    // -- 16 contiguous bytes of an NHWC row, or a quarter of a plane-image cell.  Each 32 x 32 accumulator goes through a
    // wave-private LDS patch and leaves as 16-byte stores covering whole 128-byte lines (fp32) or runs of 64-byte cells (image).
    float *chan_f = reinterpret_cast<float *>(lds);             // bias[bn]
    int *chan_e = reinterpret_cast<int *>(lds) + R::bn;         // weight exponent per channel [bn]
    for (int i = tid; i < R::bn; i += R::threads) {
        const bool ok = n0 + i < p.Cout;
        chan_f[i] = (ok && p.bias) ? p.bias[n0 + i] : 0.f;
        chan_e[i] = ok ? row_exponent(p.wt_bits[n0 + i]) : 0;
    }
    float bmax = 0.f;
    if (IMG) {
        unsigned m = 0;
        if (p.bias)
            for (int n = tid; n < p.Cout; n += R::threads) m = max(m, __float_as_uint(p.bias[n]) & 0x7fffffffu);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        unsigned *red = reinterpret_cast<unsigned *>(lds) + 2 * R::bn;
        if (lane == 0) red[wave] = m;
        __syncthreads();
        unsigned mm = 0;
#pragma unroll
        for (int w = 0; w < R::waves; ++w) mm = max(mm, red[w]);
        bmax = __uint_as_float(mm);
    }
    auto bound_bits = [&](int b) { return __float_as_uint(__uint_as_float(p.in_true_bits[b]) * p.wnorm[0] + bmax); };
    // the scale words of the output image: written by the block that finishes tile 0 (with K slices that is the tile's LAST block,
    // whichever it is -- block 0 may have left after its partial sums)
    if (IMG && t == 0 && tid < p.B) p.out_scale_bits[tid] = bound_bits(tid);
    __syncthreads();
    static_assert((2 * R::bn + 16) * 4 <= kRingStageOff, "channel tables must fit in front of the staging patches");
    const int j = lane & 31, g = lane >> 5;
    unsigned vmax[R::sm];
    int img_of[R::sm];
    if (MODE == 0) {
        float *stg = reinterpret_cast<float *>(lds + kRingStageOff) + wave * (32 * 36);      // [32 pixels][32 channels + 4 pad]
#pragma unroll
        for (int sm = 0; sm < R::sm; ++sm) {
            const long long row = m0 + wm + 32 * sm + j;
            const bool rok = row < Mtot;
            const int b = rok ? (int)((unsigned)row / uHW) : 0;
            const int ea = rok ? row_exponent(p.in_bits[b]) : 0;
            img_of[sm] = b;
            unsigned vm = 0;
#pragma unroll
            for (int sn = 0; sn < R::sn; ++sn) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = wn + 32 * sn + 8 * q + 4 * g;
                    const int4 eb = *reinterpret_cast<const int4 *>(chan_e + c);
                    const float4 bs = *reinterpret_cast<const float4 *>(chan_f + c);
                    float4 v;
                    v.x = conv_epi(__builtin_ldexpf(acc.v[sm][sn][4 * q + 0], -(ea + eb.x)) + bs.x, p.epilogue);
                    v.y = conv_epi(__builtin_ldexpf(acc.v[sm][sn][4 * q + 1], -(ea + eb.y)) + bs.y, p.epilogue);
                    v.z = conv_epi(__builtin_ldexpf(acc.v[sm][sn][4 * q + 2], -(ea + eb.z)) + bs.z, p.epilogue);
                    v.w = conv_epi(__builtin_ldexpf(acc.v[sm][sn][4 * q + 3], -(ea + eb.w)) + bs.w, p.epilogue);
                    if (rok) {                      // channels >= Cout carry zero weights and zero bias: 0 after ReLU, |0| otherwise
                        vm = max(vm, max(max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu),
                                         max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)));
                    }
                    *reinterpret_cast<float4 *>(stg + j * 36 + 8 * q + 4 * g) = v;
                }
                // the patch leaves row-major: lane -> row 8 i + lane / 8, channels 4 (lane % 8) .. + 3: eight 128-byte lines per store
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rr = 8 * i + (lane >> 3), c4 = 4 * (lane & 7);
                    const long long orow = m0 + wm + 32 * sm + rr;
                    const int col = n0 + wn + 32 * sn + c4;
                    const float4 v = *reinterpret_cast<const float4 *>(stg + rr * 36 + c4);
                    if (orow < Mtot && col < p.Cout) *reinterpret_cast<float4 *>(p.out + (size_t)orow * p.Cout + col) = v;
                }
                __builtin_amdgcn_sched_barrier(0);       // one accumulator at a time: keeps the epilogue inside the loop's register budget
            }
            vmax[sm] = vm;
        }
    } else if (MODE == 1) {
        // image output: a lane's four channels are 8 bytes of the cell's h1 half and 8 bytes of its h2 half; the wave assembles the
        // 2 chunks x 32 pixels of a 32 x 32 accumulator in an LDS patch with an 80-byte cell pitch (ds_write_b64 two-way instead of
        // eight-way conflicts) and copies the cells out as 1 KiB runs (the 64-byte cells of consecutive pixels are adjacent)
        constexpr int kPitch = 80;
        char *stg = lds + kRingStageOff + wave * (2 * 32 * kPitch);
        const long long chunk0 = (n0 + wn) / kBK;
        // Round 6: the value is formed directly at the OUTPUT image's scale -- xs = clamp(acc 2^(eo - ea - eb) + bias 2^eo), the same
        // bits as (clamp(acc 2^-(ea + eb) + bias)) 2^eo because a power-of-two factor commutes with the rounding of the sum -- which
        // drops the second ldexp of every element, and rows / channels outside the layer need no masks: their accumulators and
        // bias entries are zero and their rows' factor 2^eo is replaced by 0.  The epilogue's cost is its VALU instruction count
        // (gpurun r06_c3: 0.085 / 0.05 / 0.027 ms per 2054 tiles for this / the fp32 / the pooled epilogue at ~330 / 200 / 110
        // instructions per accumulator).
        const float lo = (p.epilogue == MH_EPI_NONE) ? -__builtin_inff() : 0.f, hi6 = (p.epilogue == MH_EPI_RELU6) ? 6.f : __builtin_inff();
#pragma unroll
        for (int sm = 0; sm < R::sm; ++sm) {
            const long long row = m0 + wm + 32 * sm + j;
            const bool rok = row < Mtot;
            const int b = rok ? (int)((unsigned)row / uHW) : 0;
            const int eo = rok ? row_exponent(bound_bits(b)) : 0;
            const int ek = eo - (rok ? row_exponent(p.in_bits[b]) : 0);
            const float so = rok ? __builtin_ldexpf(1.f, eo) : 0.f;
            const float hi = (p.epilogue == MH_EPI_RELU6) ? hi6 * so : hi6;
            img_of[sm] = b;
            unsigned vm = 0;
#pragma unroll
            for (int sn = 0; sn < R::sn; ++sn) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = wn + 32 * sn + 8 * q + 4 * g;
                    const int4 eb = *reinterpret_cast<const int4 *>(chan_e + c);
                    const float4 bs = *reinterpret_cast<const float4 *>(chan_f + c);
                    const float x0 = __builtin_amdgcn_fmed3f(__builtin_fmaf(bs.x, so, __builtin_ldexpf(acc.v[sm][sn][4 * q + 0], ek - eb.x)), lo, hi);
                    const float x1 = __builtin_amdgcn_fmed3f(__builtin_fmaf(bs.y, so, __builtin_ldexpf(acc.v[sm][sn][4 * q + 1], ek - eb.y)), lo, hi);
                    const float x2 = __builtin_amdgcn_fmed3f(__builtin_fmaf(bs.z, so, __builtin_ldexpf(acc.v[sm][sn][4 * q + 2], ek - eb.z)), lo, hi);
                    const float x3 = __builtin_amdgcn_fmed3f(__builtin_fmaf(bs.w, so, __builtin_ldexpf(acc.v[sm][sn][4 * q + 3], ek - eb.w)), lo, hi);
                    vm = max(vm, max(max(__float_as_uint(x0) & 0x7fffffffu, __float_as_uint(x1) & 0x7fffffffu),
                                     max(__float_as_uint(x2) & 0x7fffffffu, __float_as_uint(x3) & 0x7fffffffu)));
                    unsigned a1, a2, b1, b2;
                    split2_scaled(x0, x1, a1, a2);
                    split2_scaled(x2, x3, b1, b2);
                    char *cell = stg + ((q >> 1) * 32 + j) * kPitch + 16 * (q & 1) + 8 * g;
                    *reinterpret_cast<u32x2 *>(cell) = (u32x2){a1, b1};
                    *reinterpret_cast<u32x2 *>(cell + 32) = (u32x2){a2, b2};
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {       // instruction i: chunk i / 2, pixels 16 (i % 2) .. + 15: one contiguous KiB
                    const int cc = i >> 1, rr = 16 * (i & 1) + (lane >> 2), c16 = lane & 3;
                    const long long orow = m0 + wm + 32 * sm + rr, chunk = chunk0 + 2 * sn + cc;
                    const u32x4 cellv = *reinterpret_cast<const u32x4 *>(stg + (cc * 32 + rr) * kPitch + 16 * c16);
                    if (orow < Mtot && chunk * kBK < p.Cout)
                        *reinterpret_cast<u32x4 *>(p.out_cells + ((size_t)chunk * Mtot + orow) * kCell + 16 * c16) = cellv;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the true maximum of the layer's output: the largest scaled value, scale taken off (exact)
            vmax[sm] = __float_as_uint(__builtin_ldexpf(__uint_as_float(vm), -eo));
        }
    } else {
        // pooled image output (see the header): quad maxima of the raw accumulators, then lane (j, g) finishes register quad
        // q = j & 3 of pooled pixel j >> 2 -- channels 32 sn + 8 (j & 3) + 4 g + (0..3) -- so every lane scales, clamps and splits
        // four values per accumulator instead of sixteen; 2 chunks x 8 pooled pixels of cells per accumulator leave as ONE 16-byte
        // store per lane (two 512-byte runs)
        constexpr int kPitch = 80;
        char *stg = lds + kRingStageOff + wave * (2 * 8 * kPitch);
        const long long chunk0 = (n0 + wn) / kBK, Mo = Mtot >> 2;
        const int within = j & 3, pl = j >> 2;
        const float lo = (p.epilogue == MH_EPI_NONE) ? -__builtin_inff() : 0.f, hi6 = (p.epilogue == MH_EPI_RELU6) ? 6.f : __builtin_inff();
#pragma unroll
        for (int sm = 0; sm < R::sm; ++sm) {
            const long long row = m0 + wm + 32 * sm + j;
            const bool rok = row < Mtot;                                   // H and W are even: a window is inside or outside as a whole
            const int b = rok ? (int)((unsigned)(row >> 2) / uHoWo) : 0;
            const int eo = rok ? row_exponent(bound_bits(b)) : 0;
            const int ek = eo - (rok ? row_exponent(p.in_bits[b]) : 0);
            const float so = rok ? __builtin_ldexpf(1.f, eo) : 0.f;
            const float hi = (p.epilogue == MH_EPI_RELU6) ? hi6 * so : hi6;
            img_of[sm] = b;
            unsigned vm = 0;
#pragma unroll
            for (int sn = 0; sn < R::sn; ++sn) {
                float x[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float m0q = quad_max(acc.v[sm][sn][i]), m1q = quad_max(acc.v[sm][sn][4 + i]);
                    const float m2q = quad_max(acc.v[sm][sn][8 + i]), m3q = quad_max(acc.v[sm][sn][12 + i]);
                    x[i] = (within & 2) ? ((within & 1) ? m3q : m2q) : ((within & 1) ? m1q : m0q);
                }
                const int c = wn + 32 * sn + 8 * within + 4 * g;
                const int4 eb = *reinterpret_cast<const int4 *>(chan_e + c);
                const float4 bs = *reinterpret_cast<const float4 *>(chan_f + c);
                const float x0 = __builtin_amdgcn_fmed3f(__builtin_fmaf(bs.x, so, __builtin_ldexpf(x[0], ek - eb.x)), lo, hi);
                const float x1 = __builtin_amdgcn_fmed3f(__builtin_fmaf(bs.y, so, __builtin_ldexpf(x[1], ek - eb.y)), lo, hi);
                const float x2 = __builtin_amdgcn_fmed3f(__builtin_fmaf(bs.z, so, __builtin_ldexpf(x[2], ek - eb.z)), lo, hi);
                const float x3 = __builtin_amdgcn_fmed3f(__builtin_fmaf(bs.w, so, __builtin_ldexpf(x[3], ek - eb.w)), lo, hi);
                vm = max(vm, max(max(__float_as_uint(x0) & 0x7fffffffu, __float_as_uint(x1) & 0x7fffffffu),
                                 max(__float_as_uint(x2) & 0x7fffffffu, __float_as_uint(x3) & 0x7fffffffu)));
                unsigned a1, a2, b1, b2;
                split2_scaled(x0, x1, a1, a2);
                split2_scaled(x2, x3, b1, b2);
                char *cell = stg + ((within >> 1) * 8 + pl) * kPitch + 16 * (within & 1) + 8 * g;
                *reinterpret_cast<u32x2 *>(cell) = (u32x2){a1, b1};
                *reinterpret_cast<u32x2 *>(cell + 32) = (u32x2){a2, b2};
                {
                    const int cc = lane >> 5, pp = (lane >> 2) & 7, c16 = lane & 3;
                    const long long orow = ((m0 + wm + 32 * sm) >> 2) + pp, chunk = chunk0 + 2 * sn + cc;
                    const u32x4 cellv = *reinterpret_cast<const u32x4 *>(stg + (cc * 8 + pp) * kPitch + 16 * c16);
                    if (orow < Mo && chunk * kBK < p.Cout)
                        *reinterpret_cast<u32x4 *>(p.out_cells + ((size_t)chunk * Mo + orow) * kCell + 16 * c16) = cellv;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            vmax[sm] = __float_as_uint(__builtin_ldexpf(__uint_as_float(vm), -eo));
        }
    }
    if (p.out_bits) {
#pragma unroll
        for (int sm = 0; sm < R::sm; ++sm) wave_atomic_max(p.out_bits, img_of[sm], vmax[sm]);
    }
}

// C[rows][N] = epi(sum_z partial[z] + bias), and the per-image maxima of what is written (rows start at row0 of the layer)
__global__ __launch_bounds__(256) void reduce_kernel(const float *__restrict__ partial, int nslices, long long rows, int N, float *__restrict__ C,
                                                     const float *__restrict__ bias, int epilogue, long long row0, long long HW,
                                                     unsigned *__restrict__ out_bits)
{
    const long long total = rows * N;
    const size_t plane = (size_t)rows * N;
    int cur = -1;
    unsigned vmax = 0;
    // every block owns ONE contiguous range of elements, so a thread sees its image change at most a few times (a grid-stride
    // loop jumps ~1000 rows per iteration: at 37x37 that is a new image -- and a flush atomic -- almost every iteration)
    const long long per_block = ((total + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const long long lo = blockIdx.x * per_block, hi = min(total, lo + per_block);
    for (long long idx = lo + threadIdx.x; idx < hi; idx += 256) {
        const long long row = idx / N;
        const int col = (int)(idx % N);
        float v = 0.f;
        for (int z = 0; z < nslices; ++z) v += partial[z * plane + idx];
        if (bias) v += bias[col];
        v = conv_epi(v, epilogue);
        C[idx] = v;
        const int b = (int)((row0 + row) / HW);
        if (b != cur) {
            if (cur >= 0 && vmax && out_bits && may_raise(out_bits + cur, vmax)) atomicMax(out_bits + cur, vmax);
            cur = b;
            vmax = 0;
        }
        vmax = max(vmax, __float_as_uint(v) & 0x7fffffffu);
    }
    if (out_bits) wave_atomic_max(out_bits, cur < 0 ? 0 : cur, cur < 0 ? 0u : vmax);
}

// the same reduction with an IMAGE output: thread = (row, 16-channel chunk), 256 consecutive rows of one chunk per block step
// (64-byte reads per slice, 16 KB contiguous writes); the image's exponents come from the bound the conv kernel's block 0
// left in scale_bits (stream order)
__global__ __launch_bounds__(256) void reduce_planes_kernel(const float *__restrict__ partial, int nslices, long long rows, int N,
                                                            char *__restrict__ out_cells, long long Mtot, const float *__restrict__ bias,
                                                            int epilogue, long long row0, long long HW,
                                                            const unsigned *__restrict__ scale_bits, unsigned *__restrict__ out_bits)
{
    const int G = N / kBK;
    const size_t plane = (size_t)rows * N;
    const long long nblk_m = (rows + 255) / 256;
    int cur = -1;
    unsigned vmax = 0;
    const long long per_block = (nblk_m * G + gridDim.x - 1) / gridDim.x;
    for (long long blk = blockIdx.x * per_block; blk < min(nblk_m * G, (blockIdx.x + 1) * per_block); ++blk) {
        const int g = (int)(blk / nblk_m);
        const long long r = (blk % nblk_m) * 256 + threadIdx.x;
        if (r >= rows) continue;
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = bias ? bias[g * kBK + i] : 0.f;
        for (int z = 0; z < nslices; ++z) {
            const float4 *q = reinterpret_cast<const float4 *>(partial + z * plane + (size_t)r * N + g * kBK);
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float4 a = q[i]; v[4 * i] += a.x; v[4 * i + 1] += a.y; v[4 * i + 2] += a.z; v[4 * i + 3] += a.w; }
        }
        const int b = (int)((row0 + r) / HW);
        if (b != cur) {
            if (cur >= 0 && vmax && out_bits && may_raise(out_bits + cur, vmax)) atomicMax(out_bits + cur, vmax);
            cur = b;
            vmax = 0;
        }
        const int e = row_exponent(scale_bits[b]);
        unsigned h1[8], h2[8];
#pragma unroll
        for (int i = 0; i < 16; ++i) { v[i] = conv_epi(v[i], epilogue); vmax = max(vmax, __float_as_uint(v[i]) & 0x7fffffffu); }
#pragma unroll
        for (int i = 0; i < 8; ++i) split2(v[2 * i], v[2 * i + 1], e, h1[i], h2[i]);
        u32x4 *dst = reinterpret_cast<u32x4 *>(out_cells + ((size_t)g * Mtot + row0 + r) * kCell);
        dst[0] = (u32x4){h1[0], h1[1], h1[2], h1[3]};
        dst[1] = (u32x4){h1[4], h1[5], h1[6], h1[7]};
        dst[2] = (u32x4){h2[0], h2[1], h2[2], h2[3]};
        dst[3] = (u32x4){h2[4], h2[5], h2[6], h2[7]};
    }
    if (out_bits) wave_atomic_max(out_bits, cur < 0 ? 0 : cur, cur < 0 ? 0u : vmax);
}

// ------------------------------------------------------------------------------------------------- the stem: conv1_1
// NCHW image (Cin = 3) -> activation IMAGE of the 64-channel output, bias + ReLU fused: the layer is write-bound (1.2 GFLOP
// against 90 MB written per 592x592 image), so it stays on the VALU -- one thread = one pixel x one 16-channel chunk = exactly
// one 64-byte cell -- but it now writes what conv1_2 reads (no fp32 tensor, no converter pass).  Weights live in LDS as
// [tap * Cin][Cout]; the image exponents come from the bound max|x_b| * max_n sum|w_n| + max|bias| (in_bits = per-image
// largest |pixel|, from a row-maxima pass over the 25 MB input).
__global__ __launch_bounds__(256) void stem_kernel(const float *__restrict__ in, int B, int Cin, int H, int W, const float *__restrict__ w, int Cout,
                                                   const float *__restrict__ bias, int epilogue, const unsigned *__restrict__ in_bits,
                                                   char *__restrict__ out_cells, unsigned *__restrict__ scale_bits, unsigned *__restrict__ out_bits)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [9*Cin][Cout] + bias[Cout] + colsum[Cout]
    const int K = 9 * Cin;
    for (int i = threadIdx.x; i < K * Cout; i += blockDim.x) {
        const int co = i % Cout, k = i / Cout, tap = k / Cin, ci = k % Cin;
        wl[i] = w[((size_t)co * Cin + ci) * 9 + tap];
    }
    float *bl = wl + K * Cout, *cs = bl + Cout;
    for (int i = threadIdx.x; i < Cout; i += blockDim.x) bl[i] = bias ? bias[i] : 0.f;
    __syncthreads();
    for (int co = threadIdx.x; co < Cout; co += blockDim.x) {
        float a = 0.f;
        for (int k = 0; k < K; ++k) a += fabsf(wl[k * Cout + co]);
        cs[co] = a;
    }
    __syncthreads();
    float wnorm = 0.f, bmax = 0.f;
    for (int co = 0; co < Cout; ++co) { wnorm = fmaxf(wnorm, cs[co]); bmax = fmaxf(bmax, fabsf(bl[co])); }
    auto bound_bits = [&](int b) { return __float_as_uint(__uint_as_float(in_bits[b]) * wnorm + bmax); };
    if (blockIdx.x == 0 && (int)threadIdx.x < B) scale_bits[threadIdx.x] = bound_bits(threadIdx.x);
    const int G = Cout / kBK;
    const long long M = (long long)B * H * W, nblk_m = (M + 255) / 256;
    int cur = -1, e = 0;
    unsigned vmax = 0;
    const long long per_block = (nblk_m * G + gridDim.x - 1) / gridDim.x;
    for (long long blk = blockIdx.x * per_block; blk < min(nblk_m * G, (blockIdx.x + 1) * per_block); ++blk) {
        const int g = (int)(blk % G);                       // the G chunks of a pixel block run back to back: inputs hit L1
        const long long pix = (blk / G) * 256 + threadIdx.x;
        if (pix >= M) continue;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
        if (b != cur) {
            if (cur >= 0 && vmax && out_bits && may_raise(out_bits + cur, vmax)) atomicMax(out_bits + cur, vmax);
            cur = b;
            vmax = 0;
            e = row_exponent(bound_bits(b));
        }
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = bl[g * kBK + j];
        for (int ci = 0; ci < Cin; ++ci) {
            const float *plane = in + ((size_t)b * Cin + ci) * H * W;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                float v = 0.f;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = plane[(size_t)yy * W + xx];
                const float *wr = wl + (tap * Cin + ci) * Cout + g * kBK;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = fmaf(v, wr[j], acc[j]);
            }
        }
        unsigned h1[8], h2[8];
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc[j] = conv_epi(acc[j], epilogue); vmax = max(vmax, __float_as_uint(acc[j]) & 0x7fffffffu); }
#pragma unroll
        for (int j = 0; j < 8; ++j) split2(acc[2 * j], acc[2 * j + 1], e, h1[j], h2[j]);
        u32x4 *dst = reinterpret_cast<u32x4 *>(out_cells + ((size_t)g * M + pix) * kCell);
        dst[0] = (u32x4){h1[0], h1[1], h1[2], h1[3]};
        dst[1] = (u32x4){h1[4], h1[5], h1[6], h1[7]};
        dst[2] = (u32x4){h2[0], h2[1], h2[2], h2[3]};
        dst[3] = (u32x4){h2[4], h2[5], h2[6], h2[7]};
    }
    if (out_bits) wave_atomic_max(out_bits, cur < 0 ? 0 : cur, cur < 0 ? 0u : vmax);
}

// Round 6: the same layer on the matrix cores.  The VALU kernel above spends 0.40-0.43 ms on 1.2 GFLOP per image -- 1728 FMAs per
// pixel behind one LDS weight read per four of them -- against a 0.11 ms write floor.  As a product it is [pixels] x [27 -> 32] x
// [64]: two 16-k tiles, 24 MFMAs per wave and 256-pixel tile.  A block builds the im2col rows of its tile IN LDS (thread = pixel:
// 27 cached loads, scaled by the image's exponent and split into the two f16 planes: ~200 VALU instructions instead of 1728), the
// weights' plane tile is built once per block, the accumulators come out transposed (rmma's operand order) and leave through the
// image epilogue of the ring conv kernel (value formed at the output scale, cells assembled in a wave-private LDS patch, 1 KiB
// stores).  fp32 evaluation as everywhere (f16x3); scale words and true maxima as the VALU kernel writes them.
struct StemArgs {
    const float *in;            // [B][Cin][H][W]
    int B, Cin, H, W;
    const float *w;             // [64][Cin][3][3]
    const float *bias;
    int epilogue;
    const unsigned *in_bits;    // [B] largest |pixel| per image
    char *out_cells;            // [4][B*H*W][64]
    unsigned *scale_bits, *out_bits;
};
constexpr int kStemA = 2 * 256 * kCell, kStemB = 2 * 64 * kCell, kStemPatch = 2 * 32 * 80;
constexpr int kStemTab = 1280;      // bias[64] | eb[64] | reduction words [128] | per-image input / output exponents [32 + 32]
constexpr int kStemLds = kStemA + kStemB + kStemTab + 4 * kStemPatch;       // A tile | B tile | bias[64], eb[64], reduction words, per-image exponents | patches

__global__ __launch_bounds__(256, 2) void stem_mfma_kernel(const StemArgs p)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    char *At = lds, *Bt = lds + kStemA;
    float *bias_l = reinterpret_cast<float *>(lds + kStemA + kStemB);
    int *eb_l = reinterpret_cast<int *>(bias_l) + 64;
    float *red = bias_l + 128;                                            // [0..63] sum |w_n|, [64..127] |bias_n|
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = 9 * p.Cin;                                              // <= 27 (host-checked), padded to 32
    // ---- weights: thread (n = tid & 63, part = tid >> 6) writes 16-byte chunk `part` (plane part >> 1, k half part & 1) of both k-chunks
    {
        const int n = tid & 63, part = tid >> 6;
        const float *wr = p.w + (size_t)n * K;                            // w[n][ci][tap]: k = ci * 9 + tap is the row's memory order
        float wv[32];
        unsigned m = 0;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            wv[k] = k < K ? wr[k] : 0.f;
            m = max(m, __float_as_uint(wv[k]) & 0x7fffffffu);
            sum += fabsf(wv[k]);
        }
        const int e = row_exponent(m);
        if (part == 0) {
            eb_l[n] = e;
            bias_l[n] = p.bias ? p.bias[n] : 0.f;
            red[n] = sum;
            red[64 + n] = p.bias ? fabsf(p.bias[n]) : 0.f;
        }
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            unsigned d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = 16 * kc + 8 * (part & 1) + 2 * i;
                unsigned h1, h2;
                split2(wv[k], wv[k + 1], e, h1, h2);
                d[i] = (part >> 1) ? h2 : h1;
            }
            *reinterpret_cast<u32x4 *>(Bt + kc * (64 * kCell) + lds_chunk(n, part)) = (u32x4){d[0], d[1], d[2], d[3]};
        }
    }
    __syncthreads();
    float wnorm = 0.f, bmax = 0.f;
    for (int n = 0; n < 64; ++n) { wnorm = fmaxf(wnorm, red[n]); bmax = fmaxf(bmax, red[64 + n]); }
    wnorm *= 1.0001f;                                                     // a hair above the rounded sum (as weight_norm_kernel)
    auto bound_bits = [&](int b) { return __float_as_uint(__uint_as_float(p.in_bits[b]) * wnorm + bmax); };
    if (blockIdx.x == 0 && tid < p.B) p.scale_bits[tid] = bound_bits(tid);
    const unsigned uW = (unsigned)p.W, uHW = (unsigned)(p.H * p.W);
    const long long M = (long long)p.B * p.H * p.W;
    const long long ntiles = (M + 255) / 256;
    // per-image exponents in LDS (B <= 32): input scale, output scale -- looked up per row below instead of a dependent global load
    int *ea_l = reinterpret_cast<int *>(red) + 128, *eo_l = ea_l + 32;
    if (tid < 32) {
        const bool in = tid < p.B;
        ea_l[tid] = in ? row_exponent(p.in_bits[tid]) : 0;
        eo_l[tid] = in ? row_exponent(bound_bits(tid)) : 0;
    }
    __syncthreads();
    const int j = lane & 31, g = lane >> 5;
    const int wm0 = wave * 64;
    unsigned fa[2], fb[2];                                                // fragment offsets [plane], as rplan_frags
#pragma unroll
    for (int pl_ = 0; pl_ < 2; ++pl_) { fa[pl_] = lds_chunk(wm0 + j, 2 * pl_ + g); fb[pl_] = lds_chunk(j, 2 * pl_ + g); }
    char *stg = lds + kStemA + kStemB + kStemTab + wave * kStemPatch;
    const float lo = (p.epilogue == MH_EPI_NONE) ? -__builtin_inff() : 0.f, hi6 = (p.epilogue == MH_EPI_RELU6) ? 6.f : __builtin_inff();
    int cur = -1;
    unsigned vmax = 0;                                                    // running true maximum of image `cur` (this lane)
    // the 27 (padded: 32) taps of pixel m0 + tid, raw: loaded one tile AHEAD (the loads of tile t + 1 fly under the MFMAs and the
    // epilogue of tile t; the first version waited for them at the top of every tile: 21 us per tile and block, gpurun r06_c6)
    float raw[32];
    int raw_b = 0;
    auto fetch = [&](long long tile) {
        const long long m = tile * 256 + tid;
        const bool ok = tile < ntiles && m < M;
        const unsigned b = ok ? (unsigned)m / uHW : 0, rem = ok ? (unsigned)m - b * uHW : 0, y = rem / uW, x = rem - y * uW;
        const float *img = p.in + (size_t)b * p.Cin * uHW;
        raw_b = (int)b;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const int ci = k / 9, tap = k - 9 * ci, dy = tap / 3 - 1, dx = tap - 3 * (tap / 3) - 1;
            const int yy = (int)y + dy, xx = (int)x + dx;
            raw[k] = (ok && k < K && (unsigned)yy < (unsigned)p.H && (unsigned)xx < uW) ? img[(size_t)ci * uHW + (unsigned)yy * uW + (unsigned)xx] : 0.f;
        }
    };
    fetch(blockIdx.x);
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long m0 = tile * 256;
        // ---- im2col row of pixel m0 + tid, scaled by its image's exponent, split into the two planes
        {
            const int ea = ea_l[raw_b];
            unsigned h1p[16], h2p[16];
#pragma unroll
            for (int kp = 0; kp < 16; ++kp) split2(raw[2 * kp], raw[2 * kp + 1], ea, h1p[kp], h2p[kp]);
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned *src = (c >> 1) ? h2p : h1p;
                    const int q0 = 8 * kc + 4 * (c & 1);
                    *reinterpret_cast<u32x4 *>(At + kc * (256 * kCell) + lds_chunk(tid, c)) = (u32x4){src[q0], src[q0 + 1], src[q0 + 2], src[q0 + 3]};
                }
        }
        __syncthreads();
        fetch(tile + gridDim.x);
        // ---- 2 k-tiles x 3 terms x (2 x 2) accumulators, operand roles swapped as in rmma (transposed accumulators)
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            f16x8 a[2][2], bq[2][2];
#pragma unroll
            for (int sm = 0; sm < 2; ++sm)
#pragma unroll
                for (int pl_ = 0; pl_ < 2; ++pl_) a[sm][pl_] = *reinterpret_cast<const f16x8 *>(At + kc * (256 * kCell) + fa[pl_] + 2048 * sm);
#pragma unroll
            for (int sn = 0; sn < 2; ++sn)
#pragma unroll
                for (int pl_ = 0; pl_ < 2; ++pl_) bq[sn][pl_] = *reinterpret_cast<const f16x8 *>(Bt + kc * (64 * kCell) + fb[pl_] + 2048 * sn);
            constexpr int kTermA[3] = {1, 0, 0}, kTermB[3] = {0, 1, 0};
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int sm = 0; sm < 2; ++sm)
#pragma unroll
                    for (int sn = 0; sn < 2; ++sn)
                        acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bq[sn][kTermB[t]], a[sm][kTermA[t]], acc[sm][sn], 0, 0, 0);
        }
        __syncthreads();                                                  // the A tile is free for the next tile's rows
        // ---- image epilogue (conv3x3_ring_kernel, MODE 1): lane (j, g), registers 4 q .. 4 q + 3 = channels 32 sn + 8 q + 4 g + (0..3)
        // of pixel wm0 + 32 sm + j
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) {
            const long long row = m0 + wm0 + 32 * sm + j;
            const bool rok = row < M;
            const int b = rok ? (int)((unsigned)row / uHW) : 0;
            const int eo = rok ? eo_l[b] : 0;
            const int ek = eo - (rok ? ea_l[b] : 0);
            const float so = rok ? __builtin_ldexpf(1.f, eo) : 0.f;
            const float hi = (p.epilogue == MH_EPI_RELU6) ? hi6 * so : hi6;
            if (rok && b != cur) {
                if (cur >= 0 && vmax && p.out_bits && may_raise(p.out_bits + cur, vmax)) atomicMax(p.out_bits + cur, vmax);
                cur = b;
                vmax = 0;
            }
            unsigned vm = 0;
#pragma unroll
            for (int sn = 0; sn < 2; ++sn) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 32 * sn + 8 * q + 4 * g;
                    const int4 eb = *reinterpret_cast<const int4 *>(eb_l + c);
                    const float4 bs = *reinterpret_cast<const float4 *>(bias_l + c);
                    const float x0 = __builtin_amdgcn_fmed3f(__builtin_fmaf(bs.x, so, __builtin_ldexpf(acc[sm][sn][4 * q + 0], ek - eb.x)), lo, hi);
                    const float x1 = __builtin_amdgcn_fmed3f(__builtin_fmaf(bs.y, so, __builtin_ldexpf(acc[sm][sn][4 * q + 1], ek - eb.y)), lo, hi);
                    const float x2 = __builtin_amdgcn_fmed3f(__builtin_fmaf(bs.z, so, __builtin_ldexpf(acc[sm][sn][4 * q + 2], ek - eb.z)), lo, hi);
                    const float x3 = __builtin_amdgcn_fmed3f(__builtin_fmaf(bs.w, so, __builtin_ldexpf(acc[sm][sn][4 * q + 3], ek - eb.w)), lo, hi);
                    vm = max(vm, max(max(__float_as_uint(x0) & 0x7fffffffu, __float_as_uint(x1) & 0x7fffffffu),
                                     max(__float_as_uint(x2) & 0x7fffffffu, __float_as_uint(x3) & 0x7fffffffu)));
                    unsigned a1, a2, b1, b2;
                    split2_scaled(x0, x1, a1, a2);
                    split2_scaled(x2, x3, b1, b2);
                    char *cell = stg + ((q >> 1) * 32 + j) * 80 + 16 * (q & 1) + 8 * g;
                    *reinterpret_cast<u32x2 *>(cell) = (u32x2){a1, b1};
                    *reinterpret_cast<u32x2 *>(cell + 32) = (u32x2){a2, b2};
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cc = i >> 1, rr = 16 * (i & 1) + (lane >> 2), c16 = lane & 3;
                    const long long orow = m0 + wm0 + 32 * sm + rr;
                    const u32x4 cellv = *reinterpret_cast<const u32x4 *>(stg + (cc * 32 + rr) * 80 + 16 * c16);
                    if (orow < M) *reinterpret_cast<u32x4 *>(p.out_cells + ((size_t)(2 * sn + cc) * M + orow) * kCell + 16 * c16) = cellv;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (rok) vmax = max(vmax, __float_as_uint(__builtin_ldexpf(__uint_as_float(vm), -eo)));
        }
    }
    if (p.out_bits) wave_atomic_max(p.out_bits, cur, cur < 0 ? 0u : vmax);      // (a lane that never saw a pixel carries no value: its key is not used)
}

// (Round 5 built a second form of the VALU kernel -- weights as SGPR operands from a packed table read with s_load_dwordx16, inputs
// in a per-thread LDS column, v_pk_fma_f32 with an SGPR pair, no LDS weight reads: bit-identical output -- and measured it in the
// step against this one on the same boxes: 0.51 / 0.54 ms (two / one pixel per thread) against 0.42-0.43 ms.  Removed;
// profiles/r05_bench_c5_*.json, r05_bench_c6_*.json.)

// largest |x| per image of a [B][n] fp32 tensor (bits[B] zero on entry): grid (chunks, B)
__global__ __launch_bounds__(256) void image_absmax_kernel(const float *__restrict__ x, long long n, unsigned *__restrict__ bits)
{
    const float *p = x + (size_t)blockIdx.y * n;
    unsigned m = 0;
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {       // 16-byte loads (round 6: 4-byte loads ran at 1.4-2.2 TB/s, r06_c9)
        const u32x4 *q = reinterpret_cast<const u32x4 *>(p);
        const long long n4 = n >> 2;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            const u32x4 v = q[i];
            m = max(max(m, v.x & 0x7fffffffu), max(max(v.y & 0x7fffffffu, v.z & 0x7fffffffu), v.w & 0x7fffffffu));
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = max(m, __float_as_uint(p[i]) & 0x7fffffffu);
    }
    wave_atomic_max(bits, (int)blockIdx.y, m);
}

typedef Shape<256, 128, 4, 2> S256x128;
typedef Shape<128, 128, 2, 2> S128x128;
typedef Shape<256, 64, 2, 2> S256x64;
// the ring shapes: 4 ships since round 4 for Cout >= 128; 5 / 6 are the 64-channel tiles of conv1_2 (round 6: with the pooled
// image epilogue the layer's time is its K loop, which round 4 measured at 355 TF/s on this tile against 228 for the round-3 loop
// with its fp32 epilogue; profiles/r04_conv_no_epilogue.jsonl)
typedef Ring<4, 1, 2, 4, 3> CR256x128;     // shape 4: 4 waves, 64x128 wave tiles, 3 stages x 24 KB, two blocks per CU
typedef Ring<4, 1, 2, 2, 4> CR256x64;      // shape 5: 4 waves, 64x64 wave tiles, 4 stages x 20 KB, two blocks per CU
typedef Ring<4, 1, 2, 2, 3> CR256x64s3;    // shape 6: the same with 3 stages
constexpr int kMaxImages = 65535;    // grid.y of image_absmax_kernel; activation images are addressed with 32-bit byte offsets anyway
static int g_conv_shape = -1;       // mh_debug_plconv_shape
static int g_conv_flags = 0;        // mh_debug_plconv_flags
static int g_conv_splitk = 0;       // mh_debug_plconv_splitk: > 0 = every tile cut into that many K slices (measurement sweeps)

static inline size_t act_cells_bytes(long long M, int C) { return (size_t)(C / kBK) * M * kCell; }
static inline size_t wt_cells_bytes(int Cout, int Cin) { return (size_t)9 * (Cin / kBK) * Cout * kCell; }

struct Sched {
    int shape, bm, bn;
    ConvTilePlan pl;
};
// Block-tile shape and K slicing of one launch, from the r03_c10 sweep (profiles/r03_conv_sweep.jsonl; b = 6, 592 x 592):
//   * 128x128 tiles (three blocks per CU) beat 256x128 on every layer with Cout >= 128 (+6 % conv4 .. +14 % conv2_1 / conv3_1,
//     conv5: 203 vs 165 TF/s); Cout <= 64 keeps 256x64;
//   * cutting the leftover tiles of the last round into K slices (plan_conv_tiles) pays when the launch is a few rounds long
//     (conv4: 1028 tiles, 329 vs 303 TF/s) and costs 3-8 % on the long launches (conv1_2 .. conv3_1: the tail is < 1 % of the
//     work, the extra reduce launch and its partial sums are not);
//   * a launch that fits in one round is NOT split (conv5: 260 tiles, 203 TF/s whole vs 157-163 in 2-6 slices) unless it has
//     fewer tiles than a quarter of the resident slots.
static Sched schedule(long long M, int Cin, int Cout)
{
    Sched s;
    // MH_PLCONV_SHAPE=0|1: block tile of the Cout >= 128 layers for A/B runs of the whole step (0 = 256x128, 1 = 128x128)
    static const int env_shape = [] { const char *e = getenv("MH_PLCONV_SHAPE"); return e ? atoi(e) : -1; }();
    // round 4 (gpurun r04_c6, TFLOP/s fp32 / image output at b = 6): the ring kernel's 256x128 tiles (shape 4) beat the round-3
    // loop on every layer with Cout >= 128 -- conv2_1 277 / 249 vs 237 / 216, conv2_2 358 / 340 vs 313 / 301, conv3_2 381 / 369 vs
    // 331 / 324, conv4_2 393 / 378 vs 329 / 324, conv5_1 223 / 217 vs 211 / 210; Cout 64 (conv1_2: 36 k-tiles per tile, the
    // block count per CU decides) stays on the round-3 256x64 loop, whole tiles (228 vs 203 with the tail cut into slices)
    static const bool ring_off = [] { const char *e = getenv("MH_PL_RING"); return e && e[0] == '0'; }();      // A/B: MH_PL_RING=0 = round-3 loop
    static const int env64 = [] { const char *e = getenv("MH_PLCONV_SHAPE64"); return e ? atoi(e) : -1; }();      // 2 | 5 | 6 (A/B)
    const int narrow = (env64 == 2 || env64 == 5 || env64 == 6) ? env64 : (ring_off ? 2 : 5);
    if (Cout <= 64) s.shape = (g_conv_shape == 2 || g_conv_shape == 5 || g_conv_shape == 6) ? g_conv_shape : narrow;
    else {
        s.shape = (g_conv_shape >= 0) ? g_conv_shape : (env_shape >= 0 ? env_shape : (ring_off ? 1 : 4));
        if (s.shape == 2 || s.shape == 3 || s.shape > 4) s.shape = 4;
    }
    static const int bms[7] = {256, 128, 256, 256, 256, 256, 256}, bns[7] = {128, 128, 64, 256, 128, 64, 64};
    s.bm = bms[s.shape];
    s.bn = bns[s.shape];
    s.pl = plan_conv_tiles(M, Cin, Cout, s.bm, s.bn);
    const long long tiles = (long long)s.pl.tiles_m * s.pl.tiles_n;
    const int slots = resident_slots();
    const long long rounds = tiles / slots;
    const bool ring = s.shape >= 4;
    bool whole = (rounds == 0) ? tiles > slots / 4 : (rounds > 3 || s.shape == 2);
    if (ring && rounds == 0) {
        // the ring kernels add their K slices up inside the launch (round 6): a launch that leaves CUs idle or alone with one block
        // is cut until the blocks fill the resident slots (conv5: 132 tiles -> 3 slices of 11 / 11 / 10 chunks; MH_PLCONV_ONE_ROUND=n
        // forces n slices, 1 = whole tiles as in round 5)
        static const int env_sk = [] { const char *e = getenv("MH_PLCONV_ONE_ROUND"); return e ? atoi(e) : 0; }();
        const int chunks = Cin / kBK;
        int sk = env_sk > 0 ? env_sk : (int)std::max<long long>(1, slots / std::max<long long>(tiles, 1));
        sk = std::max(1, std::min(sk, chunks / 4));                    // >= 4 chunks (36 k-tiles) per slice
        s.pl.splitk = sk; s.pl.body_mtiles = s.pl.tiles_m; s.pl.tail_slices = 1;
        whole = false;
    }
    if (whole) { s.pl.splitk = 1; s.pl.body_mtiles = s.pl.tiles_m; s.pl.tail_slices = 1; }
    if (g_conv_splitk > 0) { s.pl.splitk = std::min(g_conv_splitk, 9 * (Cin / kBK)); s.pl.body_mtiles = s.pl.tiles_m; s.pl.tail_slices = 1; }
    return s;
}
// the launch's K slicing as the kernels see it (ring shapes slice at whole 16-channel chunks = 9 k-tiles)
struct Slicing {
    int ktiles_per_split, splitk, tail_ktiles, tail_slices, body_tiles, tail_tiles;
    long long tail_row0;
};
static Slicing slicing_of(const Sched &sc, long long M, int Cin)
{
    Slicing z;
    const int total_kt = 9 * (Cin / kBK);
    const int kgran = (sc.shape >= 3) ? 9 : 1;
    z.ktiles_per_split = ceil_div(ceil_div(total_kt, sc.pl.splitk), kgran) * kgran;
    z.splitk = ceil_div(total_kt, z.ktiles_per_split);
    z.tail_ktiles = ceil_div(ceil_div(total_kt, sc.pl.tail_slices), kgran) * kgran;
    z.tail_slices = ceil_div(total_kt, z.tail_ktiles);
    z.body_tiles = sc.pl.body_mtiles * sc.pl.tiles_n;
    z.tail_tiles = (sc.pl.tiles_m - sc.pl.body_mtiles) * sc.pl.tiles_n;
    z.tail_row0 = std::min<long long>(M, (long long)sc.pl.body_mtiles * sc.bm);
    if (z.tail_tiles == 0) z.tail_slices = 1;
    return z;
}
// workspace: partial sums of the body region | of the tail region | (ring shapes) the tiles' arrival counters
static void partial_bytes(const Sched &sc, long long M, int Cin, int Cout, size_t &body, size_t &tail, size_t &counters)
{
    const Slicing z = slicing_of(sc, M, Cin);
    if (sc.shape >= 4) {        // ring kernels: whole tiles of raw accumulators per slice (register order), reduced inside the launch
        const size_t tile = (size_t)sc.bm * sc.bn * sizeof(float);
        body = (z.splitk > 1) ? align_up((size_t)z.splitk * z.body_tiles * tile, 256) : 0;
        tail = (z.tail_slices > 1 && z.tail_tiles > 0) ? align_up((size_t)z.tail_slices * z.tail_tiles * tile, 256) : 0;
        counters = (body + tail) ? align_up((size_t)(z.body_tiles + z.tail_tiles) * sizeof(int), 256) : 0;
        return;
    }
    counters = 0;
    body = (z.splitk > 1) ? align_up((size_t)z.splitk * z.tail_row0 * Cout * sizeof(float), 256) : 0;
    tail = (z.tail_slices > 1 && z.tail_row0 < M) ? align_up((size_t)z.tail_slices * (M - z.tail_row0) * Cout * sizeof(float), 256) : 0;
}

}  // namespace pl
}  // namespace mh

using namespace mh;

extern "C" {

void mh_debug_plconv_shape(int shape) { pl::g_conv_shape = shape; }
void mh_debug_plconv_splitk(int splitk) { pl::g_conv_splitk = splitk; }
void mh_debug_plconv_flags(int flags) { pl::g_conv_flags = flags; }

size_t mh_act_planes_bytes(int B, int H, int W, int C)
{
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % pl::kBK) return 0;
    return align_up(pl::act_cells_bytes((long long)B * H * W, C), 256) + align_up((size_t)B * 4, 256);
}

// largest |x| per image of a [B][n] fp32 tensor as fp32 bit patterns (what mh_act_planes wants of a tensor whose producer did not
// report them: the mask tower's BatchNorm output, gradients)
int mh_image_maxbits(const float *x, int B, long long n, unsigned *bits, void *stream)
{
    MH_REQUIRE(x && bits && B > 0 && B <= pl::kMaxImages && n > 0);
    hipStream_t st = as_stream(stream);
    hipError_t e = hipMemsetAsync(bits, 0, (size_t)B * sizeof(unsigned), st);
    if (e != hipSuccess) { set_last_error("hipMemsetAsync(image maxima)", e); return (int)e; }
    const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>(64, n / 4096));
    hipLaunchKernelGGL(pl::image_absmax_kernel, dim3(gx, (unsigned)B), dim3(256), 0, st, x, n, bits);
    return check_launch("pl::image_absmax_kernel");
}

// fp32 NHWC [B,H,W,C] (+ its per-image |x| maxima) -> activation image of [B,Ho,Wo,C]; pool = the 2x2/2 max-pool first
int mh_act_planes(const float *x, const unsigned *maxbits, int B, int H, int W, int C, int pool, void *image, void *stream)
{
    MH_REQUIRE(x && maxbits && image && B > 0 && B <= pl::kMaxImages && H > 0 && W > 0 && C > 0 && C % pl::kBK == 0);
    MH_REQUIRE(!pool || (H >= 2 && W >= 2));
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(image)) & 15) == 0);
    const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
    const long long Mo = (long long)B * Ho * Wo;
    MH_REQUIRE(pl::act_cells_bytes(Mo, C) < (size_t)0x7ff00000u);
    pl::ActArgs p;
    p.x = x; p.maxbits = maxbits; p.B = B; p.H = H; p.W = W; p.C = C; p.pool = pool ? 1 : 0;
    p.cells = reinterpret_cast<char *>(image);
    p.maxbits_out = reinterpret_cast<unsigned *>(p.cells + align_up(pl::act_cells_bytes(Mo, C), 256));
    const int G = C / pl::kBK, Gb = G < 16 ? G : 16;
    const long long nblk = ((Mo + 256 / Gb - 1) / (256 / Gb)) * ((G + Gb - 1) / Gb);
    hipLaunchKernelGGL(pl::act_planes_kernel, dim3((unsigned)std::min<long long>(nblk, 256 * 64)), dim3(256), 0, as_stream(stream), p);
    return check_launch("pl::act_planes_kernel");
}

size_t mh_plconv_packed_bytes(int Cout, int Cin)
{
    if (Cout <= 0 || Cin <= 0 || Cin % pl::kBK) return 0;
    return align_up(pl::wt_cells_bytes(Cout, Cin), 256) + align_up((size_t)Cout * 4, 256) + 256;      // cells | maxbits[Cout] | wnorm
}

int mh_plconv_pack_weight(const float *w, int Cout, int Cin, int flip_transpose, void *packed, void *stream)
{
    MH_REQUIRE(w && packed && Cout > 0 && Cin > 0 && (reinterpret_cast<uintptr_t>(packed) & 255) == 0);
    const int N = flip_transpose ? Cin : Cout, K = flip_transpose ? Cout : Cin;     // the conv that consumes the image
    MH_REQUIRE(K % pl::kBK == 0);
    unsigned *cells = reinterpret_cast<unsigned *>(packed);
    unsigned *bits = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(packed) + align_up(pl::wt_cells_bytes(N, K), 256));
    hipLaunchKernelGGL(pl::weight_maxbits_kernel, dim3(N), dim3(256), 0, as_stream(stream), w, N, K, flip_transpose, Cin, bits);
    int rc = check_launch("pl::weight_maxbits_kernel");
    if (rc) return rc;
    unsigned *wnorm = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(bits) + align_up((size_t)N * 4, 256));
    hipError_t e = hipMemsetAsync(wnorm, 0, 256, as_stream(stream));
    if (e != hipSuccess) { set_last_error("hipMemsetAsync(wnorm)", e); return (int)e; }
    hipLaunchKernelGGL(pl::weight_norm_kernel, dim3(N), dim3(256), 0, as_stream(stream), w, N, K, flip_transpose, Cin, wnorm);
    rc = check_launch("pl::weight_norm_kernel");
    if (rc) return rc;
    const long long total = 9LL * (K / pl::kBK) * N * 16;
    hipLaunchKernelGGL(pl::pack_weight_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0,
                       as_stream(stream), w, N, K, flip_transpose, Cin, bits, cells);
    return check_launch("pl::pack_weight_kernel");
}

// Workspace of a sliced launch: [arrival counters | partial sums of the body | of the tail].  The counters sit FIRST, in a fixed
// reserve, so that launches of different layers (whose partial-sum regions differ) never write over them: every launch leaves its
// counters zero (the block that finishes a tile re-arms it), so a caller that hands in a buffer zeroed ONCE (MH_EPI_WS_ZEROED in
// `epilogue`) needs no memset per launch -- 14 of them per cfg2 step on the main stream (profiles/r06_step_launches_c21.txt).
static constexpr size_t kCounterReserve = 64 << 10;
size_t mh_plconv3x3_ws_bytes(int B, int H, int W, int Cin, int Cout)
{
    const long long M = (long long)B * H * W;
    if (M <= 0 || Cin <= 0 || Cout <= 0 || Cin % pl::kBK) return 0;
    const pl::Sched sc = pl::schedule(M, Cin, Cout);
    size_t body, tail, counters;
    pl::partial_bytes(sc, M, Cin, Cout, body, tail, counters);
    return body + tail + (counters ? std::max(counters, kCounterReserve) : 0);
}

static int plconv_impl(const void *in_image, const unsigned *in_true_maxbits, int B, int H, int W, int Cin, const void *packed,
                       int Cout, const float *bias, int epilogue, float *out, void *out_image, int pool, unsigned *out_maxbits,
                       void *workspace, size_t ws_bytes, void *stream)
{
    const bool ws_zeroed = (epilogue & MH_EPI_WS_ZEROED) != 0;
    epilogue &= ~MH_EPI_WS_ZEROED;
    // many small images (round 5: the 1536 7x7 RoI maps of the mask tower / the ResNet layer4 stacks): the fp32-output kernels
    // look every row's image up in in_bits; only the image-output epilogue (scale words written by one block) is limited to 256
    MH_REQUIRE(in_image && packed && (out || out_image) && B > 0 && B <= (out_image ? 256 : pl::kMaxImages) && H > 0 && W > 0);
    MH_REQUIRE(Cin > 0 && Cin % pl::kBK == 0 && Cout > 0 && Cout % 4 == 0);
    MH_REQUIRE(!out_image || (Cout % pl::kBK == 0 && in_true_maxbits));
    MH_REQUIRE(!pool || (out_image && H % 2 == 0 && W % 2 == 0));
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(in_image) | reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out) |
                 reinterpret_cast<uintptr_t>(out_image)) & 15) == 0);
    const long long M = (long long)B * H * W;
    // 32-bit offsets from a tile's first pixel: the tile, two halos, and in pool order the image rows a tile's 64 windows span
    const long long span = pool ? 2LL * W * (64 / std::max(W / 2, 1) + 2) : 256;
    MH_REQUIRE(M < (1LL << 25) && pl::act_cells_bytes(M, Cin) + (size_t)(2 * (W + 1) + span) * pl::kCell < (size_t)0x7ff00000u &&
               pl::wt_cells_bytes(Cout, Cin) < (size_t)0x7ff00000u);
    pl::ConvArgs p;
    p.in = reinterpret_cast<const char *>(in_image);
    p.in_bits = reinterpret_cast<const unsigned *>(p.in + align_up(pl::act_cells_bytes(M, Cin), 256));
    p.in_true_bits = in_true_maxbits ? in_true_maxbits : p.in_bits;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin;
    p.wt = reinterpret_cast<const char *>(packed);
    p.wt_bits = reinterpret_cast<const unsigned *>(p.wt + align_up(pl::wt_cells_bytes(Cout, Cin), 256));
    p.wnorm = reinterpret_cast<const float *>(reinterpret_cast<const char *>(p.wt_bits) + align_up((size_t)Cout * 4, 256));
    p.Cout = Cout; p.bias = bias; p.epilogue = epilogue; p.out = out; p.out_bits = out_maxbits;
    p.debug_flags = pl::g_conv_flags;
    p.out_cells = reinterpret_cast<char *>(out_image);
    p.out_scale_bits = out_image ? reinterpret_cast<unsigned *>(p.out_cells + align_up(pl::act_cells_bytes(pool ? M / 4 : M, Cout), 256)) : nullptr;
    pl::Sched sc = pl::schedule(M, Cin, Cout);
    if (pool && sc.shape < 4) sc = [&] { const int keep = pl::g_conv_shape; pl::g_conv_shape = Cout <= 64 ? 5 : 4; pl::Sched r = pl::schedule(M, Cin, Cout); pl::g_conv_shape = keep; return r; }();   // the pooled epilogue exists on the ring kernels only
    size_t body_bytes, tail_bytes, counter_bytes;
    pl::partial_bytes(sc, M, Cin, Cout, body_bytes, tail_bytes, counter_bytes);
    const size_t counter_room = counter_bytes ? std::max(counter_bytes, kCounterReserve) : 0;
    if (body_bytes + tail_bytes > 0 && (workspace == nullptr || ws_bytes < body_bytes + tail_bytes + counter_room)) {
        sc.pl.splitk = 1; sc.pl.body_mtiles = sc.pl.tiles_m; sc.pl.tail_slices = 1;     // no room for partial sums: whole tiles only
        body_bytes = tail_bytes = counter_bytes = 0;
    }
    const int total_kt = 9 * (Cin / pl::kBK);
    const pl::Slicing z = pl::slicing_of(sc, M, Cin);
    p.tiles_m = sc.pl.tiles_m; p.tiles_n = sc.pl.tiles_n;
    p.body_tiles = z.body_tiles; p.ktiles_per_split = z.ktiles_per_split; p.splitk = z.splitk;
    p.tail_tiles = z.tail_tiles; p.tail_ktiles = z.tail_ktiles; p.tail_slices = z.tail_slices; p.tail_row0 = z.tail_row0;
    const size_t counter_off = counter_bytes ? std::max(counter_bytes, kCounterReserve) : 0;       // counters first (see mh_plconv3x3_ws_bytes)
    p.counters = reinterpret_cast<int *>(workspace);
    p.partial = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + counter_off);
    p.partial_tail = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + counter_off + body_bytes);
    const long long nblocks = (long long)p.tail_tiles * p.tail_slices + (long long)p.body_tiles * p.splitk;
    MH_REQUIRE(nblocks > 0 && nblocks < (1LL << 31));
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)nblocks);
    constexpr size_t kTabMax = 1024 * 16;                          // the per-k-tile offset table lives in LDS (16 B per k-tile)
    MH_REQUIRE(total_kt <= 1024);
    const size_t tab_bytes = (size_t)total_kt * 16;
    if (sc.shape >= 4 && counter_bytes && !(ws_zeroed && counter_bytes <= kCounterReserve)) {
        // arrival counters of the sliced tiles: zero on entry (the last block of a tile re-arms its counter, but the workspace is
        // the caller's scratch: nothing promises it is still zero -- unless the caller says so, MH_EPI_WS_ZEROED)
        hipError_t e = hipMemsetAsync(p.counters, 0, counter_bytes, st);
        if (e != hipSuccess) { set_last_error("hipMemsetAsync(conv arrival counters)", e); return (int)e; }
    }
    auto ring = [&](auto tag) {
        typedef decltype(tag) R;
        if (pool) pl::launch<pl::conv3x3_ring_kernel<R, 2>>(grid, pl::ring_conv_lds_bytes<R>(2), st, p, 0, R::threads);
        else if (out_image) pl::launch<pl::conv3x3_ring_kernel<R, 1>>(grid, pl::ring_conv_lds_bytes<R>(1), st, p, 0, R::threads);
        else pl::launch<pl::conv3x3_ring_kernel<R, 0>>(grid, pl::ring_conv_lds_bytes<R>(0), st, p, 0, R::threads);
    };
    if (sc.shape == 4) { ring(pl::CR256x128()); return check_launch("pl::conv3x3_ring_kernel"); }
    if (sc.shape == 5) { ring(pl::CR256x64()); return check_launch("pl::conv3x3_ring_kernel"); }
    if (sc.shape == 6) { ring(pl::CR256x64s3()); return check_launch("pl::conv3x3_ring_kernel"); }
    if (out_image) {
        if (sc.shape == 0) pl::launch<pl::conv3x3_kernel<pl::S256x128, true>>(grid, pl::conv_lds_bytes<pl::S256x128>() + tab_bytes, st, p, pl::conv_lds_bytes<pl::S256x128>() + kTabMax);
        else if (sc.shape == 1) pl::launch<pl::conv3x3_kernel<pl::S128x128, true>>(grid, pl::conv_lds_bytes<pl::S128x128>() + tab_bytes, st, p, pl::conv_lds_bytes<pl::S128x128>() + kTabMax);
        else pl::launch<pl::conv3x3_kernel<pl::S256x64, true>>(grid, pl::conv_lds_bytes<pl::S256x64>() + tab_bytes, st, p, pl::conv_lds_bytes<pl::S256x64>() + kTabMax);
    } else {
        if (sc.shape == 0) pl::launch<pl::conv3x3_kernel<pl::S256x128, false>>(grid, pl::S256x128::lds_bytes + tab_bytes, st, p, pl::S256x128::lds_bytes + kTabMax);
        else if (sc.shape == 1) pl::launch<pl::conv3x3_kernel<pl::S128x128, false>>(grid, pl::S128x128::lds_bytes + tab_bytes, st, p, pl::S128x128::lds_bytes + kTabMax);
        else pl::launch<pl::conv3x3_kernel<pl::S256x64, false>>(grid, pl::S256x64::lds_bytes + tab_bytes, st, p, pl::S256x64::lds_bytes + kTabMax);
    }
    int rc = check_launch("pl::conv3x3_kernel");
    if (rc) return rc;
    // the round-3 loop (MH_PL_RING=0 and the A/B shapes) adds its K slices up in a second launch
    const long long HW = (long long)H * W;
    auto reduce = [&](const float *part, int slices, long long rows, long long row0) {
        if (out_image) {
            const long long nblk = ((rows + 255) / 256) * (Cout / pl::kBK);
            hipLaunchKernelGGL(pl::reduce_planes_kernel, dim3((unsigned)std::min<long long>(nblk, 256 * 8)), dim3(256), 0, st, part, slices, rows,
                               Cout, p.out_cells, M, bias, epilogue, row0, HW, p.out_scale_bits, out_maxbits);
            return check_launch("pl::reduce_planes_kernel");
        }
        const long long total = rows * Cout;
        hipLaunchKernelGGL(pl::reduce_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 256 * 8)), dim3(256), 0, st, part,
                           slices, rows, Cout, out + (size_t)row0 * Cout, bias, epilogue, row0, HW, out_maxbits);
        return check_launch("pl::reduce_kernel");
    };
    if (p.splitk > 1 && p.tail_row0 > 0) rc = reduce(p.partial, p.splitk, p.tail_row0, 0);
    if (!rc && p.tail_tiles > 0 && p.tail_slices > 1) rc = reduce(p.partial_tail, p.tail_slices, M - p.tail_row0, p.tail_row0);
    return rc;
}

// out [B,H,W,Cout] fp32 = epi(conv3x3(in image, packed weights) + bias); out_maxbits [B] (optional) receives the largest |out|
// per image -- it must be ZERO before the call (atomicMax)
int mh_plconv3x3(const void *in_image, int B, int H, int W, int Cin, const void *packed, int Cout, const float *bias,
                 int epilogue, float *out, unsigned *out_maxbits, void *workspace, size_t ws_bytes, void *stream)
{
    MH_REQUIRE(out);
    return plconv_impl(in_image, nullptr, B, H, W, Cin, packed, Cout, bias, epilogue, out, nullptr, 0, out_maxbits, workspace, ws_bytes, stream);
}

// the same conv whose output leaves the kernel AS the next layer's activation image (mh_act_planes_bytes(B, H, W, Cout)): no
// fp32 tensor, no converter pass.  in_true_maxbits [B] = the TRUE per-image maxima of the input (what its producer reported:
// the image's own scale may be a looser bound); out_maxbits as above.
int mh_plconv3x3_to_image(const void *in_image, const unsigned *in_true_maxbits, int B, int H, int W, int Cin, const void *packed,
                          int Cout, const float *bias, int epilogue, void *out_image, unsigned *out_maxbits, void *workspace,
                          size_t ws_bytes, void *stream)
{
    MH_REQUIRE(out_image && in_true_maxbits);
    return plconv_impl(in_image, in_true_maxbits, B, H, W, Cin, packed, Cout, bias, epilogue, nullptr, out_image, 0, out_maxbits, workspace,
                       ws_bytes, stream);
}

// ... and through the 2x2 / 2 max-pool that follows the layer (H, W even): out_image = the activation image of the POOLED output
// [B, H/2, W/2, Cout] (mh_act_planes_bytes(B, H / 2, W / 2, Cout)), bit-identical to mh_plconv3x3 + mh_act_planes(pool = 1) on a
// monotone epilogue (none / ReLU / ReLU6).  Replaces the fp32 tensor and the converter pass of conv1_2, conv2_2, conv3_3, conv4_3
// (reference lib/object_detector.py:110-118: the MaxPool2d modules of vgg16.features).
int mh_plconv3x3_pool_to_image(const void *in_image, const unsigned *in_true_maxbits, int B, int H, int W, int Cin, const void *packed,
                               int Cout, const float *bias, int epilogue, void *out_image, unsigned *out_maxbits, void *workspace,
                               size_t ws_bytes, void *stream)
{
    MH_REQUIRE(out_image && in_true_maxbits);
    return plconv_impl(in_image, in_true_maxbits, B, H, W, Cin, packed, Cout, bias, epilogue, nullptr, out_image, 1, out_maxbits, workspace,
                       ws_bytes, stream);
}

// conv1_1: NCHW image (Cin <= 4) -> activation image of [B,H,W,Cout], bias + activation fused (csrc/pl_conv.hip: stem_kernel);
// out_maxbits [B] zero on entry.  B <= 32 (the input's per-image maxima borrow the unused part of the image's tail).
int mh_stem_to_image(const float *in_nchw, int B, int Cin, int H, int W, const float *w, int Cout, const float *bias, int epilogue,
                     void *out_image, unsigned *out_maxbits, void *stream)
{
    MH_REQUIRE(in_nchw && w && out_image && B > 0 && B <= 32 && Cin > 0 && Cin <= 4 && H > 0 && W > 0 && Cout > 0 && Cout % pl::kBK == 0);
    MH_REQUIRE((reinterpret_cast<uintptr_t>(out_image) & 255) == 0);
    const long long M = (long long)B * H * W;
    MH_REQUIRE(pl::act_cells_bytes(M, Cout) < (size_t)0x7ff00000u);
    const size_t lds = ((size_t)9 * Cin * Cout + 2 * Cout) * sizeof(float);
    MH_REQUIRE(lds <= 64 * 1024);
    hipStream_t st = as_stream(stream);
    char *cells = reinterpret_cast<char *>(out_image);
    unsigned *scale = reinterpret_cast<unsigned *>(cells + align_up(pl::act_cells_bytes(M, Cout), 256));
    unsigned *in_bits = scale + 32;                       // scratch behind the B scale words (the tail is >= 256 bytes)
    hipError_t e = hipMemsetAsync(in_bits, 0, 32 * sizeof(unsigned), st);
    if (e != hipSuccess) { set_last_error("hipMemsetAsync(stem maxima)", e); return (int)e; }
    hipLaunchKernelGGL(pl::image_absmax_kernel, dim3(64, (unsigned)B), dim3(256), 0, st, in_nchw, (long long)Cin * H * W, in_bits);
    int rc = check_launch("pl::image_absmax_kernel");
    if (rc) return rc;
    // 64 output channels from <= 3 input channels (VGG's conv1_1): the matrix-core form (round 6); MH_STEM=valu keeps the VALU kernel (A/B)
    static const bool valu_only = [] { const char *e = getenv("MH_STEM"); return e && e[0] == 'v'; }();
    if (Cout == 64 && Cin <= 3 && M < (1LL << 31) && !valu_only) {
        pl::StemArgs p;
        p.in = in_nchw; p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.w = w; p.bias = bias; p.epilogue = epilogue; p.in_bits = in_bits;
        p.out_cells = cells; p.scale_bits = scale; p.out_bits = out_maxbits;
        // two blocks per CU = what is resident: every block walks ~16 tiles (gpurun r06_c8: 0.30 ms against 0.50-0.59 with 4-8 blocks per CU queued)
        static const int blocks_per_cu = [] { const char *e = getenv("MH_STEM_BLOCKS"); return e ? std::max(1, atoi(e)) : 2; }();
        const long long ntiles = (M + 255) / 256;
        pl::launch<pl::stem_mfma_kernel>(dim3((unsigned)std::min<long long>(ntiles, 256LL * blocks_per_cu)), (size_t)pl::kStemLds, st, p);
        return check_launch("pl::stem_mfma_kernel");
    }
    const long long nblk = ((M + 255) / 256) * (Cout / pl::kBK);
    hipLaunchKernelGGL(pl::stem_kernel, dim3((unsigned)std::min<long long>(nblk, 256 * 16)), dim3(256), lds, st, in_nchw, B, Cin, H, W, w, Cout,
                       bias, epilogue, in_bits, cells, scale, out_maxbits);
    return check_launch("pl::stem_kernel");
}

}  // extern "C"
