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

// thread = one output pixel x one 16-channel chunk; the 256 threads of a block take 256 consecutive pixels of one chunk:
// reads are 64-byte segments (4 segments per pixel when pooling), writes are 16 KB contiguous.
__global__ __launch_bounds__(256) void act_planes_kernel(const ActArgs p)
{
    const int Ho = p.pool ? p.H / 2 : p.H, Wo = p.pool ? p.W / 2 : p.W;
    const long long Mo = (long long)p.B * Ho * Wo;
    const int G = p.C / kBK;
    const long long nblk_m = (Mo + 255) / 256;
    if (blockIdx.x == 0 && threadIdx.x < p.B) p.maxbits_out[threadIdx.x] = p.maxbits[threadIdx.x];
    for (long long blk = blockIdx.x; blk < nblk_m * G; blk += gridDim.x) {
        const int g = (int)(blk / nblk_m);
        const long long m = (blk % nblk_m) * 256 + threadIdx.x;
        if (m >= Mo) continue;
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
    float *out;                // [Mtot][Cout]
    unsigned *out_bits;        // [B], zero before the launch (may be nullptr)
    int tiles_m, tiles_n;
    // tile schedule (conv.hip: ConvArgs has the long explanation): blocks [0, tail_tiles * tail_slices) = the leftover tiles
    // cut into K slices; then body_tiles * splitk blocks of whole (or uniformly split) tiles
    int body_tiles, splitk, ktiles_per_split;
    int tail_tiles, tail_slices, tail_ktiles;
    long long tail_row0;
    float *partial, *partial_tail;
};

template <class S>
__global__ __launch_bounds__(kThreads, 2) void conv3x3_kernel(const ConvArgs p)
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
    // the tile the next issue() loads: (chunk it_g, tap it_tap), advanced incrementally (no division in the loop); once the
    // last tile of the slice has been issued the state stays there (the harmless reload of the final step)
    int it_kt = kt_begin, it_g = kt_begin / 9, it_tap = kt_begin - 9 * it_g;
    auto issue = [&](Stage<S> &st) {
        const int dy = (it_tap * 11) >> 5, dx = it_tap - 3 * dy;            // tap / 3, tap % 3 for tap in [0, 9)
        const unsigned oa = (unsigned)it_g * strideA + (unsigned)(halo + (dy - 1) * p.W + (dx - 1)) * kCell;
        const unsigned ob = (unsigned)(it_tap * G + it_g) * strideB;
        const unsigned bit = 1u << it_tap;
#pragma unroll
        for (int j = 0; j < S::na; ++j) st.a[j] = load16(sa, (a_taps[j] & bit) ? cp.va[j] : kOob, oa);
#pragma unroll
        for (int j = 0; j < S::nb; ++j) st.b[j] = load16(sb, cp.vb[j], ob);
        if (it_kt + 1 < kt_end) {
            ++it_kt;
            if (++it_tap == 9) { it_tap = 0; ++it_g; }
        }
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

    // exponents of this tile's rows (their image's) and columns
    int *ex = reinterpret_cast<int *>(lds);
    for (int i = tid; i < S::bm + S::bn; i += kThreads) {
        int e = 0;
        if (i < S::bm) { if (m0 + i < Mtot) e = row_exponent(p.in_bits[(m0 + i) / HW]); }
        else if (n0 + (i - S::bm) < p.Cout) e = row_exponent(p.wt_bits[n0 + (i - S::bm)]);
        ex[i] = e;
    }
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
    acc_foreach<S>(acc, wm, wn, lane, [&](int r, int c, int sn, float v) {
        const long long row = m0 + r;
        if (row >= Mtot || n0 + c >= p.Cout) return;
        v = conv_epi(__builtin_ldexpf(v, -(ex[r] + ecol[sn])) + bcol[sn], p.epilogue);
        p.out[(size_t)row * p.Cout + n0 + c] = v;
        const unsigned bits = __float_as_uint(v) & 0x7fffffffu;
        if (b_hi - b_lo <= 1) { if (r < split) vlo = max(vlo, bits); else vhi = max(vhi, bits); }
        else if (p.out_bits) {
            unsigned m = bits;                       // one row = the 32 lanes of a half-wave: reduce, one atomic per row
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
            if ((lane & 31) == 0 && m) atomicMax(p.out_bits + row / HW, m);
        }
    });
    if (p.out_bits && b_hi - b_lo <= 1) {
        wave_atomic_max(p.out_bits, b_lo, vlo);
        if (b_hi > b_lo) wave_atomic_max(p.out_bits, b_hi, vhi);
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
            if (cur >= 0 && vmax && out_bits) atomicMax(out_bits + cur, vmax);
            cur = b;
            vmax = 0;
        }
        vmax = max(vmax, __float_as_uint(v) & 0x7fffffffu);
    }
    if (out_bits) wave_atomic_max(out_bits, cur < 0 ? 0 : cur, cur < 0 ? 0u : vmax);
}

typedef Shape<256, 128, 4, 2> S256x128;
typedef Shape<128, 128, 2, 2> S128x128;
typedef Shape<256, 64, 2, 2> S256x64;
static int g_conv_shape = -1;       // mh_debug_plconv_shape

static inline size_t act_cells_bytes(long long M, int C) { return (size_t)(C / kBK) * M * kCell; }
static inline size_t wt_cells_bytes(int Cout, int Cin) { return (size_t)9 * (Cin / kBK) * Cout * kCell; }

struct Sched {
    int shape, bm, bn;
    ConvTilePlan pl;
};
static Sched schedule(long long M, int Cin, int Cout)
{
    Sched s;
    s.shape = (g_conv_shape >= 0) ? g_conv_shape : (Cout <= 64 ? 2 : 0);
    s.bm = (s.shape == 1) ? 128 : 256;
    s.bn = (s.shape == 2) ? 64 : 128;
    s.pl = plan_conv_tiles(M, Cin, Cout, s.bm, s.bn);
    return s;
}
static void partial_bytes(const Sched &sc, long long M, int Cin, int Cout, size_t &body, size_t &tail)
{
    const int total_kt = 9 * (Cin / kBK);
    const long long row0 = std::min<long long>(M, (long long)sc.pl.body_mtiles * sc.bm);
    const int s0 = ceil_div(total_kt, ceil_div(total_kt, sc.pl.splitk));
    body = (s0 > 1) ? align_up((size_t)s0 * row0 * Cout * sizeof(float), 256) : 0;
    const int tk = ceil_div(total_kt, sc.pl.tail_slices), ts = ceil_div(total_kt, tk);
    tail = (ts > 1 && row0 < M) ? align_up((size_t)ts * (M - row0) * Cout * sizeof(float), 256) : 0;
}

}  // namespace pl
}  // namespace mh

using namespace mh;

extern "C" {

void mh_debug_plconv_shape(int shape) { pl::g_conv_shape = shape; }

size_t mh_act_planes_bytes(int B, int H, int W, int C)
{
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % pl::kBK) return 0;
    return align_up(pl::act_cells_bytes((long long)B * H * W, C), 256) + align_up((size_t)B * 4, 256);
}

// fp32 NHWC [B,H,W,C] (+ its per-image |x| maxima) -> activation image of [B,Ho,Wo,C]; pool = the 2x2/2 max-pool first
int mh_act_planes(const float *x, const unsigned *maxbits, int B, int H, int W, int C, int pool, void *image, void *stream)
{
    MH_REQUIRE(x && maxbits && image && B > 0 && B <= 256 && H > 0 && W > 0 && C > 0 && C % pl::kBK == 0);
    MH_REQUIRE(!pool || (H >= 2 && W >= 2));
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(image)) & 15) == 0);
    const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
    const long long Mo = (long long)B * Ho * Wo;
    MH_REQUIRE(pl::act_cells_bytes(Mo, C) < (size_t)0x7ff00000u);
    pl::ActArgs p;
    p.x = x; p.maxbits = maxbits; p.B = B; p.H = H; p.W = W; p.C = C; p.pool = pool ? 1 : 0;
    p.cells = reinterpret_cast<char *>(image);
    p.maxbits_out = reinterpret_cast<unsigned *>(p.cells + align_up(pl::act_cells_bytes(Mo, C), 256));
    const long long nblk = ((Mo + 255) / 256) * (C / pl::kBK);
    hipLaunchKernelGGL(pl::act_planes_kernel, dim3((unsigned)std::min<long long>(nblk, 256 * 64)), dim3(256), 0, as_stream(stream), p);
    return check_launch("pl::act_planes_kernel");
}

size_t mh_plconv_packed_bytes(int Cout, int Cin)
{
    if (Cout <= 0 || Cin <= 0 || Cin % pl::kBK) return 0;
    return align_up(pl::wt_cells_bytes(Cout, Cin), 256) + align_up((size_t)Cout * 4, 256);
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
    const long long total = 9LL * (K / pl::kBK) * N * 16;
    hipLaunchKernelGGL(pl::pack_weight_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0,
                       as_stream(stream), w, N, K, flip_transpose, Cin, bits, cells);
    return check_launch("pl::pack_weight_kernel");
}

size_t mh_plconv3x3_ws_bytes(int B, int H, int W, int Cin, int Cout)
{
    const long long M = (long long)B * H * W;
    if (M <= 0 || Cin <= 0 || Cout <= 0 || Cin % pl::kBK) return 0;
    const pl::Sched sc = pl::schedule(M, Cin, Cout);
    size_t body, tail;
    pl::partial_bytes(sc, M, Cin, Cout, body, tail);
    return body + tail;
}

// out [B,H,W,Cout] fp32 = epi(conv3x3(in image, packed weights) + bias); out_maxbits [B] (optional) receives the largest |out|
// per image -- it must be ZERO before the call (atomicMax)
int mh_plconv3x3(const void *in_image, int B, int H, int W, int Cin, const void *packed, int Cout, const float *bias,
                 int epilogue, float *out, unsigned *out_maxbits, void *workspace, size_t ws_bytes, void *stream)
{
    MH_REQUIRE(in_image && packed && out && B > 0 && H > 0 && W > 0);
    MH_REQUIRE(Cin > 0 && Cin % pl::kBK == 0 && Cout > 0 && Cout % 4 == 0);
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(in_image) | reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out)) & 15) == 0);
    const long long M = (long long)B * H * W;
    MH_REQUIRE(pl::act_cells_bytes(M, Cin) + (size_t)(2 * (W + 1) + 256) * pl::kCell < (size_t)0x7ff00000u &&
               pl::wt_cells_bytes(Cout, Cin) < (size_t)0x7ff00000u);
    pl::ConvArgs p;
    p.in = reinterpret_cast<const char *>(in_image);
    p.in_bits = reinterpret_cast<const unsigned *>(p.in + align_up(pl::act_cells_bytes(M, Cin), 256));
    p.B = B; p.H = H; p.W = W; p.Cin = Cin;
    p.wt = reinterpret_cast<const char *>(packed);
    p.wt_bits = reinterpret_cast<const unsigned *>(p.wt + align_up(pl::wt_cells_bytes(Cout, Cin), 256));
    p.Cout = Cout; p.bias = bias; p.epilogue = epilogue; p.out = out; p.out_bits = out_maxbits;
    pl::Sched sc = pl::schedule(M, Cin, Cout);
    size_t body_bytes, tail_bytes;
    pl::partial_bytes(sc, M, Cin, Cout, body_bytes, tail_bytes);
    if (body_bytes + tail_bytes > 0 && (workspace == nullptr || ws_bytes < body_bytes + tail_bytes)) {
        sc.pl.splitk = 1; sc.pl.body_mtiles = sc.pl.tiles_m; sc.pl.tail_slices = 1;     // no room for partial sums: whole tiles only
        body_bytes = tail_bytes = 0;
    }
    const int total_kt = 9 * (Cin / pl::kBK);
    p.tiles_m = sc.pl.tiles_m; p.tiles_n = sc.pl.tiles_n;
    p.body_tiles = sc.pl.body_mtiles * sc.pl.tiles_n;
    p.ktiles_per_split = ceil_div(total_kt, sc.pl.splitk);
    p.splitk = ceil_div(total_kt, p.ktiles_per_split);
    p.tail_tiles = (sc.pl.tiles_m - sc.pl.body_mtiles) * sc.pl.tiles_n;
    p.tail_ktiles = ceil_div(total_kt, sc.pl.tail_slices);
    p.tail_slices = ceil_div(total_kt, p.tail_ktiles);
    p.tail_row0 = std::min<long long>(M, (long long)sc.pl.body_mtiles * sc.bm);
    p.partial = reinterpret_cast<float *>(workspace);
    p.partial_tail = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + body_bytes);
    const long long nblocks = (long long)p.tail_tiles * p.tail_slices + (long long)p.body_tiles * p.splitk;
    MH_REQUIRE(nblocks > 0 && nblocks < (1LL << 31));
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)nblocks);
    if (sc.shape == 0) pl::launch<pl::conv3x3_kernel<pl::S256x128>>(grid, pl::S256x128::lds_bytes, st, p);
    else if (sc.shape == 1) pl::launch<pl::conv3x3_kernel<pl::S128x128>>(grid, pl::S128x128::lds_bytes, st, p);
    else pl::launch<pl::conv3x3_kernel<pl::S256x64>>(grid, pl::S256x64::lds_bytes, st, p);
    int rc = check_launch("pl::conv3x3_kernel");
    if (rc) return rc;
    const long long HW = (long long)H * W;
    auto reduce = [&](const float *part, int slices, long long rows, long long row0) {
        const long long total = rows * Cout;
        hipLaunchKernelGGL(pl::reduce_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 256 * 8)), dim3(256), 0, st, part,
                           slices, rows, Cout, out + (size_t)row0 * Cout, bias, epilogue, row0, HW, out_maxbits);
        return check_launch("pl::reduce_kernel");
    };
    if (p.splitk > 1 && p.tail_row0 > 0) rc = reduce(p.partial, p.splitk, p.tail_row0, 0);
    if (!rc && p.tail_tiles > 0 && p.tail_slices > 1) rc = reduce(p.partial_tail, p.tail_slices, M - p.tail_row0, p.tail_row0);
    return rc;
}

}  // extern "C"
