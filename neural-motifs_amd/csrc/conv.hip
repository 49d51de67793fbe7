// conv.hip -- the convolution stack in NHWC on the gfx950 matrix cores.
//
// conv3x3 (stride 1, pad 1) is an implicit GEMM on the 128x128x16 FP32 MFMA tile engine (mfma_tile.h):
//   M = B*H*W output pixels (NHWC row = pixel, so the GEMM's K-contiguous A rows are pixel channel vectors),
//   N = Cout, K = 9 taps x Cin.  For each tap the A row of pixel (b,y,x) is the channel vector of pixel
//   (b,y+dy-1,x+dx-1) or zeros outside the image: the halo is a predicate on the row pointer, no im2col buffer.
//   The K loop walks 16-channel chunks with the nine taps INSIDE a chunk, so the nine k-tiles of a chunk re-read the
//   same pixels' 64-byte segments out of L1/L2 (tap-major order sent 9.6 GB per bench step to the fabric for 2.1 GB
//   of input, this order 4.6 GB: profiles/r01_conv_traffic_*).
//   B = packed weights wt[tap][co][ci/16][h1|h2 planes + one unused slot pair].  Epilogue fuses bias + ReLU/ReLU6.
// The same kernel computes dgrad when given flip-transposed weights (mh_conv3x3_pack_weight).
#include <algorithm>
#include <cstdlib>

#include <type_traits>
#include "mfma_tile.h"
#include "pl_tile.h"     // wave_atomic_max (per-image output maxima for the plane engine, pl_conv.hip)

#ifndef MH_CONV_TAP_MAJOR
#define MH_CONV_TAP_MAJOR 0   /* 1: the K loop walks tap-major (all channels of tap 0, then tap 1, ...): A/B comparisons only */
#endif

namespace mh {

struct ConvArgs {
    const float *in;
    int B, H, W, Cin;
    const float *wt;  // [9][Cin][Cout]
    int Cout;
    const float *bias;
    int epilogue;
    float *out;
    int tiles_m, tiles_n;
    // Tile schedule (conv_schedule): the grid is 1-D.  Blocks [0, tail_tiles * tail_slices) are the TAIL: the tiles left
    // over after the last full round of resident blocks, each cut into tail_slices K slices so that the leftover keeps
    // the whole chip busy for 1/tail_slices of a block time instead of a few CUs for a whole one.  The remaining
    // body_tiles * splitk blocks are the BODY: whole tiles (splitk == 1, the normal case: direct epilogue, no partial
    // sums) or, for layers with fewer tiles than resident slots, tiles cut uniformly into splitk slices.
    int body_tiles, splitk, ktiles_per_split;
    int tail_tiles, tail_slices, tail_ktiles;
    long long tail_row0;   // first output row of the tail tiles (= body m-tiles * BM)
    float *partial;        // body:  [splitk][tail_row0][Cout]       when splitk > 1
    float *partial_tail;   // tail:  [tail_slices][M - tail_row0][Cout] when tail_slices > 1
    const int *expA;  // f16x3: exponent per OUTPUT pixel [M], covering its 3x3 input neighbourhood (pixel_exponents)
    const int *expW;  // exponent per output channel [Cout] (tail of the packed weights)
};

__device__ __forceinline__ float conv_epi(float v, int epilogue)
{
    if (epilogue == MH_EPI_RELU) return fmaxf(v, 0.f);
    if (epilogue == MH_EPI_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
    return v;
}

template <int BM, int BN>
__global__ __launch_bounds__(kThreads, MH_MINW) void conv3x3_nhwc_kernel(const ConvArgs p)
{
    constexpr int FA = TileGeom<BM, true>::floats, FB = TileGeom<BN, true>::floats;   // A: NHWC pixels, B: wt[tap][co][ci]: both K-contiguous (WM)
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x (FA + FB) floats, see launch_tile_kernel
    auto As = [&](int buf) -> float * { return lds + buf * (FA + FB); };
    auto Bs = [&](int buf) -> float * { return lds + buf * (FA + FB) + FA; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int wm, wn;
    wave_origin<BM, BN>(wave, wm, wn);
    // block -> (tile, K slice), see ConvArgs
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
    // consecutive tiles walk over Cout first: they share the same input pixels (A panel) in L2
    const long long m0 = (long long)(t / p.tiles_n) * BM;
    const int n0 = (t % p.tiles_n) * BN;
    const long long Mtot = (long long)p.B * p.H * p.W;

    // Planned loads (mfma_tile.h): per-lane byte offsets are fixed for the whole K loop, the tap shift and the channel
    // offset go into the scalar offset.  A is addressed relative to a block origin one halo (W+1 pixels) before the
    // tile's first pixel, so every tap of every valid pixel has a non-negative 32-bit offset whatever the size of
    // the input tensor; which of the 9 taps fall inside the image is a 9-bit mask per staged pixel.
    constexpr int NVA = TileGeom<BM, true>::nv, NVB = TileGeom<BN, true>::nv;
    const int halo = (p.W + 1) * p.Cin;
    const GSrc ga = make_gsrc(p.in + (ptrdiff_t)m0 * p.Cin - halo), gb = make_gsrc(p.wt);
    unsigned a_off[NVA], a_taps[NVA];
    // B = packed weights as f16 planes, wt[tap][co][ci / 16][96 B]: chunk copies, no split (mfma_tile.h: PStage)
    const unsigned b_row_bytes = (unsigned)(p.Cin / kBK) * kPlaneRowBytes;
    PPlan<BN> pb;
    plan_planes<BN>(pb, [&](int r) { return n0 + r < p.Cout; }, b_row_bytes, tid);
#pragma unroll
    for (int j = 0; j < NVA; ++j) {
        const int r = (tid + kThreads * j) >> 2;
        const long long pix = m0 + r;
        const bool row_ok = pix < Mtot;
        const int rem = (int)((row_ok ? pix : 0) % ((long long)p.H * p.W));
        const int py = rem / p.W, px = rem % p.W;
        unsigned mask = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            if (row_ok && (unsigned)(py + dy) < (unsigned)p.H && (unsigned)(px + dx) < (unsigned)p.W) mask |= 1u << tap;
        }
        a_taps[j] = mask;
        a_off[j] = (unsigned)(r * p.Cin + 4 * ((tid + kThreads * j) & 3)) * 4u;
    }

    const int kt_per_tap = p.Cin / kBK;
    const int total_kt = 9 * kt_per_tap;
    const int kt_begin = slice * kt_per_slice;
    const int kt_end = min(total_kt, kt_begin + kt_per_slice);

    // `live` = false: every load becomes a zero-returning out-of-range access (see gemm_kernel)
    typedef PStage<BN> StageB;
    auto load_tiles = [&](Stage<BM> &sa, StageB &sb, int kt, bool live) {
#if MH_CONV_TAP_MAJOR
        const int tap = min(kt / kt_per_tap, 8);
        const int g16 = kt - tap * kt_per_tap, c0 = g16 * kBK;
#else
        // k order = 16-channel chunk outer, tap inner: the nine k-tiles of a chunk read the same ~(BM + 2W + 2) pixels'
        // 64-byte channel segments, so eight of the nine reads hit L1 / L2 instead of going to the fabric
        const int g16 = kt / 9, tap = kt - 9 * g16, c0 = g16 * kBK;
#endif
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const unsigned a_soff = live ? (unsigned)(halo + (dy * p.W + dx) * p.Cin + c0) * 4u : kDeadTile;
        const unsigned bit = 1u << tap;
#pragma unroll
        for (int j = 0; j < NVA; ++j) sa.v[j] = buffer_load4(ga, (a_taps[j] & bit) ? a_off[j] : kOobOffset, a_soff);
        load_planes<BN>(sb, pb, gb, live ? (unsigned)(tap * p.Cout + n0) * b_row_bytes + (unsigned)g16 * kPlaneRowBytes : kDeadTile);
    };
    // every tap of output pixel m is staged with the exponent of pixel m (it bounds the whole 3x3 neighbourhood), so the
    // scale is constant along K = (tap, channel) and comes off in the epilogue together with the weight row's
    StageExp<BM> ea;
    load_stage_exp<BM>(ea, p.expA, m0, Mtot, true, tid);
    auto unscale = [&](long long row, int col, float v) { return __builtin_ldexpf(v, -(p.expA[row] + p.expW[col])); };
    auto store_tiles = [&](const Stage<BM> &sa, const StageB &sb, int buf) {
        store_wm<BM>(sa, As(buf), tid, ea);
        store_planes<BN>(sb, pb, Bs(buf), tid);
    };

    Acc acc;
    acc_zero(acc);
    // prefetch distance 2, branch-free half-steps (see gemm_kernel): tile kt in LDS, tile kt+1 in one register
    // stage, tile kt+2 in flight; tiles beyond kt_end are zeros
    Stage<BM> sa0, sa1;
    StageB sb0, sb1;
    load_tiles(sa0, sb0, kt_begin, true);
    store_tiles(sa0, sb0, 0);
    load_tiles(sa1, sb1, kt_begin + 1, kt_begin + 1 < kt_end);
    __syncthreads();
    auto step = [&](auto PAR, int kt) {
        constexpr int cur = decltype(PAR)::value;
        Stage<BM> &sa_next = cur ? sa0 : sa1, &sa_far = cur ? sa1 : sa0;
        StageB &sb_next = cur ? sb0 : sb1, &sb_far = cur ? sb1 : sb0;
        auto load_far = [&]() { load_tiles(sa_far, sb_far, kt + 2, kt + 2 < kt_end); };
        auto store_next = [&]() { store_tiles(sa_next, sb_next, cur ^ 1); };
        half_step<BM, BN>(load_far, store_next, As(cur), Bs(cur), wm, wn, lane, acc);
    };
    for (int kt = kt_begin; kt < kt_end; kt += 2) {
        step(std::integral_constant<int, 0>{}, kt);
        step(std::integral_constant<int, 1>{}, kt + 1);
    }

    if (nslices > 1) {
        // partial sums of this K slice: rows are numbered from the first row of the block's region (body / tail)
        const long long region_row0 = is_tail ? p.tail_row0 : 0, region_rows = is_tail ? Mtot - p.tail_row0 : p.tail_row0;
        float *dst = (is_tail ? p.partial_tail : p.partial) + (size_t)slice * region_rows * p.Cout;
        acc_foreach_pair<true, true>(acc, wm, wn, lane, [&](int r, int c0, int c1, float v0, float v1) {
            const long long row = m0 + r;
            if (row >= Mtot) return;
            float *q = dst + (size_t)(row - region_row0) * p.Cout + n0;
            if (n0 + c0 < p.Cout) v0 = unscale(row, n0 + c0, v0);
            if (n0 + c1 < p.Cout) v1 = unscale(row, n0 + c1, v1);
            if (n0 + c0 < p.Cout) q[c0] = v0;
            if (n0 + c1 < p.Cout) q[c1] = v1;
        });
        return;
    }
    // a lane holds output channels c0 and c0 + 32 of its rows: the 32 lanes of a half-wave store 128 contiguous bytes
    const int col0 = n0 + wn + (lane & 31), col1 = col0 + 32;
    const float bias0 = (p.bias && col0 < p.Cout) ? p.bias[col0] : 0.f, bias1 = (p.bias && col1 < p.Cout) ? p.bias[col1] : 0.f;
    acc_foreach_pair<true, true>(acc, wm, wn, lane, [&](int r, int, int, float v0, float v1) {
        const long long row = m0 + r;
        if (row >= Mtot) return;
        float *q = p.out + (size_t)row * p.Cout;
        if (col0 < p.Cout) v0 = unscale(row, col0, v0);
        if (col1 < p.Cout) v1 = unscale(row, col1, v1);
        if (col0 < p.Cout) q[col0] = conv_epi(v0 + bias0, p.epilogue);
        if (col1 < p.Cout) q[col1] = conv_epi(v1 + bias1, p.epilogue);
    });
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3 / stride 1 / pad 1 conv as an implicit GEMM (no patch matrix):
//     dW[co][tap][ci] = sum over pixels  gy[pix][co] * x[pix + shift(tap)][ci]        (0 where the tap leaves the image)
// M' = Cout, N' = 9*Cin, K' = B*H*W pixels.  Both operands are k-major (rows = pixels): A = gy with planned loads;
// B = x, where a staging thread's four columns lie inside one tap (Cin % 4 == 0), so its row shift is a per-thread
// constant folded into the planned offset, and which taps are valid for a pixel comes from a 9-bit mask per pixel
// (tap_mask_kernel), fetched one k-tile ahead so the address select never waits on it.  K' is huge and the output
// small, so the k range is split over blockIdx.y (partials reduced by splitk_reduce).  Operand offsets are rebased
// to the first pixel of the block's k range (32-bit offsets inside the 1 GiB descriptor).
// ---------------------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float *gy, *x;
    const unsigned short *tapmask;   // [P rounded up to 16 + 16]: bit t = tap t of this pixel reads inside the image
    int W, Cin, Cout;
    long long P;
    int tiles_m, tiles_n;
    int splitk, ktiles_per_split;
    float *out;       // [Cout][9*Cin] (splitk == 1)
    float *partial;   // [splitk][Cout][9*Cin]
    const int *expA, *expB;   // f16x3: exponent per output channel of gy [Cout] / per input channel of x [Cin], over ALL pixels
};

__global__ void tap_mask_kernel(int B, int H, int W, long long P, long long Ppad, unsigned short *__restrict__ mask)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < Ppad; i += (long long)blockDim.x * gridDim.x) {
        unsigned m = 0;
        if (i < P) {
            const int rem = (int)(i % ((long long)H * W)), y = rem / W, x = rem % W;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                if ((unsigned)(y + t / 3 - 1) < (unsigned)H && (unsigned)(x + t % 3 - 1) < (unsigned)W) m |= 1u << t;
        }
        mask[i] = (unsigned short)m;
    }
}

template <int BM, int BN>
__global__ __launch_bounds__(kThreads, MH_MINW) void conv3x3_wgrad_kernel(const WgradArgs p)
{
    constexpr int FA = TileGeom<BM, false>::floats, FB = TileGeom<BN, false>::floats;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    auto As = [&](int buf) -> float * { return lds + buf * (FA + FB); };
    auto Bs = [&](int buf) -> float * { return lds + buf * (FA + FB) + FA; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int wm, wn;
    wave_origin<BM, BN>(wave, wm, wn);
    const int t = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int m0 = (t / p.tiles_n) * BM, n0 = (t % p.tiles_n) * BN;
    const int N = 9 * p.Cin;
    const long long total_kt = (p.P + kBK - 1) / kBK;
    const long long kt_begin = (long long)blockIdx.y * p.ktiles_per_split;
    const long long kt_end = min(total_kt, kt_begin + p.ktiles_per_split);
    const long long pix_begin = kt_begin * kBK;
    const int halo = p.W + 1;

    // operands rebased to the block's first pixel (B one halo earlier: every shifted row has a non-negative offset)
    const GSrc ga = make_gsrc(p.gy + pix_begin * p.Cout + m0);
    const GSrc gb = make_gsrc(p.x + (pix_begin - halo) * p.Cin);
    const GSrc gm = make_gsrc(reinterpret_cast<const float *>(p.tapmask + pix_begin));
    const int ktail = (p.P % kBK) ? (int)(p.P % kBK) : kBK;
    Plan<BM> pa;
    plan_km<BM>(pa, p.Cout - m0, p.Cout, ktail, tid);
    constexpr int NTB = (2 * BN + kThreads - 1) / kThreads;
    unsigned b_off[2 * NTB], b_bit[NTB], m_off[NTB], m_reg[NTB];
#pragma unroll
    for (int jt = 0; jt < NTB; ++jt) {
        const int tt = tid + kThreads * jt;
        const int q = 8 * (tt >> 6) + (tt & 7), kp = (tt >> 3) & 7;
        const int col = n0 + 4 * q;
        const bool ok = (col < N) && ((2 * BN >= kThreads) || (tt < 2 * BN));
        const int tap = ok ? col / p.Cin : 0, ci = ok ? col % p.Cin : 0;
        const int shift = (tap / 3 - 1) * p.W + (tap % 3 - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) b_off[2 * jt + i] = (unsigned)((2 * kp + i + halo + shift) * p.Cin + ci) * 4u;
        b_bit[jt] = ok ? (1u << tap) : 0u;
        m_off[jt] = (unsigned)(2 * kp) * 2u;              // two consecutive 16-bit masks = one dword
    }
    auto load_mask = [&](long long kt, int jt) -> unsigned {
        const auto raw = __builtin_amdgcn_raw_buffer_load_b32(gm.rsrc, (int)m_off[jt], (int)((unsigned)(kt - kt_begin) * kBK * 2u), 0);
        return __builtin_bit_cast(unsigned, raw);
    };
#pragma unroll
    for (int jt = 0; jt < NTB; ++jt) m_reg[jt] = load_mask(kt_begin, jt);

    // tiles are requested strictly in order kt_begin, kt_begin+1, ...: m_reg always holds the mask of the tile being
    // requested and is refilled with the next tile's mask (one step of flight time)
    auto load_tiles = [&](Stage<BM> &sa, Stage<BN> &sb, long long kt, bool live) {
        const unsigned rel = (unsigned)(kt - kt_begin) * kBK;
        const bool tail = (kt + 1) * kBK > p.P;
        load_planned_km<BM>(sa, pa, ga, live ? rel * (unsigned)p.Cout * 4u : kDeadTile, tail);
        const unsigned b_soff = live ? rel * (unsigned)p.Cin * 4u : kDeadTile;
#pragma unroll
        for (int jt = 0; jt < NTB; ++jt) {
            const unsigned m = m_reg[jt];
            m_reg[jt] = load_mask(min(kt + 1, total_kt), jt);
            sb.v[2 * jt] = buffer_load4(gb, ((m & 0xffffu) & b_bit[jt]) ? b_off[2 * jt] : kOobOffset, b_soff);
            sb.v[2 * jt + 1] = buffer_load4(gb, ((m >> 16) & b_bit[jt]) ? b_off[2 * jt + 1] : kOobOffset, b_soff);
        }
    };
    // f16x3: K = pixels, so an operand "row" is a channel: gy columns scale by their channel's exponent, x columns
    // (tap, ci) by the exponent of ci (a shift does not change what the channel's maximum bounds)
    StageExp<BM> ea;
    StageExp<BN> eb;
    load_stage_exp<BM>(ea, p.expA, m0, p.Cout, false, tid, true);      // |x| maxima as bit patterns (launch_operand_absmax)
#pragma unroll
    for (int jt = 0; jt < NTB; ++jt) {
        int q, kp;
        km_task<BN>(tid + kThreads * jt, q, kp);
#pragma unroll
        for (int j = 0; j < 4; ++j) eb.km[jt][j] = (n0 + 4 * q + j < N) ? row_exponent((unsigned)p.expB[(n0 + 4 * q + j) % p.Cin]) : 0;
    }
    auto store_tiles = [&](const Stage<BM> &sa, const Stage<BN> &sb, int buf) {
        store_km<BM>(sa, As(buf), tid, ea);
        store_km<BN>(sb, Bs(buf), tid, eb);
    };

    Acc acc;
    acc_zero(acc);
    Stage<BM> sa0, sa1;
    Stage<BN> sb0, sb1;
    load_tiles(sa0, sb0, kt_begin, true);
    store_tiles(sa0, sb0, 0);
    load_tiles(sa1, sb1, kt_begin + 1, kt_begin + 1 < kt_end);
    __syncthreads();
    auto step = [&](auto PAR, long long kt) {
        constexpr int cur = decltype(PAR)::value;
        Stage<BM> &sa_next = cur ? sa0 : sa1, &sa_far = cur ? sa1 : sa0;
        Stage<BN> &sb_next = cur ? sb0 : sb1, &sb_far = cur ? sb1 : sb0;
        auto load_far = [&]() { load_tiles(sa_far, sb_far, kt + 2, kt + 2 < kt_end); };
        auto store_next = [&]() { store_tiles(sa_next, sb_next, cur ^ 1); };
        half_step<BM, BN>(load_far, store_next, As(cur), Bs(cur), wm, wn, lane, acc);
    };
    for (long long kt = kt_begin; kt < kt_end; kt += 2) {
        step(std::integral_constant<int, 0>{}, kt);
        step(std::integral_constant<int, 1>{}, kt + 1);
    }

    float *dst = (p.splitk > 1) ? p.partial + (size_t)blockIdx.y * p.Cout * N : p.out;
    acc_foreach_pair<false, false>(acc, wm, wn, lane, [&](int r, int c0, int c1, float v0, float v1) {
        const int row = m0 + r, col0 = n0 + c0, col1 = n0 + c1;
        if (row >= p.Cout) return;
        if (col0 < N) v0 = __builtin_ldexpf(v0, -(row_exponent((unsigned)p.expA[row]) + row_exponent((unsigned)p.expB[col0 % p.Cin])));
        if (col1 < N) v1 = __builtin_ldexpf(v1, -(row_exponent((unsigned)p.expA[row]) + row_exponent((unsigned)p.expB[col1 % p.Cin])));
        float *q = dst + (size_t)row * N;
        if (col1 < N) *reinterpret_cast<float2 *>(q + col0) = make_float2(v0, v1);      // N % 4 == 0: col0 is even
        else if (col0 < N) q[col0] = v0;
    });
}

// Packed weights of a 3x3 conv with N output and K input channels (for the dgrad conv the channel roles are swapped
// and the taps mirrored: flip_transpose): element (tap, n, k) = w[n][k][tap], or w[k][n][8 - tap] when flipped.
//   bf16-plane build: wt[tap][n][k / 16][24 dwords] = the LDS row image of one k-tile (hi | mid | lo, dword d of a
//   plane = k pair (2d, 2d+1)), so the kernel stages B by 16-byte copies with no split;
//   f32-MFMA build:   wt[tap][n][k] fp32 (K-contiguous rows) in the same buffer.
__global__ void pack_weight_kernel(const float *__restrict__ w, int N, int K, int flip_transpose, int src_cout,
                                   int src_cin, float *__restrict__ wt)
{
    auto src = [&](int tap, int n, int k) -> float {
        return flip_transpose ? w[((size_t)k * src_cin + n) * 9 + (8 - tap)] : w[((size_t)n * src_cin + k) * 9 + tap];
    };
    // wt[tap][n][k / 16][16 dwords] = h1 | h2 of w * 2^e[n]; the exponents e[N] (weight_exp_kernel) follow the planes
    constexpr int row_dw = kPlaneRowBytes / 4;
    const long long total = 9LL * N * (K / kBK) * row_dw;
    unsigned *out = reinterpret_cast<unsigned *>(wt);
    const int *exps = reinterpret_cast<const int *>(out + total);
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int d = idx % row_dw, plane = d / 8, kp = d % 8;
        long long t = idx / row_dw;
        const int g16 = t % (K / kBK); t /= (K / kBK);
        const int n = t % N;
        const int tap = (int)(t / N);
        unsigned pl[2];
        split_pair_f16(src(tap, n, g16 * kBK + 2 * kp), src(tap, n, g16 * kBK + 2 * kp + 1), exps[n], exps[n], pl[0], pl[1]);
        out[idx] = pl[plane];
    }
    (void)src_cout;
}

// exponent of output channel n over its 9*K weights (one block per n), written behind the planes
__global__ void weight_exp_kernel(const float *__restrict__ w, int N, int K, int flip_transpose, int src_cin,
                                  int *__restrict__ exps)
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
    if (threadIdx.x == 0) exps[n] = row_exponent(red[0]);
}
// exponent of OUTPUT pixel p = the exponent of the largest |x| over the pixels its nine taps read (pmax = per-pixel
// maxima as float bits)
__global__ void pixel_exp_kernel(const unsigned *__restrict__ pmax, int B, int H, int W, int *__restrict__ exps)
{
    const long long P = (long long)B * H * W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < P; p += (long long)blockDim.x * gridDim.x) {
        const int rem = (int)(p % ((long long)H * W)), y = rem / W, x = rem % W;
        unsigned m = 0;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx)
                if ((unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W) m = max(m, pmax[p + dy * W + dx]);
        exps[p] = row_exponent(m);
    }
}

// Stem conv: NCHW image (Cin small, e.g. 3) -> NHWC, bias + activation.  Direct VALU kernel: the layer is
// bandwidth-bound (1.2 GFLOP vs 90 MB written per 592x592 image).  One thread = one pixel x 16 output channels;
// the 16 channel-groups of a pixel sit in adjacent lanes, so stores are 64-B runs forming full lines and the
// (re-)loads of the 27 inputs hit L1.  Weights are staged in LDS as [tap*Cin][Cout].
constexpr int kStemCo = 16;
__global__ __launch_bounds__(256) void conv_first_kernel(const float *__restrict__ in, int B, int Cin, int H, int W,
                                                         const float *__restrict__ w, int Cout,
                                                         const float *__restrict__ bias, int epilogue,
                                                         float *__restrict__ out, unsigned *__restrict__ maxbits)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [9*Cin][Cout] + bias[Cout]
    const int K = 9 * Cin;
    for (int i = threadIdx.x; i < K * Cout; i += blockDim.x) {
        const int co = i % Cout, k = i / Cout;       // k = tap*Cin + ci
        const int tap = k / Cin, ci = k % Cin;
        wl[i] = w[((size_t)co * Cin + ci) * 9 + tap];
    }
    for (int i = threadIdx.x; i < Cout; i += blockDim.x) wl[K * Cout + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const int groups = Cout / kStemCo;
    const long long total = (long long)B * H * W * groups;
    int cur_b = -1;            // per-image largest |output| (maxbits != nullptr): running maximum while the image is unchanged
    unsigned vmax = 0;
    // every block owns one contiguous range of (pixel, channel group) items: neighbouring lanes share input pixels in L1 as
    // before, and a thread crosses an image boundary at most once (a grid-stride loop would flush an atomic per iteration)
    const long long per_block = ((total + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const long long lo = blockIdx.x * per_block, hi = min(total, lo + per_block);
    for (long long idx = lo + threadIdx.x; idx < hi; idx += 256) {
        const int grp = idx % groups;
        const long long pix = idx / groups;
        const int x = pix % W;
        const int y = (pix / W) % H;
        const int b = pix / ((long long)W * H);
        if (maxbits && b != cur_b) {
            if (cur_b >= 0 && vmax) atomicMax(maxbits + cur_b, vmax);
            cur_b = b;
            vmax = 0;
        }
        float acc[kStemCo];
#pragma unroll
        for (int j = 0; j < kStemCo; ++j) acc[j] = wl[K * Cout + grp * kStemCo + j];
        for (int ci = 0; ci < Cin; ++ci) {
            const float *plane = in + ((size_t)b * Cin + ci) * H * W;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                float v = 0.f;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = plane[(size_t)yy * W + xx];
                const float *wr = wl + (tap * Cin + ci) * Cout + grp * kStemCo;
#pragma unroll
                for (int j = 0; j < kStemCo; ++j) acc[j] = fmaf(v, wr[j], acc[j]);
            }
        }
        float *o = out + (size_t)pix * Cout + grp * kStemCo;
#pragma unroll
        for (int j = 0; j < kStemCo; j += 4) {
            float4 v = make_float4(conv_epi(acc[j], epilogue), conv_epi(acc[j + 1], epilogue),
                                   conv_epi(acc[j + 2], epilogue), conv_epi(acc[j + 3], epilogue));
            *reinterpret_cast<float4 *>(o + j) = v;
            vmax = max(max(vmax, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu,
                       max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)));
        }
    }
    if (maxbits) pl::wave_atomic_max(maxbits, cur_b < 0 ? 0 : cur_b, cur_b < 0 ? 0u : vmax);
}

// 2x2 stride-2 max pool, NHWC, float4 over channels (C % 4 == 0)
__global__ void maxpool2x2_nhwc_kernel(const float4 *__restrict__ in, int B, int H, int W, int C4,
                                       float4 *__restrict__ out)
{
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)B * Ho * Wo * C4;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int c = idx % C4;
        long long t = idx / C4;
        const int xo = t % Wo; t /= Wo;
        const int yo = t % Ho;
        const int b = t / Ho;
        const float4 *p = in + (((size_t)b * H + 2 * yo) * W + 2 * xo) * C4 + c;
        const float4 a = p[0], bq = p[C4], cq = p[(size_t)W * C4], d = p[(size_t)W * C4 + C4];
        float4 m;
        m.x = fmaxf(fmaxf(a.x, bq.x), fmaxf(cq.x, d.x));
        m.y = fmaxf(fmaxf(a.y, bq.y), fmaxf(cq.y, d.y));
        m.z = fmaxf(fmaxf(a.z, bq.z), fmaxf(cq.z, d.z));
        m.w = fmaxf(fmaxf(a.w, bq.w), fmaxf(cq.w, d.w));
        out[idx] = m;
    }
}

// backward of the 2x2/2 max pool: the gradient of a pooled element goes to the FIRST maximal element of its window
// in scan order (0,0),(0,1),(1,0),(1,1) (what torch's max_pool2d does); one thread per INPUT float4, so every input
// element -- including an odd trailing row / column outside all windows -- is written exactly once (no memset).
__global__ void maxpool2x2_bwd_nhwc_kernel(const float4 *__restrict__ in, const float4 *__restrict__ gout, int B, int H,
                                           int W, int C4, float4 *__restrict__ gin)
{
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)B * H * W * C4;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int c = idx % C4;
        long long t = idx / C4;
        const int x = t % W; t /= W;
        const int y = t % H;
        const int b = t / H;
        const int yo = y >> 1, xo = x >> 1;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (yo < Ho && xo < Wo) {
            const float4 *p = in + (((size_t)b * H + 2 * yo) * W + 2 * xo) * C4 + c;
            const float4 v[4] = {p[0], p[C4], p[(size_t)W * C4], p[(size_t)W * C4 + C4]};
            const float4 g = gout[(((size_t)b * Ho + yo) * Wo + xo) * C4 + c];
            const int me = (y & 1) * 2 + (x & 1);
            auto first_max = [&](float a0, float a1, float a2, float a3) {
                const float m = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
                return a0 == m ? 0 : a1 == m ? 1 : a2 == m ? 2 : 3;
            };
            if (first_max(v[0].x, v[1].x, v[2].x, v[3].x) == me) r.x = g.x;
            if (first_max(v[0].y, v[1].y, v[2].y, v[3].y) == me) r.y = g.y;
            if (first_max(v[0].z, v[1].z, v[2].z, v[3].z) == me) r.z = g.z;
            if (first_max(v[0].w, v[1].w, v[2].w, v[3].w) == me) r.w = g.w;
        }
        gin[idx] = r;
    }
}

// gradient through the fused conv epilogue: out = g where the activation was in its linear range (y = the ACTIVATED
// output: y > 0 for ReLU, 0 < y < 6 for ReLU6), 0 elsewhere
__global__ void act_bwd_kernel(const float4 *__restrict__ g, const float4 *__restrict__ y, long long n4, int epilogue,
                               float4 *__restrict__ out)
{
    const float hi = (epilogue == MH_EPI_RELU6) ? 6.f : INFINITY;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n4;
         idx += (long long)blockDim.x * gridDim.x) {
        const float4 gv = g[idx], yv = y[idx];
        out[idx] = make_float4((yv.x > 0.f && yv.x < hi) ? gv.x : 0.f, (yv.y > 0.f && yv.y < hi) ? gv.y : 0.f,
                               (yv.z > 0.f && yv.z < hi) ? gv.z : 0.f, (yv.w > 0.f && yv.w < hi) ? gv.w : 0.f);
    }
}

// generic NHWC patch matrix: out[(b,yo,xo)][(ky*kw+kx)*C + c]; columns >= kh*kw*C (padding up to ldo) are zeroed
__global__ void im2col_nhwc_kernel(const float *__restrict__ in, int B, int H, int W, int C, int kh, int kw,
                                   int stride, int pad, int Ho, int Wo, float *__restrict__ out, int ldo)
{
    const long long total = (long long)B * Ho * Wo * ldo;
    const int Kc = kh * kw * C;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int col = idx % ldo;
        const long long row = idx / ldo;
        float v = 0.f;
        if (col < Kc) {
            const int c = col % C;
            const int tap = col / C;
            const int ky = tap / kw, kx = tap % kw;
            const int xo = row % Wo;
            const int yo = (row / Wo) % Ho;
            const int b = row / ((long long)Wo * Ho);
            const int y = yo * stride - pad + ky, x = xo * stride - pad + kx;
            if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) v = in[(((size_t)b * H + y) * W + x) * C + c];
        }
        out[idx] = v;
    }
}

// [B,C,H*W] <-> [B,H*W,C] through a 32x32 LDS tile
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, int rows, int cols,
                                                        float *__restrict__ out)
{
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const float *src = in + (size_t)b * rows * cols;
    float *dst = out + (size_t)b * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < rows && c0 + tx < cols) tile[j][tx] = src[(size_t)(r0 + j) * cols + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < cols && r0 + tx < rows) dst[(size_t)(c0 + j) * rows + r0 + tx] = tile[tx][j];
}

}  // namespace mh

using namespace mh;

extern "C" {

size_t mh_conv3x3_packed_floats(int Cout, int Cin)
{
    if (Cout <= 0 || Cin <= 0) return 0;
    // planes, then a tail of 9 * Cout dwords whose first Cout hold the channel exponents (the total stays a multiple of
    // 9 * Cout, which is how callers shape the opaque container)
    return (size_t)9 * Cout * (((Cin + kBK - 1) / kBK) * (kPlaneRowBytes / 4) + 1);
}

int mh_conv3x3_pack_weight(const float *w, int Cout, int Cin, int flip_transpose, float *wt, void *stream)
{
    MH_REQUIRE(w && wt && Cout > 0 && Cin > 0);
    // the conv that consumes the packed weights has N output and K input channels
    const int N = flip_transpose ? Cin : Cout, K = flip_transpose ? Cout : Cin;
    MH_REQUIRE(K % kBK == 0);
    const long long total = (long long)mh_conv3x3_packed_floats(N, K);
    int *exps = reinterpret_cast<int *>(wt) + (size_t)9 * N * (K / kBK) * (kPlaneRowBytes / 4);
    hipLaunchKernelGGL(weight_exp_kernel, dim3(N), dim3(256), 0, as_stream(stream), w, N, K, flip_transpose, Cin, exps);
    {
        const int rc_e = check_launch("weight_exp_kernel");
        if (rc_e) return rc_e;
    }
    const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, N, K, flip_transpose, Cout,
                       Cin, wt);
    return check_launch("pack_weight_kernel");
}

}  // extern "C"

// Tile schedule of one conv launch (see ConvArgs) for block tiles of bm x bn.  kConvSlots = blocks resident at once
// (2 per CU: VGPR-bound).  Shared with conv_planes.hip.
namespace mh {
ConvTilePlan plan_conv_tiles(long long M, int Cin, int Cout, int bm, int bn)
{
    const int kSlots = resident_slots();
    ConvTilePlan sc;
    sc.tiles_m = (int)((M + bm - 1) / bm);
    sc.tiles_n = ceil_div(Cout, bn);
    const int total_kt = 9 * (Cin / kBK);
    const long long tiles = (long long)sc.tiles_m * sc.tiles_n;
    const double t1 = 2.0 * 9 * Cin * (double)bm * bn / (170e12 / 256.0);   // seconds per tile on a fully occupied CU
    const double out_bytes = (double)M * Cout * 4.0;
    static const int cand[] = {1, 2, 3, 4, 5, 6, 8};
    double best = 1e30;
    sc.splitk = 1; sc.body_mtiles = sc.tiles_m; sc.tail_slices = 1;
    // A/B switch for measurements: MH_CONV_SCHEDULE=uniform restores the round-1 schedule (every tile split alike)
    static const bool uniform = [] { const char *e = getenv("MH_CONV_SCHEDULE"); return e && e[0] == 'u'; }();
    if (uniform) {
        sc.splitk = choose_splitk_tiles(tiles, total_kt, (double)M * Cout, 2.0 * 9 * Cin * (double)Cout * M);
        return sc;
    }
    for (int s0 : cand) {
        if (s0 > 1 && (total_kt / s0 < 12 || tiles * s0 > 4 * kSlots)) break;
        const long long bpm = (long long)sc.tiles_n * s0;              // blocks per m-tile
        const long long rounds = tiles * s0 / kSlots;                  // full rounds of resident blocks
        long long body_m = (rounds >= 1) ? std::min<long long>(sc.tiles_m, rounds * kSlots / bpm) : 0;
        if (rounds == 0 || body_m <= 0) body_m = sc.tiles_m;           // everything is resident at once: no tail to cut off
        const long long body_blocks = body_m * bpm;
        const long long tail_tiles = ((long long)sc.tiles_m - body_m) * sc.tiles_n;
        // the tail's K slices (>= 8 k-tiles each): minimise  rounds(tail blocks) / slices  + the partial-sum round trip
        int tsl = 1;
        double tail_cost = 0.0;
        if (tail_tiles > 0) {
            double best_tail = 1e30;
            const int max_sl = std::max(1, std::min(total_kt / 8, 64));
            for (int c = s0; c <= std::max(s0, max_sl); ++c) {
                const long long tb = tail_tiles * c;
                const double cst = makespan_units(tb) * t1 / c + (c > 1 ? (double)tb * bm * bn * 8.0 / 4.0e12 + 3e-6 : 0.0);
                if (cst < best_tail * 0.97) { best_tail = cst; tsl = c; }
            }
            tail_cost = best_tail;
        }
        double cost;
        if (rounds == 0) cost = makespan_units(tiles * s0) * t1 / s0;
        else cost = makespan_units(body_blocks) * t1 / s0 + tail_cost;
        if (s0 > 1) cost += out_bytes * 2.0 * s0 / 4.0e12 + 4e-6;     // body partial sums: write + read
        if (cost < best * 0.97) {
            best = cost;
            sc.splitk = s0; sc.body_mtiles = (int)body_m; sc.tail_slices = tsl;
        }
    }
    if (sc.body_mtiles == sc.tiles_m) sc.tail_slices = 1;
    return sc;
}
}  // namespace mh
using namespace mh;
extern "C" {

struct ConvSchedule {
    int bm, bn, tiles_m, tiles_n;
    int splitk;          // uniform K split of the body tiles
    int body_mtiles;     // m-tiles (all their n-tiles) in the body
    int tail_slices;     // K slices of each tail tile (1 = no split)
};

static ConvSchedule conv_schedule(long long M, int Cin, int Cout)
{
    ConvSchedule sc;
    const bool narrow = (Cout <= 64);
    sc.bm = narrow ? 256 : 128;
    sc.bn = narrow ? 64 : 128;
    const ConvTilePlan pl = plan_conv_tiles(M, Cin, Cout, sc.bm, sc.bn);
    sc.tiles_m = pl.tiles_m; sc.tiles_n = pl.tiles_n; sc.splitk = pl.splitk; sc.body_mtiles = pl.body_mtiles;
    sc.tail_slices = pl.tail_slices;
    return sc;
}

// bytes of partial sums behind the pixel exponents (f16x3) in the workspace: body region, then tail region
static void conv_partial_bytes(const ConvSchedule &sc, long long M, int Cin, int Cout, size_t &body, size_t &tail)
{
    const int total_kt = 9 * (Cin / kBK);
    const long long row0 = std::min<long long>(M, (long long)sc.body_mtiles * sc.bm);
    const int s0 = ceil_div(total_kt, ceil_div(total_kt, sc.splitk));
    body = (s0 > 1) ? align_up((size_t)s0 * row0 * Cout * sizeof(float), 256) : 0;
    const int tk = ceil_div(total_kt, sc.tail_slices), ts = ceil_div(total_kt, tk);
    tail = (ts > 1 && row0 < M) ? align_up((size_t)ts * (M - row0) * Cout * sizeof(float), 256) : 0;
}

size_t mh_conv3x3_ws_bytes(int B, int H, int W, int Cin, int Cout)
{
    const long long M = (long long)B * H * W;
    if (M <= 0 || Cin <= 0 || Cout <= 0 || Cin % kBK != 0) return 0;
    const ConvSchedule sc = conv_schedule(M, Cin, Cout);
    size_t body, tail;
    conv_partial_bytes(sc, M, Cin, Cout, body, tail);
    const size_t exps = 2 * align_up((size_t)M * sizeof(int), 256);   // pixel exponents + their scratch
    return exps + body + tail;
}

int mh_conv3x3_schedule(int B, int H, int W, int Cin, int Cout, int *out8_host)
{
    const long long M = (long long)B * H * W;
    MH_REQUIRE(out8_host && M > 0 && Cin > 0 && Cin % kBK == 0 && Cout > 0);
    const ConvSchedule sc = conv_schedule(M, Cin, Cout);
    const int total_kt = 9 * (Cin / kBK);
    const int tk = ceil_div(total_kt, sc.tail_slices);
    out8_host[0] = sc.bm; out8_host[1] = sc.bn; out8_host[2] = sc.tiles_m; out8_host[3] = sc.tiles_n;
    out8_host[4] = ceil_div(total_kt, ceil_div(total_kt, sc.splitk));
    out8_host[5] = sc.body_mtiles;
    out8_host[6] = (sc.tiles_m - sc.body_mtiles) * sc.tiles_n;
    out8_host[7] = (sc.tiles_m > sc.body_mtiles) ? ceil_div(total_kt, tk) : 1;
    return MH_OK;
}

int mh_conv3x3_nhwc(const float *in, int B, int H, int W, int Cin, const float *wt, int Cout, const float *bias,
                    int epilogue, float *out, void *workspace, size_t ws_bytes, void *stream)
{
    MH_REQUIRE(in && wt && out && B > 0 && H > 0 && W > 0);
    MH_REQUIRE(Cin > 0 && Cin % kBK == 0 && Cout > 0 && Cout % 4 == 0);
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(wt) | reinterpret_cast<uintptr_t>(out)) & 15) == 0);
    // 32-bit buffer offsets: block-relative for the input (tile + two halos), absolute for the packed weights
    MH_REQUIRE((2LL * (W + 1) + 256) * Cin * 4 < (1LL << 30) && (long long)mh_conv3x3_packed_floats(Cout, Cin) * 4 < (1LL << 30));
    ConvArgs p;
    p.in = in; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.wt = wt; p.Cout = Cout; p.bias = bias;
    p.epilogue = epilogue; p.out = out;
    const long long M = (long long)B * H * W;
    ConvSchedule sc = conv_schedule(M, Cin, Cout);
    const bool narrow = (sc.bn == 64);
    p.tiles_m = sc.tiles_m;
    p.tiles_n = sc.tiles_n;
    const long long ntiles = (long long)p.tiles_m * p.tiles_n;
    MH_REQUIRE(ntiles < (1LL << 29));
    const int total_kt = 9 * (Cin / kBK);
    {   // pixel exponents at the head of the workspace (mh_conv3x3_ws_bytes counts them): mandatory in this build
        const size_t eb = align_up((size_t)M * sizeof(int), 256);
        MH_REQUIRE(workspace && ws_bytes >= 2 * eb);
        int *exps = reinterpret_cast<int *>(workspace);
        int *pmax = reinterpret_cast<int *>(reinterpret_cast<char *>(workspace) + eb);
        int rc_e = launch_row_exponents(in, true, M, Cin, Cin, pmax, as_stream(stream), /*bits_only=*/true);
        if (rc_e) return rc_e;
        hipLaunchKernelGGL(pixel_exp_kernel, dim3((unsigned)std::min<long long>((M + 255) / 256, 8192)), dim3(256), 0,
                           as_stream(stream), reinterpret_cast<const unsigned *>(pmax), B, H, W, exps);
        rc_e = check_launch("pixel_exp_kernel");
        if (rc_e) return rc_e;
        p.expA = exps;
        p.expW = reinterpret_cast<const int *>(wt) + (size_t)9 * Cout * (Cin / kBK) * (kPlaneRowBytes / 4);
        workspace = reinterpret_cast<char *>(workspace) + 2 * eb;
        ws_bytes -= 2 * eb;
    }
    size_t body_bytes, tail_bytes;
    conv_partial_bytes(sc, M, Cin, Cout, body_bytes, tail_bytes);
    if (body_bytes + tail_bytes > 0 && (workspace == nullptr || ws_bytes < body_bytes + tail_bytes)) {
        sc.splitk = 1; sc.body_mtiles = sc.tiles_m; sc.tail_slices = 1;     // no room for partial sums: whole tiles only
        body_bytes = tail_bytes = 0;
    }
    p.body_tiles = sc.body_mtiles * sc.tiles_n;
    p.ktiles_per_split = ceil_div(total_kt, sc.splitk);
    p.splitk = ceil_div(total_kt, p.ktiles_per_split);
    p.tail_tiles = (sc.tiles_m - sc.body_mtiles) * sc.tiles_n;
    p.tail_ktiles = ceil_div(total_kt, sc.tail_slices);
    p.tail_slices = ceil_div(total_kt, p.tail_ktiles);
    p.tail_row0 = std::min<long long>(M, (long long)sc.body_mtiles * sc.bm);
    p.partial = reinterpret_cast<float *>(workspace);
    p.partial_tail = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + body_bytes);
    const long long nblocks = (long long)p.tail_tiles * p.tail_slices + (long long)p.body_tiles * p.splitk;
    MH_REQUIRE(nblocks > 0 && nblocks < (1LL << 31));
    dim3 grid((unsigned)nblocks);
    if (narrow)
        launch_tile_kernel<conv3x3_nhwc_kernel<256, 64>>(grid, tile_lds_bytes<256, 64, true, true>(), as_stream(stream), p);
    else
        launch_tile_kernel<conv3x3_nhwc_kernel<128, 128>>(grid, tile_lds_bytes<128, 128, true, true>(), as_stream(stream), p);
    int rc = check_launch("conv3x3_nhwc_kernel");
    if (rc) return rc;
    if (p.splitk > 1 && p.tail_row0 > 0)
        rc = launch_splitk_reduce(p.partial, p.splitk, p.tail_row0, Cout, out, Cout, bias, epilogue, 0, as_stream(stream));
    if (!rc && p.tail_tiles > 0 && p.tail_slices > 1)
        rc = launch_splitk_reduce(p.partial_tail, p.tail_slices, M - p.tail_row0, Cout, out + (size_t)p.tail_row0 * Cout, Cout,
                                  bias, epilogue, 0, as_stream(stream));
    return rc;
}

// dW [Cout][9*Cin] (tap-major, then input channel) of the 3x3 conv from x [B,H,W,Cin] and gy [B,H,W,Cout], without
// a patch matrix.  Workspace = tap masks + split-K partials (mh_conv3x3_wgrad_ws_bytes).  Returns MH_EUNSUPPORTED in
// the f32-MFMA build (callers then use im2col + GEMM).
static int wgrad_splitk(long long P, int Cin, int Cout)
{
    const long long tiles = (long long)ceil_div(Cout, 128) * ceil_div(9 * Cin, 128);
    const long long ktiles = (P + kBK - 1) / kBK;
    int s = choose_splitk_tiles(tiles, (int)std::min<long long>(ktiles, 1 << 30), (double)Cout * 9 * Cin,
                                2.0 * 9 * Cin * (double)Cout * P);
    // 32-bit offsets inside a 1 GiB descriptor: bound the pixels a block walks over
    const long long max_pix = (1LL << 30) / ((long long)std::max(Cin, Cout) * 4) - 2 * 4096;
    while (ceil_div(ktiles, (long long)s) * kBK > max_pix && s < 65535) ++s;
    return s;
}

size_t mh_conv3x3_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout)
{
    const long long P = (long long)B * H * W;
    if (P <= 0 || Cin <= 0 || Cout <= 0) return 0;
    const size_t mask = align_up((size_t)(P + 2 * kBK + 16) * sizeof(unsigned short), 256);
    const int s = wgrad_splitk(P, Cin, Cout);
    const size_t exps = align_up((size_t)Cout * sizeof(int), 256) + align_up((size_t)Cin * sizeof(int), 256);
    return mask + exps + (s > 1 ? align_up((size_t)s * Cout * 9 * Cin * sizeof(float), 256) : 0);
}

int mh_conv3x3_wgrad(const float *x, const float *gy, int B, int H, int W, int Cin, int Cout, float *dw,
                     void *workspace, size_t ws_bytes, void *stream)
{
    MH_REQUIRE(x && gy && dw && workspace && B > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 4 == 0 && Cout > 0 && Cout % 4 == 0);
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(dw) |
                 reinterpret_cast<uintptr_t>(workspace)) & 15) == 0);
    MH_REQUIRE(ws_bytes >= mh_conv3x3_wgrad_ws_bytes(B, H, W, Cin, Cout));
    MH_REQUIRE((long long)(W + 1 + 4096) * std::max(Cin, Cout) * 4 < (1LL << 28));
    const long long P = (long long)B * H * W;
    hipStream_t st = as_stream(stream);
    unsigned short *mask = reinterpret_cast<unsigned short *>(workspace);
    const size_t mask_bytes = align_up((size_t)(P + 2 * kBK + 16) * sizeof(unsigned short), 256);
    const long long Ppad = P + 2 * kBK + 16;
    hipLaunchKernelGGL(tap_mask_kernel, dim3((unsigned)std::min<long long>((Ppad + 255) / 256, 4096)), dim3(256), 0, st, B, H,
                       W, P, Ppad, mask);
    int rc = check_launch("tap_mask_kernel");
    if (rc) return rc;
    WgradArgs p;
    p.gy = gy; p.x = x; p.tapmask = mask; p.W = W; p.Cin = Cin; p.Cout = Cout; p.P = P;
    p.tiles_m = ceil_div(Cout, 128);
    p.tiles_n = ceil_div(9 * Cin, 128);
    const long long total_kt = (P + kBK - 1) / kBK;
    int splitk = wgrad_splitk(P, Cin, Cout);
    p.ktiles_per_split = (int)ceil_div(total_kt, (long long)splitk);
    splitk = (int)ceil_div(total_kt, (long long)p.ktiles_per_split);
    MH_REQUIRE(splitk <= 65535);
    p.splitk = splitk;
    p.out = dw;
    size_t used = mask_bytes;
    {
        int *expA = reinterpret_cast<int *>(reinterpret_cast<char *>(workspace) + used);
        used += align_up((size_t)Cout * sizeof(int), 256);
        int *expB = reinterpret_cast<int *>(reinterpret_cast<char *>(workspace) + used);
        used += align_up((size_t)Cin * sizeof(int), 256);
        rc = launch_operand_absmax(gy, false, Cout, P, Cout, expA, x, false, Cin, P, Cin, expB, st);
        if (rc) return rc;
        p.expA = expA; p.expB = expB;
    }
    p.partial = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + used);
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)splitk);
    launch_tile_kernel<conv3x3_wgrad_kernel<128, 128>>(grid, tile_lds_bytes<128, 128, false, false>(), st, p);
    rc = check_launch("conv3x3_wgrad_kernel");
    if (rc || splitk == 1) return rc;
    return launch_splitk_reduce(p.partial, splitk, Cout, 9 * Cin, dw, 9 * Cin, nullptr, MH_EPI_NONE, 0, st);
}

int mh_conv_first_nchw_max(const float *in_nchw, int B, int Cin, int H, int W, const float *w, int Cout, const float *bias,
                           int epilogue, float *out_nhwc, unsigned *maxbits, void *stream);
int mh_conv_first_nchw(const float *in_nchw, int B, int Cin, int H, int W, const float *w, int Cout,
                       const float *bias, int epilogue, float *out_nhwc, void *stream)
{
    return mh_conv_first_nchw_max(in_nchw, B, Cin, H, W, w, Cout, bias, epilogue, out_nhwc, nullptr, stream);
}

// ... and the per-image largest |output| into maxbits[B] (fp32 bits; must be ZERO before the call) for mh_act_planes
int mh_conv_first_nchw_max(const float *in_nchw, int B, int Cin, int H, int W, const float *w, int Cout, const float *bias,
                           int epilogue, float *out_nhwc, unsigned *maxbits, void *stream)
{
    MH_REQUIRE(in_nchw && w && out_nhwc && B > 0 && Cin > 0 && H > 0 && W > 0);
    MH_REQUIRE(Cout > 0 && Cout % kStemCo == 0);
    MH_REQUIRE((reinterpret_cast<uintptr_t>(out_nhwc) & 15) == 0);
    const size_t lds = ((size_t)9 * Cin * Cout + Cout) * sizeof(float);
    MH_REQUIRE(lds <= 64 * 1024);
    const long long total = (long long)B * H * W * (Cout / kStemCo);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(conv_first_kernel, dim3(blocks), dim3(256), lds, as_stream(stream), in_nchw, B, Cin, H, W, w,
                       Cout, bias, epilogue, out_nhwc, maxbits);
    return check_launch("conv_first_kernel");
}

int mh_maxpool2x2_nhwc(const float *in, int B, int H, int W, int C, float *out, void *stream)
{
    MH_REQUIRE(in && out && B > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0);
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0);
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / 4);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(maxpool2x2_nhwc_kernel, dim3(blocks), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(in), B, H, W, C / 4, reinterpret_cast<float4 *>(out));
    return check_launch("maxpool2x2_nhwc_kernel");
}

int mh_maxpool2x2_bwd_nhwc(const float *in, const float *gout, int B, int H, int W, int C, float *gin, void *stream)
{
    MH_REQUIRE(in && gout && gin && B > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0);
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(gout) | reinterpret_cast<uintptr_t>(gin)) & 15) == 0);
    const long long total = (long long)B * H * W * (C / 4);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(maxpool2x2_bwd_nhwc_kernel, dim3(blocks), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(in), reinterpret_cast<const float4 *>(gout), B, H, W, C / 4,
                       reinterpret_cast<float4 *>(gin));
    return check_launch("maxpool2x2_bwd_nhwc_kernel");
}

int mh_act_bwd(const float *g, const float *y, long long n, int epilogue, float *out, void *stream)
{
    MH_REQUIRE(g && y && out && n > 0 && n % 4 == 0);
    MH_REQUIRE(epilogue == MH_EPI_RELU || epilogue == MH_EPI_RELU6);
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15) == 0);
    const int blocks = (int)std::min<long long>((n / 4 + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4 *>(g),
                       reinterpret_cast<const float4 *>(y), n / 4, epilogue, reinterpret_cast<float4 *>(out));
    return check_launch("act_bwd_kernel");
}

int mh_im2col_nhwc(const float *in, int B, int H, int W, int C, int kh, int kw, int stride, int pad, float *out,
                   int ldo, void *stream)
{
    MH_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0);
    MH_REQUIRE(ldo >= kh * kw * C);
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    MH_REQUIRE(Ho > 0 && Wo > 0);
    const long long total = (long long)B * Ho * Wo * ldo;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(im2col_nhwc_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), in, B, H, W, C, kh, kw,
                       stride, pad, Ho, Wo, out, ldo);
    return check_launch("im2col_nhwc_kernel");
}

static int launch_transpose(const float *in, int B, int rows, int cols, float *out, void *stream)
{
    MH_REQUIRE(in && out && B > 0 && rows > 0 && cols > 0 && B <= 65535);
    MH_REQUIRE(ceil_div(rows, 32) <= 65535);
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(cols, 32), ceil_div(rows, 32), B), dim3(256), 0,
                       as_stream(stream), in, rows, cols, out);
    return check_launch("transpose_kernel");
}

int mh_nchw_to_nhwc(const float *in, int B, int C, int H, int W, float *out, void *stream)
{
    return launch_transpose(in, B, C, H * W, out, stream);
}

int mh_nhwc_to_nchw(const float *in, int B, int C, int H, int W, float *out, void *stream)
{
    return launch_transpose(in, B, H * W, C, out, stream);
}

}  // extern "C"
