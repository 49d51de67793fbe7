// conv_planes.hip -- the frozen VGG trunk's 3x3 convolutions on ACTIVATION PLANES (bf16x6 build only).
//
// conv.hip's kernel reads fp32 NHWC activations and splits every element into its three bf16 terms while staging a
// k-tile (5.5 VALU per element, 9 LDS writes per thread and k-tile, two register stages).  In the trunk that work is
// redundant: a pixel's channel vector is staged by 9 taps x Cout/128 blocks, and nothing but the next conv reads it.
// Here the PRODUCING layer's epilogue writes the split once, in exactly the LDS row image of the tile engine:
//
//     planes[pixel][c / 16][ hi: 16 x bf16 | mid: 16 x bf16 | lo: 16 x bf16 ]        96 B per (pixel, 16 channels)
//
// (same exact truncation split as mfma_tile.h::split_pair; 6 B per element instead of 4), and the consuming kernel's
// staging is pure LDS-DMA: `buffer_load_dwordx4 ... lds` copies 16-byte chunks global -> LDS without touching a VGPR,
// for the pixels and for the packed weights alike.  The K loop is then
//     issue the DMA of k-tile kt+1 (other LDS buffer)  |  MI*3 + NI*3 ds_read_b128  |  MI*NI*6 MFMAs  |  vmcnt(0) + barrier
// with no split VALU, no LDS writes and no staging registers.  The arithmetic is IDENTICAL to conv.hip's kernel: same
// planes, same k order (16-channel chunk outer, tap inner), same order of the six cross terms -- outputs of whole
// (unsplit) tiles are bit-identical, which is how tests/test_gpu_ops.py checks this file.
//
// Tiles: 128x128 (4 waves of 64x64), 256x128 (4 waves of 128x64: the weight fragments are reused by four pixel
// sub-tiles) and 256x64 for 64-channel layers.  Rows of the implicit GEMM are pixels in linear order, or -- `pool` --
// in 2x2-window order (row 4w+q = pixel q of window w): the four pixels of a window then sit in ONE lane's
// accumulator registers (MFMA C layout: rows 4j..4j+3 are the consecutive registers r&3), so ReLU + 2x2 max-pool is
// a register-local max in the epilogue and the pooled tensor is the only thing written.
// Epilogues: fp32 NHWC (direct stores), or planes (per-wave LDS transpose, then 16-byte stores).
// Schedule: conv.hip's (whole tiles in the body, the leftover tiles of the last round K-split, partial sums finished
// by conv_finish_kernel, which also pools / splits).
// Border taps and rows beyond the tensor are out-of-range buffer offsets: the DMA writes zeros (raw buffer semantics).
#include <algorithm>
#include <cstdlib>

#include "mfma_tile.h"

namespace mh {
#if MH_PLANES && !MH_SPLIT_F16

constexpr int kRowB = 96;   // bytes of one operand row of one k-tile (3 planes x 16 bf16)

struct PConvArgs {
    const char *in;          // activation planes of the input [B*H*W][Cin/16][96]
    int B, H, W, Cin;
    const float *wt;         // packed weight planes (mh_conv3x3_pack_weight)
    int Cout;
    const float *bias;
    int epilogue;
    int pool;                // rows in 2x2-window order, epilogue max-pools
    float *out_f32;          // [rows_out][Cout] or null
    char *out_planes;        // [rows_out][Cout/16][96] or null
    int tiles_m, tiles_n;
    int body_tiles, splitk, ktiles_per_split;
    int tail_tiles, tail_slices, tail_ktiles;
    long long tail_row0;
    float *partial, *partial_tail;
};

// GEMM row m -> linear pixel index of the input (and its coordinates)
__device__ __forceinline__ long long row_pixel(long long m, int H, int W, int pool, int &py, int &px)
{
    if (!pool) {
        const int rem = (int)(m % ((long long)H * W));
        py = rem / W;
        px = rem % W;
        return m;
    }
    const int Wo = W >> 1, Ho = H >> 1;
    const long long w = m >> 2;
    const int q = (int)(m & 3);
    const int xo = (int)(w % Wo);
    const long long t = w / Wo;
    const int yo = (int)(t % Ho);
    const long long b = t / Ho;
    py = 2 * yo + (q >> 1);
    px = 2 * xo + (q & 1);
    return (b * H + py) * W + px;
}

typedef __attribute__((address_space(3))) void *lds_ptr_t;

template <int MI, int NI, int WM, int WN>
__global__ __launch_bounds__(kThreads, 2) void conv3x3_planes_kernel(const PConvArgs p)
{
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    constexpr int A_BYTES = BM * kRowB, B_BYTES = BN * kRowB, BUF = A_BYTES + B_BYTES;
    constexpr int NCA = BM * 6 / kThreads, NCB = (BN * 6 + kThreads - 1) / kThreads;
    static_assert(WM * WN == 4 && (BM * 6) % kThreads == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char lds[];   // 2 x BUF
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / WN) * MI * 32, wn = (wave % WN) * NI * 32;

    // block -> (tile, K slice): see ConvArgs in conv.hip
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
    const long long m0 = (long long)(t / p.tiles_n) * BM;
    const int n0 = (t % p.tiles_n) * BN;
    const long long Mrows = (long long)p.B * p.H * p.W;

    // ---- DMA plan: chunk e = tid + 256 j of an operand tile is (row e / 6, LDS slot e % 6); the lane reads the global
    // slot (e % 6) ^ swz(row), so the linear LDS image the DMA writes IS the slot-swizzled row layout of mfma_tile.h
    const unsigned a_row_bytes = (unsigned)(p.Cin / kBK) * kRowB, b_row_bytes = a_row_bytes;
    int py0, px0;
    const long long base_pix = row_pixel(m0, p.H, p.W, p.pool, py0, px0) - (p.W + 1);     // m0 < Mrows
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(p.in) + (ptrdiff_t)base_pix * a_row_bytes, 0, (int)kBufBytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.wt), 0, (int)kBufBytes, 0x00020000);
    unsigned a_voff[NCA], a_mask[NCA], b_voff[NCB];
#pragma unroll
    for (int j = 0; j < NCA; ++j) {
        const int e = tid + kThreads * j, r = e / 6, c = e % 6;
        const long long m = m0 + r;
        const bool row_ok = m < Mrows;
        int py, px;
        const long long pix = row_pixel(row_ok ? m : 0, p.H, p.W, p.pool, py, px);
        unsigned mask = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            if (row_ok && (unsigned)(py + dy) < (unsigned)p.H && (unsigned)(px + dx) < (unsigned)p.W) mask |= 1u << tap;
        }
        a_mask[j] = mask;
        a_voff[j] = row_ok ? (unsigned)(pix - base_pix) * a_row_bytes + 16u * (unsigned)(c ^ plane_swz(r)) : 0u;
    }
#pragma unroll
    for (int j = 0; j < NCB; ++j) {
        const int e = tid + kThreads * j, r = e / 6, c = e % 6;
        b_voff[j] = (e < BN * 6 && n0 + r < p.Cout) ? (unsigned)r * b_row_bytes + 16u * (unsigned)(c ^ plane_swz(r)) : kOobOffset;
    }

    const int kt_per_tap = p.Cin / kBK;
    const int total_kt = 9 * kt_per_tap;
    const int kt_begin = slice * kt_per_slice;
    const int kt_end = min(total_kt, kt_begin + kt_per_slice);
    const unsigned halo_bytes = (unsigned)(p.W + 1) * a_row_bytes;

    auto issue = [&](int kt, int buf) {
        const int g16 = kt / 9, tap = kt - 9 * g16;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const unsigned a_soff = halo_bytes + (unsigned)(dy * p.W + dx) * a_row_bytes + (unsigned)g16 * kRowB;
        const unsigned b_soff = (unsigned)(tap * p.Cout + n0) * b_row_bytes + (unsigned)g16 * kRowB;
        const unsigned bit = 1u << tap;
        char *dst = lds + buf * BUF + wave * 1024;
#pragma unroll
        for (int j = 0; j < NCA; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(dst + j * 4096), 16,
                                                     (int)((a_mask[j] & bit) ? a_voff[j] : kOobOffset), (int)a_soff, 0, 0);
#pragma unroll
        for (int j = 0; j < NCB; ++j)
            if ((BN * 6) % kThreads == 0 || kThreads * j + 64 * wave < BN * 6)      // wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(dst + A_BYTES + j * 4096), 16, (int)b_voff[j],
                                                         (int)b_soff, 0, 0);
    };

    // per-lane fragment offsets inside a buffer: row (w? + i), slot (2 plane + g) ^ swz; sub-tile s adds 32 rows
    const int fi = lane & 31, fg = lane >> 5, fs = (fi >> 3) & 1;
    unsigned offA[3], offB[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        offA[pl] = (unsigned)(wm + fi) * kRowB + 16u * (unsigned)((2 * pl + fg) ^ fs);
        offB[pl] = (unsigned)A_BYTES + (unsigned)(wn + fi) * kRowB + 16u * (unsigned)((2 * pl + fg) ^ fs);
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    issue(kt_begin, 0);
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int buf = (kt - kt_begin) & 1;
        if (kt + 1 < kt_end) issue(kt + 1, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        const char *base = lds + buf * BUF;
        bf16x8 a[MI][3], b[NI][3];
        constexpr int kOrderA[3] = {2, 0, 1}, kOrderB[3] = {0, 2, 1};   // planes in order of first use
#pragma unroll
        for (int o = 0; o < 3; ++o) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
                a[i][kOrderA[o]] = *reinterpret_cast<const bf16x8 *>(base + offA[kOrderA[o]] + i * 32 * kRowB);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                b[j][kOrderB[o]] = *reinterpret_cast<const bf16x8 *>(base + offB[kOrderB[o]] + j * 32 * kRowB);
        }
        // six cross terms per accumulator, smallest first (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi): the order of
        // mfma_tile.h::mma_frags, so the sums are bit-identical to conv.hip's kernel
        constexpr int kTermA[6] = {2, 0, 1, 1, 0, 0}, kTermB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int tt = 0; tt < 6; ++tt)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][kTermA[tt]], b[j][kTermB[tt]], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);   // keep every MFMA of this tile ahead of the wait: the DMA flies under them
        __syncthreads();      // s_waitcnt vmcnt(0) lgkmcnt(0) + barrier: the next tile has landed, this one is consumed
    }

    // ------------------------------------------------------------------------------------------------ epilogue
    const int j31 = lane & 31, g = lane >> 5;
    if (nslices > 1) {
        // partial sums of this K slice, rows in GEMM order relative to the region (body / tail)
        const long long region_row0 = is_tail ? p.tail_row0 : 0, region_rows = is_tail ? Mrows - p.tail_row0 : p.tail_row0;
        float *dst = (is_tail ? p.partial_tail : p.partial) + (size_t)slice * region_rows * p.Cout;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (row >= Mrows) continue;
                float *q = dst + (size_t)(row - region_row0) * p.Cout;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int col = n0 + wn + 32 * j + j31;
                    if (col < p.Cout) q[col] = acc[i][j][r];
                }
            }
        return;
    }
    float biasv[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int col = n0 + wn + 32 * j + j31;
        biasv[j] = (p.bias && col < p.Cout) ? p.bias[col] : 0.f;
    }
    auto epi = [&](float v, int j) {
        v += biasv[j];
        if (p.epilogue == MH_EPI_RELU) v = fmaxf(v, 0.f);
        else if (p.epilogue == MH_EPI_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
        return v;
    };
    if (p.out_f32) {
        // direct stores: the 32 lanes of a half-wave write 128 contiguous bytes of one output row
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (p.pool) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const long long row = m0 + wm + 32 * i + 8 * r4 + 4 * g;        // first row of the window
                    if (row >= Mrows) continue;
                    float *q = p.out_f32 + (size_t)(row >> 2) * p.Cout;
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int col = n0 + wn + 32 * j + j31;
                        const float v = fmaxf(fmaxf(epi(acc[i][j][4 * r4], j), epi(acc[i][j][4 * r4 + 1], j)),
                                              fmaxf(epi(acc[i][j][4 * r4 + 2], j), epi(acc[i][j][4 * r4 + 3], j)));
                        if (col < p.Cout) q[col] = v;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * g;
                    if (row >= Mrows) continue;
                    float *q = p.out_f32 + (size_t)row * p.Cout;
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int col = n0 + wn + 32 * j + j31;
                        if (col < p.Cout) q[col] = epi(acc[i][j][r], j);
                    }
                }
            }
        }
        return;
    }
    // planes out: transpose 32-row slabs of the wave's tile through a private LDS region (the K loop's buffers are
    // free: its last barrier has passed), then every lane splits 8 consecutive channels of a row and stores 3 x 16 B
    constexpr int LDW = NI * 32 + 4;
    float *sc = reinterpret_cast<float *>(lds) + wave * 32 * LDW;
    const unsigned out_row_bytes = (unsigned)(p.Cout / kBK) * kRowB;
    constexpr int CG = NI * 32 / 8;      // 8-channel groups per slab row
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if (p.pool) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    sc[(2 * r4 + g) * LDW + 32 * j + j31] =
                        fmaxf(fmaxf(epi(acc[i][j][4 * r4], j), epi(acc[i][j][4 * r4 + 1], j)),
                              fmaxf(epi(acc[i][j][4 * r4 + 2], j), epi(acc[i][j][4 * r4 + 3], j)));
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    sc[((r & 3) + 8 * (r >> 2) + 4 * g) * LDW + 32 * j + j31] = epi(acc[i][j][r], j);
        }
        const int nrows = p.pool ? 8 : 32;
        for (int pc = lane; pc < nrows * CG; pc += 64) {
            const int rl = pc / CG, cg = pc % CG;
            const long long first = m0 + wm + 32 * i;                       // first GEMM row of the slab
            const long long orow = p.pool ? (first >> 2) + rl : first + rl;
            const long long lim = p.pool ? (Mrows >> 2) : Mrows;
            const int col = n0 + wn + 8 * cg;
            if (orow >= lim || col >= p.Cout) continue;
            const float4 x0 = *reinterpret_cast<const float4 *>(sc + rl * LDW + 8 * cg);
            const float4 x1 = *reinterpret_cast<const float4 *>(sc + rl * LDW + 8 * cg + 4);
            unsigned h[4], md[4], lo[4];
            split_pair(x0.x, x0.y, h[0], md[0], lo[0]);
            split_pair(x0.z, x0.w, h[1], md[1], lo[1]);
            split_pair(x1.x, x1.y, h[2], md[2], lo[2]);
            split_pair(x1.z, x1.w, h[3], md[3], lo[3]);
            char *q = p.out_planes + (size_t)orow * out_row_bytes + (size_t)(col >> 4) * kRowB + ((col >> 3) & 1) * 16;
            *reinterpret_cast<u32x4 *>(q) = (u32x4){h[0], h[1], h[2], h[3]};
            *reinterpret_cast<u32x4 *>(q + 32) = (u32x4){md[0], md[1], md[2], md[3]};
            *reinterpret_cast<u32x4 *>(q + 64) = (u32x4){lo[0], lo[1], lo[2], lo[3]};
        }
    }
}

// Finish K-split tiles: out row o (o counted from the region's first OUTPUT row) = act(sum over slices + bias), max over
// the window's four GEMM rows when pooling; fp32 or planes.  One thread per (output row, 8 channels).
__global__ __launch_bounds__(256) void conv_finish_kernel(const float *__restrict__ partial, int nslices,
                                                          long long region_rows, int Cout, const float *__restrict__ bias,
                                                          int epilogue, int pool, float *__restrict__ out_f32,
                                                          char *__restrict__ out_planes, long long out_row0)
{
    const int cgn = Cout / 8;
    const long long orows = pool ? (region_rows >> 2) : region_rows;
    const long long total = orows * cgn;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int cg = (int)(idx % cgn);
        const long long o = idx / cgn;
        float best[8];
        const int nr = pool ? 4 : 1;
        for (int q = 0; q < nr; ++q) {
            const long long row = pool ? 4 * o + q : o;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < nslices; ++s) {
                const float4 *src = reinterpret_cast<const float4 *>(partial + ((size_t)s * region_rows + row) * Cout + 8 * cg);
                const float4 x0 = src[0], x1 = src[1];
                v[0] += x0.x; v[1] += x0.y; v[2] += x0.z; v[3] += x0.w;
                v[4] += x1.x; v[5] += x1.y; v[6] += x1.z; v[7] += x1.w;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float y = v[k] + (bias ? bias[8 * cg + k] : 0.f);
                if (epilogue == MH_EPI_RELU) y = fmaxf(y, 0.f);
                else if (epilogue == MH_EPI_RELU6) y = fminf(fmaxf(y, 0.f), 6.f);
                best[k] = (q == 0) ? y : fmaxf(best[k], y);
            }
        }
        const long long orow = out_row0 + o;
        if (out_f32) {
            float4 *dst = reinterpret_cast<float4 *>(out_f32 + (size_t)orow * Cout + 8 * cg);
            dst[0] = make_float4(best[0], best[1], best[2], best[3]);
            dst[1] = make_float4(best[4], best[5], best[6], best[7]);
        } else {
            unsigned h[4], md[4], lo[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) split_pair(best[2 * k], best[2 * k + 1], h[k], md[k], lo[k]);
            char *q = out_planes + (size_t)orow * (Cout / kBK) * kRowB + (size_t)(cg >> 1) * kRowB + (cg & 1) * 16;
            *reinterpret_cast<u32x4 *>(q) = (u32x4){h[0], h[1], h[2], h[3]};
            *reinterpret_cast<u32x4 *>(q + 32) = (u32x4){md[0], md[1], md[2], md[3]};
            *reinterpret_cast<u32x4 *>(q + 64) = (u32x4){lo[0], lo[1], lo[2], lo[3]};
        }
    }
}

// fp32 [rows][C] <-> planes [rows][C/16][96]: one thread per (row, 8 channels)
__global__ __launch_bounds__(256) void f32_to_planes_kernel(const float *__restrict__ x, long long rows, int C,
                                                            char *__restrict__ planes)
{
    const int cgn = C / 8;
    const long long total = rows * cgn;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int cg = (int)(idx % cgn);
        const long long r = idx / cgn;
        const float4 *src = reinterpret_cast<const float4 *>(x + (size_t)r * C + 8 * cg);
        const float4 x0 = src[0], x1 = src[1];
        unsigned h[4], md[4], lo[4];
        split_pair(x0.x, x0.y, h[0], md[0], lo[0]);
        split_pair(x0.z, x0.w, h[1], md[1], lo[1]);
        split_pair(x1.x, x1.y, h[2], md[2], lo[2]);
        split_pair(x1.z, x1.w, h[3], md[3], lo[3]);
        char *q = planes + (size_t)r * (C / kBK) * kRowB + (size_t)(cg >> 1) * kRowB + (cg & 1) * 16;
        *reinterpret_cast<u32x4 *>(q) = (u32x4){h[0], h[1], h[2], h[3]};
        *reinterpret_cast<u32x4 *>(q + 32) = (u32x4){md[0], md[1], md[2], md[3]};
        *reinterpret_cast<u32x4 *>(q + 64) = (u32x4){lo[0], lo[1], lo[2], lo[3]};
    }
}

__global__ __launch_bounds__(256) void planes_to_f32_kernel(const char *__restrict__ planes, long long rows, int C,
                                                            float *__restrict__ x)
{
    const int cgn = C / 8;
    const long long total = rows * cgn;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int cg = (int)(idx % cgn);
        const long long r = idx / cgn;
        const char *q = planes + (size_t)r * (C / kBK) * kRowB + (size_t)(cg >> 1) * kRowB + (cg & 1) * 16;
        const u32x4 h = *reinterpret_cast<const u32x4 *>(q), md = *reinterpret_cast<const u32x4 *>(q + 32),
                    lo = *reinterpret_cast<const u32x4 *>(q + 64);
        float o[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // hi + mid + lo is exact: the three terms are the truncation split of one fp32 number
            o[2 * k] = (__builtin_bit_cast(float, h[k] << 16) + __builtin_bit_cast(float, md[k] << 16)) +
                       __builtin_bit_cast(float, lo[k] << 16);
            o[2 * k + 1] = (__builtin_bit_cast(float, h[k] & 0xffff0000u) + __builtin_bit_cast(float, md[k] & 0xffff0000u)) +
                           __builtin_bit_cast(float, lo[k] & 0xffff0000u);
        }
        float4 *dst = reinterpret_cast<float4 *>(x + (size_t)r * C + 8 * cg);
        dst[0] = make_float4(o[0], o[1], o[2], o[3]);
        dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
}

// ---- host side ------------------------------------------------------------------------------------------------
struct PSchedule {
    int bm, bn, tiles_m, tiles_n, splitk, body_mtiles, tail_slices;
};

static PSchedule planes_schedule(long long M, int Cin, int Cout)
{
    // MH_PCONV_TILE = 128 | 256 pins the pixel-tile height of the >= 128-channel layers (A/B runs)
    static const int forced = [] { const char *e = getenv("MH_PCONV_TILE"); return e ? atoi(e) : 0; }();
    PSchedule s;
    s.bn = (Cout <= 64) ? 64 : 128;
    if (s.bn == 64) s.bm = 256;
    else if (forced == 128 || forced == 256) s.bm = forced;
    else s.bm = ((M + 255) / 256) * ceil_div(Cout, 128) >= resident_slots() ? 256 : 128;   // at least one full round of 256-row tiles
    const ConvTilePlan pl = plan_conv_tiles(M, Cin, Cout, s.bm, s.bn);
    s.tiles_m = pl.tiles_m; s.tiles_n = pl.tiles_n; s.splitk = pl.splitk; s.body_mtiles = pl.body_mtiles;
    s.tail_slices = pl.tail_slices;
    return s;
}

static void planes_partial_bytes(const PSchedule &sc, long long M, int Cin, int Cout, size_t &body, size_t &tail)
{
    const int total_kt = 9 * (Cin / kBK);
    const long long row0 = std::min<long long>(M, (long long)sc.body_mtiles * sc.bm);
    const int s0 = ceil_div(total_kt, ceil_div(total_kt, sc.splitk));
    body = (s0 > 1) ? align_up((size_t)s0 * row0 * Cout * sizeof(float), 256) : 0;
    const int tk = ceil_div(total_kt, sc.tail_slices), ts = ceil_div(total_kt, tk);
    tail = (ts > 1 && row0 < M) ? align_up((size_t)ts * (M - row0) * Cout * sizeof(float), 256) : 0;
}

template <int MI, int NI, int WM, int WN>
static void launch_planes(dim3 grid, hipStream_t st, const PConvArgs &p)
{
    constexpr size_t lds = 2 * (size_t)(WM * MI * 32 + WN * NI * 32) * kRowB;
    launch_tile_kernel<conv3x3_planes_kernel<MI, NI, WM, WN>>(grid, lds, st, p);
}

#endif  // MH_PLANES && !MH_SPLIT_F16
}  // namespace mh

using namespace mh;

extern "C" {

size_t mh_planes_bytes(long long rows, int C)
{
    if (rows <= 0 || C <= 0 || C % 16 != 0) return 0;
    return (size_t)rows * (C / 16) * 96;
}

#if MH_PLANES && !MH_SPLIT_F16
int mh_f32_to_planes(const float *x, long long rows, int C, void *planes, void *stream)
{
    MH_REQUIRE(x && planes && rows > 0 && C > 0 && C % 16 == 0);
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(planes)) & 15) == 0);
    const long long total = rows * (C / 8);
    hipLaunchKernelGGL(f32_to_planes_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 256 * 16)), dim3(256), 0,
                       as_stream(stream), x, rows, C, reinterpret_cast<char *>(planes));
    return check_launch("f32_to_planes_kernel");
}

int mh_planes_to_f32(const void *planes, long long rows, int C, float *x, void *stream)
{
    MH_REQUIRE(x && planes && rows > 0 && C > 0 && C % 16 == 0);
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(planes)) & 15) == 0);
    const long long total = rows * (C / 8);
    hipLaunchKernelGGL(planes_to_f32_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 256 * 16)), dim3(256), 0,
                       as_stream(stream), reinterpret_cast<const char *>(planes), rows, C, x);
    return check_launch("planes_to_f32_kernel");
}

size_t mh_conv3x3_planes_ws_bytes(int B, int H, int W, int Cin, int Cout)
{
    const long long M = (long long)B * H * W;
    if (M <= 0 || Cin <= 0 || Cout <= 0 || Cin % kBK != 0) return 0;
    const PSchedule sc = planes_schedule(M, Cin, Cout);
    size_t body, tail;
    planes_partial_bytes(sc, M, Cin, Cout, body, tail);
    return body + tail;
}

int mh_conv3x3_planes(const void *in_planes, int B, int H, int W, int Cin, const float *wt, int Cout, const float *bias,
                      int epilogue, int pool, void *out_planes, float *out_f32, void *workspace, size_t ws_bytes,
                      void *stream)
{
    MH_REQUIRE(in_planes && wt && B > 0 && H > 0 && W > 0);
    MH_REQUIRE((out_planes != nullptr) != (out_f32 != nullptr));
    MH_REQUIRE(Cin > 0 && Cin % kBK == 0 && Cout > 0 && Cout % (out_planes ? 16 : 4) == 0);
    MH_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0));
    MH_REQUIRE(epilogue >= MH_EPI_NONE && epilogue <= MH_EPI_RELU6);
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(in_planes) | reinterpret_cast<uintptr_t>(wt) | reinterpret_cast<uintptr_t>(out_planes) |
                 reinterpret_cast<uintptr_t>(out_f32)) & 15) == 0);
    const long long a_row_bytes = (long long)(Cin / kBK) * kRowB;
    // 32-bit buffer offsets: a tile of <= 256 rows in window order spans at most 256/2 + 2 columns of two image rows,
    // plus one halo on each side; weights absolute
    MH_REQUIRE((4LL * (W + 1) + 512) * a_row_bytes < (1LL << 30) && (long long)mh_conv3x3_packed_floats(Cout, Cin) * 4 < (1LL << 30));
    const long long M = (long long)B * H * W;
    PSchedule sc = planes_schedule(M, Cin, Cout);
    size_t body_bytes, tail_bytes;
    planes_partial_bytes(sc, M, Cin, Cout, body_bytes, tail_bytes);
    if (body_bytes + tail_bytes > 0 && (workspace == nullptr || ws_bytes < body_bytes + tail_bytes)) {
        sc.splitk = 1; sc.body_mtiles = sc.tiles_m; sc.tail_slices = 1;
        body_bytes = tail_bytes = 0;
    }
    const int total_kt = 9 * (Cin / kBK);
    PConvArgs p;
    p.in = reinterpret_cast<const char *>(in_planes);
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.wt = wt; p.Cout = Cout; p.bias = bias; p.epilogue = epilogue;
    p.pool = pool ? 1 : 0; p.out_f32 = out_f32; p.out_planes = reinterpret_cast<char *>(out_planes);
    p.tiles_m = sc.tiles_m; p.tiles_n = sc.tiles_n;
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
    MH_REQUIRE(Cout % 8 == 0 || (p.splitk == 1 && (p.tail_tiles == 0 || p.tail_slices == 1)));      // conv_finish_kernel: 8 channels per thread
    hipStream_t st = as_stream(stream);
    dim3 grid((unsigned)nblocks);
    if (sc.bn == 64) launch_planes<2, 2, 4, 1>(grid, st, p);
    else if (sc.bm == 256) launch_planes<4, 2, 2, 2>(grid, st, p);
    else launch_planes<2, 2, 2, 2>(grid, st, p);
    int rc = check_launch("conv3x3_planes_kernel");
    if (rc) return rc;
    auto finish = [&](const float *partial, int nslices, long long row0, long long rows) -> int {
        const long long orow0 = pool ? (row0 >> 2) : row0;
        const long long total = (pool ? (rows >> 2) : rows) * (Cout / 8);
        hipLaunchKernelGGL(conv_finish_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 256 * 16)), dim3(256), 0, st,
                           partial, nslices, rows, Cout, bias, epilogue, p.pool, out_f32, reinterpret_cast<char *>(out_planes), orow0);
        return check_launch("conv_finish_kernel");
    };
    if (p.splitk > 1 && p.tail_row0 > 0) rc = finish(p.partial, p.splitk, 0, p.tail_row0);
    if (!rc && p.tail_tiles > 0 && p.tail_slices > 1) rc = finish(p.partial_tail, p.tail_slices, p.tail_row0, M - p.tail_row0);
    return rc;
}
#else   // builds without bf16 planes (f32-MFMA, f16x3): the callers keep the fp32 path
int mh_f32_to_planes(const float *, long long, int, void *, void *) { return MH_EUNSUPPORTED; }
int mh_planes_to_f32(const void *, long long, int, float *, void *) { return MH_EUNSUPPORTED; }
size_t mh_conv3x3_planes_ws_bytes(int, int, int, int, int) { return 0; }
int mh_conv3x3_planes(const void *, int, int, int, int, const float *, int, const float *, int, int, void *, float *, void *,
                      size_t, void *)
{
    return MH_EUNSUPPORTED;
}
#endif

}  // extern "C"
