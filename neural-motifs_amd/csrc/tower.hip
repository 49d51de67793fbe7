// tower.hip -- fused BatchNorm(train) / ReLU / max-pool kernels for the union-box mask tower
// (lib/get_union_boxes.py:31-39: Conv7x7/2 -> ReLU -> BN -> MaxPool3x3/2 -> Conv3x3 -> ReLU -> BN, summed with the
// RoIAligned union features).  All tensors NHWC fp32; the convolutions run on the MFMA kernels with ReLU fused into
// their epilogues, so these kernels see POST-ReLU activations (the ReLU backward mask is x > 0).
//
// Why custom: the framework's channels-last BN kernels move 308 MB at < 1 TB/s and make separate passes for
// statistics, normalisation, pooling and the residual add.  Here
//   forward : stats (1 read pass, deterministic two-stage reduction, running stats updated like nn.BatchNorm)
//             -> apply fused with the 3x3/2 max-pool (first BN) or with the NHWC->NCHW residual add (second BN)
//   backward: per-channel sums (1 pass; pooled variant gathers through the saved arg-max) -> dx fused with the
//             ReLU mask.
// HBM-bound: a few passes over [n*14*14, 256] / [n*7*7, 512].
#include <algorithm>
#include <cstdlib>
#include <string>

#include "common.h"

namespace mh {

constexpr int kRowsPerBlock = 128;
constexpr int kBnGroup = 1024;       // channels one block of the statistics kernels covers (grid.y = groups)

// ---- per-channel partial sums of (x, x^2) or (g, g*xhat): block b covers rows [b*128, b*128+128) ---------------
// threads: c4 = tid % (C/4) float4 columns, lane-row = tid / (C/4); partial[b][0/1][C]
template <bool BWD, bool POOL>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                         const unsigned char *__restrict__ argmax,
                                                         const float *__restrict__ mean, const float *__restrict__ invstd,
                                                         long long M, int C, int H, int W, float *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) float red[];   // [rl][2][Cb]
    // blockIdx.y = group of up to 1024 channels (C = 2048 in the ResNet layer4 stacks: two groups); c4 indexes float4
    // columns of the WHOLE row, Cb / C4b are the group's extent
    const int c_off = (int)blockIdx.y * kBnGroup, Cb = min(kBnGroup, C - c_off), C4b = Cb >> 2;
    const int rl = threadIdx.x / C4b, nrl = blockDim.x / C4b;
    const int c4 = (c_off >> 2) + threadIdx.x % C4b, c4l = threadIdx.x % C4b;
    const long long r0 = (long long)blockIdx.x * kRowsPerBlock;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    float4 mu = s0, is = s0;
    if (BWD) {
        mu = reinterpret_cast<const float4 *>(mean)[c4];
        is = reinterpret_cast<const float4 *>(invstd)[c4];
    } else {
        mu = reinterpret_cast<const float4 *>(x)[c4];     // the pivot (row 0)
    }
    if (rl < nrl) {
        for (long long r = r0 + rl; r < std::min(M, r0 + kRowsPerBlock); r += nrl) {
            if (!BWD) {
                // sums of (x - pivot), pivot = row 0 of the tensor: a sample of the channel, so that
                // var = E[(x-p)^2] - E[x-p]^2 does not cancel catastrophically when |mean| >> std
                float4 v = reinterpret_cast<const float4 *>(x + r * C)[c4];
                v.x -= mu.x; v.y -= mu.y; v.z -= mu.z; v.w -= mu.w;
                s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
                s1.x += v.x * v.x; s1.y += v.y * v.y; s1.z += v.z * v.z; s1.w += v.w * v.w;
            } else {
                // r indexes the rows of g.  POOL: g is the pooled gradient [n, H/2, W/2, C]; the matching x element is
                // the arg-max position inside the 3x3/2 window (pad 1) of the [n,H,W,C] input.
                const float4 gv = reinterpret_cast<const float4 *>(g + r * C)[c4];
                float4 xv;
                if (POOL) {
                    const int Ho = H / 2, Wo = W / 2;
                    const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho);
                    const long long n = r / ((long long)Wo * Ho);
                    const uchar4 a = reinterpret_cast<const uchar4 *>(argmax + r * C)[c4];
                    const unsigned char aa[4] = {a.x, a.y, a.z, a.w};
                    float xs[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int iy = 2 * oy - 1 + aa[j] / 3, ix = 2 * ox - 1 + aa[j] % 3;
                        xs[j] = x[((n * H + iy) * W + ix) * C + 4 * c4 + j];
                    }
                    xv = make_float4(xs[0], xs[1], xs[2], xs[3]);
                } else {
                    xv = reinterpret_cast<const float4 *>(x + r * C)[c4];
                }
                s0.x += gv.x; s0.y += gv.y; s0.z += gv.z; s0.w += gv.w;
                s1.x += gv.x * (xv.x - mu.x) * is.x; s1.y += gv.y * (xv.y - mu.y) * is.y;
                s1.z += gv.z * (xv.z - mu.z) * is.z; s1.w += gv.w * (xv.w - mu.w) * is.w;
            }
        }
        reinterpret_cast<float4 *>(red + (size_t)(rl * 2 + 0) * Cb)[c4l] = s0;
        reinterpret_cast<float4 *>(red + (size_t)(rl * 2 + 1) * Cb)[c4l] = s1;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * Cb; i += blockDim.x) {
        float s = 0.f;
        for (int q = 0; q < nrl; ++q) s += red[(size_t)q * 2 * Cb + i];
        partial[(size_t)blockIdx.x * 2 * C + (i < Cb ? c_off + i : C + c_off + (i - Cb))] = s;
    }
}

// Column sums of the per-block partials in double: a block owns 16 channels, its 256 threads are 16 channels x 16
// slices of the nblk partial rows (64-B coalesced reads), combined through LDS.  Returns the two sums to threads
// 0..15 of the block (channel c = blockIdx.x * 16 + threadIdx.x).
constexpr int kFinCh = 16, kFinSl = 16;
__device__ __forceinline__ bool bn_partial_sums(const float *__restrict__ partial, int nblk, int C, double &s_out,
                                                double &ss_out)
{
    __shared__ double red[2][kFinSl][kFinCh];
    const int cl = threadIdx.x % kFinCh, sl = threadIdx.x / kFinCh;
    const int c = blockIdx.x * kFinCh + cl;
    double s = 0.0, ss = 0.0;
    if (c < C) {
        // four partial rows per trip, all eight loads issued before the first add (round 6: one row per trip was a chain of
        // nblk / 16 dependent L2 round trips -- 37 us for the 2352 partial rows of the mask tower's first BatchNorm, r06_c10)
        double s1 = 0.0, s2 = 0.0, s3 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
        int b = sl;
        for (; b + 3 * kFinSl < nblk; b += 4 * kFinSl) {
            const float *p0 = partial + (size_t)b * 2 * C + c, *p1 = p0 + (size_t)kFinSl * 2 * C, *p2 = p1 + (size_t)kFinSl * 2 * C,
                        *p3 = p2 + (size_t)kFinSl * 2 * C;
            const float a0 = p0[0], a1 = p1[0], a2 = p2[0], a3 = p3[0], b0 = p0[C], b1 = p1[C], b2 = p2[C], b3 = p3[C];
            s += (double)a0; s1 += (double)a1; s2 += (double)a2; s3 += (double)a3;
            ss += (double)b0; q1 += (double)b1; q2 += (double)b2; q3 += (double)b3;
        }
        for (; b < nblk; b += kFinSl) {
            s += (double)partial[(size_t)b * 2 * C + c];
            ss += (double)partial[(size_t)b * 2 * C + C + c];
        }
        s = (s + s1) + (s2 + s3);
        ss = (ss + q1) + (q2 + q3);
    }
    red[0][sl][cl] = s;
    red[1][sl][cl] = ss;
    __syncthreads();
    if (sl != 0 || c >= C) return false;
    s = 0.0; ss = 0.0;
#pragma unroll
    for (int k = 0; k < kFinSl; ++k) { s += red[0][k][cl]; ss += red[1][k][cl]; }
    s_out = s;
    ss_out = ss;
    return true;
}

// forward finalize: mean, biased var -> invstd; running stats with momentum (unbiased var), like nn.BatchNorm
__global__ __launch_bounds__(kFinCh * kFinSl) void bn_finalize_fwd_kernel(
    const float *__restrict__ partial, int nblk, int C, long long M, float eps, float momentum,
    const float *__restrict__ pivot, float *__restrict__ mean, float *__restrict__ invstd,
    float *__restrict__ running_mean, float *__restrict__ running_var)
{
    double s, ss;
    if (!bn_partial_sums(partial, nblk, C, s, ss)) return;
    const int c = blockIdx.x * kFinCh + threadIdx.x;
    const double dm = s / (double)M;                    // E[x - pivot], pivot = row 0 (see bn_partial_kernel)
    const double mu = (double)pivot[c] + dm;
    double var = ss / (double)M - dm * dm;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = (M > 1) ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}

// backward finalize: dbeta = sum g, dgamma = sum g*xhat
__global__ __launch_bounds__(kFinCh * kFinSl) void bn_finalize_bwd_kernel(const float *__restrict__ partial, int nblk,
                                                                          int C, float *__restrict__ dgamma,
                                                                          float *__restrict__ dbeta)
{
    double s, ss;
    if (!bn_partial_sums(partial, nblk, C, s, ss)) return;
    const int c = blockIdx.x * kFinCh + threadIdx.x;
    dbeta[c] = (float)s;
    dgamma[c] = (float)ss;
}

__device__ __forceinline__ float bn_affine(float x, float mu, float is, float ga, float be) { return (x - mu) * is * ga + be; }

// y = BN(x) followed by 3x3 / stride 2 / pad 1 max-pool; x [n,H,W,C] -> z [n,H/2,W/2,C], argmax in 0..8 (ky*3+kx)
__global__ void bn_pool_fwd_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                   const float *__restrict__ invstd, const float *__restrict__ gamma,
                                   const float *__restrict__ beta, long long N, int H, int W, int C,
                                   float *__restrict__ z, unsigned char *__restrict__ argmax)
{
    const int Ho = H / 2, Wo = W / 2, C4 = C >> 2;
    const long long total = N * Ho * Wo * C4;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int c4 = idx % C4;
        long long t = idx / C4;
        const int ox = t % Wo; t /= Wo;
        const int oy = t % Ho;
        const long long n = t / Ho;
        const float4 mu = reinterpret_cast<const float4 *>(mean)[c4], is = reinterpret_cast<const float4 *>(invstd)[c4];
        const float4 ga = reinterpret_cast<const float4 *>(gamma)[c4], be = reinterpret_cast<const float4 *>(beta)[c4];
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned char arg[4] = {4, 4, 4, 4};
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int iy = 2 * oy - 1 + k / 3, ix = 2 * ox - 1 + k % 3;
            if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
            const float4 v = reinterpret_cast<const float4 *>(x + ((n * H + iy) * W + ix) * C)[c4];
            const float y[4] = {bn_affine(v.x, mu.x, is.x, ga.x, be.x), bn_affine(v.y, mu.y, is.y, ga.y, be.y),
                                bn_affine(v.z, mu.z, is.z, ga.z, be.z), bn_affine(v.w, mu.w, is.w, ga.w, be.w)};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (y[j] > best[j]) { best[j] = y[j]; arg[j] = (unsigned char)k; }     // first maximum wins
        }
        reinterpret_cast<float4 *>(z)[idx] = make_float4(best[0], best[1], best[2], best[3]);
        reinterpret_cast<uchar4 *>(argmax)[idx] = make_uchar4(arg[0], arg[1], arg[2], arg[3]);
    }
}

// out[n][c][p] = residual[n][c][p] + BN(x)[n][p][c]   (x NHWC [n,P,C] -> out NCHW [n,C,P]); 32x32 LDS transpose
__global__ __launch_bounds__(256) void bn_residual_nchw_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                                               const float *__restrict__ invstd,
                                                               const float *__restrict__ gamma,
                                                               const float *__restrict__ beta,
                                                               const float *__restrict__ residual, int P, int C,
                                                               float *__restrict__ out)
{
    __shared__ float tile[32][33];
    const long long n = blockIdx.z;
    const int c0 = blockIdx.x * 32, p0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float *xs = x + n * (long long)P * C;
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        if (p < P && c < C) tile[j][tx] = bn_affine(xs[(size_t)p * C + c], mean[c], invstd[c], gamma[c], beta[c]);
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        if (p < P && c < C) {
            const size_t o = (n * C + c) * (size_t)P + p;
            out[o] = (residual ? residual[o] : 0.f) + tile[tx][j];
        }
    }
}

// y[m][c] = act(BN(x)[m][c] + residual[m][c])   (all NHWC rows; ResNet bottleneck epilogues, lib/resnet.py:25-46)
__global__ __launch_bounds__(256) void bn_apply_nhwc_kernel(const float4 *__restrict__ x, const float *__restrict__ mean,
                                                            const float *__restrict__ invstd,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta,
                                                            const float4 *__restrict__ residual, long long M, int C,
                                                            int relu, float4 *__restrict__ out)
{
    const int C4 = C >> 2;
    const long long total = M * C4;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int c4 = (int)(idx % C4);
        const float4 mu = reinterpret_cast<const float4 *>(mean)[c4], is = reinterpret_cast<const float4 *>(invstd)[c4];
        const float4 ga = reinterpret_cast<const float4 *>(gamma)[c4], be = reinterpret_cast<const float4 *>(beta)[c4];
        const float4 v = x[idx];
        float4 y = make_float4(bn_affine(v.x, mu.x, is.x, ga.x, be.x), bn_affine(v.y, mu.y, is.y, ga.y, be.y),
                               bn_affine(v.z, mu.z, is.z, ga.z, be.z), bn_affine(v.w, mu.w, is.w, ga.w, be.w));
        if (residual) {
            const float4 r = residual[idx];
            y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
        }
        if (relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
        out[idx] = y;
    }
}

// g_nhwc[n][p][c] = g_nchw[n][c][p]
__global__ __launch_bounds__(256) void nchw_to_nhwc_small_kernel(const float *__restrict__ in, int P, int C,
                                                                 float *__restrict__ out)
{
    __shared__ float tile[32][33];
    const long long n = blockIdx.z;
    const int c0 = blockIdx.x * 32, p0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        if (p < P && c < C) tile[j][tx] = in[(n * C + c) * (size_t)P + p];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        if (p < P && c < C) out[(n * P + p) * (size_t)C + c] = tile[tx][j];
    }
}

// dx = gamma*invstd * (dy - (dbeta + xhat*dgamma)/M) * [x > 0]
// POOL: dy at an input position = sum of the pooled gradients whose window arg-max is this position
template <bool POOL>
__global__ void bn_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                    const unsigned char *__restrict__ argmax, const float *__restrict__ mean,
                                    const float *__restrict__ invstd, const float *__restrict__ gamma,
                                    const float *__restrict__ dgamma, const float *__restrict__ dbeta, long long N, int H,
                                    int W, int C, int relu_mask, float *__restrict__ dx)
{
    const int C4 = C >> 2;
    const long long total = N * H * W * C4;
    const float invM = 1.f / (float)(N * H * W);
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int c4 = idx % C4;
        long long t = idx / C4;
        const int ix = t % W; t /= W;
        const int iy = t % H;
        const long long n = t / H;
        const float4 xv = reinterpret_cast<const float4 *>(x)[idx];
        float dy[4] = {0.f, 0.f, 0.f, 0.f};
        if (POOL) {
            const int Ho = H / 2, Wo = W / 2;
            // windows (oy,ox) with 2*oy-1 <= iy <= 2*oy+1
            for (int oy = (iy) / 2; oy <= (iy + 1) / 2; ++oy) {
                if (oy >= Ho) continue;
                const int ky = iy - (2 * oy - 1);
                for (int ox = (ix) / 2; ox <= (ix + 1) / 2; ++ox) {
                    if (ox >= Wo) continue;
                    const int kx = ix - (2 * ox - 1);
                    const long long o = ((n * Ho + oy) * Wo + ox) * C4 + c4;
                    const uchar4 a = reinterpret_cast<const uchar4 *>(argmax)[o];
                    const float4 gv = reinterpret_cast<const float4 *>(g)[o];
                    const int k = ky * 3 + kx;
                    if (a.x == k) dy[0] += gv.x;
                    if (a.y == k) dy[1] += gv.y;
                    if (a.z == k) dy[2] += gv.z;
                    if (a.w == k) dy[3] += gv.w;
                }
            }
        } else {
            const float4 gv = reinterpret_cast<const float4 *>(g)[idx];
            dy[0] = gv.x; dy[1] = gv.y; dy[2] = gv.z; dy[3] = gv.w;
        }
        const float4 mu = reinterpret_cast<const float4 *>(mean)[c4], is = reinterpret_cast<const float4 *>(invstd)[c4];
        const float4 ga = reinterpret_cast<const float4 *>(gamma)[c4];
        const float4 dg = reinterpret_cast<const float4 *>(dgamma)[c4], db = reinterpret_cast<const float4 *>(dbeta)[c4];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, mus[4] = {mu.x, mu.y, mu.z, mu.w}, iss[4] = {is.x, is.y, is.z, is.w};
        const float gas[4] = {ga.x, ga.y, ga.z, ga.w}, dgs[4] = {dg.x, dg.y, dg.z, dg.w}, dbs[4] = {db.x, db.y, db.z, db.w};
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xhat = (xs[j] - mus[j]) * iss[j];
            float v = gas[j] * iss[j] * (dy[j] - (dbs[j] + xhat * dgs[j]) * invM);
            if (relu_mask && !(xs[j] > 0.f)) v = 0.f;
            r[j] = v;
        }
        reinterpret_cast<float4 *>(dx)[idx] = make_float4(r[0], r[1], r[2], r[3]);
    }
}


// ---- the tower's first convolution, direct (lib/get_union_boxes.py:31: Conv2d(2, dim/2, kernel 7, stride 2, padding 3) + ReLU) ----
// As a column matrix + GEMM this layer wrote 120 MB of patches, read them back and ran a 301056 x 256 x 100 product at 48 TFLOP/s
// (K = 98 does not feed a 16-deep MFMA loop; profiles/r04_gemm_shapes.jsonl: 0.14 + 0.32 ms forward, 0.39 ms weight gradient).
// Direct form: thread = output channel (its 98 weights live in VGPRs), block = one pair's mask; the mask is read from a zero-
// padded copy [N, S+6, S+6, 2] so that a tap needs no bounds test, and every tap address is wave-uniform: the values arrive
// through the scalar cache (s_load_dwordxN) and enter v_fmac_f32 as SGPR operands -- no LDS, no per-lane address arithmetic.
// Exact fp32 FMAs (no f16 split).  VALU-bound: 98 FMA per output, 7.5 G FMA per step at b = 6.
constexpr int kT1K = 7, kT1C = 2, kT1Stride = 2, kT1Pad = 3;
constexpr int kT1Row = kT1K * kT1C;            // 14 contiguous floats of one kernel row in the NHWC mask
constexpr int kT1Taps = kT1K * kT1Row;         // 98

__global__ __launch_bounds__(256) void tower_pad_kernel(const float2 *__restrict__ in, long long N, int S, float2 *__restrict__ out)
{
    const int Sp = S + 2 * kT1Pad;
    const long long total = N * Sp * Sp;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % Sp) - kT1Pad, y = (int)((i / Sp) % Sp) - kT1Pad;
        const long long n = i / ((long long)Sp * Sp);
        out[i] = (x >= 0 && x < S && y >= 0 && y < S) ? in[(n * S + y) * S + x] : make_float2(0.f, 0.f);
    }
}

// y[n, oy, ox, c] = relu(bias[c] + sum_k xp[n, 2 oy + ky, 2 ox + kx, ci] * wk[k, c]),  k = (ky * 7 + kx) * 2 + ci.
// Two neighbouring outputs per trip: their windows share 5 of 9 columns, so one kernel row is ONE run of 18 scalar-loaded
// floats for 28 FMAs (two independent chains) -- half the scalar-cache waits per FMA of the one-output form.
constexpr int kT1Pair = kT1Row + kT1Stride * kT1C;       // 18 floats: the union of two windows' kernel row
__global__ __launch_bounds__(256) void tower_conv1_fwd_kernel(const float *__restrict__ xp, int Sp, int Ho, int Wo,
                                                              const float *__restrict__ wk, const float *__restrict__ bias, int C0,
                                                              float *__restrict__ y)
{
    const int c = (int)blockIdx.y * 256 + (int)threadIdx.x;
    float w[kT1Taps];
#pragma unroll
    for (int k = 0; k < kT1Taps; ++k) w[k] = wk[(size_t)k * C0 + c];
    const float b = bias ? bias[c] : 0.f;
    const float *__restrict__ xn = xp + (size_t)blockIdx.x * Sp * Sp * kT1C;          // wave-uniform from here on
    float *yn = y + (size_t)blockIdx.x * Ho * Wo * C0 + c;
    for (int oy = 0; oy < Ho; ++oy)
        for (int ox = 0; ox < Wo; ox += 2) {                                          // Wo is even (host check)
            const float *__restrict__ win = xn + ((size_t)(oy * kT1Stride) * Sp + ox * kT1Stride) * kT1C;
            float a0 = b, a1 = b;
#pragma unroll
            for (int ky = 0; ky < kT1K; ++ky) {
                float r[kT1Pair];
#pragma unroll
                for (int t = 0; t < kT1Pair; ++t) r[t] = win[(size_t)ky * Sp * kT1C + t];
#pragma unroll
                for (int t = 0; t < kT1Row; ++t) {
                    a0 = fmaf(r[t], w[ky * kT1Row + t], a0);
                    a1 = fmaf(r[t + kT1Stride * kT1C], w[ky * kT1Row + t], a1);
                }
            }
            yn[(size_t)(oy * Wo + ox) * C0] = fmaxf(a0, 0.f);
            yn[(size_t)(oy * Wo + ox + 1) * C0] = fmaxf(a1, 0.f);
        }
}

// partial[b][k][c] = sum over the block's pairs and their Ho*Wo positions of dy[n, p, c] * xp[n, window(p), k]  (k < 98),
// partial[b][98][c] = sum of dy[n, p, c]  (the bias gradient rides along: dy is read once).
// The gradients of one output ROW (Wo <= 16 values per thread) are fetched while the previous row's 98 x Wo FMAs run.
constexpr int kT1MaxW = 16;
__global__ __launch_bounds__(256) void tower_conv1_wgrad_kernel(const float *__restrict__ xp, int Sp, int Ho, int Wo,
                                                                const float *__restrict__ dy, long long N, int C0, int pairs_per_block,
                                                                float *__restrict__ partial)
{
    const int c = (int)blockIdx.y * 256 + (int)threadIdx.x;
    float acc[kT1Taps], accb = 0.f;
#pragma unroll
    for (int k = 0; k < kT1Taps; ++k) acc[k] = 0.f;
    const long long n0 = (long long)blockIdx.x * pairs_per_block, n1 = min(N, n0 + pairs_per_block);
    const int rows = (int)(n1 - n0) * Ho;                                  // output rows of this block, all pairs
    const float *__restrict__ d0 = dy + (size_t)n0 * Ho * Wo * C0 + c;     // row r of the block: d0 + r * Wo * C0
    float cur[kT1MaxW], nxt[kT1MaxW];
#pragma unroll
    for (int j = 0; j < kT1MaxW; ++j) cur[j] = d0[(size_t)min(j, Wo - 1) * C0];          // rows >= 1: the grid covers N exactly
    for (int r = 0; r < rows; ++r) {
        // every read is issued (columns beyond Wo and the row after the last re-read a valid address and are never used): with
        // reads under a branch the compiler cannot count what is in flight and waits for ALL of it before the first FMA
        const int rn = min(r + 1, rows - 1);
#pragma unroll
        for (int j = 0; j < kT1MaxW; ++j) nxt[j] = d0[((size_t)rn * Wo + min(j, Wo - 1)) * C0];
        __builtin_amdgcn_sched_barrier(0);         // the reads stay HERE: the scheduler otherwise sinks them below the FMAs, next to their use
        const int n = r / Ho, oy = r - n * Ho;
        const float *__restrict__ xrow = xp + ((size_t)(n0 + n) * Sp + (size_t)oy * kT1Stride) * Sp * kT1C;
#pragma unroll
        for (int j = 0; j < kT1MaxW; j += 2) {
            if (j < Wo) {                                                  // wave-uniform; Wo is even (host check)
                const float *__restrict__ win = xrow + (size_t)j * kT1Stride * kT1C;
                const float v0 = cur[j], v1 = cur[j + 1];
                accb += v0 + v1;
#pragma unroll
                for (int ky = 0; ky < kT1K; ++ky) {
                    float x[kT1Pair];
#pragma unroll
                    for (int t = 0; t < kT1Pair; ++t) x[t] = win[(size_t)ky * Sp * kT1C + t];
#pragma unroll
                    for (int t = 0; t < kT1Row; ++t) {
                        acc[ky * kT1Row + t] = fmaf(x[t], v0, acc[ky * kT1Row + t]);
                        acc[ky * kT1Row + t] = fmaf(x[t + kT1Stride * kT1C], v1, acc[ky * kT1Row + t]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kT1MaxW; ++j) cur[j] = nxt[j];
    }
    float *out = partial + (size_t)blockIdx.x * (kT1Taps + 1) * C0 + c;
#pragma unroll
    for (int k = 0; k < kT1Taps; ++k) out[(size_t)k * C0] = acc[k];
    out[(size_t)kT1Taps * C0] = accb;
}

// dwk[k][c] = sum_b partial[b][k][c]: fixed order, deterministic
__global__ __launch_bounds__(256) void tower_conv1_wgrad_reduce_kernel(const float *__restrict__ partial, int nblk, int rows_c0,
                                                                       float *__restrict__ dwk)
{
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (i >= rows_c0) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 3 < nblk; b += 4) {
        s0 += partial[(size_t)b * rows_c0 + i];
        s1 += partial[(size_t)(b + 1) * rows_c0 + i];
        s2 += partial[(size_t)(b + 2) * rows_c0 + i];
        s3 += partial[(size_t)(b + 3) * rows_c0 + i];
    }
    for (; b < nblk; ++b) s0 += partial[(size_t)b * rows_c0 + i];
    dwk[i] = (s0 + s1) + (s2 + s3);
}
constexpr int kT1PairsPerBlock = 3;

// ---- Round 6: the same layer on the matrix cores -------------------------------------------------------------------------------
// The direct kernels above are VALU-bound at the fp32 FMA rate: 7.5 G FMA per cfg2 step = 0.31-0.36 ms forward, 0.42-0.46 ms
// weight gradient (profiles/r06_bench_n1_kernel_stats_c9.csv) for 15 GFLOP each.  As products they are
//   forward          y[p][c]   = sum_k' A[p][k'] w[c][k']      p = 301056 output pixels, c = 256, k' = 7 kernel rows x 16
//   weight gradient  dw[k'][c] = sum_p  A[p][k'] dy[p][c]
// with k' = 16 ky + (2 kx + ci): one kernel row of the NHWC mask is 14 CONTIGUOUS floats of the padded copy, padded here to one
// 16-deep k-tile (the two extra columns meet zero weights / are never written out).  No column matrix and no LDS operand tiles:
// every lane builds its MFMA fragments straight from global memory --
//   forward: lane (pixel j, half g) of a 32-pixel tile reads floats 8 g .. 8 g + 7 of the seven window rows (two 16-byte loads),
//     scales by 2^14 (masks lie in [0, 1]) and splits into the two f16 planes (f16x3, as everywhere: pl_tile.h split2); the
//     weights' fragments of the wave's 64 channels (per-channel exponent, two planes, 7 k-tiles) live in 112 VGPRs for the whole
//     launch.  42 MFMAs per wave and 32-pixel tile; the accumulators come out transposed (a lane's register quad = four
//     consecutive channels of one pixel), take bias + ReLU, and leave through a wave-private LDS patch as 256-byte runs.
//     Bound by the 308 MB of y it writes.
//   weight gradient: the contraction runs over pixels, so a k-tile is ONE OUTPUT ROW (14 pixels, padded to 16 with zero
//     gradients): lane (tap row j, half g) reads the mask value under its tap for pixels 8 g .. 8 g + 7 (stride-4 floats), lane
//     (channel j, half g) the 8 gradients of its channel (stride C0: 128-byte runs across the wave).  The gradients have no
//     a-priori scale, so this product is bf16x6 (mfma_tile.h: three bf16 terms per fp32 value, six MFMAs, no maxima pass).  A wave
//     owns 64 channels x all 112 (128) tap rows = 128 accumulator registers; a block owns a contiguous range of output rows and
//     writes ONE partial [99][C0] (bias gradient = exact fp32 sums of the raw gradients, row 98) that the existing fixed-order
//     reduce kernel adds up.  48 MFMAs per wave and output row.
// MH_TOWER_CONV1=valu keeps the direct kernels (A/B).
namespace t1 {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(8)));      // a window row starts on an 8-byte boundary (33 x 2 floats per padded row)
typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(8)));

constexpr int kKy = kT1K;                      // 7 k-tiles (kernel rows)
constexpr int kMaskExp = 14;                   // masks lie in [0, 1]: scaled by 2^14 they fit f16 with the usual headroom

__device__ __forceinline__ void split2_f16(float x0, float x1, int e, unsigned &p1, unsigned &p2)
{
    const f32x2 xs = {__builtin_ldexpf(x0, e), __builtin_ldexpf(x1, e)};
    const f16x2 h1 = __builtin_convertvector(xs, f16x2);
    const f32x2 r = xs - __builtin_convertvector(h1, f32x2);     // exact
    p1 = __builtin_bit_cast(unsigned, h1);
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
__device__ __forceinline__ unsigned pack_bf16(float lo16, float hi16)
{
    const f32x2 v = {lo16, hi16};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split3_bf16(float x0, float x1, unsigned &p1, unsigned &p2, unsigned &p3)
{
    p1 = pack_bf16(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, p1 << 16);               // exact
    const float r1 = x1 - __builtin_bit_cast(float, p1 & 0xffff0000u);
    p2 = pack_bf16(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, p2 << 16);               // exact
    const float s1 = r1 - __builtin_bit_cast(float, p2 & 0xffff0000u);
    p3 = pack_bf16(s0, s1);
}
__device__ __forceinline__ int exponent_of(unsigned absmax_bits)            // pl_tile.h row_exponent
{
    const int biased = (int)(absmax_bits >> 23) & 0xff;
    if (biased == 0 || biased == 0xff) return 0;
    return 14 - (biased - 127);
}

constexpr int kFwdPatchRow = 64 + 4;           // floats per pixel row of the wave's output patch (+4: bank spread)
constexpr int kFwdPatch = 32 * kFwdPatchRow * 4;       // bytes
constexpr int kFwdLds = 4 * kFwdPatch + 256 * 8;       // four patches | bias[256], exponent[256]

// grid (pixel-tile walkers, C0 / 256), block 256: wave w = channels 64 w .. 64 w + 63 of the block's group
__global__ __launch_bounds__(256) void tower_conv1_mfma_fwd_kernel(const float *__restrict__ xp, long long N, int Sp, int Ho, int Wo,
                                                                   const float *__restrict__ wk, const float *__restrict__ bias,
                                                                   int C0, float *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, g = lane >> 5;
    const int cg = (int)blockIdx.y * 256;
    float *patch = reinterpret_cast<float *>(lds + wave * kFwdPatch);
    float *bias_l = reinterpret_cast<float *>(lds + 4 * kFwdPatch);
    int *eb_l = reinterpret_cast<int *>(bias_l + 256);
    // ---- weights of the wave's 64 channels -> per-channel exponent, two f16 planes, fragments in registers
    unsigned bq[2][kKy][2][4];                                             // [n-block][ky][plane][4 dwords = 8 k]
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int c = cg + 64 * wave + 32 * nb + j;
        float wv[kKy][8];
        unsigned m = 0;
#pragma unroll
        for (int ky = 0; ky < kKy; ++ky)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int t = 8 * g + i;                                   // column of the padded kernel row
                wv[ky][i] = t < kT1Row ? wk[(size_t)(ky * kT1Row + t) * C0 + c] : 0.f;
                m = max(m, __float_as_uint(wv[ky][i]) & 0x7fffffffu);
            }
        m = max(m, (unsigned)__shfl_xor((int)m, 32));                      // both halves of the channel's row
        const int e = exponent_of(m);
        if (g == 0) {
            eb_l[64 * wave + 32 * nb + j] = e;
            bias_l[64 * wave + 32 * nb + j] = bias ? bias[c] : 0.f;
        }
#pragma unroll
        for (int ky = 0; ky < kKy; ++ky)
#pragma unroll
            for (int i = 0; i < 4; ++i) split2_f16(wv[ky][2 * i], wv[ky][2 * i + 1], e, bq[nb][ky][0][i], bq[nb][ky][1][i]);
    }
    __syncthreads();
    const long long M = N * Ho * Wo;
    const long long ntiles = (M + 31) / 32;
    const int HW = Ho * Wo;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long p = tile * 32 + j;
        const bool ok = p < M;
        const long long pc = ok ? p : M - 1;
        const long long n = pc / HW;
        const int rem = (int)(pc - n * HW), oy = rem / Wo, ox = rem - oy * Wo;
        const float *win = xp + (((size_t)n * Sp + (size_t)oy * kT1Stride) * Sp + (size_t)ox * kT1Stride) * kT1C + 8 * g;
        f32x16 acc[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        f32x4_u v0[kKy];
        f32x2_u v1[kKy], v2[kKy];
#pragma unroll
        for (int ky = 0; ky < kKy; ++ky) {                                 // floats 8 g .. 8 g + 7 of the window row (14 are real)
            const float *row = win + (size_t)ky * Sp * kT1C;
            v0[ky] = *reinterpret_cast<const f32x4_u *>(row);
            v1[ky] = *reinterpret_cast<const f32x2_u *>(row + 4);
            v2[ky] = g == 0 ? *reinterpret_cast<const f32x2_u *>(row + 6) : (f32x2_u){0.f, 0.f};
        }
#pragma unroll
        for (int ky = 0; ky < kKy; ++ky) {
            unsigned a[2][4];
            split2_f16(v0[ky].x, v0[ky].y, kMaskExp, a[0][0], a[1][0]);
            split2_f16(v0[ky].z, v0[ky].w, kMaskExp, a[0][1], a[1][1]);
            split2_f16(v1[ky].x, v1[ky].y, kMaskExp, a[0][2], a[1][2]);
            split2_f16(v2[ky].x, v2[ky].y, kMaskExp, a[0][3], a[1][3]);
            const f16x8 a1 = __builtin_bit_cast(f16x8, (u32x4){a[0][0], a[0][1], a[0][2], a[0][3]});
            const f16x8 a2 = __builtin_bit_cast(f16x8, (u32x4){a[1][0], a[1][1], a[1][2], a[1][3]});
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const f16x8 b1 = __builtin_bit_cast(f16x8, (u32x4){bq[nb][ky][0][0], bq[nb][ky][0][1], bq[nb][ky][0][2], bq[nb][ky][0][3]});
                const f16x8 b2 = __builtin_bit_cast(f16x8, (u32x4){bq[nb][ky][1][0], bq[nb][ky][1][1], bq[nb][ky][1][2], bq[nb][ky][1][3]});
                // weights first: rows of the result = channels, columns = pixels (transposed accumulators, pl_conv.hip rmma)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a2, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b2, a1, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a1, acc[nb], 0, 0, 0);
            }
        }
        // ---- epilogue: lane (j, g), registers 4 q .. 4 q + 3 = channels 32 nb + 8 q + 4 g + (0..3) of pixel j
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 32 * nb + 8 * q + 4 * g;
                const int4 eb = *reinterpret_cast<const int4 *>(eb_l + 64 * wave + c);
                const float4 bs = *reinterpret_cast<const float4 *>(bias_l + 64 * wave + c);
                float4 o;
                o.x = fmaxf(__builtin_ldexpf(acc[nb][4 * q + 0], -kMaskExp - eb.x) + bs.x, 0.f);
                o.y = fmaxf(__builtin_ldexpf(acc[nb][4 * q + 1], -kMaskExp - eb.y) + bs.y, 0.f);
                o.z = fmaxf(__builtin_ldexpf(acc[nb][4 * q + 2], -kMaskExp - eb.z) + bs.z, 0.f);
                o.w = fmaxf(__builtin_ldexpf(acc[nb][4 * q + 3], -kMaskExp - eb.w) + bs.w, 0.f);
                *reinterpret_cast<float4 *>(patch + j * kFwdPatchRow + c) = o;
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int px = 4 * it + (lane >> 4), ch = 4 * (lane & 15);
            const float4 o = *reinterpret_cast<const float4 *>(patch + px * kFwdPatchRow + ch);
            const long long po = tile * 32 + px;
            if (po < M) *reinterpret_cast<float4 *>(y + (size_t)po * C0 + cg + 64 * wave + ch) = o;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// grid (row-range blocks, C0 / (128 NB)), block 256: wave w = channels 32 NB w .. of the block's group; partial[blockIdx.x][99][C0].
// NB = n-blocks (32 channels) per wave: 2 ships (64 channels per wave, 128 accumulator registers); NB = 1 (MH_TOWER_WGRAD_NB=1: half
// the accumulators, two blocks per CU) measured no faster (202 against 196 us per call, r06_c16).
template <int NB>
__global__ __launch_bounds__(256, 2) void tower_conv1_mfma_wgrad_kernel(const float *__restrict__ xp, long long N, int Sp, int Ho, int Wo,
                                                                                      const float *__restrict__ dy, int C0, int rows_per_block,
                                                                                      float *__restrict__ partial)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, g = lane >> 5;
    const int c0 = (int)blockIdx.y * (128 * NB) + 32 * NB * wave;
    const long long rows = N * Ho;
    const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    // tap row 32 mb + j of the padded k' order: kernel row ky = 2 mb + (j >> 4), column t = j & 15 (14 real)
    const int my_ky = 2 * wave + (j >> 4), my_t = j & 15;                     // the tap block this wave fetches for the block: mb = wave
    const bool my_tap_ok = my_ky < kKy && my_t < kT1Row;
    const int my_tap_off = my_tap_ok ? (my_ky * Sp * kT1C + my_t) : 0;
    f32x16 acc[4][NB];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
    float bsum[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bsum[nb] = 0.f;
    // The mask fragments are the same for every wave of the block: wave w fetches and splits tap block w ONLY and hands its three
    // bf16 planes to the others through LDS (two buffers, one barrier per output row).  With every wave fetching all four tap
    // blocks itself the kernel was bound by its 32 + 8 NB gather loads per row, not by the MFMAs (256 / 273 us at NB = 2 / 1,
    // r06_c10 / r06_c15).
    __shared__ u32x4 a_lds[2][4][3][64];
    float av[8], dv[NB][8];                                                // the raw values of the NEXT output row (tap block `wave`)
    // (two rows in flight were measured on this form as well, r06_c16: 240 / 206 us per call at NB = 2 / 1 against 196 / 202 --
    // the second row's registers cost more than its latency cover gives)
    auto fetch = [&](long long row) {
        const bool rok = row < r1;
        const long long rc = rok ? row : r1 - 1;
        const long long n = rc / Ho;
        const int oy = (int)(rc - n * Ho);
        const float *xrow = xp + ((size_t)n * Sp + (size_t)oy * kT1Stride) * Sp * kT1C + (size_t)(8 * g) * kT1Stride * kT1C;
        const float *drow = dy + ((size_t)rc * Wo + 8 * g) * C0 + c0 + j;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool pok = rok && 8 * g + i < Wo;                        // pixels 14, 15 of the padded row carry nothing
            av[i] = (pok && my_tap_ok) ? xrow[my_tap_off + i * kT1Stride * kT1C] : 0.f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) dv[nb][i] = pok ? drow[(size_t)i * C0 + 32 * nb] : 0.f;
        }
    };
    if (r0 < r1) fetch(r0);
    int buf = 0;
    for (long long row = r0; row < r1; ++row, buf ^= 1) {
        unsigned d[NB][3][4];
        {
            unsigned mine[3][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) split3_bf16(av[2 * i], av[2 * i + 1], mine[0][i], mine[1][i], mine[2][i]);
#pragma unroll
            for (int pl_ = 0; pl_ < 3; ++pl_) a_lds[buf][wave][pl_][lane] = (u32x4){mine[pl_][0], mine[pl_][1], mine[pl_][2], mine[pl_][3]};
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) split3_bf16(dv[nb][2 * i], dv[nb][2 * i + 1], d[nb][0][i], d[nb][1][i], d[nb][2][i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) bsum[nb] += dv[nb][i];
        }
        __syncthreads();
        fetch(row + 1);                                                    // flies under the MFMAs below
        u32x4 a[4][3];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int pl_ = 0; pl_ < 3; ++pl_) a[mb][pl_] = a_lds[buf][mb][pl_][lane];
        constexpr int kTa[6] = {2, 0, 1, 1, 0, 0}, kTd[6] = {0, 2, 1, 0, 1, 0};   // b3 d1, b1 d3, b2 d2, b2 d1, b1 d2, b1 d1
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const bf16x8 fa = __builtin_bit_cast(bf16x8, a[mb][kTa[t]]);
                    const bf16x8 fd = __builtin_bit_cast(bf16x8, (u32x4){d[nb][kTd[t]][0], d[nb][kTd[t]][1], d[nb][kTd[t]][2], d[nb][kTd[t]][3]});
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fd, acc[mb][nb], 0, 0, 0);
                }
    }
    // ---- partial[b][k][c]: lane (channel j, g), register r of block mb = tap row 32 mb + 8 (r / 4) + 4 g + r % 4
    float *out = partial + (size_t)blockIdx.x * (kT1Taps + 1) * C0 + c0 + j;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kp = 32 * mb + 8 * (r >> 2) + 4 * g + (r & 3);
            const int ky = kp >> 4, t = kp & 15;
            if (ky < kKy && t < kT1Row) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) out[(size_t)(ky * kT1Row + t) * C0 + 32 * nb] = acc[mb][nb][r];
            }
        }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float s = bsum[nb] + __shfl_xor(bsum[nb], 32);
        if (g == 0) out[(size_t)kT1Taps * C0 + 32 * nb] = s;
    }
}
constexpr int kWgradBlocks = 512;              // at most two row ranges per CU (the workspace bound); MH_TOWER_WGRAD_BLOCKS picks fewer
}  // namespace t1

static int grid_for(long long total) { return (int)std::min<long long>((total + 255) / 256, 256 * 16); }

}  // namespace mh

using namespace mh;

extern "C" {

size_t mh_bn_ws_bytes(long long M, int C)
{
    if (M <= 0 || C <= 0) return 0;
    const long long nblk = (M + kRowsPerBlock - 1) / kRowsPerBlock;
    return align_up((size_t)nblk * 2 * C * sizeof(float), 256);
}

static int check_bn_args(long long M, int C)
{
    MH_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 8192 && (C <= kBnGroup || C % kBnGroup == 0));
    return MH_OK;
}

// statistics of x [M,C] (NHWC rows): mean, invstd (biased variance, eps); running stats updated when given
int mh_bn_stats(const float *x, long long M, int C, float eps, float momentum, float *mean, float *invstd,
                float *running_mean, float *running_var, void *workspace, size_t ws_bytes, void *stream)
{
    int rc = check_bn_args(M, C);
    if (rc) return rc;
    MH_REQUIRE(x && mean && invstd && workspace && ws_bytes >= mh_bn_ws_bytes(M, C));
    MH_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    hipStream_t st = as_stream(stream);
    const int nblk = (int)((M + kRowsPerBlock - 1) / kRowsPerBlock);
    float *partial = reinterpret_cast<float *>(workspace);
    const int Cb = std::min(C, kBnGroup), nrl = 256 / (Cb / 4);
    hipLaunchKernelGGL((bn_partial_kernel<false, false>), dim3(nblk, ceil_div(C, kBnGroup)), dim3(256), (size_t)nrl * 2 * Cb * sizeof(float), st, x,
                       (const float *)nullptr, (const unsigned char *)nullptr, (const float *)nullptr,
                       (const float *)nullptr, M, C, 0, 0, partial);
    rc = check_launch("bn_partial_kernel<fwd>");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(ceil_div(C, kFinCh)), dim3(kFinCh * kFinSl), 0, st, partial, nblk, C, M, eps, momentum,
                       x, mean, invstd, running_mean, running_var);
    return check_launch("bn_finalize_fwd_kernel");
}

int mh_bn_pool_fwd(const float *x, long long N, int H, int W, int C, const float *mean, const float *invstd,
                   const float *gamma, const float *beta, float *z, unsigned char *argmax, void *stream)
{
    int rc = check_bn_args(N * H * W, C);
    if (rc) return rc;
    MH_REQUIRE(x && mean && invstd && gamma && beta && z && argmax && H % 2 == 0 && W % 2 == 0);
    hipLaunchKernelGGL(bn_pool_fwd_kernel, dim3(grid_for(N * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0, as_stream(stream),
                       x, mean, invstd, gamma, beta, N, H, W, C, z, argmax);
    return check_launch("bn_pool_fwd_kernel");
}

int mh_bn_residual_nchw(const float *x, long long N, int P, int C, const float *mean, const float *invstd,
                        const float *gamma, const float *beta, const float *residual_nchw, float *out_nchw, void *stream)
{
    int rc = check_bn_args(N * P, C);
    if (rc) return rc;
    MH_REQUIRE(x && mean && invstd && gamma && beta && out_nchw && P > 0 && N <= 65535);
    hipLaunchKernelGGL(bn_residual_nchw_kernel, dim3(ceil_div(C, 32), ceil_div(P, 32), (unsigned)N), dim3(256), 0,
                       as_stream(stream), x, mean, invstd, gamma, beta, residual_nchw, P, C, out_nchw);
    return check_launch("bn_residual_nchw_kernel");
}

int mh_bn_apply_nhwc(const float *x, long long M, int C, const float *mean, const float *invstd, const float *gamma,
                     const float *beta, const float *residual, int relu, float *out, void *stream)
{
    int rc = check_bn_args(M, C);
    if (rc) return rc;
    MH_REQUIRE(x && mean && invstd && gamma && beta && out);
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0);
    hipLaunchKernelGGL(bn_apply_nhwc_kernel, dim3(grid_for(M * (C / 4))), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(x), mean, invstd, gamma, beta,
                       reinterpret_cast<const float4 *>(residual), M, C, relu, reinterpret_cast<float4 *>(out));
    return check_launch("bn_apply_nhwc_kernel");
}

int mh_nchw_to_nhwc_small(const float *in_nchw, long long N, int P, int C, float *out_nhwc, void *stream)
{
    MH_REQUIRE(in_nchw && out_nhwc && N > 0 && N <= 65535 && P > 0 && C > 0);
    hipLaunchKernelGGL(nchw_to_nhwc_small_kernel, dim3(ceil_div(C, 32), ceil_div(P, 32), (unsigned)N), dim3(256), 0,
                       as_stream(stream), in_nchw, P, C, out_nhwc);
    return check_launch("nchw_to_nhwc_small_kernel");
}

// BN backward (+ ReLU mask of the producer).  pooled != 0: g is the gradient of the POOLED output [N,H/2,W/2,C] and
// argmax the indices saved by mh_bn_pool_fwd; otherwise g is dense [N,H,W,C].  Outputs dx [N,H,W,C], dgamma, dbeta [C].
int mh_bn_bwd(const float *x, const float *g, const unsigned char *argmax, long long N, int H, int W, int C,
              const float *mean, const float *invstd, const float *gamma, int pooled, int relu_mask, float *dx,
              float *dgamma, float *dbeta, void *workspace, size_t ws_bytes, void *stream)
{
    const long long M = N * H * W;
    int rc = check_bn_args(M, C);
    if (rc) return rc;
    MH_REQUIRE(x && g && mean && invstd && gamma && dx && dgamma && dbeta && workspace);
    MH_REQUIRE(!pooled || (argmax && H % 2 == 0 && W % 2 == 0));
    const long long Mg = pooled ? N * (H / 2) * (W / 2) : M;
    MH_REQUIRE(ws_bytes >= mh_bn_ws_bytes(Mg, C));
    hipStream_t st = as_stream(stream);
    const int nblk = (int)((Mg + kRowsPerBlock - 1) / kRowsPerBlock);
    float *partial = reinterpret_cast<float *>(workspace);
    const int Cb = std::min(C, kBnGroup), nrl = 256 / (Cb / 4);
    const size_t lds = (size_t)nrl * 2 * Cb * sizeof(float);
    const dim3 pgrid(nblk, ceil_div(C, kBnGroup));
    if (pooled)
        hipLaunchKernelGGL((bn_partial_kernel<true, true>), pgrid, dim3(256), lds, st, x, g, argmax, mean, invstd, Mg,
                           C, H, W, partial);
    else
        hipLaunchKernelGGL((bn_partial_kernel<true, false>), pgrid, dim3(256), lds, st, x, g, argmax, mean, invstd, Mg,
                           C, H, W, partial);
    rc = check_launch("bn_partial_kernel<bwd>");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(ceil_div(C, kFinCh)), dim3(kFinCh * kFinSl), 0, st, partial, nblk, C, dgamma, dbeta);
    rc = check_launch("bn_finalize_bwd_kernel");
    if (rc) return rc;
    if (pooled)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<true>), dim3(grid_for(M * (C / 4))), dim3(256), 0, st, x, g, argmax, mean,
                           invstd, gamma, dgamma, dbeta, N, H, W, C, relu_mask, dx);
    else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<false>), dim3(grid_for(M * (C / 4))), dim3(256), 0, st, x, g, argmax, mean,
                           invstd, gamma, dgamma, dbeta, N, H, W, C, relu_mask, dx);
    return check_launch("bn_bwd_apply_kernel");
}

// ---- the tower's first convolution: matrix cores (t1::, round 6) unless MH_TOWER_CONV1=valu (the direct kernels) ----
static bool tower_conv1_on_mfma()
{
    const char *e = getenv("MH_TOWER_CONV1");          // read per call: the tests run both forms in one process
    return !(e && std::string(e) == "valu");
}
// pixel-tile walkers of the forward kernel: every block first builds the weight fragments of its 256 channels (56 strided loads
// and 28 splits per lane), so few long-lived blocks (two per CU) beat many short ones; MH_TOWER_FWD_BLOCKS for A/B
static long long tower_fwd_blocks()
{
    static const int v = [] { const char *e = getenv("MH_TOWER_FWD_BLOCKS"); const int x = e ? atoi(e) : 0; return x > 0 ? x : 512; }();
    return v;
}
static int check_t1(long long N, int S, int C0)
{
    MH_REQUIRE(N > 0 && N <= 0x7fffffffLL / 4 && S >= kT1K - 2 * kT1Pad && S <= 4096 && C0 > 0 && C0 % 256 == 0 && C0 <= 256 * 65535);
    MH_REQUIRE((S + 2 * kT1Pad - kT1K) % kT1Stride == 0);      // the last window ends on the last padded column (S = 27: 14 x 14 outputs)
    const int Ho = (S + 2 * kT1Pad - kT1K) / kT1Stride + 1;
    MH_REQUIRE(Ho % 2 == 0 && Ho <= kT1MaxW);                   // outputs are produced in pairs; a row of gradients lives in registers
    return MH_OK;
}
int mh_tower_conv1_out_size(int S) { return (S + 2 * kT1Pad - kT1K) / kT1Stride + 1; }
size_t mh_tower_conv1_padded_bytes(long long N, int S)
{
    if (N <= 0 || S <= 0) return 0;
    const size_t Sp = (size_t)S + 2 * kT1Pad;
    return align_up((size_t)N * Sp * Sp * kT1C * sizeof(float), 256);
}
size_t mh_tower_conv1_wgrad_ws_bytes(long long N, int C0)
{
    if (N <= 0 || C0 <= 0) return 0;
    // the direct kernel: one partial per 3 pairs; the matrix-core kernel: one per row range, at most t1::kWgradBlocks
    const size_t nblk = std::max<size_t>((size_t)ceil_div(N, (long long)kT1PairsPerBlock), (size_t)std::min<long long>(t1::kWgradBlocks, N * kT1MaxW));
    return align_up(nblk * (kT1Taps + 1) * (size_t)C0 * sizeof(float), 256);
}
int mh_tower_conv1_pad(const float *rects_nhwc, long long N, int S, float *padded, void *stream)
{
    int rc = check_t1(N, S, 256);
    if (rc) return rc;
    MH_REQUIRE(rects_nhwc && padded && ((reinterpret_cast<uintptr_t>(rects_nhwc) | reinterpret_cast<uintptr_t>(padded)) & 7) == 0);
    const long long Sp = S + 2 * kT1Pad;
    hipLaunchKernelGGL(tower_pad_kernel, dim3(grid_for(N * Sp * Sp)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2 *>(rects_nhwc), N, S, reinterpret_cast<float2 *>(padded));
    return check_launch("tower_pad_kernel");
}
int mh_tower_conv1_fwd(const float *padded, long long N, int S, const float *w_kc, const float *bias, int C0, float *y_nhwc,
                       void *stream)
{
    int rc = check_t1(N, S, C0);
    if (rc) return rc;
    MH_REQUIRE(padded && w_kc && y_nhwc);
    const int Ho = mh_tower_conv1_out_size(S);
    if (tower_conv1_on_mfma()) {
        MH_REQUIRE(((reinterpret_cast<uintptr_t>(padded) | reinterpret_cast<uintptr_t>(y_nhwc)) & 15) == 0);
        const long long ntiles = (N * Ho * Ho + 31) / 32;
        hipLaunchKernelGGL(t1::tower_conv1_mfma_fwd_kernel, dim3((unsigned)std::min<long long>(ntiles, tower_fwd_blocks()), (unsigned)(C0 / 256)), dim3(256),
                           (size_t)t1::kFwdLds, as_stream(stream), padded, N, S + 2 * kT1Pad, Ho, Ho, w_kc, bias, C0, y_nhwc);
        return check_launch("tower_conv1_mfma_fwd_kernel");
    }
    hipLaunchKernelGGL(tower_conv1_fwd_kernel, dim3((unsigned)N, (unsigned)(C0 / 256)), dim3(256), 0, as_stream(stream), padded,
                       S + 2 * kT1Pad, Ho, Ho, w_kc, bias, C0, y_nhwc);
    return check_launch("tower_conv1_fwd_kernel");
}
int mh_tower_conv1_wgrad(const float *padded, const float *dy_nhwc, long long N, int S, int C0, float *dw_kc, void *workspace,
                         size_t ws_bytes, void *stream)
{
    int rc = check_t1(N, S, C0);
    if (rc) return rc;
    MH_REQUIRE(padded && dy_nhwc && dw_kc && workspace && ws_bytes >= mh_tower_conv1_wgrad_ws_bytes(N, C0));
    hipStream_t st = as_stream(stream);
    const int Ho = mh_tower_conv1_out_size(S);
    float *partial = reinterpret_cast<float *>(workspace);
    if (tower_conv1_on_mfma()) {
        const long long rows = N * Ho;
        static const int want = [] { const char *e = getenv("MH_TOWER_WGRAD_BLOCKS"); const int v = e ? atoi(e) : 0; return (v >= 1 && v <= t1::kWgradBlocks) ? v : 256; }();
        const int per = (int)ceil_div(rows, (long long)want), nb = (int)ceil_div(rows, (long long)per);
        static const bool wide = [] { const char *e = getenv("MH_TOWER_WGRAD_NB"); return !(e && atoi(e) == 1); }();     // default 64 channels per wave, one block per CU (=1: 32 channels, two blocks per CU)
        if (wide)
            hipLaunchKernelGGL(t1::tower_conv1_mfma_wgrad_kernel<2>, dim3((unsigned)nb, (unsigned)(C0 / 256)), dim3(256), 0, st, padded, N,
                               S + 2 * kT1Pad, Ho, Ho, dy_nhwc, C0, per, partial);
        else
            hipLaunchKernelGGL(t1::tower_conv1_mfma_wgrad_kernel<1>, dim3((unsigned)nb, (unsigned)(C0 / 128)), dim3(256), 0, st, padded, N,
                               S + 2 * kT1Pad, Ho, Ho, dy_nhwc, C0, per, partial);
        rc = check_launch("tower_conv1_mfma_wgrad_kernel");
        if (rc) return rc;
        const int rows_c0 = (kT1Taps + 1) * C0;
        hipLaunchKernelGGL(tower_conv1_wgrad_reduce_kernel, dim3(ceil_div(rows_c0, 256)), dim3(256), 0, st, partial, nb, rows_c0, dw_kc);
        return check_launch("tower_conv1_wgrad_reduce_kernel");
    }
    const int nblk = (int)ceil_div(N, (long long)kT1PairsPerBlock);
    hipLaunchKernelGGL(tower_conv1_wgrad_kernel, dim3((unsigned)nblk, (unsigned)(C0 / 256)), dim3(256), 0, st, padded, S + 2 * kT1Pad, Ho,
                       Ho, dy_nhwc, N, C0, kT1PairsPerBlock, partial);
    rc = check_launch("tower_conv1_wgrad_kernel");
    if (rc) return rc;
    const int rows_c0 = (kT1Taps + 1) * C0;
    hipLaunchKernelGGL(tower_conv1_wgrad_reduce_kernel, dim3(ceil_div(rows_c0, 256)), dim3(256), 0, st, partial, nblk, rows_c0, dw_kc);
    return check_launch("tower_conv1_wgrad_reduce_kernel");
}

}  // extern "C"
