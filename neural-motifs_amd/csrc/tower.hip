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
    if (c < C)
        for (int b = sl; b < nblk; b += kFinSl) {
            s += (double)partial[(size_t)b * 2 * C + c];
            ss += (double)partial[(size_t)b * 2 * C + C + c];
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

// ---- the tower's first convolution, direct: see tower_conv1_fwd_kernel ----
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
    const size_t nblk = (size_t)ceil_div(N, (long long)kT1PairsPerBlock);
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
    const int nblk = (int)ceil_div(N, (long long)kT1PairsPerBlock);
    float *partial = reinterpret_cast<float *>(workspace);
    hipLaunchKernelGGL(tower_conv1_wgrad_kernel, dim3((unsigned)nblk, (unsigned)(C0 / 256)), dim3(256), 0, st, padded, S + 2 * kT1Pad, Ho,
                       Ho, dy_nhwc, N, C0, kT1PairsPerBlock, partial);
    rc = check_launch("tower_conv1_wgrad_kernel");
    if (rc) return rc;
    const int rows_c0 = (kT1Taps + 1) * C0;
    hipLaunchKernelGGL(tower_conv1_wgrad_reduce_kernel, dim3(ceil_div(rows_c0, 256)), dim3(256), 0, st, partial, nblk, rows_c0, dw_kc);
    return check_launch("tower_conv1_wgrad_reduce_kernel");
}

}  // extern "C"
