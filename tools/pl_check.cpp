// Device check of the round-3 plane GEMM (csrc/pl_gemm.hip) through the C ABI, torch-free:
//   * accuracy of mh_gemm_f32 (operand preparation + plane GEMM) in all four storage orientations, ragged and aligned
//     shapes, bias / ReLU / accumulate / forced split-K, against a float64 CPU product -- next to the round-2 in-loop-split
//     kernel (mh_gemm_f32_v2) on the same inputs;
//   * mh_make_planes + mh_gemm_planes on persistent images (the cached-weight path);
//   * speed on the step's big shapes (fc6 forward / input gradient / weight gradient, fc7, the 120-row object fc6, 4096^3):
//     v2 | v3 end to end (absmax + split + product) | v3 product only on ready images, per block-tile shape and split-K.
//   hipcc -O2 tools/pl_check.cpp -o tools/_bin/pl_check -ldl
//   tools/_bin/pl_check neural-motifs_amd/csrc/libmotifs_hip.so [--quick]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <vector>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef int (*gemm_fn)(int, int, int, int, int, const float *, int, const float *, int, float *, int, const float *, int, int, int, void *, size_t, void *);
typedef size_t (*ws_fn)(int, int, int, int);
typedef size_t (*pbytes_fn)(long long, long long);
typedef int (*mkplanes_fn)(const float *, int, long long, long long, long long, void *, void *);
typedef int (*gemmpl_fn)(int, int, int, const void *, const void *, float *, int, const float *, int, int, int, void *, size_t, void *);
typedef void (*shape_fn)(int);
typedef const char *(*err_fn)(void);

static gemm_fn g3, g2;
static ws_fn ws3, ws2, wspl;
static pbytes_fn pbytes;
static mkplanes_fn mkplanes;
static gemmpl_fn gemmpl;
static shape_fn set_shape;
static err_fn last_err;

struct Dev {
    void *p = nullptr;
    size_t n = 0;
    explicit Dev(size_t bytes) : n(bytes) { HIP_OK(hipMalloc(&p, std::max<size_t>(bytes, 256))); }
    ~Dev() { (void)hipFree(p); }
    float *f() { return reinterpret_cast<float *>(p); }
};

static void fill(std::vector<float> &v, std::mt19937 &rng, bool wide)
{
    std::normal_distribution<float> nrm(0.f, 1.f);
    std::uniform_real_distribution<float> expo(-8.f, 8.f);
    for (auto &x : v) x = wide ? nrm(rng) * std::exp2(expo(rng)) : nrm(rng);
}

// C = epi(op(A) op(B) + bias) (+ C0) in float64
static void ref_gemm(int tA, int tB, int M, int N, int K, const std::vector<float> &A, int lda, const std::vector<float> &B, int ldb,
                     const float *bias, int epi, const std::vector<float> *C0, int ldc, std::vector<double> &R)
{
    R.assign((size_t)M * N, 0.0);
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) {
            const double a = tA ? A[(size_t)k * lda + m] : A[(size_t)m * lda + k];
            double *r = &R[(size_t)m * N];
            if (tB) for (int n = 0; n < N; ++n) r[n] += a * (double)B[(size_t)n * ldb + k];
            else for (int n = 0; n < N; ++n) r[n] += a * (double)B[(size_t)k * ldb + n];
        }
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double v = R[(size_t)m * N + n] + (bias ? bias[n] : 0.0);
            if (epi == 1) v = std::max(v, 0.0);
            if (C0) v += (*C0)[(size_t)m * ldc + n];
            R[(size_t)m * N + n] = v;
        }
}

struct Err { double rms_rel, max_rel; };
static Err compare(const std::vector<float> &C, int ldc, const std::vector<double> &R, int M, int N)
{
    double ss = 0, se = 0, mx = 0;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            const double r = R[(size_t)m * N + n], e = (double)C[(size_t)m * ldc + n] - r;
            ss += r * r; se += e * e; mx = std::fmax(mx, std::fabs(e));
        }
    const double rms = std::sqrt(ss / ((double)M * N)) + 1e-300;
    return {std::sqrt(se / ((double)M * N)) / rms, mx / rms};
}

static int accuracy_case(const char *name, int tA, int tB, int M, int N, int K, int padA, int padB, int padC, bool wide, int use_bias,
                         int epi, int accum, int splitk, unsigned seed)
{
    std::mt19937 rng(seed);
    const int lda = (tA ? M : K) + padA, ldb = (tB ? K : N) + padB, ldc = N + padC;
    std::vector<float> A((size_t)(tA ? K : M) * lda), B((size_t)(tB ? N : K) * ldb), C0((size_t)M * ldc), bias(N);
    fill(A, rng, wide); fill(B, rng, wide); fill(C0, rng, false); fill(bias, rng, false);
    Dev dA(A.size() * 4), dB(B.size() * 4), dC(C0.size() * 4), dbias(N * 4);
    HIP_OK(hipMemcpy(dA.p, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dB.p, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dbias.p, bias.data(), N * 4, hipMemcpyHostToDevice));
    std::vector<double> R;
    ref_gemm(tA, tB, M, N, K, A, lda, B, ldb, use_bias ? bias.data() : nullptr, epi, accum ? &C0 : nullptr, ldc, R);
    Err e[2];
    int rc[2];
    for (int v = 0; v < 2; ++v) {
        gemm_fn g = v ? g2 : g3;
        const size_t wsb = (v ? ws2 : ws3)(M, N, K, splitk);
        Dev ws(wsb);
        HIP_OK(hipMemcpy(dC.p, C0.data(), C0.size() * 4, hipMemcpyHostToDevice));
        rc[v] = g(tA, tB, M, N, K, dA.f(), lda, dB.f(), ldb, dC.f(), ldc, use_bias ? dbias.f() : nullptr, epi, accum, splitk, ws.p, ws.n, nullptr);
        HIP_OK(hipDeviceSynchronize());
        std::vector<float> C(C0.size());
        HIP_OK(hipMemcpy(C.data(), dC.p, C.size() * 4, hipMemcpyDeviceToHost));
        e[v] = compare(C, ldc, R, M, N);
    }
    const bool ok = rc[0] == 0 && e[0].rms_rel < 2e-6 && e[0].max_rel < 4e-5;
    printf("{\"check\": \"accuracy\", \"case\": \"%s\", \"tA\": %d, \"tB\": %d, \"M\": %d, \"N\": %d, \"K\": %d, \"splitk\": %d, \"rc\": %d, "
           "\"v3_rms_rel\": %.3g, \"v3_max_rel\": %.3g, \"v2_rms_rel\": %.3g, \"v2_max_rel\": %.3g, \"ok\": %s}\n",
           name, tA, tB, M, N, K, splitk, rc[0], e[0].rms_rel, e[0].max_rel, e[1].rms_rel, e[1].max_rel, ok ? "true" : "false");
    if (rc[0]) printf("{\"error\": \"%s\"}\n", last_err());
    fflush(stdout);
    return ok ? 0 : 1;
}

// persistent images: A [M,K] K-contiguous, B given as [K,N] (k-major) -> both through mh_make_planes, every forced shape
static int image_case(int M, int N, int K, unsigned seed)
{
    std::mt19937 rng(seed);
    std::vector<float> A((size_t)M * K), B((size_t)K * N);
    fill(A, rng, true); fill(B, rng, true);
    Dev dA(A.size() * 4), dB(B.size() * 4), dC((size_t)M * N * 4), ia(pbytes(M, K)), ib(pbytes(N, K));
    HIP_OK(hipMemcpy(dA.p, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dB.p, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    int rc = mkplanes(dA.f(), 1, M, K, K, ia.p, nullptr);
    rc |= mkplanes(dB.f(), 0, N, K, N, ib.p, nullptr);
    std::vector<double> R;
    ref_gemm(0, 0, M, N, K, A, K, B, N, nullptr, 0, nullptr, N, R);
    int bad = 0;
    for (int shape = -1; shape <= 4; ++shape) {
        set_shape(shape);
        for (int sk : {0, 1, 3}) {
            Dev ws(wspl(M, N, K, sk));
            HIP_OK(hipMemset(dC.p, 0xff, (size_t)M * N * 4));
            const int r2 = gemmpl(M, N, K, ia.p, ib.p, dC.f(), N, nullptr, 0, 0, sk, ws.p, ws.n, nullptr);
            HIP_OK(hipDeviceSynchronize());
            std::vector<float> C((size_t)M * N);
            HIP_OK(hipMemcpy(C.data(), dC.p, C.size() * 4, hipMemcpyDeviceToHost));
            const Err e = compare(C, N, R, M, N);
            const bool ok = (rc | r2) == 0 && e.rms_rel < 2e-6 && e.max_rel < 4e-5;
            bad += !ok;
            printf("{\"check\": \"images\", \"M\": %d, \"N\": %d, \"K\": %d, \"shape\": %d, \"splitk\": %d, \"rc\": %d, \"rms_rel\": %.3g, \"max_rel\": %.3g, \"ok\": %s}\n",
                   M, N, K, shape, sk, rc | r2, e.rms_rel, e.max_rel, ok ? "true" : "false");
        }
    }
    set_shape(-1);
    fflush(stdout);
    return bad;
}

// mh_make_planes_both: A [M,K] -> its row image is the A operand; B given as [K,N] -> its COLUMN image is the B operand
typedef int (*mkboth_fn)(const float *, long long, long long, long long, void *, void *, void *);
static mkboth_fn mkboth;
static int both_case(int M, int N, int K, unsigned seed)
{
    std::mt19937 rng(seed);
    std::vector<float> A((size_t)M * K), B((size_t)K * N);
    fill(A, rng, true); fill(B, rng, true);
    Dev dA(A.size() * 4), dB(B.size() * 4), dC((size_t)M * N * 4), ia(pbytes(M, K)), iat(pbytes(K, M)), ib(pbytes(K, N)), ibt(pbytes(N, K));
    HIP_OK(hipMemcpy(dA.p, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dB.p, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    int rc = mkboth(dA.f(), M, K, K, ia.p, iat.p, nullptr);
    rc |= mkboth(dB.f(), K, N, N, ib.p, ibt.p, nullptr);
    std::vector<double> R;
    ref_gemm(0, 0, M, N, K, A, K, B, N, nullptr, 0, nullptr, N, R);
    Dev ws(wspl(M, N, K, 0));
    rc |= gemmpl(M, N, K, ia.p, ibt.p, dC.f(), N, nullptr, 0, 0, 0, ws.p, ws.n, nullptr);
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> C((size_t)M * N);
    HIP_OK(hipMemcpy(C.data(), dC.p, C.size() * 4, hipMemcpyDeviceToHost));
    const Err e = compare(C, N, R, M, N);
    // and the transposed product from the two OTHER images: C^T [N,M] = B^T A^T -> rows image of ... = (ib as [K rows,N]) no:
    // A^T image (rows = K, K = M) x B image (rows = K, K = N) is not a product; instead check C2 [K,K] = A^T A via iat x iat
    std::vector<double> R2((size_t)K * K, 0.0);
    for (int m = 0; m < M; ++m) for (int i = 0; i < K; ++i) { const double a = A[(size_t)m * K + i]; for (int j = 0; j < K; ++j) R2[(size_t)i * K + j] += a * (double)A[(size_t)m * K + j]; }
    Dev dC2((size_t)K * K * 4), ws2(wspl(K, K, M, 0));
    rc |= gemmpl(K, K, M, iat.p, iat.p, dC2.f(), K, nullptr, 0, 0, 0, ws2.p, ws2.n, nullptr);
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> C2((size_t)K * K);
    HIP_OK(hipMemcpy(C2.data(), dC2.p, C2.size() * 4, hipMemcpyDeviceToHost));
    const Err e2 = compare(C2, K, R2, K, K);
    const bool ok = rc == 0 && e.rms_rel < 2e-6 && e.max_rel < 4e-5 && e2.rms_rel < 2e-6 && e2.max_rel < 4e-5;
    printf("{\"check\": \"make_planes_both\", \"M\": %d, \"N\": %d, \"K\": %d, \"rc\": %d, \"rms_rel\": %.3g, \"max_rel\": %.3g, \"AtA_rms_rel\": %.3g, \"AtA_max_rel\": %.3g, \"ok\": %s}\n",
           M, N, K, rc, e.rms_rel, e.max_rel, e2.rms_rel, e2.max_rel, ok ? "true" : "false");
    fflush(stdout);
    return ok ? 0 : 1;
}

static float time_ms(int iters, const std::function<void()> &fn)
{
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    for (int i = 0; i < 12; ++i) fn();         // the clock needs a few ms of the load it is measured under (r06_c3: the first config after a light one read 10-15 % slow)
    HIP_OK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) fn();
    HIP_OK(hipEventRecord(e1, nullptr));
    HIP_OK(hipEventSynchronize(e1));
    float ms;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    HIP_OK(hipEventDestroy(e0)); HIP_OK(hipEventDestroy(e1));
    return ms / iters;
}

static void fill_dev(float *d, size_t n, unsigned seed)
{
    std::vector<float> h(std::min<size_t>(n, (size_t)1 << 24));
    std::mt19937 rng(seed);
    std::normal_distribution<float> nrm(0.f, 1.f);
    for (auto &v : h) v = nrm(rng);
    for (size_t o = 0; o < n; o += h.size()) HIP_OK(hipMemcpy(d + o, h.data(), std::min(h.size(), n - o) * 4, hipMemcpyHostToDevice));
}

static void speed_case(const char *name, int tA, int tB, int M, int N, int K, int iters)
{
    const size_t na = (size_t)M * K, nb = (size_t)N * K;
    Dev dA(na * 4), dB(nb * 4), dC((size_t)M * N * 4);
    fill_dev(dA.f(), na, 1); fill_dev(dB.f(), nb, 2);
    const int lda = tA ? M : K, ldb = tB ? K : N;
    const double flops = 2.0 * M * N * (double)K;
    {
        Dev ws(ws2(M, N, K, 0));
        const float ms = time_ms(iters, [&] { g2(tA, tB, M, N, K, dA.f(), lda, dB.f(), ldb, dC.f(), N, nullptr, 0, 0, 0, ws.p, ws.n, nullptr); });
        printf("{\"check\": \"speed\", \"case\": \"%s\", \"engine\": \"v2 in-loop split\", \"M\": %d, \"N\": %d, \"K\": %d, \"ms\": %.4f, \"tflops\": %.1f}\n", name, M, N, K, ms, flops / ms * 1e-9);
    }
    {
        Dev ws(ws3(M, N, K, 0));
        const float ms = time_ms(iters, [&] { g3(tA, tB, M, N, K, dA.f(), lda, dB.f(), ldb, dC.f(), N, nullptr, 0, 0, 0, ws.p, ws.n, nullptr); });
        printf("{\"check\": \"speed\", \"case\": \"%s\", \"engine\": \"v3 end to end (absmax + planes + product)\", \"M\": %d, \"N\": %d, \"K\": %d, \"ms\": %.4f, \"tflops\": %.1f}\n", name, M, N, K, ms, flops / ms * 1e-9);
    }
    Dev ia(pbytes(M, K)), ib(pbytes(N, K));
    {
        const float ms_a = time_ms(iters, [&] { mkplanes(dA.f(), !tA, M, K, lda, ia.p, nullptr); });
        const float ms_b = time_ms(iters, [&] { mkplanes(dB.f(), tB, N, K, ldb, ib.p, nullptr); });
        printf("{\"check\": \"speed\", \"case\": \"%s\", \"engine\": \"make_planes\", \"ms_A\": %.4f, \"GBps_A\": %.0f, \"ms_B\": %.4f, \"GBps_B\": %.0f}\n", name, ms_a,
               na * 12.0 / ms_a * 1e-6, ms_b, nb * 12.0 / ms_b * 1e-6);
    }
    if (!tB) {      // B stored [K][N]: both of its images from one read (what a trainable weight / an activation costs per step)
        Dev i1(pbytes(K, N)), i2(pbytes(N, K));
        const float ms = time_ms(iters, [&] { mkboth(dB.f(), K, N, ldb, i1.p, i2.p, nullptr); });
        printf("{\"check\": \"speed\", \"case\": \"%s\", \"engine\": \"make_planes_both(B)\", \"ms\": %.4f, \"GBps\": %.0f}\n", name, ms, nb * 16.0 / ms * 1e-6);
    }
    for (int shape = -1; shape <= 1; ++shape) {
        if (shape == 0 && M <= 128) continue;
        set_shape(shape);
        for (int sk : {0, 1, 2, 4, 8, 16}) {
            if (sk > 1 && K / 16 / sk < 8) continue;
            Dev ws(wspl(M, N, K, sk));
            const float ms = time_ms(iters, [&] { gemmpl(M, N, K, ia.p, ib.p, dC.f(), N, nullptr, 0, 0, sk, ws.p, ws.n, nullptr); });
            printf("{\"check\": \"speed\", \"case\": \"%s\", \"engine\": \"v3 product on images\", \"shape\": %d, \"splitk\": %d, \"ms\": %.4f, \"tflops\": %.1f}\n", name, shape, sk, ms,
                   flops / ms * 1e-9);
            fflush(stdout);
        }
    }
    set_shape(-1);
}

// ------------------------------------------------------------------------------------------------- plane conv (pl_conv.hip)
typedef size_t (*sz4_fn)(int, int, int, int);
typedef size_t (*sz5_fn)(int, int, int, int, int);
typedef size_t (*sz2_fn)(int, int);
typedef int (*actpl_fn)(const float *, const unsigned *, int, int, int, int, int, void *, void *);
typedef int (*plpack_fn)(const float *, int, int, int, void *, void *);
typedef int (*plconv_fn)(const void *, int, int, int, int, const void *, int, const float *, int, float *, unsigned *, void *, size_t, void *);
typedef int (*v2pack_fn)(const float *, int, int, int, float *, void *);
typedef int (*v2conv_fn)(const float *, int, int, int, int, const float *, int, const float *, int, float *, void *, size_t, void *);
static sz4_fn act_bytes;
static sz2_fn plpacked_bytes, v2packed_floats;
static sz5_fn plconv_ws, v2conv_ws;
static actpl_fn act_planes;
static plpack_fn plpack;
static plconv_fn plconv;
static v2pack_fn v2pack;
static v2conv_fn v2conv;
static shape_fn set_conv_shape, set_conv_splitk, set_conv_flags;

static unsigned fbits(float v) { unsigned u; memcpy(&u, &v, 4); return u & 0x7fffffffu; }

// B x H x W x Cin -> Cout, optional 2x2 pool in front; reference = float64 loops (small) or the round-2 kernel (big)
static int conv_case(const char *name, int B, int H, int W, int Cin, int Cout, int pool, bool cpu_ref, unsigned seed)
{
    std::mt19937 rng(seed);
    std::normal_distribution<float> nrm(0.f, 1.f);
    const int Hi = pool ? 2 * H : H, Wi = pool ? 2 * W : W;
    std::vector<float> x((size_t)B * Hi * Wi * Cin), w((size_t)Cout * Cin * 9), bias(Cout);
    for (size_t i = 0; i < x.size(); ++i) { const float v = nrm(rng) * (1.f + 3.f * ((i / ((size_t)Hi * Wi * Cin)) % 3)); x[i] = v > 0 ? v : 0.f; }
    for (auto &v : w) v = nrm(rng) * 0.05f;
    for (auto &v : bias) v = nrm(rng) * 0.1f;
    std::vector<unsigned> mb(B, 0);
    for (int b = 0; b < B; ++b)
        for (size_t i = 0; i < (size_t)Hi * Wi * Cin; ++i) mb[b] = std::max(mb[b], fbits(x[(size_t)b * Hi * Wi * Cin + i]));
    // pooled input on the host (what the conv sees)
    std::vector<float> xp((size_t)B * H * W * Cin);
    for (int b = 0; b < B; ++b) for (int y = 0; y < H; ++y) for (int xx = 0; xx < W; ++xx) for (int c = 0; c < Cin; ++c) {
        auto at = [&](int yy, int xq) { return x[(((size_t)b * Hi + yy) * Wi + xq) * Cin + c]; };
        xp[(((size_t)b * H + y) * W + xx) * Cin + c] = pool ? std::max(std::max(at(2 * y, 2 * xx), at(2 * y, 2 * xx + 1)), std::max(at(2 * y + 1, 2 * xx), at(2 * y + 1, 2 * xx + 1))) : at(y, xx);
    }
    Dev dx(x.size() * 4), dxp(xp.size() * 4), dw(w.size() * 4), db(Cout * 4), dmb(B * 4), dmbo(B * 4), dout((size_t)B * H * W * Cout * 4), dref((size_t)B * H * W * Cout * 4);
    HIP_OK(hipMemcpy(dx.p, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dxp.p, xp.data(), xp.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dw.p, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(db.p, bias.data(), Cout * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dmb.p, mb.data(), B * 4, hipMemcpyHostToDevice));
    Dev img(act_bytes(B, H, W, Cin)), pk(plpacked_bytes(Cout, Cin)), ws(plconv_ws(B, H, W, Cin, Cout));
    int rc = act_planes(dx.f(), (const unsigned *)dmb.p, B, Hi, Wi, Cin, pool, img.p, nullptr);
    rc |= plpack(dw.f(), Cout, Cin, 0, pk.p, nullptr);
    // reference output
    std::vector<double> R;
    std::vector<float> Rf((size_t)B * H * W * Cout);
    if (cpu_ref) {
        for (int b = 0; b < B; ++b) for (int y = 0; y < H; ++y) for (int xx = 0; xx < W; ++xx) for (int co = 0; co < Cout; ++co) {
            double a = bias[co];
            for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
                const int yy = y + ky - 1, xq = xx + kx - 1;
                if (yy < 0 || yy >= H || xq < 0 || xq >= W) continue;
                for (int ci = 0; ci < Cin; ++ci) a += (double)xp[(((size_t)b * H + yy) * W + xq) * Cin + ci] * (double)w[((size_t)co * Cin + ci) * 9 + ky * 3 + kx];
            }
            Rf[(((size_t)b * H + y) * W + xx) * Cout + co] = (float)std::max(a, 0.0);
        }
    } else {
        Dev pk2(v2packed_floats(Cout, Cin) * 4), ws2(v2conv_ws(B, H, W, Cin, Cout));
        rc |= v2pack(dw.f(), Cout, Cin, 0, pk2.f(), nullptr);
        rc |= v2conv(dxp.f(), B, H, W, Cin, pk2.f(), Cout, db.f(), 1, dref.f(), ws2.p, ws2.n, nullptr);
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(Rf.data(), dref.p, Rf.size() * 4, hipMemcpyDeviceToHost));
    }
    int bad = 0;
    for (int shape = -1; shape <= 4; ++shape) {
        if (shape == 3) continue;
        set_conv_shape(shape);
        Dev ws3(plconv_ws(B, H, W, Cin, Cout));
        HIP_OK(hipMemset(dmbo.p, 0, B * 4));
        HIP_OK(hipMemset(dout.p, 0xff, dout.n));
        const int r2 = plconv(img.p, B, H, W, Cin, pk.p, Cout, db.f(), 1, dout.f(), (unsigned *)dmbo.p, ws3.p, ws3.n, nullptr);
        HIP_OK(hipDeviceSynchronize());
        std::vector<float> O(Rf.size());
        std::vector<unsigned> mbo(B);
        HIP_OK(hipMemcpy(O.data(), dout.p, O.size() * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(mbo.data(), dmbo.p, B * 4, hipMemcpyDeviceToHost));
        double ss = 0, se = 0, mx = 0;
        int mb_bad = 0;
        for (int b = 0; b < B; ++b) {
            unsigned m = 0;
            for (size_t i = 0; i < (size_t)H * W * Cout; ++i) {
                const size_t k = (size_t)b * H * W * Cout + i;
                const double e = (double)O[k] - (double)Rf[k];
                ss += (double)Rf[k] * Rf[k]; se += e * e; mx = std::fmax(mx, std::fabs(e));
                m = std::max(m, fbits(O[k]));
            }
            mb_bad += (m != mbo[b]);
        }
        const double rms = std::sqrt(ss / O.size()) + 1e-300;
        const bool ok = (rc | r2) == 0 && std::sqrt(se / O.size()) / rms < 2e-6 && mx / rms < 4e-5 && mb_bad == 0;
        bad += !ok;
        printf("{\"check\": \"conv\", \"case\": \"%s\", \"B\": %d, \"H\": %d, \"W\": %d, \"Cin\": %d, \"Cout\": %d, \"pool\": %d, \"ref\": \"%s\", \"shape\": %d, "
               "\"rc\": %d, \"rms_rel\": %.3g, \"max_rel\": %.3g, \"maxbits_wrong\": %d, \"ok\": %s}\n", name, B, H, W, Cin, Cout, pool,
               cpu_ref ? "float64" : "round-2 kernel", shape, rc | r2, std::sqrt(se / O.size()) / rms, mx / rms, mb_bad, ok ? "true" : "false");
        if (rc | r2) printf("{\"error\": \"%s\"}\n", last_err());
        fflush(stdout);
    }
    set_conv_shape(-1);
    return bad;
}

typedef int (*plconv_img_fn)(const void *, const unsigned *, int, int, int, int, const void *, int, const float *, int, void *, unsigned *, void *, size_t, void *);
typedef int (*stem_img_fn)(const float *, int, int, int, int, const float *, int, const float *, int, void *, unsigned *, void *);
typedef int (*stem_max_fn)(const float *, int, int, int, int, const float *, int, const float *, int, float *, unsigned *, void *);
static plconv_img_fn plconv_img, plconv_pool;
static stem_img_fn stem_img;
static stem_max_fn stem_max;

// image-output epilogue: [stem ->] conv A (image out) -> conv B (fp32 out)  against  the same chain through fp32 tensors and
// the converter; both are f16x3 evaluations of the same math, so they agree to a few 1e-6 of the result's rms
static int chain_case(const char *name, int B, int H, int W, int C0, int C1, int C2, bool with_stem, unsigned seed)
{
    std::mt19937 rng(seed);
    std::normal_distribution<float> nrm(0.f, 1.f);
    const size_t M = (size_t)B * H * W;
    std::vector<float> x(with_stem ? (size_t)B * 3 * H * W : M * C0), ws_((size_t)C0 * 27), bs(C0), wa((size_t)C1 * C0 * 9), ba(C1), wb((size_t)C2 * C1 * 9), bb(C2);
    for (size_t i = 0; i < x.size(); ++i) { const float v = nrm(rng) * (1.f + 2.f * ((i / (x.size() / B)) % 3)); x[i] = with_stem ? v : (v > 0 ? v : 0.f); }
    for (auto &v : ws_) v = nrm(rng) * 0.2f;
    for (auto &v : bs) v = nrm(rng) * 0.1f;
    for (auto &v : wa) v = nrm(rng) * 0.05f;
    for (auto &v : ba) v = nrm(rng) * 0.1f;
    for (auto &v : wb) v = nrm(rng) * 0.05f;
    for (auto &v : bb) v = nrm(rng) * 0.1f;
    Dev dx(x.size() * 4), dws(ws_.size() * 4), dbs(C0 * 4), dwa(wa.size() * 4), dba(C1 * 4), dwb(wb.size() * 4), dbb(C2 * 4);
    HIP_OK(hipMemcpy(dx.p, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dws.p, ws_.data(), ws_.size() * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(dbs.p, bs.data(), C0 * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dwa.p, wa.data(), wa.size() * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(dba.p, ba.data(), C1 * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dwb.p, wb.data(), wb.size() * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(dbb.p, bb.data(), C2 * 4, hipMemcpyHostToDevice));
    Dev pka(plpacked_bytes(C1, C0)), pkb(plpacked_bytes(C2, C1));
    int rc = plpack(dwa.f(), C1, C0, 0, pka.p, nullptr) | plpack(dwb.f(), C2, C1, 0, pkb.p, nullptr);
    Dev mb(3 * B * 4), mb2(3 * B * 4);
    HIP_OK(hipMemset(mb.p, 0, mb.n)); HIP_OK(hipMemset(mb2.p, 0, mb2.n));
    unsigned *m0 = (unsigned *)mb.p, *m1 = m0 + B, *m2 = m1 + B, *n0_ = (unsigned *)mb2.p, *n1 = n0_ + B, *n2 = n1 + B;
    Dev y0(M * C0 * 4), y1(M * C1 * 4), outA(M * C2 * 4), outB(M * C2 * 4);
    Dev img0(act_bytes(B, H, W, C0)), img1(act_bytes(B, H, W, C1)), jmg0(act_bytes(B, H, W, C0)), jmg1(act_bytes(B, H, W, C1));
    Dev wsa(plconv_ws(B, H, W, C0, C1)), wsb(plconv_ws(B, H, W, C1, C2));
    // path 1: through fp32 tensors + converters
    if (with_stem) {
        rc |= stem_max(dx.f(), B, 3, H, W, dws.f(), C0, dbs.f(), 1, y0.f(), m0, nullptr);
        rc |= act_planes(y0.f(), m0, B, H, W, C0, 0, img0.p, nullptr);
    } else {
        std::vector<unsigned> hb(B, 0);
        for (int b = 0; b < B; ++b) for (size_t i = 0; i < (size_t)H * W * C0; ++i) hb[b] = std::max(hb[b], fbits(x[(size_t)b * H * W * C0 + i]));
        HIP_OK(hipMemcpy(m0, hb.data(), B * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(n0_, hb.data(), B * 4, hipMemcpyHostToDevice));
        rc |= act_planes(dx.f(), m0, B, H, W, C0, 0, img0.p, nullptr);
    }
    rc |= plconv(img0.p, B, H, W, C0, pka.p, C1, dba.f(), 1, y1.f(), m1, wsa.p, wsa.n, nullptr);
    rc |= act_planes(y1.f(), m1, B, H, W, C1, 0, img1.p, nullptr);
    rc |= plconv(img1.p, B, H, W, C1, pkb.p, C2, dbb.f(), 1, outA.f(), m2, wsb.p, wsb.n, nullptr);
    // path 2: image outputs
    const void *in2 = img0.p;
    if (with_stem) { rc |= stem_img(dx.f(), B, 3, H, W, dws.f(), C0, dbs.f(), 1, jmg0.p, n0_, nullptr); in2 = jmg0.p; }
    int bad = 0;
    for (int shape = -1; shape <= 4; ++shape) {
        if (shape == 3) continue;
        set_conv_shape(shape);
        HIP_OK(hipMemset(n1, 0, B * 4)); HIP_OK(hipMemset(n2, 0, B * 4));
        Dev wsa2(plconv_ws(B, H, W, C0, C1)), wsb2(plconv_ws(B, H, W, C1, C2));
        int r2 = plconv_img(in2, n0_, B, H, W, C0, pka.p, C1, dba.f(), 1, jmg1.p, n1, wsa2.p, wsa2.n, nullptr);
        r2 |= plconv(jmg1.p, B, H, W, C1, pkb.p, C2, dbb.f(), 1, outB.f(), n2, wsb2.p, wsb2.n, nullptr);
        HIP_OK(hipDeviceSynchronize());
        std::vector<float> A(M * C2), Bv(M * C2);
        std::vector<unsigned> ma(3 * B), mbv(3 * B);
        HIP_OK(hipMemcpy(A.data(), outA.p, A.size() * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(Bv.data(), outB.p, Bv.size() * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(ma.data(), mb.p, ma.size() * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(mbv.data(), mb2.p, mbv.size() * 4, hipMemcpyDeviceToHost));
        double ss = 0, se = 0, mx = 0;
        for (size_t i = 0; i < A.size(); ++i) { const double e = (double)A[i] - Bv[i]; ss += (double)A[i] * A[i]; se += e * e; mx = std::fmax(mx, std::fabs(e)); }
        const double rms = std::sqrt(ss / A.size()) + 1e-300;
        int mdiff = 0;      // true maxima of the middle layer agree up to the two evaluations' rounding
        for (int b = 0; b < B; ++b) { float fa, fb; memcpy(&fa, &ma[B + b], 4); memcpy(&fb, &mbv[B + b], 4); mdiff += std::fabs(fa - fb) > 1e-5 * std::fabs(fa); }
        const bool ok = (rc | r2) == 0 && std::sqrt(se / A.size()) / rms < 3e-6 && mx / rms < 6e-5 && mdiff == 0;
        bad += !ok;
        printf("{\"check\": \"conv image output\", \"case\": \"%s\", \"B\": %d, \"H\": %d, \"W\": %d, \"C\": [%d, %d, %d], \"stem\": %d, \"shape\": %d, \"rc\": %d, "
               "\"rms_rel\": %.3g, \"max_rel\": %.3g, \"mid_maxima_differ\": %d, \"ok\": %s}\n", name, B, H, W, C0, C1, C2, with_stem ? 1 : 0, shape, rc | r2,
               std::sqrt(se / A.size()) / rms, mx / rms, mdiff, ok ? "true" : "false");
        if (rc | r2) printf("{\"error\": \"%s\"}\n", last_err());
        fflush(stdout);
    }
    set_conv_shape(-1);
    return bad;
}

// pooled-image epilogue (round 6): the image mh_plconv3x3_pool_to_image writes must be, cell for cell, the 2x2 maximum of the image
// mh_plconv3x3_to_image writes for the same launch shape (same K order, same per-image scale from the same bound, the split is a
// monotone function of the value): bitwise.  True maxima and scale words must be equal too.
static int pool_case(const char *name, int B, int H, int W, int Cin, int Cout, int splitk, unsigned seed)
{
    if (!plconv_pool) { printf("{\"check\": \"conv pooled image\", \"error\": \"no mh_plconv3x3_pool_to_image\"}\n"); return 1; }
    std::mt19937 rng(seed);
    std::normal_distribution<float> nrm(0.f, 1.f);
    const size_t M = (size_t)B * H * W, Mo = M / 4;
    std::vector<float> x(M * Cin), w((size_t)Cout * Cin * 9), bias(Cout);
    for (size_t i = 0; i < x.size(); ++i) { const float v = nrm(rng) * (1.f + 3.f * ((i / ((size_t)H * W * Cin)) % 3)); x[i] = v > 0 ? v : 0.f; }
    for (auto &v : w) v = nrm(rng) * 0.05f;
    for (auto &v : bias) v = nrm(rng) * 0.1f;
    std::vector<unsigned> mb(B, 0);
    for (int b = 0; b < B; ++b)
        for (size_t i = 0; i < (size_t)H * W * Cin; ++i) mb[b] = std::max(mb[b], fbits(x[(size_t)b * H * W * Cin + i]));
    Dev dx(x.size() * 4), dw(w.size() * 4), db(Cout * 4), dmb(B * 4), dm1(B * 4), dm2(B * 4);
    HIP_OK(hipMemcpy(dx.p, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dw.p, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(db.p, bias.data(), Cout * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dmb.p, mb.data(), B * 4, hipMemcpyHostToDevice));
    Dev img(act_bytes(B, H, W, Cin)), pk(plpacked_bytes(Cout, Cin)), i1(act_bytes(B, H, W, Cout)), i2(act_bytes(B, H / 2, W / 2, Cout));
    int rc = act_planes(dx.f(), (const unsigned *)dmb.p, B, H, W, Cin, 0, img.p, nullptr);
    rc |= plpack(dw.f(), Cout, Cin, 0, pk.p, nullptr);
    int bad = 0;
    for (int shape : {Cout <= 64 ? 5 : 4, Cout <= 64 ? 6 : -1}) {
        set_conv_shape(shape); set_conv_splitk(splitk);
        Dev ws3(plconv_ws(B, H, W, Cin, Cout));
        HIP_OK(hipMemset(dm1.p, 0, B * 4)); HIP_OK(hipMemset(dm2.p, 0, B * 4));
        HIP_OK(hipMemset(i1.p, 0xee, i1.n)); HIP_OK(hipMemset(i2.p, 0xee, i2.n));
        int r2 = plconv_img(img.p, (const unsigned *)dmb.p, B, H, W, Cin, pk.p, Cout, db.f(), 1, i1.p, (unsigned *)dm1.p, ws3.p, ws3.n, nullptr);
        r2 |= plconv_pool(img.p, (const unsigned *)dmb.p, B, H, W, Cin, pk.p, Cout, db.f(), 1, i2.p, (unsigned *)dm2.p, ws3.p, ws3.n, nullptr);
        HIP_OK(hipDeviceSynchronize());
        std::vector<unsigned char> h1(i1.n), h2(i2.n);
        std::vector<unsigned> m1(B), m2(B);
        HIP_OK(hipMemcpy(h1.data(), i1.p, i1.n, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(h2.data(), i2.p, i2.n, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(m1.data(), dm1.p, B * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(m2.data(), dm2.p, B * 4, hipMemcpyDeviceToHost));
        auto half = [](const unsigned char *p) { unsigned short u; memcpy(&u, p, 2); const int e = (u >> 10) & 31, m = u & 1023; float v = e ? std::ldexp(1.f + m / 1024.f, e - 15) : std::ldexp(m / 1024.f, -14); return (u & 0x8000) ? -v : v; };
        const bool tails = splitk == 0 && (long long)((M + 255) / 256) * ((Cout + (Cout <= 64 ? 63 : 127)) / (Cout <= 64 ? 64 : 128)) > 512;
        auto cell = [&](const std::vector<unsigned char> &im, size_t Mtot, size_t pix, int c) { return im.data() + ((size_t)(c / 16) * Mtot + pix) * 64 + 2 * (c % 16); };
        long long wrong = 0;
        for (int b = 0; b < B; ++b) for (int yo = 0; yo < H / 2; ++yo) for (int xo = 0; xo < W / 2; ++xo) for (int c = 0; c < Cout; ++c) {
            double best = -1e300; const unsigned char *bp = nullptr;
            for (int dy = 0; dy < 2; ++dy) for (int dxx = 0; dxx < 2; ++dxx) {
                const unsigned char *q = cell(h1, M, ((size_t)b * H + 2 * yo + dy) * W + 2 * xo + dxx, c);
                const double v = (double)half(q) + (double)half(q + 32);
                if (v > best) { best = v; bp = q; }
            }
            const unsigned char *q2 = cell(h2, Mo, ((size_t)b * (H / 2) + yo) * (W / 2) + xo, c);
            const bool differ = (memcmp(bp, q2, 2) != 0) || (memcmp(bp + 32, q2 + 32, 2) != 0);
            // a launch with sliced TAIL tiles: the two row orders put different pixels into the tail, whose sums are added up in
            // another order -- those cells may differ in the last bits of the two-term value (<= 2^-20 of the scaled range)
            if (differ && tails) { const double v2 = (double)half(q2) + (double)half(q2 + 32); wrong += std::fabs(v2 - best) > 32768.0 * 1e-6; }
            else wrong += differ;
        }
        const size_t c1 = ((size_t)(Cout / 16) * M * 64 + 255) / 256 * 256, c2 = ((size_t)(Cout / 16) * Mo * 64 + 255) / 256 * 256;
        const int scale_diff = memcmp(h1.data() + c1, h2.data() + c2, (size_t)B * 4) != 0;
        int mdiff = 0;
        for (int b = 0; b < B; ++b) mdiff += m1[b] != m2[b];
        const bool ok = (rc | r2) == 0 && wrong == 0 && !scale_diff && mdiff == 0;
        bad += !ok;
        printf("{\"check\": \"conv pooled image\", \"case\": \"%s\", \"B\": %d, \"H\": %d, \"W\": %d, \"Cin\": %d, \"Cout\": %d, \"shape\": %d, \"splitk\": %d, \"rc\": %d, "
               "\"cells_wrong\": %lld, \"scales_differ\": %d, \"maxima_differ\": %d, \"ok\": %s}\n", name, B, H, W, Cin, Cout, shape, splitk, rc | r2, wrong, scale_diff, mdiff,
               ok ? "true" : "false");
        if (rc | r2) printf("{\"error\": \"%s\"}\n", last_err());
        fflush(stdout);
        if (Cout > 64) break;
    }
    set_conv_shape(-1); set_conv_splitk(0);
    return bad;
}

static bool g_sweep_quick = false, g_sweep_ring_only = false;
// shape x split-K sweep of one layer (the planner's choice is the row with splitk 0)
static void conv_sweep(const char *name, int B, int H, int W, int Cin, int Cout, int iters)
{
    const size_t nx = (size_t)B * H * W * Cin;
    Dev dx(nx * 4), dw((size_t)Cout * Cin * 9 * 4), db(Cout * 4), dmb(B * 4), dmbo(B * 4), dout((size_t)B * H * W * Cout * 4);
    fill_dev(dx.f(), nx, 3); fill_dev(dw.f(), (size_t)Cout * Cin * 9, 5); fill_dev(db.f(), Cout, 6);
    std::vector<unsigned> mb(B, fbits(6.0f));
    HIP_OK(hipMemcpy(dmb.p, mb.data(), B * 4, hipMemcpyHostToDevice));
    const double flops = 2.0 * 9 * Cin * (double)Cout * B * H * W;
    Dev img(act_bytes(B, H, W, Cin)), pk(plpacked_bytes(Cout, Cin)), oimg(act_bytes(B, H, W, Cout));
    plpack(dw.f(), Cout, Cin, 0, pk.p, nullptr);
    act_planes(dx.f(), (const unsigned *)dmb.p, B, H, W, Cin, 0, img.p, nullptr);
    for (int shape = 0; shape <= 6; ++shape) {
        if (shape == 3 || (shape == 2 && Cout > 64)) continue;
        if (Cout <= 64 ? (shape != 2 && shape < 5) : shape > 4) continue;
        if (g_sweep_ring_only && shape < 4 && Cout > 64) continue;
        for (int sk : {0, 1, 2, 3, 4, 6}) {
            if (g_sweep_quick && sk != 0 && sk != 1 && !(H <= 74 && (sk == 2 || sk == 3 || sk == 4))) continue;
            set_conv_shape(shape); set_conv_splitk(sk);
            Dev ws3(plconv_ws(B, H, W, Cin, Cout));
            const float ms = time_ms(iters, [&] { plconv(img.p, B, H, W, Cin, pk.p, Cout, db.f(), 1, dout.f(), (unsigned *)dmbo.p, ws3.p, ws3.n, nullptr); });
            const float msi = time_ms(iters, [&] { plconv_img(img.p, (const unsigned *)dmb.p, B, H, W, Cin, pk.p, Cout, db.f(), 1, oimg.p, (unsigned *)dmbo.p, ws3.p, ws3.n, nullptr); });
            float msp = 0.f;      // output through the 2x2 pool as the next layer's image (ring shapes, even maps)
            if (plconv_pool && shape >= 4 && H % 2 == 0 && W % 2 == 0)
                msp = time_ms(iters, [&] { plconv_pool(img.p, (const unsigned *)dmb.p, B, H, W, Cin, pk.p, Cout, db.f(), 1, oimg.p, (unsigned *)dmbo.p, ws3.p, ws3.n, nullptr); });
            printf("{\"check\": \"conv sweep\", \"case\": \"%s\", \"shape\": %d, \"splitk\": %d, \"ms\": %.4f, \"tflops\": %.1f, \"ms_image_out\": %.4f, \"tflops_image_out\": %.1f, "
                   "\"ms_pooled_image_out\": %.4f, \"tflops_pooled_image_out\": %.1f}\n",
                   name, shape, sk, ms, flops / ms * 1e-9, msi, flops / msi * 1e-9, msp, msp > 0 ? flops / msp * 1e-9 : 0.0);
            fflush(stdout);
        }
    }
    set_conv_shape(-1); set_conv_splitk(0);
}

static void conv_speed(const char *name, int B, int H, int W, int Cin, int Cout, int pool_in_front, int iters)
{
    const int Hi = pool_in_front ? 2 * H : H, Wi = pool_in_front ? 2 * W : W;
    const size_t nx = (size_t)B * Hi * Wi * Cin, nxp = (size_t)B * H * W * Cin;
    Dev dx(nx * 4), dxp(nxp * 4), dw((size_t)Cout * Cin * 9 * 4), db(Cout * 4), dmb(B * 4), dmbo(B * 4), dout((size_t)B * H * W * Cout * 4);
    fill_dev(dx.f(), nx, 3); fill_dev(dxp.f(), nxp, 4); fill_dev(dw.f(), (size_t)Cout * Cin * 9, 5); fill_dev(db.f(), Cout, 6);
    std::vector<unsigned> mb(B, fbits(6.0f));
    HIP_OK(hipMemcpy(dmb.p, mb.data(), B * 4, hipMemcpyHostToDevice));
    const double flops = 2.0 * 9 * Cin * (double)Cout * B * H * W;
    {
        Dev pk2(v2packed_floats(Cout, Cin) * 4), ws2(v2conv_ws(B, H, W, Cin, Cout));
        v2pack(dw.f(), Cout, Cin, 0, pk2.f(), nullptr);
        const float ms = time_ms(iters, [&] { v2conv(dxp.f(), B, H, W, Cin, pk2.f(), Cout, db.f(), 1, dout.f(), ws2.p, ws2.n, nullptr); });
        printf("{\"check\": \"conv speed\", \"case\": \"%s\", \"engine\": \"v2 (split in the loop, incl. exponent passes)\", \"ms\": %.4f, \"tflops\": %.1f}\n", name, ms, flops / ms * 1e-9);
    }
    Dev img(act_bytes(B, H, W, Cin)), pk(plpacked_bytes(Cout, Cin));
    plpack(dw.f(), Cout, Cin, 0, pk.p, nullptr);
    {
        const float ms = time_ms(iters, [&] { act_planes(dx.f(), (const unsigned *)dmb.p, B, Hi, Wi, Cin, pool_in_front, img.p, nullptr); });
        printf("{\"check\": \"conv speed\", \"case\": \"%s\", \"engine\": \"act_planes%s\", \"ms\": %.4f, \"GBps\": %.0f}\n", name, pool_in_front ? " (2x2 pool fused)" : "", ms,
               (nx + nxp) * 4.0 / ms * 1e-6);
    }
    for (int shape = -1; shape <= 2; ++shape) {
        if (shape == 2 && Cout > 64) continue;
        set_conv_shape(shape);
        Dev ws3(plconv_ws(B, H, W, Cin, Cout));
        const float ms = time_ms(iters, [&] { plconv(img.p, B, H, W, Cin, pk.p, Cout, db.f(), 1, dout.f(), (unsigned *)dmbo.p, ws3.p, ws3.n, nullptr); });
        printf("{\"check\": \"conv speed\", \"case\": \"%s\", \"engine\": \"v3 plane conv\", \"shape\": %d, \"ms\": %.4f, \"tflops\": %.1f}\n", name, shape, ms, flops / ms * 1e-9);
        fflush(stdout);
    }
    set_conv_shape(-1);
}


// --ring: the round-4 ring kernels (shapes 3..5) against the round-3 loop (shapes 0, 1) on ready images: accuracy on ragged
// shapes, then speed on the step's big products
static void ring_speed(const char *name, int M, int N, int K, int iters)
{
    Dev dA((size_t)M * K * 4), dB((size_t)N * K * 4), dC((size_t)M * N * 4), ia(pbytes(M, K)), ib(pbytes(N, K));
    fill_dev(dA.f(), (size_t)M * K, 1); fill_dev(dB.f(), (size_t)N * K, 2);
    mkplanes(dA.f(), 1, M, K, K, ia.p, nullptr);
    mkplanes(dB.f(), 1, N, K, K, ib.p, nullptr);
    const double flops = 2.0 * M * N * (double)K;
    for (int shape : {1, 0, 3, 4}) {
        if (shape == 0 && M <= 128) continue;
        set_shape(shape);
        for (int sk : {1, 2, 3, 4, 8}) {
            if (sk > 1 && K / 16 / sk < 8) continue;
            Dev ws(wspl(M, N, K, sk));
            const float ms = time_ms(iters, [&] { gemmpl(M, N, K, ia.p, ib.p, dC.f(), N, nullptr, 0, 0, sk, ws.p, ws.n, nullptr); });
            printf("{\"check\": \"ring speed\", \"case\": \"%s\", \"M\": %d, \"N\": %d, \"K\": %d, \"shape\": %d, \"splitk\": %d, \"ms\": %.4f, \"tflops\": %.1f}\n", name, M, N, K,
                   shape, sk, ms, flops / ms * 1e-9);
            fflush(stdout);
        }
    }
    set_shape(-1);
}

// --order-ab: the planner's own launch of each big product of the step under the round-3 tile numbering (order 0) and the
// XCD-banded order (1), alternating, on ready images
static shape_fn set_order;
static void order_ab(const char *name, int M, int N, int K, int iters)
{
    Dev dA((size_t)M * K * 4), dB((size_t)N * K * 4), dC((size_t)M * N * 4), ia(pbytes(M, K)), ib(pbytes(N, K));
    fill_dev(dA.f(), (size_t)M * K, 1); fill_dev(dB.f(), (size_t)N * K, 2);
    mkplanes(dA.f(), 1, M, K, K, ia.p, nullptr);
    mkplanes(dB.f(), 1, N, K, K, ib.p, nullptr);
    const double flops = 2.0 * M * N * (double)K;
    for (int rep = 0; rep < 2; ++rep)
        for (int order : {0, 1}) {
            set_order(order);
            for (int shape : {-1, 3, 4}) {
                if (rep == 1 && shape >= 0) continue;
                set_shape(shape);
                for (int sk : {0, 1, 2, 4, 8}) {
                    if (sk > 1 && K / 16 / sk < 8) continue;
                    if (rep == 1 && sk != 0) continue;
                    Dev ws(wspl(M, N, K, sk));
                    const float ms = time_ms(iters, [&] { gemmpl(M, N, K, ia.p, ib.p, dC.f(), N, nullptr, 0, 0, sk, ws.p, ws.n, nullptr); });
                    printf("{\"check\": \"order ab\", \"case\": \"%s\", \"M\": %d, \"N\": %d, \"K\": %d, \"order\": %d, \"shape\": %d, \"splitk\": %d, \"ms\": %.4f, \"tflops\": %.1f}\n",
                           name, M, N, K, order, shape, sk, ms, flops / ms * 1e-9);
                    fflush(stdout);
                }
            }
        }
    set_order(1); set_shape(-1);
}

int main(int argc, char **argv)
{
    if (argc < 2) { printf("usage: pl_check <libmotifs_hip.so> [--quick] [--speed-only]\n"); return 1; }
    const bool quick = argc > 2 && !strcmp(argv[2], "--quick");
    const bool speed_only = argc > 2 && !strcmp(argv[2], "--speed-only");
    const bool pmc = argc > 2 && !strcmp(argv[2], "--pmc");
    void *h = dlopen(argv[1], RTLD_NOW);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 2; }
    g3 = (gemm_fn)dlsym(h, "mh_gemm_f32"); g2 = (gemm_fn)dlsym(h, "mh_gemm_f32_v2");
    ws3 = (ws_fn)dlsym(h, "mh_gemm_ws_bytes"); ws2 = (ws_fn)dlsym(h, "mh_gemm_ws_bytes_v2"); wspl = (ws_fn)dlsym(h, "mh_gemm_planes_ws_bytes");
    pbytes = (pbytes_fn)dlsym(h, "mh_planes_bytes"); mkplanes = (mkplanes_fn)dlsym(h, "mh_make_planes");
    gemmpl = (gemmpl_fn)dlsym(h, "mh_gemm_planes"); set_shape = (shape_fn)dlsym(h, "mh_debug_pl_shape"); last_err = (err_fn)dlsym(h, "mh_last_error");
    mkboth = (mkboth_fn)dlsym(h, "mh_make_planes_both");
    if (!g3 || !g2 || !ws3 || !ws2 || !wspl || !pbytes || !mkplanes || !mkboth || !gemmpl || !set_shape || !last_err) { printf("missing symbol\n"); return 2; }
    act_bytes = (sz4_fn)dlsym(h, "mh_act_planes_bytes"); plpacked_bytes = (sz2_fn)dlsym(h, "mh_plconv_packed_bytes");
    v2packed_floats = (sz2_fn)dlsym(h, "mh_conv3x3_packed_floats"); plconv_ws = (sz5_fn)dlsym(h, "mh_plconv3x3_ws_bytes");
    v2conv_ws = (sz5_fn)dlsym(h, "mh_conv3x3_ws_bytes"); act_planes = (actpl_fn)dlsym(h, "mh_act_planes"); plpack = (plpack_fn)dlsym(h, "mh_plconv_pack_weight");
    plconv = (plconv_fn)dlsym(h, "mh_plconv3x3"); v2pack = (v2pack_fn)dlsym(h, "mh_conv3x3_pack_weight"); v2conv = (v2conv_fn)dlsym(h, "mh_conv3x3_nhwc");
    set_conv_shape = (shape_fn)dlsym(h, "mh_debug_plconv_shape");
    set_conv_splitk = (shape_fn)dlsym(h, "mh_debug_plconv_splitk");
    set_conv_flags = (shape_fn)dlsym(h, "mh_debug_plconv_flags");
    plconv_img = (plconv_img_fn)dlsym(h, "mh_plconv3x3_to_image"); plconv_pool = (plconv_img_fn)dlsym(h, "mh_plconv3x3_pool_to_image"); stem_img = (stem_img_fn)dlsym(h, "mh_stem_to_image"); stem_max = (stem_max_fn)dlsym(h, "mh_conv_first_nchw_max");
    if (!act_bytes || !plpacked_bytes || !v2packed_floats || !plconv_ws || !v2conv_ws || !act_planes || !plpack || !plconv || !v2pack || !v2conv || !set_conv_shape || !plconv_img || !stem_img || !stem_max) { printf("missing conv symbol\n"); return 2; }
    if (argc > 2 && !strcmp(argv[2], "--conv-replay")) {
        // the 12 trunk launches of one bench step (b = 6, 592x592), each once: the target of the rocprofv3 --pmc FETCH_SIZE /
        // WRITE_SIZE passes behind bench.py's roofline.traffic (tools/r03/traffic.sh); prints the algorithmic bytes per launch
        // round 6: every layer with the epilogue the step uses -- 1 = output as the next layer's image, 2 = through the 2x2 pool as the
        // next layer's image, 0 = fp32 NHWC (the last layer); the input image of a layer is made by the converter here
        struct L { const char *name; int H, Cin, Cout, mode; } layers[] = {
            {"conv1_2", 592, 64, 64, 2}, {"conv2_1", 296, 64, 128, 1}, {"conv2_2", 296, 128, 128, 2}, {"conv3_1", 148, 128, 256, 1},
            {"conv3_2", 148, 256, 256, 1}, {"conv3_3", 148, 256, 256, 2}, {"conv4_1", 74, 256, 512, 1}, {"conv4_2", 74, 512, 512, 1},
            {"conv4_3", 74, 512, 512, 2}, {"conv5_1", 37, 512, 512, 1}, {"conv5_2", 37, 512, 512, 1}, {"conv5_3", 37, 512, 512, 0}};
        const int B = 6;
        // --conv-replay N: every layer N times back to back (the counter passes then take the LAST launch of each layer: warm
        // caches and clocks, as in the step, instead of the first launch after an idle gap)
        const int reps = argc > 3 ? std::max(1, atoi(argv[3])) : 1;
        if (argc > 4) set_conv_splitk(atoi(argv[4]));          // measurement: every tile in that many K slices (1 = whole tiles, no tail slicing)
        for (const L &l : layers) {
            const size_t nx = (size_t)B * l.H * l.H * l.Cin;
            Dev dx(nx * 4), dw((size_t)l.Cout * l.Cin * 9 * 4), db(l.Cout * 4), dmb(B * 4), dmbo(B * 4), dout((size_t)B * l.H * l.H * l.Cout * 4);
            fill_dev(dx.f(), nx, 3); fill_dev(dw.f(), (size_t)l.Cout * l.Cin * 9, 5); fill_dev(db.f(), l.Cout, 6);
            std::vector<unsigned> mb(B, fbits(6.0f));
            HIP_OK(hipMemcpy(dmb.p, mb.data(), B * 4, hipMemcpyHostToDevice));
            HIP_OK(hipMemset(dmbo.p, 0, B * 4));
            Dev img(act_bytes(B, l.H, l.H, l.Cin)), pk(plpacked_bytes(l.Cout, l.Cin)), ws3(plconv_ws(B, l.H, l.H, l.Cin, l.Cout));
            plpack(dw.f(), l.Cout, l.Cin, 0, pk.p, nullptr);
            act_planes(dx.f(), (const unsigned *)dmb.p, B, l.H, l.H, l.Cin, 0, img.p, nullptr);
            int rc = 0;
            for (int rep = 0; rep < reps; ++rep) {
                if (l.mode == 0) rc = plconv(img.p, B, l.H, l.H, l.Cin, pk.p, l.Cout, db.f(), 1, dout.f(), (unsigned *)dmbo.p, ws3.p, ws3.n, nullptr);
                else if (l.mode == 1) rc = plconv_img(img.p, (const unsigned *)dmb.p, B, l.H, l.H, l.Cin, pk.p, l.Cout, db.f(), 1, dout.p, (unsigned *)dmbo.p, ws3.p, ws3.n, nullptr);
                else rc = plconv_pool(img.p, (const unsigned *)dmb.p, B, l.H, l.H, l.Cin, pk.p, l.Cout, db.f(), 1, dout.p, (unsigned *)dmbo.p, ws3.p, ws3.n, nullptr);
            }
            HIP_OK(hipDeviceSynchronize());
            const double M = (double)B * l.H * l.H;
            printf("{\"layer\": \"%s\", \"H\": %d, \"Cin\": %d, \"Cout\": %d, \"epilogue\": \"%s\", \"rc\": %d, \"algorithmic_read_bytes\": %.0f, \"algorithmic_write_bytes\": %.0f, \"flops\": %.0f}\n",
                   l.name, l.H, l.Cin, l.Cout, l.mode == 0 ? "fp32" : (l.mode == 1 ? "image" : "pooled image"), rc, 4.0 * (M * l.Cin + 9.0 * l.Cin * l.Cout),
                   4.0 * M * l.Cout / (l.mode == 2 ? 4 : 1), 2.0 * 9 * l.Cin * l.Cout * M);
        }
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "--conv-noepi")) {
        // what the K loop alone costs: debug flag 2 = the ring kernel returns without its epilogue (no output)
        for (int fl : {0, 2}) {
            set_conv_flags(fl);
            printf("{\"debug_flags\": %d}\n", fl);
            g_sweep_quick = true;
            conv_sweep("conv1_2", 6, 592, 592, 64, 64, 5);
            conv_sweep("conv2_1", 6, 296, 296, 64, 128, 5);
            conv_sweep("conv2_2", 6, 296, 296, 128, 128, 5);
            conv_sweep("conv3_2", 6, 148, 148, 256, 256, 5);
        }
        set_conv_flags(0);
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "--stem")) {
        // conv1_1 -> image at the bench size (b = 6, 592 x 592): MH_STEM=valu in the environment selects the VALU kernel
        const int B = 6, H = 592, W = 592, C0 = 64;
        const size_t M = (size_t)B * H * W;
        Dev dx(M * 3 * 4), dw((size_t)C0 * 27 * 4), db(C0 * 4), mb(64 * 4), img(act_bytes(B, H, W, C0));
        fill_dev(dx.f(), M * 3, 3); fill_dev(dw.f(), (size_t)C0 * 27, 5); fill_dev(db.f(), C0, 6);
        const float ms = time_ms(20, [&] { HIP_OK(hipMemsetAsync(mb.p, 0, 64 * 4, nullptr)); stem_img(dx.f(), B, 3, H, W, dw.f(), C0, db.f(), 1, img.p, (unsigned *)mb.p, nullptr); });
        printf("{\"check\": \"stem speed\", \"variant\": \"%s\", \"ms\": %.4f, \"image_GBps\": %.0f}\n", getenv("MH_STEM") ? getenv("MH_STEM") : "mfma", ms, M * C0 * 4.0 / ms * 1e-6);
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "--conv-r06")) {
        // round 6: the ring shapes only -- 64-channel tiles (5, 6) next to the round-3 loop (2) on conv1_2, pooled image output,
        // K slices added up inside the launch (conv5: 1 .. 6 slices), with and without the epilogue (debug flag 2)
        g_sweep_ring_only = true;
        for (int fl : {0, 2}) {
            set_conv_flags(fl);
            printf("{\"debug_flags\": %d}\n", fl);
            g_sweep_quick = fl != 0;
            conv_sweep("conv1_2", 6, 592, 592, 64, 64, 5);
            conv_sweep("conv2_1", 6, 296, 296, 64, 128, 5);
            conv_sweep("conv2_2", 6, 296, 296, 128, 128, 5);
            conv_sweep("conv3_1", 6, 148, 148, 128, 256, 5);
            conv_sweep("conv3_3", 6, 148, 148, 256, 256, 5);
            conv_sweep("conv4_1", 6, 74, 74, 256, 512, 5);
            conv_sweep("conv4_3", 6, 74, 74, 512, 512, 5);
            g_sweep_quick = fl != 0 && fl != 2;
            conv_sweep("conv5_1", 6, 37, 37, 512, 512, 10);
        }
        set_conv_flags(0);
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "--conv-sweep")) {
        g_sweep_quick = argc > 3 && !strcmp(argv[3], "--quick");
        conv_sweep("conv1_2", 6, 592, 592, 64, 64, 5);
        conv_sweep("conv2_1", 6, 296, 296, 64, 128, 5);
        conv_sweep("conv2_2", 6, 296, 296, 128, 128, 5);
        conv_sweep("conv3_1", 6, 148, 148, 128, 256, 5);
        conv_sweep("conv3_2", 6, 148, 148, 256, 256, 5);
        conv_sweep("conv4_1", 6, 74, 74, 256, 512, 5);
        conv_sweep("conv4_2", 6, 74, 74, 512, 512, 5);
        conv_sweep("conv5_1", 6, 37, 37, 512, 512, 10);
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "--conv")) {
        int badc = 0;
        badc += conv_case("small", 3, 20, 20, 32, 64, 0, true, 31);
        badc += conv_case("small pooled", 2, 13, 17, 16, 128, 1, true, 32);
        badc += conv_case("ragged rows", 3, 23, 19, 64, 256, 0, true, 33);
        badc += conv_case("conv5 (split-K)", 6, 37, 37, 512, 512, 0, false, 34);
        badc += conv_case("conv4 pooled (tail slices)", 6, 74, 74, 256, 512, 1, false, 35);
        badc += conv_case("conv2 pooled", 2, 296, 296, 64, 128, 1, false, 36);
        badc += chain_case("small", 3, 20, 20, 32, 64, 128, false, 41);
        badc += chain_case("ragged rows, Cout 64", 2, 23, 19, 64, 64, 64, false, 42);
        badc += chain_case("stem", 3, 24, 31, 64, 64, 128, true, 43);
        badc += chain_case("conv5-like (split-K)", 6, 37, 37, 512, 512, 512, false, 44);
        badc += chain_case("conv4-like (tail slices)", 6, 74, 74, 256, 512, 256, false, 45);
        badc += pool_case("small", 3, 20, 20, 32, 64, 0, 51);
        badc += pool_case("ragged windows", 2, 26, 18, 64, 128, 0, 52);
        badc += pool_case("three images per tile", 5, 6, 10, 16, 256, 0, 53);
        badc += pool_case("conv4_3-like (tail slices)", 6, 74, 74, 256, 512, 0, 54);
        badc += pool_case("K slices added up in the launch", 3, 36, 36, 256, 128, 3, 55);
        badc += pool_case("conv1_2-like", 1, 120, 136, 64, 64, 0, 56);
        printf("{\"check\": \"conv summary\", \"failed\": %d}\n", badc);
        if (argc > 3 && !strcmp(argv[3], "--speed")) {
            conv_speed("conv1_2", 6, 592, 592, 64, 64, 0, 5);
            conv_speed("conv2_1", 6, 296, 296, 64, 128, 1, 5);
            conv_speed("conv2_2", 6, 296, 296, 128, 128, 0, 5);
            conv_speed("conv3_1", 6, 148, 148, 128, 256, 1, 5);
            conv_speed("conv3_2", 6, 148, 148, 256, 256, 0, 5);
            conv_speed("conv4_1", 6, 74, 74, 256, 512, 1, 5);
            conv_speed("conv4_2", 6, 74, 74, 512, 512, 0, 5);
            conv_speed("conv5_1", 6, 37, 37, 512, 512, 1, 5);
        }
        return badc ? 1 : 0;
    }
    if (argc > 2 && !strcmp(argv[2], "--order-ab")) {
        set_order = (shape_fn)dlsym(h, "mh_debug_pl_order");
        if (!set_order) { printf("missing mh_debug_pl_order\n"); return 2; }
        order_ab("fc6 forward", 1536, 4096, 25088, 5);
        order_ab("fc6 input gradient", 1536, 25088, 4096, 5);
        order_ab("fc6 weight gradient", 4096, 25088, 1536, 5);
        order_ab("fc7 forward", 1536, 4096, 4096, 10);
        order_ab("fc7 weight gradient", 4096, 4096, 1536, 10);
        order_ab("4096^3", 4096, 4096, 4096, 5);
        return 0;
    }
    int bad = 0;
    if (argc > 2 && !strcmp(argv[2], "--ring")) {
        bad += image_case(520, 300, 1040, 21);
        bad += image_case(130, 60, 200, 22);
        bad += image_case(257, 513, 16 * 37, 25);
        bad += image_case(1000, 1000, 16 * 4 + 5, 26);
        printf("{\"check\": \"ring accuracy summary\", \"failed\": %d}\n", bad);
        if (argc > 3 && !strcmp(argv[3], "--accuracy")) return bad ? 1 : 0;
        ring_speed("4096^3", 4096, 4096, 4096, 5);
        ring_speed("fc6 forward", 1536, 4096, 25088, 5);
        ring_speed("fc7 forward", 1536, 4096, 4096, 10);
        ring_speed("fc6 weight gradient", 4096, 25088, 1536, 5);
        ring_speed("fc6 input gradient", 1536, 25088, 4096, 5);
        return bad ? 1 : 0;
    }
    if (pmc) {      // a few launches of the product kernel on ready images, nothing else: the target of rocprofv3 --pmc
        const int M = 1536, N = 4096, K = 25088;
        Dev dA((size_t)M * K * 4), dB((size_t)N * K * 4), dC((size_t)M * N * 4), ia(pbytes(M, K)), ib(pbytes(N, K));
        fill_dev(dA.f(), (size_t)M * K, 1); fill_dev(dB.f(), (size_t)N * K, 2);
        mkplanes(dA.f(), 1, M, K, K, ia.p, nullptr);
        mkplanes(dB.f(), 1, N, K, K, ib.p, nullptr);
        const int shape = argc > 3 ? atoi(argv[3]) : -1, sk = argc > 4 ? atoi(argv[4]) : 0;      // --pmc [shape] [splitk]
        set_shape(shape);
        Dev ws(wspl(M, N, K, sk));
        for (int i = 0; i < 3; ++i) gemmpl(M, N, K, ia.p, ib.p, dC.f(), N, nullptr, 0, 0, sk, ws.p, ws.n, nullptr);
        HIP_OK(hipDeviceSynchronize());
        return 0;
    }
    if (!speed_only) {
        bad += accuracy_case("NT aligned", 0, 1, 256, 256, 512, 0, 0, 0, true, 0, 0, 0, 0, 1);
        bad += accuracy_case("NN aligned", 0, 0, 256, 384, 512, 0, 0, 0, true, 0, 0, 0, 0, 2);
        bad += accuracy_case("TN aligned", 1, 0, 384, 256, 512, 0, 0, 0, true, 0, 0, 0, 0, 3);
        bad += accuracy_case("TT aligned", 1, 1, 256, 256, 512, 0, 0, 0, true, 0, 0, 0, 0, 4);
        bad += accuracy_case("NT ragged bias relu", 0, 1, 301, 151, 1003, 1, 3, 5, true, 1, 1, 0, 0, 5);
        bad += accuracy_case("NN ragged accumulate", 0, 0, 77, 259, 130, 3, 1, 0, false, 1, 0, 1, 0, 6);
        bad += accuracy_case("TN ragged splitk 3", 1, 0, 203, 190, 999, 1, 2, 1, true, 0, 0, 0, 3, 7);
        bad += accuracy_case("TT ragged", 1, 1, 130, 67, 75, 2, 1, 3, false, 1, 0, 0, 0, 8);
        bad += accuracy_case("NT narrow N=51", 0, 1, 700, 51, 4096, 0, 0, 0, false, 1, 0, 0, 0, 9);
        bad += accuracy_case("NT 120 rows", 0, 1, 120, 512, 2048, 0, 0, 0, false, 1, 1, 0, 0, 10);
        bad += accuracy_case("NT K=7", 0, 1, 40, 33, 7, 0, 0, 0, false, 0, 0, 0, 0, 11);
        bad += accuracy_case("NT 600x600x600 splitk 5 accumulate", 0, 1, 600, 600, 600, 0, 0, 0, true, 1, 1, 1, 5, 12);
        bad += image_case(520, 300, 1040, 21);
        bad += image_case(130, 60, 200, 22);
        bad += both_case(300, 260, 333, 23);
        bad += both_case(64, 517, 70, 24);
        printf("{\"check\": \"accuracy summary\", \"failed\": %d}\n", bad);
    }
    if (!quick) {
        speed_case("fc6 forward", 0, 1, 1536, 4096, 25088, 5);
        speed_case("fc6 input gradient", 0, 0, 1536, 25088, 4096, 5);
        speed_case("fc6 weight gradient", 1, 0, 4096, 25088, 1536, 5);
        speed_case("fc7 forward", 0, 1, 1536, 4096, 4096, 10);
        speed_case("object fc6 (120 rows)", 0, 1, 120, 4096, 25088, 10);
        speed_case("4096^3", 0, 1, 4096, 4096, 4096, 5);
    }
    return bad ? 1 : 0;
}
