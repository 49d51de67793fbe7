// Device check of the two exact bf16 splits behind mh_gemm_f32 (truncation = default build, round-to-nearest =
// MH_SPLIT_RN=1 build): error against a float64 CPU product, the sign bias on all-positive data, and GEMM speed.
//   hipcc -O2 tools/split_check.cpp -o tools/_bin/split_check -ldl
//   tools/_bin/split_check <libmotifs_hip.so> [<another build> ...]
// No torch, no Python: starts in milliseconds on a fresh GPU box.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef int (*gemm_fn)(int, int, int, int, int, const float *, int, const float *, int, float *, int, const float *,
                       int, int, int, void *, size_t, void *);
typedef int (*int_fn)(void);
typedef size_t (*gemm_ws_fn)(int, int, int, int);
static gemm_ws_fn g_gemm_ws = nullptr;      // mh_gemm_ws_bytes of the library under test (the f16x3 build keeps row exponents there)

struct GemmWs {      // workspace for an unsplit GEMM of the given shape (nullptr / 0 where the build needs none)
    void *p = nullptr;
    size_t n = 0;
    GemmWs(int M, int N, int K) { n = g_gemm_ws ? g_gemm_ws(M, N, K, 1) : 0; if (n && hipMalloc(&p, n) != hipSuccess) { printf("hipMalloc ws\n"); exit(2); } }
    ~GemmWs() { if (p) (void)hipFree(p); }
};

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

struct Err { double max_rel, rms_rel, mean_signed; };

static Err run_case(gemm_fn gemm, int M, int N, int K, bool positive, unsigned seed)
{
    std::mt19937 rng(seed);
    std::normal_distribution<float> nrm(0.f, 1.f);
    std::uniform_real_distribution<float> expo(-8.f, 8.f);
    std::vector<float> A((size_t)M * K), B((size_t)K * N), C((size_t)M * N);
    for (auto &v : A) { v = nrm(rng) * std::exp2(expo(rng)); if (positive) v = std::fabs(v); }
    for (auto &v : B) { v = nrm(rng) * std::exp2(expo(rng)); if (positive) v = std::fabs(v); }
    float *dA, *dB, *dC;
    HIP_OK(hipMalloc(&dA, A.size() * 4)); HIP_OK(hipMalloc(&dB, B.size() * 4)); HIP_OK(hipMalloc(&dC, C.size() * 4));
    HIP_OK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    GemmWs ws(M, N, K);
    int rc = gemm(0, 0, M, N, K, dA, K, dB, N, dC, N, nullptr, 0, 0, 1, ws.p, ws.n, nullptr);
    if (rc) { printf("mh_gemm_f32 rc=%d\n", rc); exit(3); }
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> R((size_t)M * N, 0.0);
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) {
            const double a = A[(size_t)m * K + k];
            const float *b = &B[(size_t)k * N];
            double *r = &R[(size_t)m * N];
            for (int n = 0; n < N; ++n) r[n] += a * (double)b[n];
        }
    double ss = 0, se = 0, mx = 0, sg = 0;
    for (size_t i = 0; i < R.size(); ++i) {
        const double e = (double)C[i] - R[i];
        ss += R[i] * R[i]; se += e * e; mx = std::fmax(mx, std::fabs(e));
        if (R[i] != 0) sg += e / R[i];
    }
    const double rms = std::sqrt(ss / R.size());
    HIP_OK(hipFree(dA)); HIP_OK(hipFree(dB)); HIP_OK(hipFree(dC));
    return {mx / rms, std::sqrt(se / R.size()) / rms, sg / R.size()};
}

static double speed(gemm_fn gemm, int S, int iters)
{
    float *dA, *dB, *dC;
    HIP_OK(hipMalloc(&dA, (size_t)S * S * 4)); HIP_OK(hipMalloc(&dB, (size_t)S * S * 4)); HIP_OK(hipMalloc(&dC, (size_t)S * S * 4));
    std::vector<float> h((size_t)S * S);
    std::mt19937 rng(7); std::normal_distribution<float> nrm(0.f, 1.f);
    for (auto &v : h) v = nrm(rng);
    HIP_OK(hipMemcpy(dA, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dB, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    GemmWs ws(S, S, S);
    for (int i = 0; i < 3; ++i) gemm(0, 1, S, S, S, dA, S, dB, S, dC, S, nullptr, 0, 0, 1, ws.p, ws.n, nullptr);
    HIP_OK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) gemm(0, 1, S, S, S, dA, S, dB, S, dC, S, nullptr, 0, 0, 1, ws.p, ws.n, nullptr);
    HIP_OK(hipEventRecord(e1, nullptr)); HIP_OK(hipEventSynchronize(e1));
    float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    HIP_OK(hipFree(dA)); HIP_OK(hipFree(dB)); HIP_OK(hipFree(dC));
    return 2.0 * S * S * S * iters / (ms * 1e-3) / 1e12;
}

typedef size_t (*sz2_fn)(int, int);
typedef size_t (*sz5_fn)(int, int, int, int, int);
typedef int (*pack_fn)(const float *, int, int, int, float *, void *);
typedef int (*conv_fn)(const float *, int, int, int, int, const float *, int, const float *, int, float *, void *, size_t, void *);
typedef int (*wgrad_fn)(const float *, const float *, int, int, int, int, int, float *, void *, size_t, void *);

// a permutation matrix must carry every mantissa bit of the other operand through the matrix cores
static int exact_copies(gemm_fn gemm)
{
    const int S = 128;
    std::mt19937 rng(11);
    std::vector<float> X((size_t)S * S), P((size_t)S * S, 0.f), C((size_t)S * S);
    for (auto &v : X) { unsigned u = (rng() & 0x007fffffu) | ((100u + rng() % 56u) << 23) | ((rng() & 1u) << 31); v = *(float *)&u; }
    for (int i = 0; i < S; ++i) P[(size_t)i * S + (i * 37 + 5) % S] = 1.f;
    float *dX, *dP, *dC;
    HIP_OK(hipMalloc(&dX, X.size() * 4)); HIP_OK(hipMalloc(&dP, P.size() * 4)); HIP_OK(hipMalloc(&dC, C.size() * 4));
    HIP_OK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice));
    int bad = 0;
    GemmWs ws(S, S, S);
    gemm(0, 0, S, S, S, dX, S, dP, S, dC, S, nullptr, 0, 0, 1, ws.p, ws.n, nullptr);      // C[m][(k*37+5)%S] = X[m][k]
    HIP_OK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    for (int m = 0; m < S; ++m) for (int k = 0; k < S; ++k) bad += C[(size_t)m * S + (k * 37 + 5) % S] != X[(size_t)m * S + k];
    gemm(0, 0, S, S, S, dP, S, dX, S, dC, S, nullptr, 0, 0, 1, ws.p, ws.n, nullptr);      // C[i][n] = X[(i*37+5)%S][n]
    HIP_OK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < S; ++i) for (int n = 0; n < S; ++n) bad += C[(size_t)i * S + n] != X[(size_t)((i * 37 + 5) % S) * S + n];
    HIP_OK(hipFree(dX)); HIP_OK(hipFree(dP)); HIP_OK(hipFree(dC));
    return bad;
}

// 3x3/1/1 convolution and its weight gradient against float64 loops; returns rms errors over rms of the result
static void conv_case(void *h, double &conv_rms, double &conv_max, double &wg_rms, double &wg_max, int &wg_rc,
                      int B = 2, int H = 20, int W = 24, int Ci = 64, int Co = 64)
{
    sz2_fn packed = (sz2_fn)dlsym(h, "mh_conv3x3_packed_floats");
    pack_fn pack = (pack_fn)dlsym(h, "mh_conv3x3_pack_weight");
    sz5_fn cws = (sz5_fn)dlsym(h, "mh_conv3x3_ws_bytes"), gws = (sz5_fn)dlsym(h, "mh_conv3x3_wgrad_ws_bytes");
    conv_fn conv = (conv_fn)dlsym(h, "mh_conv3x3_nhwc");
    wgrad_fn wgrad = (wgrad_fn)dlsym(h, "mh_conv3x3_wgrad");
    std::mt19937 rng(5); std::normal_distribution<float> nrm(0.f, 1.f); std::uniform_real_distribution<float> expo(-4.f, 4.f);
    const size_t np = (size_t)B * H * W;
    std::vector<float> x(np * Ci), w((size_t)Co * Ci * 9), gy(np * Co), y(np * Co), dw((size_t)Co * 9 * Ci);
    for (auto &v : x) v = std::fabs(nrm(rng)) * std::exp2(expo(rng));      // post-ReLU-like input
    for (auto &v : w) v = nrm(rng) * 0.05f;
    for (auto &v : gy) v = nrm(rng) * std::exp2(expo(rng));
    float *dx, *dwt, *dwp, *dy, *dgy, *ddw; void *ws = nullptr;
    HIP_OK(hipMalloc(&dx, x.size() * 4)); HIP_OK(hipMalloc(&dwt, w.size() * 4)); HIP_OK(hipMalloc(&dwp, packed(Co, Ci) * 4));
    HIP_OK(hipMalloc(&dy, y.size() * 4)); HIP_OK(hipMalloc(&dgy, gy.size() * 4)); HIP_OK(hipMalloc(&ddw, dw.size() * 4));
    HIP_OK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dwt, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dgy, gy.data(), gy.size() * 4, hipMemcpyHostToDevice));
    size_t wsb = std::max(cws(B, H, W, Ci, Co), gws(B, H, W, Ci, Co));
    if (wsb) HIP_OK(hipMalloc(&ws, wsb));
    int rc = pack(dwt, Co, Ci, 0, dwp, nullptr);
    rc |= conv(dx, B, H, W, Ci, dwp, Co, nullptr, 0, dy, ws, wsb, nullptr);
    if (rc) { printf("conv rc=%d\n", rc); exit(3); }
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
    wg_rc = wgrad(dx, dgy, B, H, W, Ci, Co, ddw, ws, wsb, nullptr);
    HIP_OK(hipDeviceSynchronize());
    if (!wg_rc) HIP_OK(hipMemcpy(dw.data(), ddw, dw.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> yr(np * Co, 0.0), dwr((size_t)Co * 9 * Ci, 0.0);
    for (int b = 0; b < B; ++b) for (int i = 0; i < H; ++i) for (int j = 0; j < W; ++j) {
        const size_t p = ((size_t)b * H + i) * W + j;
        for (int t = 0; t < 9; ++t) {
            const int ii = i + t / 3 - 1, jj = j + t % 3 - 1;
            if (ii < 0 || ii >= H || jj < 0 || jj >= W) continue;
            const float *xs = &x[(((size_t)b * H + ii) * W + jj) * Ci];
            for (int co = 0; co < Co; ++co) {
                double acc = 0; const double g = gy[p * Co + co];
                double *dr = &dwr[((size_t)co * 9 + t) * Ci];
                for (int ci = 0; ci < Ci; ++ci) { acc += (double)xs[ci] * (double)w[((size_t)co * Ci + ci) * 9 + t]; dr[ci] += g * (double)xs[ci]; }
                yr[p * Co + co] += acc;
            }
        }
    }
    auto cmp = [](const std::vector<float> &a, const std::vector<double> &r, double &rms_e, double &max_e) {
        double ss = 0, se = 0, mx = 0;
        for (size_t i = 0; i < r.size(); ++i) { const double e = a[i] - r[i]; ss += r[i] * r[i]; se += e * e; mx = std::fmax(mx, std::fabs(e)); }
        const double rms = std::sqrt(ss / r.size()); rms_e = std::sqrt(se / r.size()) / rms; max_e = mx / rms;
    };
    cmp(y, yr, conv_rms, conv_max);
    if (!wg_rc) cmp(dw, dwr, wg_rms, wg_max); else wg_rms = wg_max = -1;
    HIP_OK(hipFree(dx)); HIP_OK(hipFree(dwt)); HIP_OK(hipFree(dwp)); HIP_OK(hipFree(dy)); HIP_OK(hipFree(dgy)); HIP_OK(hipFree(ddw));
    if (ws) HIP_OK(hipFree(ws));
}

int main(int argc, char **argv)
{
    bool conv_only = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--conv-only")) { conv_only = true; continue; }
        void *h = dlopen(argv[i], RTLD_NOW | RTLD_LOCAL);
        if (!h) { printf("dlopen %s: %s\n", argv[i], dlerror()); return 1; }
        gemm_fn gemm = (gemm_fn)dlsym(h, "mh_gemm_f32");
        g_gemm_ws = (gemm_ws_fn)dlsym(h, "mh_gemm_ws_bytes");
        int_fn split = (int_fn)dlsym(h, "mh_mfma_split"), rne = (int_fn)dlsym(h, "mh_split_rne");
        if (!gemm || !split || !rne) { printf("missing symbols in %s\n", argv[i]); return 1; }
        if (conv_only) {      // several conv shapes: 256x64 and 128x128 tiles, split-K, ragged Cout, many small images
            const int shp[][5] = {{2, 20, 24, 64, 64}, {1, 19, 23, 256, 192}, {5, 14, 14, 128, 256}, {1, 37, 37, 512, 64}, {1, 9, 70, 32, 100}};
            for (const auto &q : shp) {
                double c_rms, c_max, g_rms, g_max; int g_rc;
                conv_case(h, c_rms, c_max, g_rms, g_max, g_rc, q[0], q[1], q[2], q[3], q[4]);
                printf("{\"lib\": \"%s\", \"conv\": [%d, %d, %d, %d, %d], \"max_err_over_rms\": %.3e, \"rms_err_over_rms\": %.3e, "
                       "\"wgrad_rc\": %d, \"wgrad_max_err_over_rms\": %.3e}\n", argv[i], q[0], q[1], q[2], q[3], q[4], c_max, c_rms, g_rc, g_max);
                fflush(stdout);
            }
            continue;
        }
        const Err mixed = run_case(gemm, 256, 256, 4096, false, 1), pos = run_case(gemm, 256, 256, 4096, true, 2);
        const double tf = speed(gemm, 4096, 20);
        const int bad = exact_copies(gemm);
        double c_rms, c_max, g_rms, g_max; int g_rc;
        conv_case(h, c_rms, c_max, g_rms, g_max, g_rc);
        printf("{\"lib\": \"%s\", \"mfma_split\": %d, \"split_rne\": %d, \"K\": 4096, "
               "\"mixed_sign\": {\"max_err_over_rms\": %.3e, \"rms_err_over_rms\": %.3e}, "
               "\"all_positive\": {\"max_err_over_rms\": %.3e, \"rms_err_over_rms\": %.3e, \"mean_signed_rel_err\": %.3e}, "
               "\"gemm_4096_nt_tflops\": %.1f, \"permutation_copy_mismatches\": %d, "
               "\"conv3x3_64x64\": {\"max_err_over_rms\": %.3e, \"rms_err_over_rms\": %.3e}, "
               "\"conv3x3_wgrad\": {\"rc\": %d, \"max_err_over_rms\": %.3e, \"rms_err_over_rms\": %.3e}}\n",
               argv[i], split(), rne(), mixed.max_rel, mixed.rms_rel, pos.max_rel, pos.rms_rel, pos.mean_signed, tf, bad, c_max, c_rms, g_rc, g_max, g_rms);
        fflush(stdout);
    }
    return 0;
}
