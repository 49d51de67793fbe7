// One launch of each fc6 GEMM of the bench step (forward, input gradient, weight gradient at M = 1536 relation rows, and the
// 120-row object forward) plus one streaming kernel of known byte count (calibration), through the C ABI, without torch --
// meant to run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (tools/traffic_run.sh gemm).  MH_GEMM_PATCH=rows in
// the environment replays the round-1 tile order.  Buffers are left uninitialised (HBM traffic does not depend on values).
//   hipcc -O2 tools/gemm_traffic.cpp -o tools/_bin/gemm_traffic -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>

typedef int (*gemm_fn)(int, int, int, int, int, const float *, int, const float *, int, float *, int, const float *, int, int, int,
                       void *, size_t, void *);
typedef size_t (*ws_fn)(int, int, int, int);
typedef int (*auto_fn)(int, int, int);
typedef int (*act_fn)(const float *, const float *, long long, int, float *, void *);
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

struct Shape { const char *name; int ta, tb, M, N, K; };

int main(int argc, char **argv)
{
    void *h = dlopen(argc > 1 ? argv[1] : "neural-motifs_amd/csrc/libmotifs_hip.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
    gemm_fn gemm = (gemm_fn)dlsym(h, "mh_gemm_f32");
    ws_fn wsb = (ws_fn)dlsym(h, "mh_gemm_ws_bytes");
    auto_fn autos = (auto_fn)dlsym(h, "mh_gemm_auto_splitk");
    act_fn act = (act_fn)dlsym(h, "mh_act_bwd");
    if (!gemm || !wsb || !autos || !act) { printf("missing symbols\n"); return 1; }
    const Shape shapes[] = {
        {"fc6_fwd_M1536", 0, 1, 1536, 4096, 25088}, {"fc6_dgrad_M1536", 0, 0, 1536, 25088, 4096},
        {"fc6_wgrad_M1536", 1, 0, 4096, 25088, 1536}, {"fc6_fwd_M120", 0, 1, 120, 4096, 25088},
        {"fc7_fwd_M1536", 0, 1, 1536, 4096, 4096},
    };
    const size_t big = (size_t)4096 * 25088 * 4;                 // the fc6 weight
    float *dA, *dB, *dC, *dthird; void *ws = nullptr;
    size_t ws_b = 0;
    for (const Shape &s : shapes) ws_b = std::max(ws_b, wsb(s.M, s.N, s.K, autos(s.M, s.N, s.K)));
    HIP_OK(hipMalloc(&dA, big)); HIP_OK(hipMalloc(&dB, big)); HIP_OK(hipMalloc(&dC, big));
    const long long n_cal = 64ll << 20;
    HIP_OK(hipMalloc(&dthird, (size_t)n_cal * 4));
    if (ws_b) HIP_OK(hipMalloc(&ws, ws_b));
    int idx = 0;
    for (const Shape &s : shapes) {
        const int lda = s.ta ? s.M : s.K, ldb = s.tb ? s.K : s.N;
        const int sk = autos(s.M, s.N, s.K);
        const int rc = gemm(s.ta, s.tb, s.M, s.N, s.K, dA, lda, dB, ldb, dC, s.N, nullptr, 0, 0, sk, ws, wsb(s.M, s.N, s.K, sk), nullptr);
        HIP_OK(hipDeviceSynchronize());
        printf("{\"launch\": %d, \"name\": \"%s\", \"rc\": %d, \"M\": %d, \"N\": %d, \"K\": %d, \"splitk\": %d, "
               "\"read_bytes_algorithmic\": %zu, \"write_bytes_algorithmic\": %zu, \"splitk_ws_bytes\": %zu, \"flops\": %.0f}\n",
               idx++, s.name, rc, s.M, s.N, s.K, sk, ((size_t)s.M * s.K + (size_t)s.N * s.K) * 4, (size_t)s.M * s.N * 4,
               sk > 1 ? (size_t)sk * s.M * s.N * 4 : (size_t)0, 2.0 * s.M * s.N * s.K);
        fflush(stdout);
    }
    const int rc = act(dA, dthird, n_cal, 1, dC, nullptr);
    HIP_OK(hipDeviceSynchronize());
    printf("{\"launch\": %d, \"name\": \"calibration_act_bwd\", \"rc\": %d, \"read_bytes_algorithmic\": %lld, \"write_bytes_algorithmic\": %lld}\n",
           idx, rc, 2 * n_cal * 4, n_cal * 4);
    return 0;
}
