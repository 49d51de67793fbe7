// One launch of every conv3x3 implicit-GEMM shape of the bench step (12 VGG16 layers at 6 x 592 x 592, the union
// tower's forward and input-gradient conv) plus one streaming kernel of known byte count (calibration), through the C
// ABI, without torch -- meant to run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (tools/traffic_run.sh):
// a PMC pass over these 15 dispatches takes seconds, a pass over the whole bench step does not finish.
// Buffers are left uninitialised on purpose (no fill kernels in the trace; HBM traffic does not depend on the values).
//   hipcc -O2 tools/conv_traffic.cpp -o tools/_bin/conv_traffic -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>

typedef size_t (*sz2_fn)(int, int);
typedef size_t (*sz5_fn)(int, int, int, int, int);
typedef int (*conv_fn)(const float *, int, int, int, int, const float *, int, const float *, int, float *, void *, size_t, void *);
typedef int (*act_fn)(const float *, const float *, long long, int, float *, void *);

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

struct Shape { const char *name; int B, H, W, Ci, Co; };

int main(int argc, char **argv)
{
    void *h = dlopen(argc > 1 ? argv[1] : "neural-motifs_amd/csrc/libmotifs_hip.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
    sz2_fn packed = (sz2_fn)dlsym(h, "mh_conv3x3_packed_floats");
    sz5_fn cws = (sz5_fn)dlsym(h, "mh_conv3x3_ws_bytes");
    conv_fn conv = (conv_fn)dlsym(h, "mh_conv3x3_nhwc");
    act_fn act = (act_fn)dlsym(h, "mh_act_bwd");
    if (!packed || !cws || !conv || !act) { printf("missing symbols\n"); return 1; }
    const Shape shapes[] = {
        {"conv1_2", 6, 592, 592, 64, 64},   {"conv2_1", 6, 296, 296, 64, 128},  {"conv2_2", 6, 296, 296, 128, 128},
        {"conv3_1", 6, 148, 148, 128, 256}, {"conv3_2", 6, 148, 148, 256, 256}, {"conv3_3", 6, 148, 148, 256, 256},
        {"conv4_1", 6, 74, 74, 256, 512},   {"conv4_2", 6, 74, 74, 512, 512},   {"conv4_3", 6, 74, 74, 512, 512},
        {"conv5_1", 6, 37, 37, 512, 512},   {"conv5_2", 6, 37, 37, 512, 512},   {"conv5_3", 6, 37, 37, 512, 512},
        {"tower_fwd", 1536, 14, 14, 128, 256}, {"tower_dgrad", 1536, 14, 14, 256, 128},
    };
    size_t in_b = 0, out_b = 0, wt_b = 0, ws_b = 0;
    for (const Shape &s : shapes) {
        const size_t px = (size_t)s.B * s.H * s.W;
        in_b = std::max(in_b, px * s.Ci * 4); out_b = std::max(out_b, px * s.Co * 4);
        wt_b = std::max(wt_b, packed(s.Co, s.Ci) * 4); ws_b = std::max(ws_b, cws(s.B, s.H, s.W, s.Ci, s.Co));
    }
    const long long n_cal = 64ll << 20;                       // calibration: reads 2 x 256 MiB, writes 256 MiB
    in_b = std::max(in_b, (size_t)n_cal * 4); out_b = std::max(out_b, (size_t)n_cal * 4);
    float *din, *dout, *dwt, *dthird; void *ws = nullptr;
    HIP_OK(hipMalloc(&din, in_b)); HIP_OK(hipMalloc(&dout, out_b)); HIP_OK(hipMalloc(&dwt, wt_b));
    HIP_OK(hipMalloc(&dthird, (size_t)n_cal * 4));
    if (ws_b) HIP_OK(hipMalloc(&ws, ws_b));
    int idx = 0;
    for (const Shape &s : shapes) {
        const size_t px = (size_t)s.B * s.H * s.W;
        const size_t wsb = cws(s.B, s.H, s.W, s.Ci, s.Co);
        const int rc = conv(din, s.B, s.H, s.W, s.Ci, dwt, s.Co, nullptr, 1, dout, ws, wsb, nullptr);
        HIP_OK(hipDeviceSynchronize());
        printf("{\"launch\": %d, \"name\": \"%s\", \"rc\": %d, \"B\": %d, \"H\": %d, \"W\": %d, \"Cin\": %d, \"Cout\": %d, "
               "\"read_bytes_algorithmic\": %zu, \"write_bytes_algorithmic\": %zu, \"splitk_ws_bytes\": %zu, \"flops\": %.0f}\n",
               idx++, s.name, rc, s.B, s.H, s.W, s.Ci, s.Co, px * s.Ci * 4 + packed(s.Co, s.Ci) * 4, px * s.Co * 4, wsb,
               2.0 * px * s.Ci * s.Co * 9);
        fflush(stdout);
    }
    const int rc = act(din, dthird, n_cal, 1, dout, nullptr);
    HIP_OK(hipDeviceSynchronize());
    printf("{\"launch\": %d, \"name\": \"calibration_act_bwd\", \"rc\": %d, \"read_bytes_algorithmic\": %lld, \"write_bytes_algorithmic\": %lld}\n",
           idx, rc, 2 * n_cal * 4, n_cal * 4);
    return 0;
}
