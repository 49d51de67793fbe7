// optim.hip -- the tail of the training step (models/train_rels.py:143-150): global gradient-norm clipping
// (lib/pytorch_misc.py:416-455) and SGD with momentum + weight decay, as multi-tensor kernels.
//
// The reference computes one `.norm()` per parameter with a host sync each, then scales every gradient, then runs
// torch's SGD (several elementwise passes per tensor).  Here:
//   mh_multi_sumsq   : one launch over a chunk table covering all gradients -> per-chunk partial sums of squares
//                      (fixed order => deterministic), second tiny launch -> total sum of squares on the DEVICE
//   mh_multi_sgd_step: one launch; every element does
//                         g' = g * clip_coef            clip_coef = min(1, max_norm / (sqrt(sumsq) + 1e-6))
//                         d  = g' + wd * p
//                         buf = first_step ? d : momentum * buf + d
//                         p  = p - lr * buf
//                      reading p,g,buf and writing p,buf once (5 x 4 B per element: HBM-bound), no host round trip.
// The chunk table (device array of {p, g, buf, n, lr}) is expanded ON THE DEVICE from the per-parameter list
// (mh_opt_build_chunks): the list travels in the kernel arguments, so a new set of gradient addresses (autograd hands out
// fresh .grad tensors every step) costs one tiny launch and no host->device copy -- a 140 KB pinned copy in front of the
// clip kernels was the one place where the asynchronous step waited on the copy engine.
#include <algorithm>

#include "common.h"

namespace mh {

struct OptChunk {
    float *p;
    const float *g;
    float *buf;
    int n;       // elements in this chunk (<= kChunkElems)
    float lr;
};
static_assert(sizeof(OptChunk) == 32, "OptChunk layout is part of the C ABI (4 x 8 bytes)");

constexpr int kChunkElems = 65536;

__global__ __launch_bounds__(256) void multi_sumsq_kernel(const OptChunk *__restrict__ chunks, float *__restrict__ partial)
{
    __shared__ float red[256];
    const OptChunk c = chunks[blockIdx.x];
    float s = 0.f;
    const int n4 = c.n >> 2;
    const bool vec = (reinterpret_cast<uintptr_t>(c.g) & 15) == 0;
    if (vec) {
        const float4 *g4 = reinterpret_cast<const float4 *>(c.g);
        for (int i = threadIdx.x; i < n4; i += 256) {
            const float4 v = g4[i];
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (int i = 4 * n4 + threadIdx.x; i < c.n; i += 256) s += c.g[i] * c.g[i];
    } else {
        for (int i = threadIdx.x; i < c.n; i += 256) s += c.g[i] * c.g[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float *__restrict__ partial, int n, float *__restrict__ out)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

// Device-side guard: a step whose squared gradient norm is not finite (NaN-poisoned outputs of a timed-out persistent
// LSTM launch, an overflow) must not touch the weights -- the host runs several steps ahead and could not stop it in time
// (ADVICE r02).  Such a step is SKIPPED here: weights and momentum stay as they are (a skipped first step zeroes the
// momentum buffer, which is what the next step then builds on), and one lane counts the skip in a host-pinned word that
// mh_opt_skipped_steps() reads without synchronising.
__global__ __launch_bounds__(256) void multi_sgd_kernel(const OptChunk *__restrict__ chunks,
                                                        const float *__restrict__ sumsq, float max_norm, float momentum,
                                                        float weight_decay, int first_step, unsigned *skip_word)
{
    const OptChunk c = chunks[blockIdx.x];
    float coef = 1.f;
    if (sumsq != nullptr && max_norm > 0.f) {
        const float ss = sumsq[0];
        if (!(ss >= 0.f && ss < INFINITY)) {          // NaN or inf: wave-uniform
            if (blockIdx.x == 0 && threadIdx.x == 0 && skip_word) atomicAdd(skip_word, 1u);
            if (first_step)
                for (int i = threadIdx.x; i < c.n; i += 256) c.buf[i] = 0.f;
            return;
        }
        const float total_norm = sqrtf(ss);
        const float cc = max_norm / (total_norm + 1e-6f);
        coef = cc < 1.f ? cc : 1.f;
    }
    const bool vec = ((reinterpret_cast<uintptr_t>(c.g) | reinterpret_cast<uintptr_t>(c.p) |
                       reinterpret_cast<uintptr_t>(c.buf)) & 15) == 0;
    auto upd = [&](float p, float g, float b, float &pn, float &bn) {
        const float d = g * coef + weight_decay * p;
        bn = first_step ? d : momentum * b + d;
        pn = p - c.lr * bn;
    };
    const int n4 = vec ? (c.n >> 2) : 0;
    float4 *p4 = reinterpret_cast<float4 *>(c.p);
    const float4 *g4 = reinterpret_cast<const float4 *>(c.g);
    float4 *b4 = reinterpret_cast<float4 *>(c.buf);
    for (int i = threadIdx.x; i < n4; i += 256) {
        float4 p = p4[i], b = b4[i];
        const float4 g = g4[i];
        upd(p.x, g.x, b.x, p.x, b.x);
        upd(p.y, g.y, b.y, p.y, b.y);
        upd(p.z, g.z, b.z, p.z, b.z);
        upd(p.w, g.w, b.w, p.w, b.w);
        p4[i] = p;
        b4[i] = b;
    }
    for (int i = 4 * n4 + threadIdx.x; i < c.n; i += 256) upd(c.p[i], c.g[i], c.buf[i], c.p[i], c.buf[i]);
}

// Per-parameter records of one expand launch, passed BY VALUE (kernel-argument segment: 96 x 32 B + 2 ints < 4 KB).
constexpr int kParamsPerLaunch = 96;
struct OptParamList {
    OptChunk prm[kParamsPerLaunch];   // n = elements of the WHOLE parameter
    int count;
    int total_chunks;                 // chunks of these `count` parameters
};
__global__ __launch_bounds__(256) void expand_chunks_kernel(const OptParamList L, OptChunk *__restrict__ chunks)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= L.total_chunks) return;
    int rem = c;
    bool found = false;
    OptChunk o = L.prm[0];
    for (int i = 0; i < L.count && !found; ++i) {  // every active lane is at the same i: L.prm[i] is a scalar load
        const OptChunk q = L.prm[i];
        const int nc = (q.n + kChunkElems - 1) / kChunkElems;
        if (rem < nc) {
            o = q;
            found = true;
        } else {
            rem -= nc;
        }
    }
    const long long off = (long long)rem * kChunkElems;
    OptChunk r;
    r.p = o.p + off;
    r.g = o.g + off;
    r.buf = o.buf + off;
    r.n = (int)min((long long)kChunkElems, (long long)o.n - off);
    r.lr = o.lr;
    chunks[c] = r;
}

}  // namespace mh

using namespace mh;

// one host-pinned, device-mapped counter per device (like the fault words of lstm.hip)
constexpr int kMaxSkipDevices = 64;
static unsigned *g_skip_words = nullptr;
static unsigned *skip_word_for_current_device()
{
    if (!g_skip_words) {
        void *p = nullptr;
        if (hipHostMalloc(&p, kMaxSkipDevices * sizeof(unsigned), hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess)
            return nullptr;
        for (int i = 0; i < kMaxSkipDevices; ++i) reinterpret_cast<unsigned *>(p)[i] = 0u;
        g_skip_words = reinterpret_cast<unsigned *>(p);
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxSkipDevices) return nullptr;
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, &g_skip_words[dev], 0) != hipSuccess) return nullptr;
    return reinterpret_cast<unsigned *>(dp);
}

extern "C" {

int mh_opt_chunk_elems(void) { return kChunkElems; }

int mh_opt_skipped_steps(void)
{
    const unsigned *w = g_skip_words;
    if (!w) return 0;
    long long n = 0;
    for (int i = 0; i < kMaxSkipDevices; ++i) n += __atomic_load_n(&w[i], __ATOMIC_RELAXED);
    return (int)std::min<long long>(n, 0x7fffffff);
}

int mh_opt_skipped_clear(void)
{
    unsigned *w = g_skip_words;
    if (w)
        for (int i = 0; i < kMaxSkipDevices; ++i) __atomic_store_n(&w[i], 0u, __ATOMIC_RELAXED);
    return MH_OK;
}

int mh_opt_build_chunks(const void *params_host, int nparams, void *chunks, int nchunks, void *stream)
{
    MH_REQUIRE(nparams >= 0 && nchunks >= 0);
    if (nparams == 0) return MH_OK;
    MH_REQUIRE(params_host && chunks);
    const OptChunk *prm = reinterpret_cast<const OptChunk *>(params_host);
    OptChunk *out = reinterpret_cast<OptChunk *>(chunks);
    long long done = 0;
    for (int i0 = 0; i0 < nparams; i0 += kParamsPerLaunch) {
        OptParamList L;
        L.count = std::min(kParamsPerLaunch, nparams - i0);
        long long tot = 0;
        for (int i = 0; i < L.count; ++i) {
            MH_REQUIRE(prm[i0 + i].n > 0 && prm[i0 + i].p && prm[i0 + i].g && prm[i0 + i].buf);
            L.prm[i] = prm[i0 + i];
            tot += (prm[i0 + i].n + kChunkElems - 1) / kChunkElems;
        }
        MH_REQUIRE(done + tot <= nchunks);
        L.total_chunks = (int)tot;
        hipLaunchKernelGGL(expand_chunks_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, as_stream(stream), L, out + done);
        int rc = check_launch("expand_chunks_kernel");
        if (rc) return rc;
        done += tot;
    }
    MH_REQUIRE(done == nchunks);
    return MH_OK;
}

int mh_multi_sumsq(const void *chunks, int nchunks, float *partial, float *sumsq_out, void *stream)
{
    MH_REQUIRE(nchunks >= 0 && sumsq_out);
    hipStream_t st = as_stream(stream);
    if (nchunks == 0) return (int)hipMemsetAsync(sumsq_out, 0, sizeof(float), st);
    MH_REQUIRE(chunks && partial);
    hipLaunchKernelGGL(multi_sumsq_kernel, dim3(nchunks), dim3(256), 0, st, reinterpret_cast<const OptChunk *>(chunks),
                       partial);
    int rc = check_launch("multi_sumsq_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, partial, nchunks, sumsq_out);
    return check_launch("sum_partials_kernel");
}

int mh_multi_sgd_step(const void *chunks, int nchunks, const float *sumsq, float max_norm, float momentum,
                      float weight_decay, int first_step, void *stream)
{
    MH_REQUIRE(nchunks >= 0);
    if (nchunks == 0) return MH_OK;
    MH_REQUIRE(chunks);
    unsigned *skip = skip_word_for_current_device();
    if (sumsq && !skip) { set_last_error("skip counter (hipHostMalloc / hipGetDevice)", hipErrorOutOfMemory); return (int)hipErrorOutOfMemory; }
    hipLaunchKernelGGL(multi_sgd_kernel, dim3(nchunks), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const OptChunk *>(chunks), sumsq, max_norm, momentum, weight_decay, first_step, skip);
    return check_launch("multi_sgd_kernel");
}

}  // extern "C"
