// common.h -- shared helpers for the gfx950 kernels of libmotifs_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/motifs_hip.h"

namespace mh {

// thread-local record of the last launch failure (mh_last_error)
void set_last_error(const char *what, hipError_t e);

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error(what, e);
        return (int)e;
    }
    return MH_OK;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

template <typename T>
__host__ __device__ inline T ceil_div(T a, T b) { return (a + b - 1) / b; }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

constexpr int kWave = 64;  // CDNA wavefront

// the small-product engine (gemm.hip; include/motifs_hip.h: mh_gemm_small_f32) and the library-internal form of mh_gemm_f32
// (pl_gemm.hip) that hands split-K arrival counters (zero on entry, left zero) to it when the product is a small one
constexpr int kGemmCounters = 4096;
int gemm_small(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
               const float *bias, int epilogue, int accumulate, void *workspace, size_t ws_bytes, int *counters, int n_counters,
               void *stream);
int gemm_f32_ctr(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
                 const float *bias, int epilogue, int accumulate, void *workspace, size_t ws_bytes, int *counters, int n_counters,
                 void *stream);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

}  // namespace mh

#define MH_REQUIRE(cond)                                  \
    do {                                                  \
        if (!(cond)) {                                    \
            mh::set_last_error("bad argument: " #cond, hipErrorInvalidValue); \
            return MH_EINVAL;                             \
        }                                                 \
    } while (0)
