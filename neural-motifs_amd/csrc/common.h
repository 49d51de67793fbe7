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
