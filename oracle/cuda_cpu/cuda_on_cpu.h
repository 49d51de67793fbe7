/* cuda_on_cpu.h -- TEST INFRASTRUCTURE ONLY (oracle/): a minimal serial CPU stand-in for the slice of the CUDA runtime /
 * cuBLAS API that the reference's three kernel files use, so that those files can be compiled FROM WHERE THEY LIE under
 * /root/reference (oracle/build_ref_cuda.py; outputs only into oracle/_ref/) and run here as the reference's own
 * arithmetic:
 *     lib/fpn/nms/src/cuda/nms_kernel.cu                       (devIoU, nms_kernel, ApplyNMSGPU incl. the host sweep)
 *     lib/fpn/roi_align/src/cuda/roi_align_kernel.cu           (ROIAlignForward / ROIAlignBackward + launchers)
 *     lib/lstm/highway_lstm_cuda/src/highway_lstm_kernel.cu    (elementWise_fp / _bp + both host loops over cuBLAS)
 * This header is our own code; no reference source is copied.  What it emulates and how:
 *   - a kernel launch `k<<<grid, block[, shmem, stream]>>>(args)` is rewritten by the build script into
 *     cuda_cpu::launch([&] { k(args); }, grid, block, ...), which runs the blocks one after another and, inside a block,
 *     the threads one after another in increasing threadIdx.x (blockIdx / threadIdx / blockDim / gridDim are globals);
 *   - __shared__ becomes `static`; __syncthreads() records that the block has a barrier and the block is then run a
 *     SECOND time: legal for kernels whose post-barrier code only reads shared memory written before the barrier and whose
 *     global writes are idempotent -- true for nms_kernel (the only kernel here with a barrier); kernels without a
 *     barrier run exactly once (ROIAlignBackward accumulates, so this matters);
 *   - atomicAdd(float*) is a plain read-modify-write: the serial thread order (increasing output index) fixes the
 *     summation order, the same order oracle/native_ops.c uses;
 *   - cudaMalloc / cudaMemcpy / cudaFree are malloc / memcpy / free; streams, events and device selection are no-ops;
 *   - cublasSgemm / cublasSgemv are column-major triple loops with fp32 accumulation in increasing k (cuBLAS' own
 *     summation order is unspecified: results are compared at a tolerance, not bit for bit);
 *   - min / max are the CUDA overloads for int and float (fminf / fmaxf semantics for float).
 * Compile with -ffp-contract=off: nvcc would contract a*b+c into FMAs at its own discretion; the pinned semantics are the
 * source's expression order without contraction (what oracle/native_ops.c and csrc/exact_ops.hip implement). */
#ifndef ORACLE_CUDA_ON_CPU_H
#define ORACLE_CUDA_ON_CPU_H
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static

struct uint3_cpu { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern uint3_cpu blockIdx, threadIdx;
extern dim3 blockDim, gridDim;
namespace cuda_cpu { extern bool saw_barrier; }
static inline void __syncthreads() { cuda_cpu::saw_barrier = true; }

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline float atomicAdd(float *p, float v) { const float old = *p; *p = old + v; return old; }

typedef int cudaError_t;
enum { cudaSuccess = 0 };
typedef void *cudaStream_t;
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };
static inline const char *cudaGetErrorString(cudaError_t) { return "cuda-on-cpu: no error"; }
static inline cudaError_t cudaGetLastError() { return cudaSuccess; }
static inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
static inline cudaError_t cudaGetDevice(int *d) { *d = 0; return cudaSuccess; }
static inline cudaError_t cudaSetDevice(int) { return cudaSuccess; }
static inline cudaError_t cudaStreamCreate(cudaStream_t *s) { *s = 0; return cudaSuccess; }
static inline cudaError_t cudaStreamDestroy(cudaStream_t) { return cudaSuccess; }
template <typename T>
static inline cudaError_t cudaMalloc(T **p, size_t n) { *p = (T *)malloc(n ? n : 1); return cudaSuccess; }
static inline cudaError_t cudaFree(void *p) { free(p); return cudaSuccess; }
static inline cudaError_t cudaMemcpy(void *d, const void *s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return cudaSuccess; }

typedef void *cublasHandle_t;
typedef int cublasStatus_t;
enum { CUBLAS_STATUS_SUCCESS = 0 };
enum cublasOperation_t { CUBLAS_OP_N = 0, CUBLAS_OP_T = 1 };
static inline cublasStatus_t cublasSetStream(cublasHandle_t, cudaStream_t) { return CUBLAS_STATUS_SUCCESS; }
/* column-major: C[m x n] = alpha * op(A)[m x k] * op(B)[k x n] + beta * C */
static inline cublasStatus_t cublasSgemm(cublasHandle_t, cublasOperation_t ta, cublasOperation_t tb, int m, int n, int k,
                                         const float *alpha, const float *A, int lda, const float *B, int ldb,
                                         const float *beta, float *C, int ldc)
{
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) {
            float acc = 0.f;
            for (int l = 0; l < k; ++l) {
                const float a = (ta == CUBLAS_OP_N) ? A[i + (size_t)l * lda] : A[l + (size_t)i * lda];
                const float b = (tb == CUBLAS_OP_N) ? B[l + (size_t)j * ldb] : B[j + (size_t)l * ldb];
                acc += a * b;
            }
            float *c = C + i + (size_t)j * ldc;
            *c = (*beta == 0.f) ? *alpha * acc : *alpha * acc + *beta * *c;
        }
    return CUBLAS_STATUS_SUCCESS;
}
/* y = alpha * op(A)[m x n] * x + beta * y (column-major A) */
static inline cublasStatus_t cublasSgemv(cublasHandle_t, cublasOperation_t ta, int m, int n, const float *alpha, const float *A,
                                         int lda, const float *x, int incx, const float *beta, float *y, int incy)
{
    const int rows = (ta == CUBLAS_OP_N) ? m : n, cols = (ta == CUBLAS_OP_N) ? n : m;
    for (int i = 0; i < rows; ++i) {
        float acc = 0.f;
        for (int l = 0; l < cols; ++l) acc += ((ta == CUBLAS_OP_N) ? A[i + (size_t)l * lda] : A[l + (size_t)i * lda]) * x[(size_t)l * incx];
        float *c = y + (size_t)i * incy;
        *c = (*beta == 0.f) ? *alpha * acc : *alpha * acc + *beta * *c;
    }
    return CUBLAS_STATUS_SUCCESS;
}

namespace cuda_cpu {
template <typename F>
static inline void launch(F body, dim3 grid, dim3 block, size_t = 0, cudaStream_t = 0)
{
    const dim3 saved_b = ::blockDim, saved_g = ::gridDim;
    ::blockDim = block;
    ::gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                ::blockIdx.x = bx; ::blockIdx.y = by; ::blockIdx.z = bz;
                saw_barrier = false;
                for (int pass = 0; pass < 2; ++pass) {
                    for (unsigned tz = 0; tz < block.z; ++tz)
                        for (unsigned ty = 0; ty < block.y; ++ty)
                            for (unsigned tx = 0; tx < block.x; ++tx) {
                                ::threadIdx.x = tx; ::threadIdx.y = ty; ::threadIdx.z = tz;
                                body();
                            }
                    if (!saw_barrier) break;      /* no barrier in this block: one pass is the whole execution */
                }
            }
    ::blockDim = saved_b;
    ::gridDim = saved_g;
}
}  // namespace cuda_cpu

#ifdef CUDA_ON_CPU_DEFINE_GLOBALS
uint3_cpu blockIdx, threadIdx;
dim3 blockDim, gridDim;
namespace cuda_cpu { bool saw_barrier = false; }
#endif
#endif
