// LDS-DMA probe (buffer_load_dwordx4 ... lds) on gfx950: what the activation-plane conv kernel assumed, checked one
// assumption at a time.  Prints one JSON line per experiment.
//   hipcc --offload-arch=gfx950 -O2 tools/dma_probe.cpp -o tools/_bin/dma_probe && tools/_bin/dma_probe
//  E1  linear image: lane l of instruction j lands at m0_base + j*1024 + 16*l (per-lane SOURCE offsets arbitrary)
//  E2  out-of-range voffset (>= num_records): does the DMA write zeros, or leave the LDS bytes untouched?
//  E3  soffset participates in the address but NOT in the range check?
//  E4  LDS destinations beyond 64 KiB (M0 wider than 16 bits?)
//  E5  s_waitcnt vmcnt(0) + s_barrier is enough to read the data from another wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void *lds_ptr_t;
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

struct Args {
    const unsigned *src;       // n dwords, src[i] = i + 1
    unsigned num_records;      // bytes
    unsigned *out;             // dump of the LDS region [nbytes / 4]
    unsigned lds_base;         // byte offset of the destination region inside the dynamic LDS
    unsigned nbytes;           // region size (multiple of 4096)
    unsigned oob_every;        // lanes with (lane % oob_every == 1) use an out-of-range voffset (0 = none)
    unsigned soffset;          // scalar offset added to every load
    unsigned voff_bias;        // subtracted from the per-lane offset (so that voffset + soffset is the intended address)
    unsigned fill;             // LDS pre-fill pattern
    unsigned reverse;          // source permutation: lane l reads chunk (63 - l) of its kilobyte
};

__global__ __launch_bounds__(256) void probe(const Args a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned *l32 = reinterpret_cast<unsigned *>(lds);
    for (unsigned i = tid; i < (a.lds_base + a.nbytes) / 4; i += 256) l32[i] = a.fill;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(a.src), 0, (int)a.num_records, 0x00020000);
    const int ninst = a.nbytes / 4096;              // 4 waves x 1 KiB per instruction index
    for (int j = 0; j < ninst; ++j) {
        const unsigned chunk = 256u * j + 64u * wave + (a.reverse ? 63u - lane : lane);
        unsigned voff = chunk * 16u - a.voff_bias;
        if (a.oob_every && (lane % a.oob_every) == 1) voff = 0x80000000u;
        char *dst = lds + a.lds_base + j * 4096 + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)dst, 16, (int)voff, (int)a.soffset, 0, 0);
    }
    __syncthreads();                                // vmcnt(0) + barrier
    // every thread dumps words written by OTHER waves too
    for (unsigned i = tid; i < a.nbytes / 4; i += 256) a.out[i] = l32[a.lds_base / 4 + i];
}

static void run(const char *name, Args a, size_t lds_bytes, const unsigned *dsrc, unsigned *dout)
{
    a.src = dsrc; a.out = dout;
    HIP_OK(hipMemset(dout, 0xEE, a.nbytes));
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), lds_bytes, 0, a);
    HIP_OK(hipDeviceSynchronize());
    std::vector<unsigned> h(a.nbytes / 4);
    HIP_OK(hipMemcpy(h.data(), dout, a.nbytes, hipMemcpyDeviceToHost));
    long ok = 0, zero = 0, fill = 0, other = 0;
    for (unsigned c = 0; c < a.nbytes / 16; ++c) {
        const unsigned j = c / 256, w = (c / 64) % 4, l = c % 64;
        const unsigned src_chunk = 256u * j + 64u * w + (a.reverse ? 63u - l : l);
        const bool oob = a.oob_every && (l % a.oob_every) == 1;
        for (int k = 0; k < 4; ++k) {
            const unsigned v = h[c * 4 + k], want = src_chunk * 4 + k + 1;
            if (!oob && v == want) ++ok;
            else if (v == 0) ++zero;
            else if (v == a.fill) ++fill;
            else ++other;
        }
    }
    printf("{\"experiment\": \"%s\", \"dwords\": %u, \"as_intended\": %ld, \"zero\": %ld, \"untouched_fill\": %ld, \"other\": %ld}\n",
           name, a.nbytes / 4, ok, zero, fill, other);
}

int main()
{
    const unsigned n = 1u << 20;                     // 4 MiB of source
    std::vector<unsigned> src(n);
    for (unsigned i = 0; i < n; ++i) src[i] = i + 1;
    unsigned *dsrc, *dout;
    HIP_OK(hipMalloc(&dsrc, n * 4)); HIP_OK(hipMalloc(&dout, 160 * 1024));
    HIP_OK(hipMemcpy(dsrc, src.data(), n * 4, hipMemcpyHostToDevice));
    Args a = {};
    a.num_records = n * 4; a.nbytes = 16384; a.fill = 0xABABABABu;
    run("E1 linear image, identity source order", a, 65536, dsrc, dout);
    a.reverse = 1;
    run("E1b linear image, reversed source order inside each KiB", a, 65536, dsrc, dout);
    a.reverse = 0; a.oob_every = 4;
    run("E2 every 4th lane out of range (voffset 0x80000000): zero or untouched?", a, 65536, dsrc, dout);
    a.oob_every = 0; a.num_records = 8192;           // only the first 8 KiB are in range
    run("E2b num_records = 8 KiB of a 16 KiB tile (upper half out of range)", a, 65536, dsrc, dout);
    a.num_records = n * 4; a.soffset = 1u << 20; a.voff_bias = 1u << 20;   // voffset wraps negative, soffset brings it back
    run("E3 voffset = intended - 1 MiB (wrapped), soffset = +1 MiB", a, 65536, dsrc, dout);
    a.soffset = 4096; a.voff_bias = 4096; a.num_records = 16384;          // voff max = 12 KiB-16 in range; address up to 16 KiB
    run("E3b num_records = tile size, soffset = 4096, voffset = intended - 4096 (first KiBs wrap)", a, 65536, dsrc, dout);
    a.soffset = 0; a.voff_bias = 0; a.num_records = n * 4;
    a.lds_base = 96 * 1024;
    run("E4 destination at LDS byte 96 KiB", a, 160 * 1024 - 4096, dsrc, dout);
    a.lds_base = 60 * 1024;
    run("E4b destination straddling 64 KiB", a, 160 * 1024 - 4096, dsrc, dout);
    return 0;
}
