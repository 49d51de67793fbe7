/* stub for oracle/build_ref_cuda.py: the API slice lives in cuda_on_cpu.h (TEST INFRASTRUCTURE ONLY) */
#include "cuda_on_cpu.h"
