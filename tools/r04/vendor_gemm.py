#!/usr/bin/env python3
"""Independent matrix-core ceiling (VERDICT r03 missing #3; test-only A/B allowed by SURVEY.md section 7): the vendor's f16 / bf16
GEMM (torch.matmul -> hipBLASLt / rocBLAS) on RANDOM data on THIS box, timed with HIP events next to our own plane GEMM on ready
images.  f16x3 issues 3 f16 MFMAs per fp32 product, so a vendor rate V TFLOP/s bounds our fp32-equivalent rate by V / 3.

    python tools/r04/vendor_gemm.py > gpurun_out/r04_vendor_gemm.jsonl
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'neural-motifs_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    from lib import _hip
    dev = torch.device('cuda', 0)
    for (M, N, K) in ((4096, 4096, 4096), (8192, 8192, 8192), (1536, 4096, 25088), (4096, 25088, 1536)):
        for dt in (torch.float16, torch.bfloat16):
            a = torch.randn(M, K, device=dev).to(dt)
            b = torch.randn(N, K, device=dev).to(dt)
            for zero in (False, True):
                if zero:
                    a.zero_(); b.zero_()
                ms = timed(lambda: torch.matmul(a, b.t()), 20)
                print(json.dumps({'engine': 'vendor (torch.matmul)', 'dtype': str(dt), 'M': M, 'N': N, 'K': K, 'data': 'zeros' if zero else 'randn',
                                  'ms': round(ms, 4), 'tflops': round(2.0 * M * N * K / ms * 1e-9, 1),
                                  'f16x3_equivalent_bound': round(2.0 * M * N * K / ms * 1e-9 / 3, 1)}), flush=True)
        a = torch.randn(M, K, device=dev)
        b = torch.randn(N, K, device=dev)
        ia, ib = _hip.make_planes(a, True), _hip.make_planes(b, True)
        out = torch.empty(M, N, device=dev)
        for shape in (-1, 0, 3, 4):
            _hip.lib().mh_debug_pl_shape(shape)
            ms = timed(lambda: _hip.gemm_planes(ia, ib, out=out), 10)
            print(json.dumps({'engine': 'plane GEMM on ready images', 'shape': shape, 'M': M, 'N': N, 'K': K, 'ms': round(ms, 4),
                              'tflops_fp32_equivalent': round(2.0 * M * N * K / ms * 1e-9, 1),
                              'f16_mfma_tflops': round(3 * 2.0 * M * N * K / ms * 1e-9, 1)}), flush=True)
        _hip.lib().mh_debug_pl_shape(-1)
        # fp32 vendor GEMM for reference (what `nn.Linear` in fp32 would run)
        ms = timed(lambda: torch.matmul(a, b.t()), 5)
        print(json.dumps({'engine': 'vendor (torch.matmul)', 'dtype': 'torch.float32', 'M': M, 'N': N, 'K': K, 'data': 'randn', 'ms': round(ms, 4),
                          'tflops': round(2.0 * M * N * K / ms * 1e-9, 1)}), flush=True)


if __name__ == '__main__':
    main()
