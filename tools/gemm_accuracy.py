#!/usr/bin/env python3
"""Error of the MFMA GEMM against an fp64 reference (run on the GPU box).  Prints max / rms error relative to the
rms magnitude of the result, for the library as built (MH_MFMA_SPLIT=0: f32-input MFMA, 6: bf16x6, 3: bf16x3) and for
torch's own fp32 matmul (rocBLAS) as a yardstick."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'neural-motifs_amd'))
from lib import _hip
torch.manual_seed(0)
for (M, N, K, scale) in [(512, 512, 4096, 1.0), (256, 1024, 25088, 1.0), (512, 512, 4096, 1e3)]:
    a = (torch.randn(M, K, device='cuda') * scale); b = torch.randn(N, K, device='cuda')
    a[:, ::7] *= 1e-3; b[:, ::5] *= 1e4                      # wide dynamic range inside a row
    ref = a.double() @ b.double().t()
    rms = ref.pow(2).mean().sqrt().item()
    for name, out in (('mh_gemm', _hip.gemm(a, b, False, True)), ('torch fp32 matmul', a @ b.t())):
        err = (out.double() - ref)
        print('%-18s M%d N%d K%d scale %g: max err / rms(ref) = %.3e   rms err / rms(ref) = %.3e' % (
            name, M, N, K, scale, err.abs().max().item() / rms, err.pow(2).mean().sqrt().item() / rms))
