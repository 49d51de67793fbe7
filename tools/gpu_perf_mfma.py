#!/usr/bin/env python3
"""quick A/B of the MFMA engine: a few GEMM + conv shapes (prints TF/s)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'neural-motifs_amd'))
from lib import _hip
def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
dev='cuda'
tot=0
for name, M, N, K, ta, tb in [('sq4096',4096,4096,4096,0,1),('fc6_fwd',1536,4096,25088,0,1),('fc6_dgrad',1536,25088,4096,0,0),('fc6_wgrad',4096,25088,1536,1,0),('tower_wgrad',512,2304,75264,1,0)]:
    a = torch.randn((K, M) if ta else (M, K), device=dev); b = torch.randn((N, K) if tb else (K, N), device=dev); out = torch.empty(M, N, device=dev)
    ms = timeit(lambda: _hip.gemm(a, b, bool(ta), bool(tb), out=out), iters=5)
    tot+=ms
    print('GEMM %-12s %8.3f ms %7.2f TF/s' % (name, ms, 2.0*M*N*K/ms/1e9), flush=True)
    del a,b,out
B=6
for (S, ci, co, mult) in [(592,64,64,1),(296,64,128,1),(296,128,128,1),(148,128,256,1),(148,256,256,2),(74,256,512,1),(74,512,512,2),(37,512,512,3)]:
    x = torch.randn(B, S, S, ci, device=dev); wt = _hip.conv3x3_pack_weight(torch.randn(co, ci, 3, 3, device=dev) * 0.05); bias = torch.randn(co, device=dev)
    ms = timeit(lambda: _hip.conv3x3_nhwc(x, wt, bias, 1), iters=5)
    tot+=ms*mult
    print('CONV %4d %3d->%3d %8.3f ms %7.2f TF/s' % (S, ci, co, ms, 2.0*B*S*S*ci*co*9/ms/1e9), flush=True)
    del x, wt
print('TOTAL weighted ms: %.2f' % tot)
