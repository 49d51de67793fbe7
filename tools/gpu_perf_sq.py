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
for name, M, N, K, ta, tb in [('sq4096',4096,4096,4096,0,1),('sq8192',8192,8192,4096,0,1),('fc6_wgrad',4096,25088,1536,1,0)]:
    a = torch.randn((K, M) if ta else (M, K), device='cuda'); b = torch.randn((N, K) if tb else (K, N), device='cuda'); out = torch.empty(M, N, device='cuda')
    ms = timeit(lambda: _hip.gemm(a, b, bool(ta), bool(tb), out=out, splitk=1), iters=5)
    print('%s GEMM %-10s %8.3f ms %7.2f TF/s' % (os.environ.get('TAG',''), name, ms, 2.0*M*N*K/ms/1e9), flush=True)
