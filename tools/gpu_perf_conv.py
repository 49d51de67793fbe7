#!/usr/bin/env python3
"""Per-layer timing of the trunk's conv kernel on the bench shapes (b = 6, 592x592) and of the fc6 / fc7 GEMMs.
Environment switches are read once per process by the library, so run one process per variant:
    MH_CONV_SCHEDULE=uniform | MH_SLOTS=768 | MH_GEMM_PATCH=rows | MOTIFS_HIP_LIB=<variant .so>
Prints TFLOP/s (fp32-equivalent) per layer and the weighted trunk total, then one JSON line."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'neural-motifs_amd'))
from lib import _hip


def timeit(fn, iters=8, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


B = 6
LAYERS = [('conv1_2', 592, 64, 64, True, 1), ('conv2_1', 296, 64, 128, False, 1), ('conv2_2', 296, 128, 128, True, 1),
          ('conv3_1', 148, 128, 256, False, 1), ('conv3_2', 148, 256, 256, False, 1), ('conv3_3', 148, 256, 256, True, 1),
          ('conv4_1', 74, 256, 512, False, 1), ('conv4_2', 74, 512, 512, False, 1), ('conv4_3', 74, 512, 512, True, 1),
          ('conv5_x', 37, 512, 512, False, 3)]
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
out = {'env': {k: os.environ.get(k) for k in ('MH_CONV_SCHEDULE', 'MH_SLOTS', 'MH_GEMM_PATCH', 'MOTIFS_HIP_LIB')},
       'mfma_split': _hip.lib().mh_mfma_split(), 'f16': _hip.lib().mh_split_f16()}
tot = {'conv': 0.0, 'conv+pool': 0.0}
for name, S, ci, co, pooled, mult in (LAYERS if which != 'gemm' else []):
    x = torch.randn(B, S, S, ci, device='cuda')
    wt = _hip.conv3x3_pack_weight(torch.randn(co, ci, 3, 3, device='cuda') * 0.05)
    bias = torch.randn(co, device='cuda')
    fl = 2.0 * B * S * S * ci * co * 9
    ms_f = timeit(lambda: _hip.conv3x3_nhwc(x, wt, bias, 1))
    y = _hip.conv3x3_nhwc(x, wt, bias, 1)
    ms_pool = timeit(lambda: _hip.maxpool2x2_nhwc(y)) if pooled else 0.0
    tot['conv'] += ms_f * mult
    tot['conv+pool'] += (ms_f + ms_pool) * mult
    out[name] = {'ms': ms_f, 'tf': fl / ms_f / 1e9, 'pool_ms': ms_pool, 'schedule': _hip.conv3x3_schedule(B, S, S, ci, co)}
    print('%-8s %4d %3d->%3d  %7.3f ms %6.1f TF | pool kernel %.3f ms' % (name, S, ci, co, ms_f, fl / ms_f / 1e9, ms_pool), flush=True)
    del x, wt, y
trunk_flops = sum(2.0 * B * S * S * ci * co * 9 * m for _, S, ci, co, _, m in LAYERS)
for k, v in tot.items():
    if v:
        print('TRUNK %-10s %8.3f ms   %6.1f TF' % (k, v, trunk_flops / v / 1e9))
out['trunk_ms'] = tot
if which in ('all', 'gemm'):
    for name, M, N, K, ta, tb in [('fc6_fwd', 1536, 4096, 25088, 0, 1), ('fc6_dgrad', 1536, 25088, 4096, 0, 0),
                                  ('fc6_wgrad', 4096, 25088, 1536, 1, 0), ('fc7_fwd', 1536, 4096, 4096, 0, 1),
                                  ('fc6_obj_fwd', 120, 4096, 25088, 0, 1)]:
        a = torch.randn((K, M) if ta else (M, K), device='cuda')
        b = torch.randn((N, K) if tb else (K, N), device='cuda')
        o = torch.empty(M, N, device='cuda')
        ms = timeit(lambda: _hip.gemm(a, b, bool(ta), bool(tb), out=o), iters=5)
        out[name] = {'ms': ms, 'tf': 2.0 * M * N * K / ms / 1e9}
        print('GEMM %-12s %8.3f ms %7.2f TF/s' % (name, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
        del a, b, o
print(json.dumps(out))
