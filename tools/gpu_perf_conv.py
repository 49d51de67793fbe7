#!/usr/bin/env python3
"""Per-layer A/B of the trunk's conv kernels on the bench shapes (b = 6, 592x592): the fp32-activation kernel
(mh_conv3x3_nhwc) against the activation-plane kernel (mh_conv3x3_planes, optionally with the fused 2x2 pool), and the
fc6 GEMMs.  Environment switches are read once per process by the library, so run one process per variant:
    MH_CONV_SCHEDULE=uniform | MH_PCONV_TILE=128|256 | MH_GEMM_PATCH=rows
Prints TFLOP/s (fp32-equivalent) per layer and the weighted trunk total."""
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
out = {'env': {k: os.environ.get(k) for k in ('MH_CONV_SCHEDULE', 'MH_PCONV_TILE', 'MH_GEMM_PATCH')}}
tot = {'fp32': 0.0, 'planes': 0.0, 'fp32+pool': 0.0, 'planes_fused': 0.0}
for name, S, ci, co, pooled, mult in (LAYERS if which != 'gemm' else []):
    x = torch.randn(B, S, S, ci, device='cuda')
    wt = _hip.conv3x3_pack_weight(torch.randn(co, ci, 3, 3, device='cuda') * 0.05)
    bias = torch.randn(co, device='cuda')
    fl = 2.0 * B * S * S * ci * co * 9
    ms_f = timeit(lambda: _hip.conv3x3_nhwc(x, wt, bias, 1))
    y = _hip.conv3x3_nhwc(x, wt, bias, 1)
    ms_pool = timeit(lambda: _hip.maxpool2x2_nhwc(y)) if pooled else 0.0
    row = {'fp32_ms': ms_f, 'fp32_tf': fl / ms_f / 1e9, 'pool_ms': ms_pool}
    if _hip.planes_supported():
        xp = _hip.f32_to_planes(x)
        ms_p = timeit(lambda: _hip.conv3x3_planes(xp, wt, bias, 1, pool=False, out_fp32=False))
        row.update(planes_ms=ms_p, planes_tf=fl / ms_p / 1e9)
        if pooled and S % 2 == 0:
            ms_pp = timeit(lambda: _hip.conv3x3_planes(xp, wt, bias, 1, pool=True, out_fp32=False))
            row.update(planes_pool_ms=ms_pp, planes_pool_tf=fl / ms_pp / 1e9)
        else:
            ms_pp = ms_p
        tot['planes'] += ms_p * mult
        tot['planes_fused'] += ms_pp * mult
        del xp
    tot['fp32'] += ms_f * mult
    tot['fp32+pool'] += (ms_f + ms_pool) * mult
    row['schedule'] = _hip.conv3x3_schedule(B, S, S, ci, co)
    out[name] = row
    print('%-8s %4d %3d->%3d  fp32 %7.3f ms %6.1f TF | planes %7.3f ms %6.1f TF | planes+pool %s | pool kernel %.3f ms' % (
        name, S, ci, co, ms_f, row['fp32_tf'], row.get('planes_ms', 0), row.get('planes_tf', 0),
        ('%7.3f ms %6.1f TF' % (row['planes_pool_ms'], row['planes_pool_tf'])) if 'planes_pool_ms' in row else '   -   ', ms_pool), flush=True)
    del x, wt, y
trunk_flops = sum(2.0 * B * S * S * ci * co * 9 * m for _, S, ci, co, _, m in LAYERS)
for k, v in tot.items():
    if v:
        print('TRUNK %-13s %8.3f ms   %6.1f TF' % (k, v, trunk_flops / v / 1e9))
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
