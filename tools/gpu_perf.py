#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels on the GPU box (writes gpurun_out/perf_ops.json).
    python tools/gpu_perf.py [--quick]
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'neural-motifs_amd'))
from lib import _hip  # noqa: E402

PEAK_TF = 157.3


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms


def main():
    quick = '--quick' in sys.argv
    res = {}
    dev = 'cuda'
    print('device:', torch.cuda.get_device_name(0), flush=True)

    # ---- GEMMs of cfg2 (b=6: 120 objects, 1536 relations)
    gemms = [
        ('fc6_rel_fwd   [1536x25088]x[4096x25088]^T', 1536, 4096, 25088, 0, 1),
        ('fc6_rel_dgrad [1536x4096]x[4096x25088]', 1536, 25088, 4096, 0, 0),
        ('fc6_rel_wgrad [1536x4096]^Tx[1536x25088]', 4096, 25088, 1536, 1, 0),
        ('fc7_rel_fwd   [1536x4096]x[4096x4096]^T', 1536, 4096, 4096, 0, 1),
        ('fc6_obj_fwd   [120x25088]x[4096x25088]^T', 120, 4096, 25088, 0, 1),
        ('fc6_obj_wgrad [120x4096]^Tx[120x25088]', 4096, 25088, 120, 1, 0),
        ('lstm_inproj   [120x4424]x[4424x3072]', 120, 3072, 4424, 0, 0),
        ('post_lstm     [120x512]x[8192x512]^T', 120, 8192, 512, 0, 1),
        ('rel_compress  [1536x4096]x[51x4096]^T', 1536, 51, 4096, 0, 1),
        ('square 4096', 4096, 4096, 4096, 0, 1),
    ]
    if quick:
        gemms = gemms[:2] + gemms[-1:]
    for name, M, N, K, ta, tb in gemms:
        a = torch.randn((K, M) if ta else (M, K), device=dev)
        b = torch.randn((N, K) if tb else (K, N), device=dev)
        out = torch.empty(M, N, device=dev)
        best = None
        sk_auto = _hip.lib().mh_gemm_auto_splitk(M, N, K)
        for sk in sorted(set([1, sk_auto])):
            ms = timeit(lambda: _hip.gemm(a, b, bool(ta), bool(tb), out=out, splitk=sk), iters=5 if M * N * K > 1e11 else 20)
            tf = 2.0 * M * N * K / ms / 1e9
            print('GEMM %-46s splitk=%2d  %8.3f ms  %7.2f TF/s  (%.1f%% of fp32 MFMA peak)' % (name, sk, ms, tf, 100 * tf / PEAK_TF), flush=True)
            if best is None or ms < best[0]:
                best = (ms, sk, tf)
        res['gemm:' + name] = dict(ms=best[0], splitk=best[1], tflops=best[2], auto_splitk=sk_auto)
        del a, b, out

    # ---- VGG16 trunk layers at 592x592, batch 6
    B = 2 if quick else 6
    layers = [(592, 64, 64), (296, 64, 128), (296, 128, 128), (148, 128, 256), (148, 256, 256), (74, 256, 512),
              (74, 512, 512), (37, 512, 512)]
    tot_ms, tot_fl = 0.0, 0.0
    for (S, ci, co) in layers:
        x = torch.randn(B, S, S, ci, device=dev)
        wt = _hip.conv3x3_pack_weight(torch.randn(co, ci, 3, 3, device=dev) * 0.05)
        bias = torch.randn(co, device=dev)
        ms = timeit(lambda: _hip.conv3x3_nhwc(x, wt, bias, 1), iters=5)
        fl = 2.0 * B * S * S * ci * co * 9
        tf = fl / ms / 1e9
        print('CONV3x3 %4dx%-4d %3d->%3d  B=%d  %8.3f ms  %7.2f TF/s (%.1f%%)' % (S, S, ci, co, B, ms, tf, 100 * tf / PEAK_TF), flush=True)
        res['conv:%d_%d_%d' % (S, ci, co)] = dict(ms=ms, tflops=tf, batch=B)
        mult = {(592, 64, 64): 1, (296, 64, 128): 1, (296, 128, 128): 1, (148, 128, 256): 1, (148, 256, 256): 2,
                (74, 256, 512): 1, (74, 512, 512): 2, (37, 512, 512): 3}[(S, ci, co)]
        tot_ms += ms * mult
        tot_fl += fl * mult
        del x, wt
    x = torch.randn(B, 3, 592, 592, device=dev)
    w = torch.randn(64, 3, 3, 3, device=dev)
    bias = torch.randn(64, device=dev)
    ms = timeit(lambda: _hip.conv_first_nchw(x, w, bias, 1), iters=5)
    print('CONV stem 3->64 592^2 B=%d  %8.3f ms  (%.2f GB/s written)' % (B, ms, B * 592 * 592 * 64 * 4 / ms / 1e6), flush=True)
    res['conv:stem'] = dict(ms=ms)
    y = torch.randn(B, 592, 592, 64, device=dev)
    ms = timeit(lambda: _hip.maxpool2x2_nhwc(y), iters=5)
    print('MAXPOOL 592^2x64 B=%d %8.3f ms  (%.2f GB/s)' % (B, ms, B * 592 * 592 * 64 * 4 * 1.25 / ms / 1e6), flush=True)
    res['pool:592'] = dict(ms=ms)
    print('TRUNK conv3x3 layers total: %.2f ms for B=%d -> %.1f TF/s (%.1f%% of peak)' % (tot_ms, B, tot_fl / tot_ms / 1e9, 100 * tot_fl / tot_ms / 1e9 / PEAK_TF), flush=True)
    res['trunk_conv3x3'] = dict(ms=tot_ms, tflops=tot_fl / tot_ms / 1e9, batch=B)
    del x, y

    # ---- RoIAlign (union boxes, cfg2) and NMS
    feat = torch.randn(6, 37, 37, 512, device=dev)
    rois = torch.rand(1536, 5, device=dev) * 500
    rois[:, 0] = torch.randint(0, 6, (1536,), device=dev).float()
    rois[:, 3:] = rois[:, 1:3] + 60
    ms = timeit(lambda: _hip.roi_align_fwd(feat, rois, 7, 7, 1 / 16, True))
    byts = 1536 * 512 * 49 * 4 + 6 * 512 * 37 * 37 * 4
    print('ROIALIGN nhwc N=1536 C=512: %.3f ms  %.1f GB/s algorithmic' % (ms, byts / ms / 1e6), flush=True)
    res['roialign:1536'] = dict(ms=ms, gbps=byts / ms / 1e6)
    for n in (1000, 6000):
        b = torch.rand(n, 4, device=dev) * 400
        b[:, 2:] = b[:, :2] + 30 + torch.rand(n, 2, device=dev) * 100
        ms = timeit(lambda: _hip.nms(b, 0.7))
        print('NMS n=%d: %.3f ms' % (n, ms), flush=True)
        res['nms:%d' % n] = dict(ms=ms)
    pairs = torch.rand(1536, 8, device=dev) * 300
    pairs[:, 2:4] += 300
    pairs[:, 6:8] += 300
    ms = timeit(lambda: _hip.draw_union_boxes(pairs, 27, -0.5, True))
    print('DRAW masks n=1536: %.3f ms' % ms, flush=True)
    res['draw:1536'] = dict(ms=ms)

    # ---- LSTM (obj ctx: T=20,B=6,in=4424,H=512,L=2)
    for name, ins, L_ in (('obj_ctx', 4424, 2), ('edge_ctx', 712, 2), ('edge_ctx_L4', 712, 4)):
        T, Bb, H = 20, 6, 512
        x = torch.randn(T, Bb, ins, device=dev)
        wtot = sum(6 * H * (ins if l == 0 else H) + 5 * H * H for l in range(L_))
        wgt = torch.randn(wtot, device=dev) * 0.02
        bias = torch.zeros(5 * H * L_, device=dev)
        drop = torch.ones(L_, Bb, H, device=dev)
        lengths = [T] * Bb
        ms_f = timeit(lambda: _hip.hwlstm_fwd(x, lengths, wgt, bias, drop, H, L_, True), iters=5)
        h, c, g = _hip.hwlstm_fwd(x, lengths, wgt, bias, drop, H, L_, True)
        go = torch.randn(T, Bb, H, device=dev)
        ms_b = timeit(lambda: _hip.hwlstm_bwd(go, x, lengths, wgt, drop, H, L_, h, c, g), iters=5)
        print('LSTM %-12s fwd %.3f ms (%.1f us/step)  bwd %.3f ms (%.1f us/step)' % (name, ms_f, 1e3 * ms_f / (T * L_), ms_b, 1e3 * ms_b / (T * L_)), flush=True)
        res['lstm:' + name] = dict(fwd_ms=ms_f, bwd_ms=ms_b)

    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'perf_ops.json'), 'w') as f:
        json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
