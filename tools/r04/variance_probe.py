#!/usr/bin/env python3
"""Round-4 step-time probe: ONE process, the bench's cfg2 step, UNPROFILED, many timed regions.

  * per-step HIP-event times of R regions x S steps for each setting, settings ALTERNATED region by region on one box:
        overlap (two HIP streams) x late_vr (union-box backward first)   -> 4 arms
  * host-side evidence per step: enqueue time, caching-allocator segment events (hipMalloc / hipFree), gc collections,
    FusedClipSGD run-ahead waits
  * prints one JSON object per region and a summary table; `--out` writes the lines to a file

    python tools/r04/variance_probe.py --regions 3 --steps 20 --out gpurun_out/r04_variance.jsonl
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'neural-motifs_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import bench as B  # noqa: E402


def pct(xs, q):
    return B.pct(xs, q)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--regions', type=int, default=3)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--out', default=None)
    ap.add_argument('--arms', default='11,01,10,00', help='overlap,late_vr bits per arm')
    args = ap.parse_args()
    from dataloaders.synthetic import SyntheticVG, make_blob
    from lib.optim import FusedClipSGD
    from lib.rel_model import RelModel
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    torch.manual_seed(1234)
    np.random.seed(1434)
    n_img = B.BATCH * 4
    ds = SyntheticVG(num_images=n_img, seed=1434, n_boxes=B.N_BOXES, n_rels=B.N_RELS)
    model = RelModel(classes=ds.ind_to_classes, rel_classes=ds.ind_to_predicates, mode='sgcls', num_gpus=1, **B.MODEL_KW)
    for _, p in model.detector.named_parameters():
        p.requires_grad = False
    model.to(dev).train()
    lr = 1e-3 * B.BATCH
    fc = [p for n, p in model.named_parameters() if n.startswith('roi_fmap') and p.requires_grad]
    rest = [p for n, p in model.named_parameters() if not n.startswith('roi_fmap') and p.requires_grad]
    opt = FusedClipSGD([{'params': fc, 'lr': lr / 10.0}, {'params': rest}], lr=lr, momentum=0.9, weight_decay=1e-4)
    blobs = [make_blob(ds, range(i * B.BATCH, (i + 1) * B.BATCH), is_train=True) for i in range(n_img // B.BATCH)]
    for b in blobs:
        b.scatter()

    def step(i):
        res = model[blobs[i % len(blobs)]]
        loss = F.cross_entropy(res.rm_obj_dists, res.rm_obj_labels) + F.cross_entropy(res.rel_dists, res.rel_labels[:, -1])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step(max_norm=5.0)
        return loss

    def seg():
        st = torch.cuda.memory_stats()
        return (st.get('segment.all.allocated', 0), st.get('segment.all.freed', 0), st.get('num_alloc_retries', 0))

    def region(overlap, late, label):
        model.overlap_streams = bool(overlap)
        model.late_vr_backward = 'auto' if late else '0'
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        host, segs, gcs = [], [], []
        g0 = gc.get_stats()
        t0 = time.perf_counter()
        for i in range(args.steps):
            ev[i].record()
            s0 = seg()
            t1 = time.perf_counter()
            step(args.warmup + i)
            host.append(1e3 * (time.perf_counter() - t1))
            segs.append(tuple(b - a for a, b in zip(s0, seg())))
        ev[args.steps].record()
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t0) / args.steps
        g1 = gc.get_stats()
        gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
        return {'arm': label, 'overlap': overlap, 'late_vr': late, 'wall_ms_per_step': round(wall, 3), 'img_s': round(1e3 * B.BATCH / wall, 1),
                'gpu_p50': round(pct(gpu, .5), 3), 'gpu_p90': round(pct(gpu, .9), 3), 'gpu_max': round(max(gpu), 3), 'gpu_min': round(min(gpu), 3),
                'host_p50': round(pct(host, .5), 3), 'host_max': round(max(host), 3),
                'gpu_ms': [round(x, 2) for x in gpu], 'host_ms': [round(x, 2) for x in host],
                'alloc_events': [s for s in segs if any(s)],
                'gc_collections': [b['collections'] - a['collections'] for a, b in zip(g0, g1)]}

    arms = [(int(a[0]), int(a[1])) for a in args.arms.split(',')]
    lines = []
    for r in range(args.regions):
        for ov, late in arms:
            d = region(ov, late, 'ov%d_late%d' % (ov, late))
            d['region'] = r
            lines.append(d)
            print(json.dumps(d), flush=True)
    # gc off, shipped arm: does Python's cyclic collector show up?
    gc.disable()
    for r in range(2):
        d = region(arms[0][0], arms[0][1], 'ov%d_late%d_gc_off' % arms[0])
        d['region'] = r
        lines.append(d)
        print(json.dumps(d), flush=True)
    gc.enable()
    summary = {}
    for d in lines:
        summary.setdefault(d['arm'], []).append(d['wall_ms_per_step'])
    table = {k: {'wall_ms': v, 'mean': round(sum(v) / len(v), 3), 'spread_pct': round(100 * (max(v) - min(v)) / (sum(v) / len(v)), 2)} for k, v in summary.items()}
    print(json.dumps({'summary': table}), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, 'w') as f:
            for d in lines:
                f.write(json.dumps(d) + '\n')
            f.write(json.dumps({'summary': table}) + '\n')


if __name__ == '__main__':
    main()
