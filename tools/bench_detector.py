#!/usr/bin/env python3
"""Throughput of the detector pre-training step (models/train_detector.py: ObjectDetector in
'rpntrain' mode, trainable VGG16 trunk, 4 losses, clip + SGD) on one GPU, synthetic 592x592 images.  Prints one JSON
line.  (bench.py stays the headline SGCls benchmark; this is the secondary measurement of the widened scope.)"""
import argparse, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'neural-motifs_amd'))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=6)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    args = ap.parse_args()
    from dataloaders.synthetic import SyntheticVG, make_blob
    from lib.detector_loss import detector_losses
    from lib.object_detector import ObjectDetector
    from lib.optim import FusedClipSGD
    torch.manual_seed(0); np.random.seed(0)
    ds = SyntheticVG(num_images=args.batch * 2, seed=77, n_boxes=20, n_rels=4)
    det = ObjectDetector(classes=ds.ind_to_classes, mode='rpntrain').cuda().train()
    opt = FusedClipSGD([p for p in det.parameters() if p.requires_grad], lr=1e-3 * args.batch, momentum=0.9, weight_decay=1e-4)
    blobs = [make_blob(ds, range(i * args.batch, (i + 1) * args.batch), is_train=True, mode='det') for i in range(2)]
    for b in blobs:
        b.scatter()
    def step(i):
        b = blobs[i % 2]
        res = det[b]
        losses = detector_losses(res, b.train_anchor_labels, b.train_anchors)
        opt.zero_grad(set_to_none=True)
        losses['total'].backward()
        opt.step(max_norm=5.0)
        return losses['total']
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(json.dumps({'metric': 'images/sec detector pre-training step (VGG16 rpntrain, fwd+bwd+clip+SGD)',
                      'value': args.batch * args.steps / dt, 'unit': 'img/s', 'ms_per_step': 1e3 * dt / args.steps,
                      'batch': args.batch, 'steps': args.steps, 'final_loss': float(loss),
                      'peak_mem_gb': torch.cuda.max_memory_allocated() / 2 ** 30}))

if __name__ == '__main__':
    main()
