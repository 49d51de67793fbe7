"""
Detector pre-training driver -- the flow of the reference's models/train_detector.py (ObjectDetector in 'rpntrain'
mode, trunk trainable, loss = RoI class + RoI box + RPN class + RPN box, global grad-clip, SGD momentum 0.9) on the
MI355X implementation.  The step before the relation-model hot path: it produces the detector
checkpoint every train_rels run starts from, and it is the only consumer of the trunk's backward pass.

    python models/train_detector.py -b 6 -lr 1e-3 -ngpu 1 -nepoch 1 -clip 5 -max_iters 20
    torchrun --nproc-per-node 8 models/train_detector.py ... -ngpu 8      # one process per GPU, RCCL all-reduce

Per epoch, like the reference (train_detector.py:158-181, :222-233): validation detections of every rank are gathered,
rank 0 computes the COCO-protocol box mAP on the VG ground truth (lib/evaluation/det_map.py stands in for pycocotools),
mAP at IoU .5 drives ReduceLROnPlateau, and the checkpoint is saved in the reference's format
({'epoch', 'state_dict', 'optimizer'}).
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import pandas as pd
import torch

from torch.optim.lr_scheduler import ReduceLROnPlateau

from config import ModelConfig, BOX_SCALE, IM_SCALE
from dataloaders.visual_genome import VGDataLoader, VG
from lib import dist as D
from lib.detector_loss import detector_losses
from lib.evaluation.det_map import detection_rows, evaluate_bbox, summarize
from lib.object_detector import ObjectDetector
from lib.optim import FusedClipSGD
from lib.pytorch_misc import optimistic_restore, print_para

conf = ModelConfig()
rank, world, local_rank = D.init_from_env()
if world > 1 and conf.num_gpus != world:
    raise ValueError('-ngpu {} but WORLD_SIZE={}'.format(conf.num_gpus, world))
torch.cuda.set_device(local_rank)
np.random.seed(conf.seed + rank)
torch.manual_seed(conf.seed)

train, val, _ = VG.splits(num_val_im=conf.val_size, filter_non_overlap=False, filter_empty_rels=False,
                          use_proposals=conf.use_proposals, seed=conf.seed)
train_loader, val_loader = VGDataLoader.splits(train, val, mode='det', batch_size=conf.batch_size,
                                               num_workers=conf.num_workers, num_gpus=1, rank=rank, world_size=world)

detector = ObjectDetector(classes=train.ind_to_classes, num_gpus=1,
                          mode='rpntrain' if not conf.use_proposals else 'proposals', use_resnet=conf.use_resnet)
detector.cuda()
if conf.use_proposals:                      # "stanford" setup: lower layers frozen
    for n, param in detector.named_parameters():
        if n.startswith('features'):
            param.requires_grad = False
if rank == 0:
    print(print_para(detector), flush=True)

optimizer = FusedClipSGD([p for p in detector.parameters() if p.requires_grad], weight_decay=conf.l2,
                         lr=conf.lr * world * conf.batch_size, momentum=0.9)
scheduler = ReduceLROnPlateau(optimizer, 'max', patience=3, factor=0.1, threshold=0.001, threshold_mode='abs', cooldown=1)
reducer = D.OverlappedGradReducer([p for p in detector.parameters() if p.requires_grad])      # inert at world 1

start_epoch = -1
if conf.ckpt is not None and os.path.exists(conf.ckpt):
    ckpt = torch.load(conf.ckpt, map_location='cpu')
    if optimistic_restore(detector, ckpt['state_dict']):
        start_epoch = ckpt['epoch']


def train_batch(b):
    result = detector[b]
    if conf.use_proposals:
        losses = detector_losses(result)
    else:
        losses = detector_losses(result, b.train_anchor_labels, b.train_anchors)
    optimizer.zero_grad(set_to_none=True)
    reducer.prepare()
    losses['total'].backward()
    reducer.finish()
    optimizer.step(max_norm=conf.clip)
    return pd.Series({k: float(v) for k, v in losses.items()})


def train_epoch(epoch_num):
    detector.train()
    if hasattr(getattr(train_loader, 'sampler', None), 'set_epoch'):
        train_loader.sampler.set_epoch(epoch_num)        # reshuffle every epoch (reference: DataLoader(shuffle=True))
    tr, start = [], time.time()
    for b, batch in enumerate(train_loader):
        tr.append(train_batch(batch))
        if conf.max_iters and b + 1 >= conf.max_iters:
            break
        if b % conf.print_interval == 0 and b >= conf.print_interval and rank == 0:
            mn = pd.concat(tr[-conf.print_interval:], axis=1).mean(1)
            time_per_batch = (time.time() - start) / conf.print_interval
            print("\ne{:2d}b{:5d}/{:5d} {:.3f}s/batch, {:.1f}m/epoch".format(
                epoch_num, b, len(train_loader), time_per_batch, len(train_loader) * time_per_batch / 60))
            print(mn)
            print('-----------', flush=True)
            start = time.time()
    return pd.concat(tr, axis=1) if tr else pd.DataFrame()


def val_epoch():
    detector.eval()
    rows, n_batches = [], 0
    with torch.no_grad():
        for val_b, batch in enumerate(val_loader):
            if conf.max_iters and val_b >= conf.max_iters:
                break
            first_image = (val_b * world + rank) * conf.batch_size                     # _RankSampler's sharding
            rows.append(detection_rows(detector[batch], first_image, BOX_SCALE / IM_SCALE))
            n_batches += 1
    dets = np.concatenate(rows, 0) if rows else np.zeros((0, 7))
    if world > 1:
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, dets)
        dets = np.concatenate(gathered, 0)
    mAp = 0.0
    if rank == 0:
        if dets.shape[0] == 0:
            print("No detections anywhere")
        else:
            # the reference scores every image of `val` (its loader drops the last partial batch, those images count
            # as missed); with several ranks or a capped epoch only the images that were actually run are scored
            covered = n_batches * world * conf.batch_size
            img_ids = range(len(val)) if world == 1 and not conf.max_iters else range(min(covered, len(val)))
            stats = evaluate_bbox(val.coco, dets, img_ids)
            print(summarize(stats), flush=True)
            mAp = float(stats[1])
    if world > 1:
        t = torch.tensor([mAp], dtype=torch.float64, device='cuda')
        torch.distributed.broadcast(t, 0)
        mAp = float(t.item())
    return mAp


if __name__ == '__main__':
    if rank == 0:
        print("Training starts now!")
    from lib.pytorch_misc import quiet_gc
    quiet_gc()
    for epoch in range(start_epoch + 1, start_epoch + 1 + conf.num_epochs):
        rez = train_epoch(epoch)
        if rank == 0:
            print("overall{:2d}: ({:.3f})\n{}".format(epoch, rez.mean(1)['total'], rez.mean(1)), flush=True)
        mAp = val_epoch()
        quiet_gc()
        scheduler.step(mAp)
        if rank == 0 and conf.save_dir is not None:
            os.makedirs(conf.save_dir, exist_ok=True)
            torch.save({'epoch': epoch, 'state_dict': detector.state_dict(), 'optimizer': optimizer.state_dict()},
                       os.path.join(conf.save_dir, '{}-{}.tar'.format('vgdet', epoch)))
    if world > 1:
        torch.distributed.destroy_process_group()
