"""
Detector pre-training driver -- the flow of the reference's models/train_detector.py (ObjectDetector in 'rpntrain'
mode, trunk trainable, loss = RoI class + RoI box + RPN class + RPN box, global grad-clip, SGD momentum 0.9) on the
MI355X implementation.  The step before the relation-model hot path: it produces the detector
checkpoint every train_rels run starts from, and it is the only consumer of the trunk's backward pass.

    python models/train_detector.py -b 6 -lr 1e-3 -ngpu 1 -nepoch 1 -clip 5 -max_iters 20
    torchrun --nproc-per-node 8 models/train_detector.py ... -ngpu 8      # one process per GPU, RCCL all-reduce

Validation by COCO mAP (pycocotools) is not part of this environment: per epoch the driver reports the training
losses and saves the checkpoint in the reference's format ({'epoch', 'state_dict', 'optimizer'}).
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import pandas as pd
import torch

from config import ModelConfig
from dataloaders.visual_genome import VGDataLoader, VG
from lib import dist as D
from lib.detector_loss import detector_losses
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
    return pd.concat(tr, axis=1)


if __name__ == '__main__':
    if rank == 0:
        print("Training starts now!")
    for epoch in range(start_epoch + 1, start_epoch + 1 + conf.num_epochs):
        rez = train_epoch(epoch)
        if rank == 0:
            print("overall{:2d}: ({:.3f})\n{}".format(epoch, rez.mean(1)['total'], rez.mean(1)), flush=True)
            if conf.save_dir is not None:
                os.makedirs(conf.save_dir, exist_ok=True)
                torch.save({'epoch': epoch, 'state_dict': detector.state_dict(), 'optimizer': optimizer.state_dict()},
                           os.path.join(conf.save_dir, '{}-{}.tar'.format('vgdet', epoch)))
    if world > 1:
        torch.distributed.destroy_process_group()
