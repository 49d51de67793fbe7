"""
Relation-model training driver -- the flow of the reference's models/train_rels.py (build RelModel, freeze the
detector, SGD with 1/10 learning rate on the fc layers, loss = CE(objects) + CE(relations), global grad-clip,
per-epoch checkpoint + Recall@K validation, LR-on-plateau) on the MI355X implementation.

    python models/train_rels.py -m sgcls -model motifnet -order leftright -nl_obj 2 -nl_edge 2 -b 6 -clip 5 \
        -hidden_dim 512 -pooling_dim 4096 -lr 1e-3 -ngpu 1 -use_bias -nepoch 1 -max_iters 20
    torchrun --nproc-per-node 8 models/train_rels.py ... -ngpu 8        # one process per GPU, RCCL all-reduce
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import pandas as pd
import torch
from torch import optim
from torch.nn import functional as F
from torch.optim.lr_scheduler import ReduceLROnPlateau

from config import ModelConfig, BOX_SCALE, IM_SCALE
from dataloaders.visual_genome import VGDataLoader, VG
from lib import dist as D
from lib.evaluation.sg_eval import BasicSceneGraphEvaluator
from lib.optim import FusedClipSGD
from lib.losses import relation_losses
from lib.pytorch_misc import restore_rel_checkpoint, clip_grad_norm, print_para, quiet_gc, with_ahead

conf = ModelConfig()
if conf.model == 'motifnet':
    from lib.rel_model import RelModel
elif conf.model == 'stanford':              # message-passing baseline (reference models/train_rels.py:22-27)
    from lib.rel_model_stanford import RelModelStanford as RelModel
else:
    raise ValueError('unknown model %r' % conf.model)

rank, world, local_rank = D.init_from_env()
if world > 1 and conf.num_gpus != world:
    raise ValueError('-ngpu {} but WORLD_SIZE={}'.format(conf.num_gpus, world))
torch.cuda.set_device(local_rank)
np.random.seed(conf.seed + rank)
torch.manual_seed(conf.seed)

train, val, _ = VG.splits(num_val_im=conf.val_size, filter_duplicate_rels=True, use_proposals=conf.use_proposals,
                          filter_non_overlap=conf.mode == 'sgdet', seed=conf.seed)
train_loader, val_loader = VGDataLoader.splits(train, val, mode='rel', batch_size=conf.batch_size,
                                               num_workers=conf.num_workers, num_gpus=1, rank=rank, world_size=world)

# FrequencyBias statistics (reference lib/sparse_targets.py:20: get_counts over the VG training split without duplicate
# filtering -- what the constructor does itself when the VG files are here); the synthetic stand-in is scanned directly
freq_counts = None
if conf.use_bias and not isinstance(train, VG):
    from lib.get_dataset_counts import get_counts
    freq_counts = get_counts(train, must_overlap=True)
detector = RelModel(classes=train.ind_to_classes, rel_classes=train.ind_to_predicates, num_gpus=1, mode=conf.mode,
                    require_overlap_det=True, use_resnet=conf.use_resnet, order=conf.order, nl_edge=conf.nl_edge,
                    nl_obj=conf.nl_obj, hidden_dim=conf.hidden_dim, use_proposals=conf.use_proposals,
                    pass_in_obj_feats_to_decoder=conf.pass_in_obj_feats_to_decoder,
                    pass_in_obj_feats_to_edge=conf.pass_in_obj_feats_to_edge, pooling_dim=conf.pooling_dim,
                    rec_dropout=conf.rec_dropout, use_bias=conf.use_bias, use_tanh=conf.use_tanh,
                    limit_vision=conf.limit_vision,
                    freq_counts=freq_counts)

for n, param in detector.detector.named_parameters():      # freeze the detector
    param.requires_grad = False
if rank == 0:
    print(print_para(detector), flush=True)


def get_optim(lr):
    fc_params = [p for n, p in detector.named_parameters() if n.startswith('roi_fmap') and p.requires_grad]
    non_fc_params = [p for n, p in detector.named_parameters() if not n.startswith('roi_fmap') and p.requires_grad]
    params = [{'params': fc_params, 'lr': lr / 10.0}, {'params': non_fc_params}]
    if conf.adam:
        optimizer = optim.Adam(params, weight_decay=conf.l2, lr=lr, eps=1e-3)
    else:
        optimizer = FusedClipSGD(params, weight_decay=conf.l2, lr=lr, momentum=0.9)      # clip + SGD fused on the GPU
    scheduler = ReduceLROnPlateau(optimizer, 'max', patience=3, factor=0.1, threshold=0.0001, threshold_mode='abs',
                                  cooldown=1)
    return optimizer, scheduler


start_epoch = -1
if conf.ckpt is not None:
    start_epoch = restore_rel_checkpoint(detector, torch.load(conf.ckpt, map_location='cpu'), conf.ckpt)

detector.cuda()
reducer = D.OverlappedGradReducer([p for p in detector.parameters() if p.requires_grad])   # inert at world 1


# SGDet: the frozen detector stage of the next batches (RPN -> NMS -> RoI head -> per-class NMS -> GT matching: where the host
# waits for the device) runs ahead on its own thread and stream (lib/rel_model.py: RelModel.detect_ahead; bench.py cfg3: 217-223 img/s
# against 160 in line).  GT-box modes have no waits in their detector stage: in line.  MOTIFS_DETECT_AHEAD=0 turns it off, =N keeps
# N batches in flight
DETECT_AHEAD = int(os.environ.get('MOTIFS_DETECT_AHEAD', '2')) if conf.mode == 'sgdet' else 0
if rank == 0:
    print('detector stage: %s' % ('%d batch(es) ahead of the step' % DETECT_AHEAD if DETECT_AHEAD else 'in line'), flush=True)


def train_batch(b, verbose=False, start_ahead=(), upload_ahead=()):
    result = detector[b]
    for nb in start_ahead:
        detector.detect_ahead_blob(nb)
    for nb in upload_ahead:          # the next batch's host -> HBM copies start now, on the copy stream (dataloaders/blob.py: Blob.prefetch)
        if hasattr(nb, 'prefetch'):
            nb.prefetch()
    ls = relation_losses(result)          # [class_loss, rel_loss] (reference :140-141) as one autograd node: lib/losses.py
    if world > 1:      # global-mean loss semantics of the single-process reference
        w = D.global_row_weights([result.rm_obj_labels.shape[0], result.rel_labels.shape[0]], ls.device)
        loss = (ls * w).sum()
    else:
        loss = ls.sum()
    optimizer.zero_grad(set_to_none=True)
    reducer.prepare()
    loss.backward()          # world > 1: gradient buckets are all-reduced (RCCL) while backward is still running
    reducer.finish()
    if isinstance(optimizer, FusedClipSGD):
        optimizer.step(max_norm=conf.clip)
        if verbose and rank == 0:
            print('---Total norm {:.3f}'.format(optimizer.last_total_norm()), flush=True)
    else:
        clip_grad_norm([(n, p) for n, p in detector.named_parameters() if p.grad is not None], max_norm=conf.clip,
                       verbose=verbose and rank == 0, clip=True)
        optimizer.step()
    # the three losses stay on the device: reading them here (the reference's `.data[0]`, :150) would drain the GPU queue
    # every step and serialise the host's launch work of the next step with this step's kernels; train_epoch reads a whole
    # print interval at once
    return torch.cat((ls.detach(), ls.detach().sum()[None]))        # [class_loss, rel_loss, total]


LOSS_KEYS = ('class_loss', 'rel_loss', 'total')


def _loss_frame(steps):
    """[3] device tensors of some steps -> DataFrame (rows = LOSS_KEYS, one column per step): ONE device->host copy"""
    return pd.DataFrame(torch.stack(steps, 1).cpu().numpy(), index=LOSS_KEYS)


def _device_health():
    """after a synchronisation point: raise if a persistent launch timed out or the optimizer skipped a step on the device
    (lib/_hip.py: check_faults / check_skipped_steps) -- nothing derived from such a run may be saved"""
    from lib import _hip
    if torch.cuda.is_available():
        _hip.check_faults()
        _hip.check_skipped_steps()


def train_epoch(epoch_num):
    detector.train()
    if hasattr(getattr(train_loader, 'sampler', None), 'set_epoch'):
        train_loader.sampler.set_epoch(epoch_num)        # reshuffle every epoch (reference: DataLoader(shuffle=True))
    # losses of the steps since the last print stay on the device (no per-step sync); every print interval they move to the
    # host in ONE copy and the device tensors are dropped (an epoch-long list of 512-byte blocks fragments the allocator)
    pending, frames, start = [], [], time.time()
    for b, (batch, following) in enumerate(with_ahead(train_loader, max(DETECT_AHEAD, 1))):
        if conf.max_iters and b >= conf.max_iters:
            break
        pending.append(train_batch(batch, verbose=b % (conf.print_interval * 10) == 0, start_ahead=following if DETECT_AHEAD else (),
                                   upload_ahead=() if DETECT_AHEAD else following))
        if b % conf.print_interval == 0 and b >= conf.print_interval:
            frames.append(_loss_frame(pending))
            pending = []
            _device_health()                               # the copy above synchronised: faults of those steps are visible
            if rank == 0:
                mn = frames[-1].iloc[:, -conf.print_interval:].mean(1)
                tpb = (time.time() - start) / conf.print_interval
                print("\ne{:2d}b{:5d}/{:5d} {:.3f}s/batch, {:.1f}m/epoch".format(epoch_num, b, len(train_loader), tpb,
                                                                                 len(train_loader) * tpb / 60))
                print(mn)
                print('-----------', flush=True)
            start = time.time()
    if pending:
        frames.append(_loss_frame(pending))
    detector.ahead_discard()                               # a stage started for a batch the loop did not reach (max_iters)
    _device_health()
    return pd.concat(frames, axis=1, ignore_index=True) if frames else pd.DataFrame()      # a loader that yields no batch


def val_batch(batch_num, b, evaluator):
    det_res = [detector[b]]
    for i, (boxes_i, objs_i, obj_scores_i, rels_i, pred_scores_i) in enumerate(det_res):
        gt_entry = {'gt_classes': val.gt_classes[batch_num + i].copy(),
                    'gt_relations': val.relationships[batch_num + i].copy(),
                    'gt_boxes': val.gt_boxes[batch_num + i].copy()}
        assert np.all(objs_i[rels_i[:, 0]] > 0) and np.all(objs_i[rels_i[:, 1]] > 0)
        pred_entry = {'pred_boxes': boxes_i * BOX_SCALE / IM_SCALE, 'pred_classes': objs_i, 'pred_rel_inds': rels_i,
                      'obj_scores': obj_scores_i, 'rel_scores': pred_scores_i}
        evaluator[conf.mode].evaluate_scene_graph_entry(gt_entry, pred_entry)


def val_epoch():
    detector.eval()
    evaluator = BasicSceneGraphEvaluator.all_modes()
    with torch.no_grad():
        for val_b, (batch, following) in enumerate(with_ahead(val_loader, max(DETECT_AHEAD, 1))):
            for nb in (following if DETECT_AHEAD else ()):
                detector.detect_ahead_blob(nb)
            val_batch((val_b * world + rank), batch, evaluator)
    detector.ahead_discard()
    recalls = evaluator[conf.mode].result_dict[conf.mode + '_recall']
    if world > 1:                                            # every rank evaluated its own images: merge the lists
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, recalls)
        for k in recalls:
            recalls[k] = [x for g in gathered for x in g[k]]
    _device_health()
    if rank == 0:
        evaluator[conf.mode].print_stats()
    return float(np.mean(recalls[100])) if len(recalls[100]) else 0.0


if rank == 0:
    print("Training starts now!")
optimizer, scheduler = get_optim(conf.lr * world * conf.batch_size)
quiet_gc()        # model, optimizer and loaders are built: keep full garbage collections out of the steps (lib/pytorch_misc.py)
for epoch in range(start_epoch + 1, start_epoch + 1 + conf.num_epochs):
    rez = train_epoch(epoch)
    if rank == 0:
        print("overall{:2d}: ({:.3f})\n{}".format(epoch, rez.mean(1)['total'], rez.mean(1)), flush=True)
        if conf.save_dir is not None:
            if isinstance(optimizer, FusedClipSGD):
                optimizer.synchronize()          # (a step deferred to the optimizer's own stream must have landed)
            torch.save({'epoch': epoch, 'state_dict': detector.state_dict()},
                       os.path.join(conf.save_dir, '{}-{}.tar'.format('vgrel', epoch)))
    mAp = val_epoch()
    quiet_gc()        # what the validation pass left behind joins the permanent generation; what died is collected here
    scheduler.step(mAp)
    if any(pg['lr'] <= (conf.lr * world * conf.batch_size) / 99.0 for pg in optimizer.param_groups):
        print("exiting training early", flush=True)
        break
if world > 1:
    torch.distributed.destroy_process_group()
