"""
Relation-model evaluation driver (flow of the reference's models/eval_rels.py): one image per step, the eval 5-tuple
goes to the numpy Recall@K evaluator; predictions can be cached with dill.

    python models/eval_rels.py -m predcls -model motifnet -order leftright -nl_obj 2 -nl_edge 2 -hidden_dim 512 \
        -pooling_dim 4096 -ngpu 1 -use_bias [-ckpt checkpoints/motifnet/vgrel-7.tar] [-cache preds.pkl]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import dill as pkl
import numpy as np
import torch
from tqdm import tqdm

from config import ModelConfig, BOX_SCALE, IM_SCALE
from dataloaders.visual_genome import VGDataLoader, VG
from lib.evaluation.sg_eval import BasicSceneGraphEvaluator
from lib.pytorch_misc import optimistic_restore, with_ahead

conf = ModelConfig()
if conf.model == 'motifnet':
    from lib.rel_model import RelModel
elif conf.model == 'stanford':              # message-passing baseline (reference models/train_rels.py:22-27)
    from lib.rel_model_stanford import RelModelStanford as RelModel
else:
    raise ValueError('unknown model %r' % conf.model)

train, val, test = VG.splits(num_val_im=conf.val_size, filter_duplicate_rels=True, use_proposals=conf.use_proposals,
                             filter_non_overlap=conf.mode == 'sgdet', seed=conf.seed)
if conf.test:
    val = test
train_loader, val_loader = VGDataLoader.splits(train, val, mode='rel', batch_size=conf.batch_size,
                                               num_workers=conf.num_workers, num_gpus=1)

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
detector.cuda()
if conf.ckpt is not None:
    optimistic_restore(detector, torch.load(conf.ckpt, map_location='cpu')['state_dict'])

all_pred_entries = []


DEVICE_EVAL = bool(os.environ.get('MOTIFS_DEVICE_EVAL')) and conf.mode in ('sgdet', 'sgcls', 'predcls') and not conf.multi_pred
device_recalls = {20: [], 50: [], 100: []}


def val_batch_device(batch_num, b):
    """Recall@K without moving the [Nrel,51] score matrix to the host (lib/evaluation/sg_eval_device.py)"""
    from lib.evaluation.sg_eval_device import recall_at_k
    boxes_i, objs_i, obj_scores_i, rels_i, pred_scores_i = detector[b]
    t = lambda a: torch.from_numpy(np.asarray(a))
    gt_boxes = t(val.gt_boxes[batch_num]).float()
    if conf.mode == 'predcls':
        pred_boxes, pred_classes = gt_boxes.cuda(), t(val.gt_classes[batch_num]).cuda()
    elif conf.mode == 'sgcls':
        pred_boxes, pred_classes = gt_boxes.cuda(), objs_i
    else:
        pred_boxes, pred_classes = boxes_i * (BOX_SCALE / IM_SCALE), objs_i
    rec, _ = recall_at_k(t(val.relationships[batch_num]), gt_boxes, t(val.gt_classes[batch_num]), rels_i, pred_scores_i,
                         pred_boxes, pred_classes)
    for k, v in rec.items():
        device_recalls[k].append(v)


def val_batch(batch_num, b, evaluator):
    det_res = [detector[b]]
    for i, (boxes_i, objs_i, obj_scores_i, rels_i, pred_scores_i) in enumerate(det_res):
        gt_entry = {'gt_classes': val.gt_classes[batch_num + i].copy(),
                    'gt_relations': val.relationships[batch_num + i].copy(),
                    'gt_boxes': val.gt_boxes[batch_num + i].copy()}
        assert np.all(objs_i[rels_i[:, 0]] > 0) and np.all(objs_i[rels_i[:, 1]] > 0)
        pred_entry = {'pred_boxes': boxes_i * BOX_SCALE / IM_SCALE, 'pred_classes': objs_i, 'pred_rel_inds': rels_i,
                      'obj_scores': obj_scores_i, 'rel_scores': pred_scores_i}
        all_pred_entries.append(pred_entry)
        evaluator[conf.mode].evaluate_scene_graph_entry(gt_entry, pred_entry)


evaluator = BasicSceneGraphEvaluator.all_modes(multiple_preds=conf.multi_pred)
if conf.cache is not None and os.path.exists(conf.cache):
    print("Found {}! Loading from it".format(conf.cache))
    with open(conf.cache, 'rb') as f:
        all_pred_entries = pkl.load(f)
    for i, pred_entry in enumerate(tqdm(all_pred_entries)):
        gt_entry = {'gt_classes': val.gt_classes[i].copy(), 'gt_relations': val.relationships[i].copy(),
                    'gt_boxes': val.gt_boxes[i].copy()}
        evaluator[conf.mode].evaluate_scene_graph_entry(gt_entry, pred_entry)
    evaluator[conf.mode].print_stats()
else:
    detector.eval()
    detector.eval_on_device = DEVICE_EVAL
    with torch.no_grad():
        # SGDet: the detector stage of the next images runs beside this image's relation stage (RelModel.detect_ahead; bench.py
        # cfg5: 77 img/s against 66 in line; GT-box modes have no waits in that stage and stay in line).
        # MOTIFS_DETECT_AHEAD=0: in line, =N: N images in flight
        ahead = int(os.environ.get('MOTIFS_DETECT_AHEAD', '2')) if conf.mode == 'sgdet' else 0
        for val_b, (batch, following) in enumerate(with_ahead(tqdm(val_loader), max(ahead, 1))):
            for nb in (following if ahead else ()):
                detector.detect_ahead_blob(nb)
            if DEVICE_EVAL:
                val_batch_device(val_b, batch)
            else:
                val_batch(val_b, batch, evaluator)
    if DEVICE_EVAL:
        print('======================' + conf.mode + ' (device evaluator)============================')
        for k, v in device_recalls.items():
            print('R@%i: %f' % (k, float(np.mean(v)) if v else 0.0))
    else:
        evaluator[conf.mode].print_stats()
    if conf.cache is not None:
        with open(conf.cache, 'wb') as f:
            pkl.dump(all_pred_entries, f)
