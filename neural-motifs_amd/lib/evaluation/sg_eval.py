"""
Recall@{20,50,100} for scene graphs (reference lib/evaluation/sg_eval.py): match predicted
(subject, predicate, object) triplets to ground truth by class equality and per-box IoU >= 0.5.
Pure numpy on the host, consumes the eval tuples of RelModel.forward.
"""
from functools import reduce

import numpy as np

from config import MODES
from lib.fpn.box_intersections_cpu.bbox import bbox_overlaps
from lib.pytorch_misc import intersect_2d, argsort_desc

np.set_printoptions(precision=3)


class BasicSceneGraphEvaluator:
    def __init__(self, mode, multiple_preds=False):
        self.result_dict = {mode + '_recall': {20: [], 50: [], 100: []}}
        self.mode = mode
        self.multiple_preds = multiple_preds

    @classmethod
    def all_modes(cls, **kwargs):
        return {m: cls(mode=m, **kwargs) for m in MODES}

    @classmethod
    def vrd_modes(cls, **kwargs):
        return {m: cls(mode=m, multiple_preds=True, **kwargs) for m in ('preddet', 'phrdet')}

    def evaluate_scene_graph_entry(self, gt_entry, pred_scores, viz_dict=None, iou_thresh=0.5):
        return evaluate_from_dict(gt_entry, pred_scores, self.mode, self.result_dict, viz_dict=viz_dict,
                                  iou_thresh=iou_thresh, multiple_preds=self.multiple_preds)

    def save(self, fn):
        np.save(fn, self.result_dict)

    def print_stats(self):
        print('======================' + self.mode + '============================')
        for k, v in self.result_dict[self.mode + '_recall'].items():
            print('R@%i: %f' % (k, np.mean(v)))


def evaluate_from_dict(gt_entry, pred_entry, mode, result_dict, multiple_preds=False, viz_dict=None, **kwargs):
    gt_rels = gt_entry['gt_relations']
    gt_boxes = gt_entry['gt_boxes'].astype(float)
    gt_classes = gt_entry['gt_classes']
    pred_rel_inds = pred_entry['pred_rel_inds']
    rel_scores = pred_entry['rel_scores']
    recalls = result_dict[mode + '_recall']

    if mode == 'predcls':
        pred_boxes, pred_classes, obj_scores = gt_boxes, gt_classes, np.ones(gt_classes.shape[0])
    elif mode == 'sgcls':
        pred_boxes, pred_classes, obj_scores = gt_boxes, pred_entry['pred_classes'], pred_entry['obj_scores']
    elif mode in ('sgdet', 'phrdet'):
        pred_boxes = pred_entry['pred_boxes'].astype(float)
        pred_classes, obj_scores = pred_entry['pred_classes'], pred_entry['obj_scores']
    elif mode == 'preddet':
        prc = intersect_2d(pred_rel_inds, gt_rels[:, :2])
        if prc.size == 0:
            for k in recalls:
                recalls[k].append(0.0)
            return None, None, None
        sel = prc.argmax(0)
        pred_rel_inds, rel_scores = pred_rel_inds[sel], rel_scores[sel]
        ranked = argsort_desc(rel_scores[:, 1:])
        ranked[:, 1] += 1
        ranked = np.column_stack((pred_rel_inds[ranked[:, 0]], ranked[:, 1]))
        matches = intersect_2d(ranked, gt_rels)
        for k in recalls:
            recalls[k].append(float(matches[:k].any(0).sum()) / float(gt_rels.shape[0]))
        return None, None, None
    else:
        raise ValueError('invalid mode')

    if multiple_preds:
        overall = obj_scores[pred_rel_inds].prod(1)[:, None] * rel_scores[:, 1:]
        top = argsort_desc(overall)[:100]
        pred_rels = np.column_stack((pred_rel_inds[top[:, 0]], top[:, 1] + 1))
        predicate_scores = rel_scores[top[:, 0], top[:, 1] + 1]
    else:
        pred_rels = np.column_stack((pred_rel_inds, 1 + rel_scores[:, 1:].argmax(1)))
        predicate_scores = rel_scores[:, 1:].max(1)

    pred_to_gt, pred_5ples, rel_scores = evaluate_recall(
        gt_rels, gt_boxes, gt_classes, pred_rels, pred_boxes, pred_classes, predicate_scores, obj_scores,
        phrdet=(mode == 'phrdet'), **kwargs)
    for k in recalls:
        match = reduce(np.union1d, pred_to_gt[:k])
        recalls[k].append(float(len(match)) / float(gt_rels.shape[0]))
    return pred_to_gt, pred_5ples, rel_scores


def evaluate_recall(gt_rels, gt_boxes, gt_classes, pred_rels, pred_boxes, pred_classes, rel_scores=None,
                    cls_scores=None, iou_thresh=0.5, phrdet=False):
    if pred_rels.size == 0:
        return [[]], np.zeros((0, 5)), np.zeros(0)
    assert gt_rels.shape[0] != 0
    gt_triplets, gt_triplet_boxes, _ = _triplet(gt_rels[:, 2], gt_rels[:, :2], gt_classes, gt_boxes)
    assert pred_rels[:, :2].max() < pred_classes.shape[0]
    assert np.all(pred_rels[:, 2] > 0)
    pred_triplets, pred_triplet_boxes, relation_scores = _triplet(
        pred_rels[:, 2], pred_rels[:, :2], pred_classes, pred_boxes, rel_scores, cls_scores)
    scores_overall = relation_scores.prod(1)
    if not np.all(scores_overall[1:] <= scores_overall[:-1] + 1e-5):
        print("Somehow the relations weren't sorted properly: \n{}".format(scores_overall))
    pred_to_gt = _compute_pred_matches(gt_triplets, pred_triplets, gt_triplet_boxes, pred_triplet_boxes, iou_thresh,
                                       phrdet=phrdet)
    pred_5ples = np.column_stack((pred_rels[:, :2], pred_triplets[:, [0, 2, 1]]))
    return pred_to_gt, pred_5ples, relation_scores


def _triplet(predicates, relations, classes, boxes, predicate_scores=None, class_scores=None):
    assert predicates.shape[0] == relations.shape[0]
    so = classes[relations[:, :2]]
    triplets = np.column_stack((so[:, 0], predicates, so[:, 1]))
    triplet_boxes = np.column_stack((boxes[relations[:, 0]], boxes[relations[:, 1]]))
    triplet_scores = None
    if predicate_scores is not None and class_scores is not None:
        triplet_scores = np.column_stack((class_scores[relations[:, 0]], class_scores[relations[:, 1]],
                                          predicate_scores))
    return triplets, triplet_boxes, triplet_scores


def _compute_pred_matches(gt_triplets, pred_triplets, gt_boxes, pred_boxes, iou_thresh, phrdet=False):
    """for every prediction the list of GT relations it matches"""
    keeps = intersect_2d(gt_triplets, pred_triplets)
    gt_has_match = keeps.any(1)
    pred_to_gt = [[] for _ in range(pred_boxes.shape[0])]
    for gt_ind, gt_box, keep_inds in zip(np.where(gt_has_match)[0], gt_boxes[gt_has_match], keeps[gt_has_match]):
        boxes = pred_boxes[keep_inds]
        if phrdet:
            g = gt_box.reshape((2, 4))
            g_union = np.concatenate((g.min(0)[:2], g.max(0)[2:]), 0)
            b = boxes.reshape((-1, 2, 4))
            b_union = np.concatenate((b.min(1)[:, :2], b.max(1)[:, 2:]), 1)
            inds = bbox_overlaps(g_union[None], b_union)[0] >= iou_thresh
        else:
            sub_iou = bbox_overlaps(gt_box[None, :4], boxes[:, :4])[0]
            obj_iou = bbox_overlaps(gt_box[None, 4:], boxes[:, 4:])[0]
            inds = (sub_iou >= iou_thresh) & (obj_iou >= iou_thresh)
        for i in np.where(keep_inds)[0][inds]:
            pred_to_gt[i].append(int(gt_ind))
    return pred_to_gt
