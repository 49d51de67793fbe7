"""
Recall@K of one image without leaving the GPU: the ranking and triplet matching of
lib/evaluation/sg_eval.py (`evaluate_from_dict` for the sgdet / sgcls / predcls modes, `_compute_pred_matches`) on
device tensors, so that an evaluation step transfers K-many recall numbers instead of the [Nrel, 51] score matrix.

Inputs are what `RelModel` returns in eval mode BEFORE the `.cpu().numpy()` of the driver: boxes, classes, object
scores, relation pairs (already sorted by overall score, lib/surgery.py: filter_dets) and predicate probabilities.
Matching (labels equal, float64 IoU of subject and object boxes >= 0.5) is the `mh_triplet_match` kernel.
"""
import torch

from lib import _hip


def recall_at_k(gt_rels, gt_boxes, gt_classes, pred_rel_inds, rel_scores, pred_boxes, pred_classes, ks=(20, 50, 100),
                iou_thresh=0.5):
    """
    :param gt_rels: [G,3] (subject idx, object idx, predicate); gt_boxes [n,4]; gt_classes [n]
    :param pred_rel_inds: [P,2] ranked relation pairs; rel_scores [P,51]; pred_boxes [m,4]; pred_classes [m]
    :return: (dict k -> recall, nmatch [P] int32 device tensor = matched GT relations per prediction)
    """
    dev = pred_boxes.device
    gt_rels, gt_classes = gt_rels.to(dev).long(), gt_classes.to(dev).long()
    gt_boxes = gt_boxes.to(dev).float()
    pred_rel_inds, pred_classes = pred_rel_inds.to(dev).long(), pred_classes.to(dev).long()
    predicates = 1 + rel_scores[:, 1:].argmax(1)                         # sg_eval.py:85 (single prediction per pair)
    gt_trip = torch.stack((gt_classes[gt_rels[:, 0]], gt_rels[:, 2], gt_classes[gt_rels[:, 1]]), 1)
    gt_tb = torch.cat((gt_boxes[gt_rels[:, 0]], gt_boxes[gt_rels[:, 1]]), 1)
    pr_trip = torch.stack((pred_classes[pred_rel_inds[:, 0]], predicates, pred_classes[pred_rel_inds[:, 1]]), 1)
    pr_tb = torch.cat((pred_boxes[pred_rel_inds[:, 0]], pred_boxes[pred_rel_inds[:, 1]]), 1)
    first, nmatch = _hip.triplet_match(gt_trip, gt_tb, pr_trip, pr_tb, iou_thresh)
    G = max(int(gt_rels.shape[0]), 1)
    counts = torch.stack([(first < k).sum() for k in ks]).tolist()      # the only device -> host transfer
    return {k: c / float(G) for k, c in zip(ks, counts)}, nmatch
