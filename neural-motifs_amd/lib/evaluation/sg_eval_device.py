"""
Recall@K of one image without leaving the GPU: the ranking and triplet matching of
lib/evaluation/sg_eval.py (`evaluate_from_dict` for the sgdet / sgcls / predcls modes, `_compute_pred_matches`) on
device tensors, so that an evaluation step transfers K-many recall numbers instead of the [Nrel, 51] score matrix.

Inputs are what `RelModel` returns in eval mode BEFORE the `.cpu().numpy()` of the driver: boxes, classes, object
scores, relation pairs (already sorted by overall score, lib/surgery.py: filter_dets) and predicate probabilities.
Matching (labels equal, float64 IoU of subject and object boxes >= 0.5) is the `mh_triplet_match` kernel.
The `multiple_preds` ranking (top 100 of all pair x predicate scores) and the phrase-detection (`phrdet`) union-box
criterion of the reference evaluator are served by the same kernel.
"""
import torch

from lib import _hip


def _union(tb):
    """[n,8] (subject box, object box) -> [n,8] (union box, union box): phrase detection scores the union region, and
    the matching kernel tests the subject and the object IoU -- both are then the union IoU"""
    u = torch.cat((torch.min(tb[:, :2], tb[:, 4:6]), torch.max(tb[:, 2:4], tb[:, 6:8])), 1)
    return torch.cat((u, u), 1)


def recall_at_k(gt_rels, gt_boxes, gt_classes, pred_rel_inds, rel_scores, pred_boxes, pred_classes, ks=(20, 50, 100),
                iou_thresh=0.5, multiple_preds=False, obj_scores=None, phrdet=False):
    """
    :param gt_rels: [G,3] (subject idx, object idx, predicate); gt_boxes [n,4]; gt_classes [n]
    :param pred_rel_inds: [P,2] ranked relation pairs; rel_scores [P,51]; pred_boxes [m,4]; pred_classes [m]
    :param multiple_preds: every (pair, predicate) combination competes: the 100 best by
        obj_score(subj) * obj_score(obj) * predicate prob are scored (reference sg_eval.py:78-84; needs `obj_scores` [m])
    :param phrdet: phrase detection -- a prediction matches when the UNION box of its pair overlaps the union box of the
        GT pair (reference sg_eval.py:262-270)
    :return: (dict k -> recall, nmatch [P'] int32 device tensor = matched GT relations per scored prediction)
    """
    dev = pred_boxes.device
    gt_rels, gt_classes = gt_rels.to(dev).long(), gt_classes.to(dev).long()
    gt_boxes = gt_boxes.to(dev).float()
    pred_rel_inds, pred_classes = pred_rel_inds.to(dev).long(), pred_classes.to(dev).long()
    if multiple_preds:
        if obj_scores is None:
            raise ValueError('multiple_preds ranks by object scores: pass obj_scores')
        os_ = obj_scores.to(dev).float()
        overall = (os_[pred_rel_inds[:, 0]] * os_[pred_rel_inds[:, 1]])[:, None] * rel_scores[:, 1:]
        top = torch.sort(overall.reshape(-1), descending=True, stable=True)[1][:100]
        npred = rel_scores.shape[1] - 1
        pred_rel_inds, predicates = pred_rel_inds[top // npred], top % npred + 1
    else:
        predicates = 1 + rel_scores[:, 1:].argmax(1)                     # sg_eval.py:85 (single prediction per pair)
    gt_trip = torch.stack((gt_classes[gt_rels[:, 0]], gt_rels[:, 2], gt_classes[gt_rels[:, 1]]), 1)
    gt_tb = torch.cat((gt_boxes[gt_rels[:, 0]], gt_boxes[gt_rels[:, 1]]), 1)
    pr_trip = torch.stack((pred_classes[pred_rel_inds[:, 0]], predicates, pred_classes[pred_rel_inds[:, 1]]), 1)
    pr_tb = torch.cat((pred_boxes[pred_rel_inds[:, 0]], pred_boxes[pred_rel_inds[:, 1]]), 1)
    if phrdet:
        gt_tb, pr_tb = _union(gt_tb), _union(pr_tb)
    first, nmatch = _hip.triplet_match(gt_trip, gt_tb, pr_trip, pr_tb, iou_thresh)
    G = max(int(gt_rels.shape[0]), 1)
    counts = torch.stack([(first < k).sum() for k in ks]).tolist()      # the only device -> host transfer
    return {k: c / float(G) for k, c in zip(ks, counts)}, nmatch
