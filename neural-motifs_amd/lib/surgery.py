"""
Eval post-processing (reference lib/surgery.py:21-59): rank candidate relations by
max_pred_score * subject_score * object_score and hand everything to the numpy evaluator.
"""
import torch


def filter_dets(boxes, obj_scores, obj_classes, rel_inds, pred_scores, to_numpy=True):
    """
    boxes [num_box,4], obj_scores [num_box], obj_classes [num_box], rel_inds [num_rel,2], pred_scores [num_rel,51]
    -> (boxes, classes, obj_scores, rels sorted by triple score, pred_scores sorted) as numpy arrays (the reference's
    contract), or as device tensors with to_numpy=False (on-device evaluation: lib/evaluation/sg_eval_device.py)
    """
    if boxes.dim() != 2:
        raise ValueError("Boxes needs to be [num_box, 4] but its {}".format(boxes.size()))
    num_box = boxes.size(0)
    assert obj_scores.size(0) == num_box and obj_classes.size() == obj_scores.size()
    assert rel_inds.size(1) == 2 and pred_scores.size(0) == rel_inds.size(0)
    with torch.no_grad():
        best_pred = pred_scores[:, 1:].max(1)[0]
        triple = best_pred * obj_scores[rel_inds[:, 0]] * obj_scores[rel_inds[:, 1]]
        _, order = torch.sort(triple.view(-1), dim=0, descending=True, stable=True)
        if not to_numpy:
            return boxes.detach(), obj_classes.detach(), obj_scores.detach(), rel_inds[order], pred_scores[order]
        rels = rel_inds[order].cpu().numpy()
        pred_sorted = pred_scores[order].cpu().numpy()
    return boxes.detach().cpu().numpy(), obj_classes.detach().cpu().numpy(), obj_scores.detach().cpu().numpy(), \
        rels, pred_sorted
