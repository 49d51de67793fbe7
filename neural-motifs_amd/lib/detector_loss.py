"""
The four losses of detector pre-training (reference models/train_detector.py:100-140): RoI classification + box
regression on the sampled RoIs, RPN objectness + anchor regression on the sampled anchors.  Host-side glue over small
tensors (a few hundred rows); kept as one function so the driver, the tests and the oracle-side restatement
(oracle/model.py: detector_losses) state the same arithmetic.
"""
import torch
from torch.nn import functional as F

from config import FG_FRACTION, RPN_FG_FRACTION
from lib.fpn.box_utils import bbox_loss


def detector_losses(result, train_anchor_labels=None, train_anchors=None):
    """:param result: ObjectDetector Result of a training forward in mode 'rpntrain' (or 'proposals')
    :param train_anchor_labels: Blob.train_anchor_labels [k,5] (img, h, w, A, label); None -> no RPN terms
    :param train_anchors: Blob.train_anchors [k,8] (anchor box, matched GT box)
    :return: dict of scalar losses incl. 'total'"""
    scores, box_deltas, labels = result.od_obj_dists, result.od_box_deltas, result.od_obj_labels
    roi_boxes, bbox_targets = result.od_box_priors, result.od_box_targets
    valid_inds = (labels != 0).nonzero().squeeze(1)
    fg_cnt = valid_inds.size(0)
    bg_cnt = labels.size(0) - fg_cnt
    out = {'class_loss': F.cross_entropy(scores, labels)}
    box_reg_mult = 2 * (1. / FG_FRACTION) * fg_cnt / (fg_cnt + bg_cnt + 1e-4)
    twod_inds = valid_inds * box_deltas.size(1) + labels[valid_inds]
    out['box_loss'] = bbox_loss(roi_boxes[valid_inds], box_deltas.reshape(-1, 4)[twod_inds],
                                bbox_targets[valid_inds]) * box_reg_mult
    total = out['class_loss'] + out['box_loss']
    if train_anchor_labels is not None:
        anchor_labels = train_anchor_labels[:, -1]
        anchors, anchor_targets = train_anchors[:, :4], train_anchors[:, 4:]
        pos = (anchor_labels == 1).nonzero().squeeze(1)
        out['rpn_class_loss'] = F.cross_entropy(result.rpn_scores, anchor_labels)
        rpn_box_mult = 2 * (1. / RPN_FG_FRACTION) * pos.size(0) / (anchor_labels.size(0) + 1e-4)
        out['rpn_box_loss'] = bbox_loss(anchors[pos], result.rpn_box_deltas[pos], anchor_targets[pos]) * rpn_box_mult
        total = total + out['rpn_class_loss'] + out['rpn_box_loss']
    out['total'] = total
    return out
