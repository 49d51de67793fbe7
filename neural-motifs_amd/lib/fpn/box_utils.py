"""
Box algebra with the reference's names and conventions (lib/fpn/box_utils.py): inclusive-pixel boxes
(width = x2 - x1 + 1), (cx,cy,w,h) <-> (x1,y1,x2,y2) codecs, delta decoding, pairwise IoU.
Tensor inputs stay on their device; the dense [A,B] fp32 IoU runs on the HIP kernel (mh_bbox_overlaps) when the
inputs live on the GPU; numpy inputs go to the float64 host routine, as in the reference.
"""
import numpy as np
import torch
from torch.nn import functional as F

from lib.fpn.box_intersections_cpu.bbox import bbox_overlaps as bbox_overlaps_np
from lib.fpn.box_intersections_cpu.bbox import bbox_intersections as bbox_intersections_np


def center_size(boxes):
    """(x1,y1,x2,y2) -> (cx,cy,w,h)"""
    wh = boxes[:, 2:] - boxes[:, :2] + 1.0
    ctr = boxes[:, :2] + 0.5 * wh
    if isinstance(boxes, np.ndarray):
        return np.column_stack((ctr, wh))
    return torch.cat((ctr, wh), 1)


def point_form(boxes):
    """(cx,cy,w,h) -> (x1,y1,x2,y2)"""
    lo = boxes[:, :2] - 0.5 * boxes[:, 2:]
    hi = boxes[:, :2] + 0.5 * (boxes[:, 2:] - 2.0)
    if isinstance(boxes, np.ndarray):
        return np.column_stack((lo, hi))
    return torch.cat((lo, hi), 1)


def bbox_preds(boxes, deltas):
    """apply (tx,ty,tw,th) to prior boxes (x1,y1,x2,y2)"""
    if boxes.size(0) == 0:
        return boxes
    prior = center_size(boxes)
    xys = prior[:, :2] + prior[:, 2:] * deltas[:, :2]
    whs = torch.exp(deltas[:, 2:]) * prior[:, 2:]
    return point_form(torch.cat((xys, whs), 1))


def bbox_loss(prior_boxes, deltas, gt_boxes, eps=1e-4, scale_before=1):
    prior = center_size(prior_boxes)
    gt = center_size(gt_boxes)
    targets = torch.cat(((gt[:, :2] - prior[:, :2]) / prior[:, 2:], torch.log(gt[:, 2:]) - torch.log(prior[:, 2:])), 1)
    return F.smooth_l1_loss(deltas, targets, reduction='sum') / (eps + prior.size(0))


def bbox_intersections(box_a, box_b):
    if isinstance(box_a, np.ndarray):
        assert isinstance(box_b, np.ndarray)
        return bbox_intersections_np(box_a, box_b)
    hi = torch.min(box_a[:, None, 2:], box_b[None, :, 2:])
    lo = torch.max(box_a[:, None, :2], box_b[None, :, :2])
    inter = torch.clamp(hi - lo + 1.0, min=0)
    return inter[:, :, 0] * inter[:, :, 1]


def bbox_overlaps(box_a, box_b):
    """[A,4] x [B,4] -> [A,B] IoU"""
    if isinstance(box_a, np.ndarray):
        assert isinstance(box_b, np.ndarray)
        return bbox_overlaps_np(box_a, box_b)
    if box_a.is_cuda and box_a.dtype == torch.float32 and box_a.size(0) > 0 and box_b.size(0) > 0:
        from lib import _hip
        return _hip.bbox_overlaps(box_a.contiguous(), box_b.contiguous())
    inter = bbox_intersections(box_a, box_b)
    area_a = ((box_a[:, 2] - box_a[:, 0] + 1.0) * (box_a[:, 3] - box_a[:, 1] + 1.0))[:, None]
    area_b = ((box_b[:, 2] - box_b[:, 0] + 1.0) * (box_b[:, 3] - box_b[:, 1] + 1.0))[None, :]
    return inter / (area_a + area_b - inter)


def nms_overlaps(boxes):
    """per-class IoU: boxes [N,nc,4] -> [N,N,nc]"""
    assert boxes.dim() == 3
    hi = torch.min(boxes[:, None, :, 2:], boxes[None, :, :, 2:])
    lo = torch.max(boxes[:, None, :, :2], boxes[None, :, :, :2])
    inter = torch.clamp(hi - lo + 1.0, min=0)
    inters = inter[..., 0] * inter[..., 1]
    areas = (boxes[..., 2] - boxes[..., 0] + 1.0) * (boxes[..., 3] - boxes[..., 1] + 1.0)    # [N,nc]
    union = -inters + areas[None] + areas[:, None]
    return inters / union
