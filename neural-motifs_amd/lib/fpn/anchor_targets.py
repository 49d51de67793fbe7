"""
RPN anchor targets, produced on the host while a batch is collated (behaviour of the reference's
lib/fpn/anchor_targets.py:16-105, which dataloaders/blob.py:91-102 calls per image).

Index work: the bar is bit-exact against the reference's own function (tests/test_det_samplers.py, goldens from
tests/golden/make_golden.py), including WHICH anchors survive the sub-sampling -- so the two numpy draws are made in
the reference's order (surplus foreground first, then surplus background) through `rs` (default: the global numpy RNG).
"""
import numpy as np

from config import IM_SCALE, RPN_NEGATIVE_OVERLAP, RPN_POSITIVE_OVERLAP, RPN_BATCHSIZE, RPN_FG_FRACTION, ANCHOR_SIZE, \
    ANCHOR_SCALES, ANCHOR_RATIOS
from lib.fpn.box_intersections_cpu.bbox import bbox_overlaps
from lib.fpn.generate_anchors import generate_anchors

DONT_CARE, NEGATIVE, POSITIVE = -1, 0, 1


def _anchors_inside(grid_flat, height, width, border):
    x1, y1, x2, y2 = grid_flat.T
    return np.flatnonzero((x1 >= -border) & (y1 >= -border) & (x2 < width + border) & (y2 < height + border))


def _initial_labels(iou):
    """positive: IoU >= 0.7 with some GT box, or the best anchor(s) of a GT box; negative: best IoU < 0.3"""
    best_gt = iou.argmax(axis=1)
    best_iou = iou[np.arange(iou.shape[0]), best_gt]
    top_per_gt = iou[iou.argmax(axis=0), np.arange(iou.shape[1])]
    state = np.full(iou.shape[0], DONT_CARE, dtype=np.int64)
    state[best_iou < RPN_NEGATIVE_OVERLAP] = NEGATIVE
    state[np.where(iou == top_per_gt)[0]] = POSITIVE          # ties included; may overwrite a negative
    state[best_iou >= RPN_POSITIVE_OVERLAP] = POSITIVE
    return state, best_gt


def _drop_surplus(state, kind, quota, rs):
    members = np.flatnonzero(state == kind)
    if members.size > quota:
        state[rs.choice(members, size=members.size - quota, replace=False)] = DONT_CARE


def anchor_target_layer(gt_boxes, im_size, allowed_border=0, rs=None):
    """
    :param gt_boxes: [n,4] x1,y1,x2,y2 at IM_SCALE;  im_size: (h, w) at IM_SCALE
    :return: anchors [k,4], anchor_inds [k,3] = (h, w, A) grid position, bbox_targets [k,4] = matched GT box,
             labels [k] in {0,1}  (k <= RPN_BATCHSIZE, at most half positive)
    """
    rs = np.random if rs is None else rs
    if max(im_size) != IM_SCALE:
        raise ValueError("im size is {}".format(im_size))
    height, width = im_size
    grid = generate_anchors(base_size=ANCHOR_SIZE, feat_stride=16, anchor_scales=ANCHOR_SCALES,
                            anchor_ratios=ANCHOR_RATIOS)                    # [h, w, A, 4]
    grid_flat = grid.reshape(-1, 4)
    inside = _anchors_inside(grid_flat, height, width, allowed_border)
    if inside.size == 0:
        raise ValueError("There were no good anchors for an image of size {} with boxes {}".format(im_size, gt_boxes))
    candidates = grid_flat[inside]
    state, best_gt = _initial_labels(bbox_overlaps(candidates, gt_boxes))

    _drop_surplus(state, POSITIVE, int(RPN_FG_FRACTION * RPN_BATCHSIZE), rs)
    _drop_surplus(state, NEGATIVE, RPN_BATCHSIZE - int(np.sum(state == POSITIVE)), rs)

    used = np.flatnonzero(state >= 0)                                           # ascending = raster order of the grid
    full = np.full(grid_flat.shape[0], DONT_CARE, dtype=np.int64)
    full[inside] = state
    anchor_inds = np.column_stack(np.where(full.reshape(grid.shape[:-1]) >= 0))
    labels = state[used]
    assert np.all(labels >= 0)
    return candidates[used], anchor_inds, gt_boxes[best_gt[used]], labels
