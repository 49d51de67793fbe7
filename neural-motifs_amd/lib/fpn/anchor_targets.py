"""
RPN anchor targets, produced on the host while a batch is collated (reference lib/fpn/anchor_targets.py:16-105, called
from dataloaders/blob.py:91-102).  Integer / index work: the oracle bar is bit-exact, so the sampling order and the
numpy.random draw sequence of the reference are kept (fg subsample first, then bg); `rs` makes the draws injectable
(default: the global numpy RNG, like the reference).
"""
import numpy as np

from config import IM_SCALE, RPN_NEGATIVE_OVERLAP, RPN_POSITIVE_OVERLAP, RPN_BATCHSIZE, RPN_FG_FRACTION, ANCHOR_SIZE, \
    ANCHOR_SCALES, ANCHOR_RATIOS
from lib.fpn.box_intersections_cpu.bbox import bbox_overlaps
from lib.fpn.generate_anchors import generate_anchors


def anchor_target_layer(gt_boxes, im_size, allowed_border=0, rs=None):
    """
    :param gt_boxes: [n,4] x1,y1,x2,y2 at IM_SCALE
    :param im_size: (h, w) at IM_SCALE
    :return: anchors [k,4], anchor_inds [k,3] (h, w, A), bbox_targets [k,4] (the matched GT box), labels [k] in {0,1}
    """
    rs = np.random if rs is None else rs
    if max(im_size) != IM_SCALE:
        raise ValueError("im size is {}".format(im_size))
    h, w = im_size
    ans_np = generate_anchors(base_size=ANCHOR_SIZE, feat_stride=16, anchor_scales=ANCHOR_SCALES,
                              anchor_ratios=ANCHOR_RATIOS)
    ans_np_flat = ans_np.reshape((-1, 4))
    inds_inside = np.where((ans_np_flat[:, 0] >= -allowed_border) & (ans_np_flat[:, 1] >= -allowed_border) &
                           (ans_np_flat[:, 2] < w + allowed_border) & (ans_np_flat[:, 3] < h + allowed_border))[0]
    good_ans_flat = ans_np_flat[inds_inside]
    if good_ans_flat.size == 0:
        raise ValueError("There were no good anchors for an image of size {} with boxes {}".format(im_size, gt_boxes))

    overlaps = bbox_overlaps(good_ans_flat, gt_boxes)                     # float64 [anchors, gt]
    anchor_to_gtbox = overlaps.argmax(axis=1)
    max_overlaps = overlaps[np.arange(anchor_to_gtbox.shape[0]), anchor_to_gtbox]
    gtbox_to_anchor = overlaps.argmax(axis=0)
    gt_max_overlaps = overlaps[gtbox_to_anchor, np.arange(overlaps.shape[1])]
    gt_argmax_overlaps = np.where(overlaps == gt_max_overlaps)[0]

    # 1 positive, 0 negative, -1 don't care; bg first so that positives clobber them
    labels = (-1) * np.ones(overlaps.shape[0], dtype=np.int64)
    labels[max_overlaps < RPN_NEGATIVE_OVERLAP] = 0
    labels[gt_argmax_overlaps] = 1
    labels[max_overlaps >= RPN_POSITIVE_OVERLAP] = 1

    num_fg = int(RPN_FG_FRACTION * RPN_BATCHSIZE)
    fg_inds = np.where(labels == 1)[0]
    if len(fg_inds) > num_fg:
        labels[rs.choice(fg_inds, size=(len(fg_inds) - num_fg), replace=False)] = -1
    num_bg = RPN_BATCHSIZE - np.sum(labels == 1)
    bg_inds = np.where(labels == 0)[0]
    if len(bg_inds) > num_bg:
        labels[rs.choice(bg_inds, size=(len(bg_inds) - num_bg), replace=False)] = -1

    labels_unmap = (-1) * np.ones(ans_np_flat.shape[0], dtype=np.int64)
    labels_unmap[inds_inside] = labels
    labels_unmap_res = labels_unmap.reshape(ans_np.shape[:-1])            # h, w, A
    anchor_inds = np.column_stack(np.where(labels_unmap_res >= 0))

    anchor_inds_flat = np.where(labels >= 0)[0]
    anchors = good_ans_flat[anchor_inds_flat]
    bbox_targets = gt_boxes[anchor_to_gtbox[anchor_inds_flat]]
    labels = labels[anchor_inds_flat]
    assert np.all(labels >= 0)
    return anchors, anchor_inds, bbox_targets, labels
