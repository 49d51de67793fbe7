"""
Relation sampling for SGDet training (reference lib/fpn/proposal_assignments/rel_assignments.py:15-145).

For every image: detections that match a GT box (same label, IoU >= fg_thresh) inherit that box's annotated
relations -> foreground candidates, sampled per GT relation in proportion to the IoU product and capped at
round(REL_FG_FRACTION * 64) = 16 per image; background = pairs of distinct, overlapping (0 < IoU < 1), non-background
detections that are not foreground, filling up to 64 rows per image (the literal 64 of the reference, :31/:112).
Rows come out sorted by (subject, object) within each image.  Host numpy, like the reference; `rs` makes the draw
reproducible (defaults to numpy's global RNG).
"""
import numpy as np
import torch

from config import REL_FG_FRACTION
from lib.fpn.box_intersections_cpu.bbox import bbox_overlaps
from lib.pytorch_misc import h2d, host_np, set_host

RELS_PER_IMG_SGDET = 64


def rel_assignments(im_inds, rpn_rois, roi_gtlabels, gt_boxes, gt_classes, gt_rels, image_offset, fg_thresh=0.5,
                    num_sample_per_gt=4, filter_non_overlap=True, rs=None):
    """
    im_inds [n] image of each detection, rpn_rois [n,4] boxes, roi_gtlabels [n] assigned GT label (0 = background),
    gt_boxes [g,4], gt_classes [g,2] (im, class), gt_rels [r,4] (im, box0, box1, predicate; within-image indices).
    Returns LongTensor [m,4] (im, subject det, object det, predicate) with GLOBAL detection indices.
    """
    rs = np.random if rs is None else rs
    dev = rpn_rois.device
    fg_per_image = int(np.round(REL_FG_FRACTION * RELS_PER_IMG_SGDET))
    # ONE device->host copy for what only the device knows (detection boxes + their matched labels; fp32 holds a class id
    # exactly); image indices and the ground truth come from their host mirrors when the caller attached them
    packed = torch.cat((rpn_rois.detach().float(), roi_gtlabels.detach().float()[:, None]), 1).cpu().numpy()
    det_boxes = packed[:, :4].astype(np.float64)
    det_labels = packed[:, 4].astype(np.int64)
    det_im = host_np(im_inds)
    gtb = host_np(gt_boxes).astype(np.float64)
    gtc = host_np(gt_classes).copy()
    gtr = host_np(gt_rels).copy()
    gtc[:, 0] -= image_offset
    gtr[:, 0] -= image_offset
    num_im = int(gtc[:, 0].max()) + 1

    out, seen = [], 0
    for im in range(num_im):
        det = np.where(det_im == im)[0]
        g = np.where(gtc[:, 0] == im)[0]
        boxes_i, labels_i = det_boxes[det], det_labels[det]
        gt_boxes_i, gt_classes_i = gtb[g], gtc[g, 1]
        rels_i = gtr[gtr[:, 0] == im, 1:]
        n = boxes_i.shape[0]

        ious = bbox_overlaps(boxes_i, gt_boxes_i)
        is_match = (labels_i[:, None] == gt_classes_i[None]) & (ious >= fg_thresh)
        self_iou = bbox_overlaps(boxes_i, boxes_i)
        if filter_non_overlap:
            possible = (self_iou < 1) & (self_iou > 0)
        else:
            possible = ~np.eye(n, dtype=bool)
        possible = possible.copy()
        possible[labels_i == 0] = False
        possible[:, labels_i == 0] = False

        fg = []
        for (src, dst, pred) in rels_i:
            cands, weights = [], []
            for a in np.where(is_match[:, src])[0]:
                for b in np.where(is_match[:, dst])[0]:
                    if a != b:
                        cands.append((a, b, pred))
                        weights.append(ious[a, src] * ious[b, dst])
                        possible[a, b] = False
            if not cands:
                continue
            p = np.asarray(weights)
            p = p / p.sum()
            k = min(len(cands), num_sample_per_gt)
            for j in rs.choice(len(cands), p=p, size=k, replace=False):
                fg.append(cands[j])
        fg = np.asarray(fg, dtype=np.int64).reshape(-1, 3)
        if fg.shape[0] > fg_per_image:
            fg = fg[rs.choice(fg.shape[0], size=fg_per_image, replace=False)]

        bg = np.column_stack(np.where(possible))
        bg = np.column_stack((bg, np.zeros(bg.shape[0], dtype=np.int64)))
        if bg.shape[0] > 0:
            bg = bg[rs.choice(bg.shape[0], size=min(RELS_PER_IMG_SGDET - fg.shape[0], bg.shape[0]), replace=False)]
        if fg.shape[0] == 0 and bg.shape[0] == 0:
            bg = np.array([[0, 0, 0]], dtype=np.int64)          # keep the image represented (reference :124-126)
        rows = np.concatenate((fg, bg.astype(np.int64)), 0)
        rows[:, 0:2] += seen
        rows = rows[np.lexsort((rows[:, 1], rows[:, 0]))]
        out.append(np.column_stack((np.full(rows.shape[0], im, dtype=np.int64), rows)))
        seen += n
    rows_all = np.ascontiguousarray(np.concatenate(out, 0), dtype=np.int64)
    return set_host(h2d(rows_all, dev), rows_all)
