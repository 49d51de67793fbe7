"""
RoI sampling for detector pre-training (behaviour of the reference's
lib/fpn/proposal_assignments/proposal_assignments_det.py:12-118): per image the RPN proposals AND the GT boxes are
candidates; each is labelled with its best-overlapping GT box; up to ROIS_PER_IMG * FG_FRACTION foreground RoIs
(IoU >= fg_thresh) are drawn first, the rest of the ROIS_PER_IMG budget is filled with background RoIs
(BG_THRESH_LO <= IoU < BG_THRESH_HI), whose labels become 0.

The sampling core `_sel_inds` reproduces the reference's two numpy draws in order (pinned bit-exactly by
tests/test_det_samplers.py); `rs` injects the RNG.  Tensors stay on the device they arrive on.
"""
import numpy as np
import torch

from config import BG_THRESH_HI, BG_THRESH_LO, FG_FRACTION, ROIS_PER_IMG
from lib.fpn.box_utils import bbox_overlaps
from lib.pytorch_misc import h2d


def _sel_inds(max_overlaps, fg_thresh=0.5, fg_rois_per_image=128, rois_per_image=256, rs=None):
    """indices (foreground first) into one image's candidates and the number of foreground picks"""
    rs = np.random if rs is None else rs
    fg = np.where(max_overlaps >= fg_thresh)[0]
    n_fg = min(fg_rois_per_image, fg.shape[0])
    if fg.size > 0:
        fg = rs.choice(fg, size=n_fg, replace=False)
    bg = np.where((max_overlaps < BG_THRESH_HI) & (max_overlaps >= BG_THRESH_LO))[0]
    n_bg = min(rois_per_image - n_fg, bg.size)
    if bg.size > 0:
        bg = rs.choice(bg, size=n_bg, replace=False)
    return np.append(fg, bg), n_fg


def proposal_assignments_det(rpn_rois, gt_boxes, gt_classes, image_offset, fg_thresh=0.5, rs=None):
    """
    :param rpn_rois: [n,5] (img_ind, x1, y1, x2, y2); gt_boxes [g,4]; gt_classes [g,2] (img_ind, class), image-sorted
    :return: rois [k,5], labels [k] int64, bbox_targets [k,4]
    """
    fg_quota = int(np.round(ROIS_PER_IMG * FG_FRACTION))
    gt_image = gt_classes[:, 0] - image_offset
    # candidates = proposals followed by the GT boxes, grouped by image (stable: proposals stay ahead of GT boxes)
    cand_image, order = torch.sort(torch.cat([rpn_rois[:, 0].long(), gt_image], 0), dim=0, stable=True)
    cand_boxes = torch.cat([rpn_rois[:, 1:], gt_boxes], 0)[order]
    picked_rois, picked_labels, picked_targets = [], [], []
    for im in range(int(cand_image[-1]) + 1):
        gt_rows = (gt_image == im).nonzero()
        if gt_rows.numel() == 0:
            continue
        g0, g1 = int(gt_rows[0]), int(gt_rows[-1]) + 1              # GT rows of an image are contiguous
        rows = (cand_image == im).nonzero().squeeze(1)
        c0, c1 = int(rows[0]), int(rows[-1]) + 1
        best_iou, best_gt = bbox_overlaps(cand_boxes[c0:c1], gt_boxes[g0:g1]).max(1)
        keep_np, n_fg = _sel_inds(best_iou.cpu().numpy(), fg_thresh, fg_quota, ROIS_PER_IMG, rs)
        if keep_np.size == 0:
            continue
        keep = h2d(keep_np.astype(np.int64), rpn_rois.device)
        matched = best_gt[keep] + g0
        lab = gt_classes[:, 1][matched].clone()
        lab[n_fg:] = 0                                               # everything after the foreground picks is background
        picked_labels.append(lab)
        picked_targets.append(gt_boxes[matched])
        picked_rois.append(torch.cat((cand_image[c0:c1, None][keep].float(), cand_boxes[c0:c1][keep]), 1))
    return torch.cat(picked_rois, 0), torch.cat(picked_labels, 0), torch.cat(picked_targets, 0)
