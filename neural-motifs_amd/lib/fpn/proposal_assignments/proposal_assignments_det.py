"""
Detector-training proposal assignment (reference lib/fpn/proposal_assignments/proposal_assignments_det.py:12-118):
per image, RPN proposals + GT boxes are labelled by their best-overlapping GT box, <= 25 % foreground of 256 RoIs are
sampled (numpy draws: fg first, then bg, as in `_sel_inds` :94-118), background labels clamped to 0.  Host logic on
device tensors; `rs` injects the RNG (default: global numpy RNG like the reference).
"""
import numpy as np
import torch

from config import BG_THRESH_HI, BG_THRESH_LO, FG_FRACTION, ROIS_PER_IMG
from lib.fpn.box_utils import bbox_overlaps


def proposal_assignments_det(rpn_rois, gt_boxes, gt_classes, image_offset, fg_thresh=0.5, rs=None):
    """
    :param rpn_rois: [n,5] (img_ind, x1, y1, x2, y2)
    :param gt_boxes: [g,4]; gt_classes: [g,2] (img_ind, class), sorted by image
    :return: rois [k,5], labels [k] int64, bbox_targets [k,4]
    """
    fg_rois_per_image = int(np.round(ROIS_PER_IMG * FG_FRACTION))
    gt_img_inds = gt_classes[:, 0] - image_offset
    all_boxes = torch.cat([rpn_rois[:, 1:], gt_boxes], 0)
    ims_per_box = torch.cat([rpn_rois[:, 0].long(), gt_img_inds], 0)
    im_sorted, idx = torch.sort(ims_per_box, dim=0, stable=True)           # tie rule: DESIGN.md §4
    all_boxes = all_boxes[idx]
    num_images = int(im_sorted[-1]) + 1
    labels, rois, bbox_targets = [], [], []
    for im_ind in range(num_images):
        g_inds = (gt_img_inds == im_ind).nonzero()
        if g_inds.numel() == 0:
            continue
        g_inds = g_inds.squeeze(1)
        g_start, g_end = int(g_inds[0]), int(g_inds[-1]) + 1
        t_inds = (im_sorted == im_ind).nonzero().squeeze(1)
        t_start, t_end = int(t_inds[0]), int(t_inds[-1]) + 1
        ious = bbox_overlaps(all_boxes[t_start:t_end], gt_boxes[g_start:g_end])
        max_overlaps, gt_assignment = ious.max(1)
        gt_assignment = gt_assignment + g_start
        keep_inds_np, num_fg = _sel_inds(max_overlaps.cpu().numpy(), fg_thresh, fg_rois_per_image, ROIS_PER_IMG, rs)
        if keep_inds_np.size == 0:
            continue
        keep_inds = torch.from_numpy(keep_inds_np.astype(np.int64)).to(rpn_rois.device)
        labels_ = gt_classes[:, 1][gt_assignment[keep_inds]].clone()
        bbox_target_ = gt_boxes[gt_assignment[keep_inds]]
        if num_fg < labels_.size(0):
            labels_[num_fg:] = 0
        rois_ = torch.cat((im_sorted[t_start:t_end, None][keep_inds].float(), all_boxes[t_start:t_end][keep_inds]), 1)
        labels.append(labels_)
        rois.append(rois_)
        bbox_targets.append(bbox_target_)
    return torch.cat(rois, 0), torch.cat(labels, 0), torch.cat(bbox_targets, 0)


def _sel_inds(max_overlaps, fg_thresh=0.5, fg_rois_per_image=128, rois_per_image=256, rs=None):
    rs = np.random if rs is None else rs
    fg_inds = np.where(max_overlaps >= fg_thresh)[0]
    fg_rois_per_this_image = min(fg_rois_per_image, fg_inds.shape[0])
    if fg_inds.size > 0:
        fg_inds = rs.choice(fg_inds, size=fg_rois_per_this_image, replace=False)
    bg_inds = np.where((max_overlaps < BG_THRESH_HI) & (max_overlaps >= BG_THRESH_LO))[0]
    bg_rois_per_this_image = min(rois_per_image - fg_rois_per_this_image, bg_inds.size)
    if bg_inds.size > 0:
        bg_inds = rs.choice(bg_inds, size=bg_rois_per_this_image, replace=False)
    return np.append(fg_inds, bg_inds), fg_rois_per_this_image
