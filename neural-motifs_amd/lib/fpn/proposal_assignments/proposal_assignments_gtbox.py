"""
Relation sampling for GT-box training (SGCls / PredCls), reference
lib/fpn/proposal_assignments/proposal_assignments_gtbox.py:9-87: keep at most RELS_PER_IMG*REL_FG_FRACTION*num_im
foreground (annotated) relations, fill up to RELS_PER_IMG*num_im with background pairs (same image, i != j, not
annotated), sort rows by (image, subject, object).  Host-side index work (a few hundred rows); the numpy RNG can be
passed in so that runs are reproducible (`rs`), default is the global numpy RNG like the reference.
"""
import numpy as np
import torch

from config import RELS_PER_IMG, REL_FG_FRACTION
from lib.pytorch_misc import enumerate_by_image, host_np, h2d, set_host


def proposal_assignments_gtbox(rois, gt_boxes, gt_classes, gt_rels, image_offset, fg_thresh=0.5, rs=None):
    """
    rois [n,5] (im, box), gt_classes [n,2] (im, class), gt_rels [r,4] (im, box0, box1, predicate) with within-image
    box indices.  Returns (rois, labels [n], rel_labels [m,4] with GLOBAL box indices).
    """
    rs = np.random if rs is None else rs
    dev = rois.device
    im_inds = host_np(rois)[:, 0].astype(np.int64)          # no device round trip when the Blob's host mirrors are attached
    n = im_inds.shape[0]
    num_im = int(im_inds[-1]) + 1

    fg = host_np(gt_rels).astype(np.int64).copy()
    fg[:, 0] -= image_offset
    first_box = {i: s for i, s, e in enumerate_by_image(im_inds)}
    for r in range(fg.shape[0]):
        fg[r, 1:3] += first_box[int(fg[r, 0])]

    cand = im_inds[:, None] == im_inds[None, :]
    np.fill_diagonal(cand, False)
    cand[fg[:, 1], fg[:, 2]] = False                     # annotated pairs are not background
    bg_pairs = np.column_stack(np.nonzero(cand))

    num_fg = min(fg.shape[0], int(RELS_PER_IMG * REL_FG_FRACTION * num_im))
    if num_fg < fg.shape[0]:
        fg = fg[rs.choice(fg.shape[0], size=num_fg, replace=False)]
    num_bg = min(bg_pairs.shape[0], int(RELS_PER_IMG * num_im) - num_fg)
    if num_bg > 0:
        bg = np.column_stack((im_inds[bg_pairs[:, 0]], bg_pairs, np.zeros(bg_pairs.shape[0], dtype=np.int64)))
        if num_bg < bg.shape[0]:
            bg = bg[rs.choice(bg.shape[0], size=num_bg, replace=False)]
        rel_labels = np.concatenate((fg, bg), 0)
    else:
        rel_labels = fg
    key = rel_labels[:, 0] * (n ** 2) + rel_labels[:, 1] * n + rel_labels[:, 2]
    rel_labels = rel_labels[np.argsort(key, kind='stable')]
    labels = gt_classes[:, 1].contiguous()
    rel_labels = np.ascontiguousarray(rel_labels, dtype=np.int64)
    return rois, labels, set_host(h2d(rel_labels, dev), rel_labels)      # (mirror: the union-box geometry is computed on the host from it)
