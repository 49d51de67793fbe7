"""
oracle/boxes.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the box algebra, anchor grid, NMS wrapper, RPN proposal decode and
per-image detection filter of the reference.

  box codec            lib/fpn/box_utils.py:28-78
  pairwise IoU (fp32)  lib/fpn/box_utils.py:85-131
  anchors              lib/fpn/generate_anchors.py:39-126, config.py:57-61
  apply_nms            lib/fpn/nms/functions/nms.py:7-45
  roi_proposals        lib/object_detector.py:560-612
  nms_boxes/filter_det lib/object_detector.py:363-408, :425-485

Tie rule: the reference sorts scores with an unstable torch.sort; oracle
and HIP path both use (score descending, original index ascending).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import native

ANCHOR_SIZE = 16
ANCHOR_RATIOS = (0.23232838, 0.63365731, 1.28478321, 3.15089189)
ANCHOR_SCALES = (2.22152954, 4.12315647, 7.21692515, 12.60263013, 22.7102731)
IM_SCALE = 592


def center_size(boxes):
    wh = boxes[:, 2:] - boxes[:, :2] + 1.0
    return torch.cat((boxes[:, :2] + 0.5 * wh, wh), 1)


def point_form(boxes):
    return torch.cat((boxes[:, :2] - 0.5 * boxes[:, 2:],
                      boxes[:, :2] + 0.5 * (boxes[:, 2:] - 2.0)), 1)


def bbox_preds(boxes, deltas):
    if boxes.size(0) == 0:
        return boxes
    prior_centers = center_size(boxes)
    xys = prior_centers[:, :2] + prior_centers[:, 2:] * deltas[:, :2]
    whs = torch.exp(deltas[:, 2:]) * prior_centers[:, 2:]
    return point_form(torch.cat((xys, whs), 1))


def bbox_intersections(box_a, box_b):
    A, B = box_a.size(0), box_b.size(0)
    max_xy = torch.min(box_a[:, 2:].unsqueeze(1).expand(A, B, 2), box_b[:, 2:].unsqueeze(0).expand(A, B, 2))
    min_xy = torch.max(box_a[:, :2].unsqueeze(1).expand(A, B, 2), box_b[:, :2].unsqueeze(0).expand(A, B, 2))
    inter = torch.clamp((max_xy - min_xy + 1.0), min=0)
    return inter[:, :, 0] * inter[:, :, 1]


def bbox_overlaps(box_a, box_b):
    inter = bbox_intersections(box_a, box_b)
    area_a = ((box_a[:, 2] - box_a[:, 0] + 1.0) * (box_a[:, 3] - box_a[:, 1] + 1.0)).unsqueeze(1).expand_as(inter)
    area_b = ((box_b[:, 2] - box_b[:, 0] + 1.0) * (box_b[:, 3] - box_b[:, 1] + 1.0)).unsqueeze(0).expand_as(inter)
    union = area_a + area_b - inter
    return inter / union


# ----------------------------------------------------------------------------- anchors
def _whctrs(anchor):
    w = anchor[2] - anchor[0] + 1
    h = anchor[3] - anchor[1] + 1
    return w, h, anchor[0] + 0.5 * (w - 1), anchor[1] + 0.5 * (h - 1)


def _mkanchors(ws, hs, x_ctr, y_ctr):
    ws, hs = ws[:, None], hs[:, None]
    return np.hstack((x_ctr - 0.5 * (ws - 1), y_ctr - 0.5 * (hs - 1),
                      x_ctr + 0.5 * (ws - 1), y_ctr + 0.5 * (hs - 1)))


def generate_base_anchors(base_size=16, ratios=ANCHOR_RATIOS, scales=ANCHOR_SCALES):
    ratios, scales = np.array(ratios), np.array(scales)
    base_anchor = np.array([1, 1, base_size, base_size]) - 1
    w, h, x_ctr, y_ctr = _whctrs(base_anchor)
    size_ratios = (w * h) / ratios
    ws = np.sqrt(size_ratios)          # no rounding (generate_anchors.py:110)
    hs = ws * ratios
    ratio_anchors = _mkanchors(ws, hs, x_ctr, y_ctr)
    out = []
    for i in range(ratio_anchors.shape[0]):
        w, h, x_ctr, y_ctr = _whctrs(ratio_anchors[i])
        out.append(_mkanchors(w * scales, h * scales, x_ctr, y_ctr))
    return np.vstack(out)


def generate_anchors(base_size=ANCHOR_SIZE, feat_stride=16, anchor_scales=ANCHOR_SCALES,
                     anchor_ratios=ANCHOR_RATIOS, im_scale=IM_SCALE):
    anchors = generate_base_anchors(base_size, anchor_ratios, anchor_scales)
    shift_x = np.arange(0, im_scale // feat_stride) * feat_stride
    shift_x, shift_y = np.meshgrid(shift_x, shift_x)
    shifts = np.stack([shift_x, shift_y, shift_x, shift_y], -1)
    return shifts[:, :, None] + anchors[None, None]          # [h, w, A, 4] float64


# ----------------------------------------------------------------------------- NMS wrapper
def nms_single_im(scores, boxes, pre_nms_topn=12000, post_nms_topn=2000, nms_thresh=0.7):
    """nms.py:35-45.  Returns indices (int64) into the unsorted input, in score order."""
    _, idx = torch.sort(scores, dim=0, descending=True, stable=True)
    if idx.size(0) > pre_nms_topn:
        idx = idx[:pre_nms_topn]
    boxes_sorted = boxes[idx].contiguous()
    keep = native.nms(boxes_sorted.numpy(), nms_thresh)
    keep = keep[:min(len(keep), post_nms_topn)]
    return idx[torch.from_numpy(keep.astype(np.int64))]


def apply_nms(scores, boxes, pre_nms_topn=12000, post_nms_topn=2000, boxes_per_im=None,
              nms_thresh=0.7):
    just_inds = boxes_per_im is None
    if boxes_per_im is None:
        boxes_per_im = [boxes.size(0)]
    s, keep, im_per = 0, [], []
    for bpi in boxes_per_im:
        e = s + int(bpi)
        keep_im = nms_single_im(scores[s:e], boxes[s:e], pre_nms_topn, post_nms_topn, nms_thresh)
        keep.append(keep_im + s)
        im_per.append(keep_im.size(0))
        s = e
    inds = torch.cat(keep, 0)
    if just_inds:
        return inds
    return inds, im_per


# ----------------------------------------------------------------------------- RPN proposals
def roi_proposals(rpn_feats, anchors, im_sizes, nms_thresh=0.7, pre_nms_topn=6000,
                  post_nms_topn=1000, stride=16):
    """
    object_detector.py:560-612.
    rpn_feats [B,h,w,A,6] (2 class logits + 4 deltas), anchors [h,w,A,4] fp32,
    im_sizes [B,3] (h,w,scale) -> rois [n,5] (im, x1,y1,x2,y2)
    """
    class_fmap = rpn_feats[..., :2].contiguous()
    class_preds = F.softmax(class_fmap, 4)[..., 1].contiguous().clone()
    box_fmap = rpn_feats[..., 2:].contiguous()
    anchor_stacked = torch.cat([anchors[None]] * rpn_feats.size(0), 0)
    box_preds = bbox_preds(anchor_stacked.view(-1, 4), box_fmap.view(-1, 4)).view(*box_fmap.size())
    for i, (h, w, scale) in enumerate(im_sizes):
        h_end, w_end = int(h) // stride, int(w) // stride
        if h_end < class_preds.size(1):
            class_preds[i, h_end:] = -0.01
        if w_end < class_preds.size(2):
            class_preds[i, :, w_end:] = -0.01
        box_preds[i, :, :, :, 0].clamp_(min=0, max=w - 1)
        box_preds[i, :, :, :, 1].clamp_(min=0, max=h - 1)
        box_preds[i, :, :, :, 2].clamp_(min=0, max=w - 1)
        box_preds[i, :, :, :, 3].clamp_(min=0, max=h - 1)
    sizes = center_size(box_preds.view(-1, 4))
    class_preds.view(-1)[(sizes[:, 2] < 4) | (sizes[:, 3] < 4)] = -0.01
    per_im = int(np.prod(box_preds.size()[1:-1]))
    inds, im_per = apply_nms(class_preds.view(-1), box_preds.view(-1, 4),
                             pre_nms_topn=pre_nms_topn, post_nms_topn=post_nms_topn,
                             boxes_per_im=[per_im] * rpn_feats.size(0), nms_thresh=nms_thresh)
    img_inds = torch.cat([val * torch.ones(i) for val, i in enumerate(im_per)], 0)
    return torch.cat((img_inds[:, None], box_preds.view(-1, 4)[inds]), 1)


# ----------------------------------------------------------------------------- detection filter
def filter_det(scores, boxes, start_ind=0, max_per_img=100, thresh=0.001, pre_nms_topn=6000,
               post_nms_topn=300, nms_thresh=0.3, nms_filter_duplicates=True):
    """object_detector.py:425-485 for one image. scores [n,C] (softmaxed), boxes [n,C,4]."""
    valid_cls = (scores[:, 1:].max(0)[0] > thresh).nonzero() + 1
    if valid_cls.numel() == 0:
        return None
    nms_mask = torch.zeros_like(scores)
    for c_i in valid_cls.squeeze(1).tolist():
        keep = apply_nms(scores[:, c_i], boxes[:, c_i], pre_nms_topn=pre_nms_topn,
                         post_nms_topn=post_nms_topn, nms_thresh=nms_thresh)
        nms_mask[:, c_i][keep] = 1
    dists_all = nms_mask * scores
    if nms_filter_duplicates:
        scores_pre, labels_pre = dists_all.max(1)
        inds_all = scores_pre.nonzero().squeeze(1)
        labels_all = labels_pre[inds_all]
        scores_all = scores_pre[inds_all]
    else:
        nz = nms_mask.nonzero()
        inds_all, labels_all = nz[:, 0], nz[:, 1]
        scores_all = scores.reshape(-1)[inds_all * scores.size(1) + labels_all]
    vs, idx = torch.sort(scores_all, dim=0, descending=True, stable=True)
    idx = idx[vs > thresh]
    if max_per_img < idx.size(0):
        idx = idx[:max_per_img]
    return inds_all[idx] + start_ind, scores_all[idx], labels_all[idx]


def enumerate_by_image(im_inds):
    """lib/pytorch_misc.py:278-287"""
    im_inds_np = np.asarray(im_inds)
    initial_ind = int(im_inds_np[0])
    s = 0
    for i, val in enumerate(im_inds_np):
        if val != initial_ind:
            yield initial_ind, s, i
            initial_ind = int(val)
            s = i
    yield initial_ind, s, len(im_inds_np)


def nms_boxes(obj_dists, rois, box_deltas, im_sizes, nms_filter_duplicates=True, max_per_img=64,
              thresh=0.01):
    """object_detector.py:363-408"""
    boxes = bbox_preds(rois[:, None, 1:].expand_as(box_deltas).contiguous().view(-1, 4),
                       box_deltas.reshape(-1, 4)).view(*box_deltas.size()).clone()
    inds = rois[:, 0].long().contiguous()
    dets = []
    for i, s, e in enumerate_by_image(inds.numpy()):
        h, w = im_sizes[i, :2]
        boxes[s:e, :, 0].clamp_(min=0, max=w - 1)
        boxes[s:e, :, 1].clamp_(min=0, max=h - 1)
        boxes[s:e, :, 2].clamp_(min=0, max=w - 1)
        boxes[s:e, :, 3].clamp_(min=0, max=h - 1)
        d = filter_det(F.softmax(obj_dists[s:e], 1), boxes[s:e], start_ind=s,
                       nms_filter_duplicates=nms_filter_duplicates, max_per_img=max_per_img,
                       thresh=thresh)
        if d is not None:
            dets.append(d)
    if len(dets) == 0:
        return None
    nms_inds, nms_scores, nms_labels = [torch.cat(x, 0) for x in zip(*dets)]
    twod_inds = nms_inds * boxes.size(1) + nms_labels
    nms_boxes_assign = boxes.view(-1, 4)[twod_inds]
    nms_boxes_ = torch.cat((rois[:, 1:][nms_inds][:, None], boxes[nms_inds][:, 1:]), 1)
    return nms_inds, nms_scores, nms_labels, nms_boxes_assign, nms_boxes_, inds[nms_inds]
