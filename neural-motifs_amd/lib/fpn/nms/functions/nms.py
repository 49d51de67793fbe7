"""
Greedy NMS wrapper with the reference's signature (lib/fpn/nms/functions/nms.py:7-45) over the on-device gfx950 NMS
(mh_nms / mh_nms_batched: 64x64 wavefront-bitmask IoU tiles + single-workgroup sweep, no D2H of the mask).

Tie rule: scores are sorted (value descending, index ascending) -- the reference's torch.sort was unstable.
"""
import torch

from lib import _hip


def _nms_launch(scores, boxes, pre_nms_topn, nms_thresh):
    """sort + suppression of one image, nothing read back: (score order, keep list, device count)"""
    _, idx = torch.sort(scores, dim=0, descending=True, stable=True)
    if idx.size(0) > pre_nms_topn:
        idx = idx[:pre_nms_topn]
    boxes_sorted = boxes[idx].contiguous().float()
    keep, num = _hip.nms(boxes_sorted, float(nms_thresh))
    return idx, keep, num


def apply_nms(scores, boxes, pre_nms_topn=12000, post_nms_topn=2000, boxes_per_im=None, nms_thresh=0.7):
    """indices (into `scores`) of the kept boxes in score order; with `boxes_per_im` also the count per image.
    The suppression of EVERY image is enqueued first; the result sizes come back in ONE device->host copy (the reference
    reads one count per image, lib/fpn/nms/functions/nms.py:18 -- six queue drains per SGDet step at b = 6)."""
    just_inds = boxes_per_im is None
    if boxes_per_im is None:
        boxes_per_im = [boxes.size(0)]
    s, launched = 0, []
    for bpi in boxes_per_im:
        e = s + int(bpi)
        launched.append(_nms_launch(scores[s:e], boxes[s:e], pre_nms_topn, nms_thresh) + (s,))
        s = e
    counts = torch.cat([num.view(-1)[:1] for _, _, num, _ in launched]).cpu().tolist()      # the only host sync
    keep, im_per = [], []
    for (idx, kp, _, s0), cnt in zip(launched, counts):
        num_out = min(int(cnt), post_nms_topn)
        keep.append(idx[kp[:num_out].long()] + s0)
        im_per.append(num_out)
    inds = torch.cat(keep, 0)
    if just_inds:
        return inds
    return inds, im_per


def nms_mask_per_class(scores, boxes, class_ids, nms_thresh, post_nms_topn):
    """Batched per-class NMS for one image (the 150-launch loop of object_detector.py:445-452 in ONE launch pair).
    scores [n,C], boxes [n,C,4], class_ids LongTensor [k] -> float mask [n,C] with 1 at kept (roi, class)."""
    n, C = scores.shape
    k = class_ids.numel()
    mask = torch.zeros_like(scores)
    if k == 0 or n == 0:
        return mask
    sc = scores[:, class_ids]                                             # [n,k]
    _, idx = torch.sort(sc, dim=0, descending=True, stable=True)          # per class order
    bsel = boxes[:, class_ids]                                            # [n,k,4]
    sorted_boxes = torch.gather(bsel, 0, idx[:, :, None].expand(n, k, 4)).permute(1, 0, 2).contiguous().float()
    offs = torch.arange(0, (k + 1) * n, n, dtype=torch.int32, device=scores.device)
    keep, num = _hip.nms_batched(sorted_boxes.view(-1, 4), offs, n, float(nms_thresh))
    keep = keep.view(k, n).long()
    num = num[:k].clamp(max=post_nms_topn).long()
    valid = torch.arange(n, device=scores.device)[None, :] < num[:, None]                 # [k,n]
    rows = torch.gather(idx.t(), 1, keep.clamp(min=0, max=n - 1))                          # original roi ids
    cls = class_ids[:, None].expand(k, n)
    # scatter without a boolean gather (rows[valid] would read the count back): kept entries add 1, the others add 0
    mask.view(-1).index_add_(0, (rows * C + cls).reshape(-1), valid.reshape(-1).to(mask.dtype))
    mask.clamp_(max=1.0)
    return mask
