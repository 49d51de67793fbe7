"""
Detection mAP for the detector pre-training driver (reference models/train_detector.py:158-181: `COCOeval(val.coco,
val.coco.loadRes(dets), 'bbox')`, `mAp = coco_eval.stats[1]`, and dataloaders/visual_genome.py:103-127, the
"fauxcoco" ground truth built from the VG boxes).

The reference gets the evaluation from a third-party package that is neither vendored in its tree nor installed here:
`pycocotools` (cocodataset/cocoapi, PythonAPI/pycocotools/cocoeval.py + the C mask API for box IoU; unpinned in the
reference, which predates its 2.0.1 release).  This module restates that published bbox protocol in numpy:

  * ground truth / detections per (image, category); detections by descending score (stable), at most 100;
  * IoU on [x, y, w, h] boxes with continuous areas (w * h); a crowd ground truth uses inter / area(det);
  * ten IoU thresholds .50:.05:.95; greedy matching in score order: a detection takes the unmatched ground truth of
    highest IoU >= threshold, regular ones before ignored ones (ignored = crowd or area outside the range);
    unmatched detections whose own area is outside the range are ignored;
  * precision at 101 recall points, made monotone from the right, averaged over the entries that exist.

One behaviour is the reference's own and is kept by default (`first_ann_id=0`): its faux-COCO numbers annotations from
0 and cocoeval stores the matched annotation id in `dtMatches` and tests it for truth, so the detection matched to the
dataset's very first annotation counts as unmatched.  `first_ann_id=1` gives the textbook result.

PARITY UNPINNED: pycocotools is absent from this image, so there are no golden vectors from it.  tests/test_det_map.py
checks this vectorised implementation against hand-computed cases and against a separate loop-by-loop restatement of
the matching rule.
"""
import numpy as np

IOU_THRS = np.linspace(0.5, 0.95, 10)
REC_THRS = np.linspace(0.0, 1.0, 101)
MAX_DETS = (1, 10, 100)
AREA_RNG = ((0.0, 1e5 ** 2), (0.0, 32.0 ** 2), (32.0 ** 2, 96.0 ** 2), (96.0 ** 2, 1e5 ** 2))      # all, small, medium, large
STAT_NAMES = ('AP', 'AP50', 'AP75', 'APs', 'APm', 'APl', 'AR1', 'AR10', 'AR100', 'ARs', 'ARm', 'ARl')


class FauxCoco(object):
    """Ground truth of a VG split in the form the reference hands to COCOeval (visual_genome.py:103-127): per image
    the boxes as [x, y, w, h] with w = x2 - x1 + 1, area = w * h, iscrowd 0, annotation ids counted over the dataset."""

    def __init__(self, gt_classes, gt_boxes, num_classes, first_ann_id=0):
        self.num_images = len(gt_classes)
        self.cat_ids = list(range(1, num_classes))                                   # '__background__' has no category
        self.cls, self.xywh, self.area, self.ann_id, self.crowd = [], [], [], [], []
        next_id = first_ann_id
        for c, b in zip(gt_classes, gt_boxes):
            c = np.asarray(c).reshape(-1).astype(np.int64)
            b = np.asarray(b, dtype=np.float64).reshape(-1, 4)
            wh = b[:, 2:4] - b[:, 0:2] + 1.0
            self.cls.append(c)
            self.xywh.append(np.concatenate((b[:, 0:2], wh), 1))
            self.area.append(wh[:, 0] * wh[:, 1])
            self.ann_id.append(np.arange(next_id, next_id + c.shape[0], dtype=np.int64))
            self.crowd.append(np.zeros(c.shape[0], dtype=bool))
            next_id += c.shape[0]


def box_iou_xywh(dt, gt, crowd):
    """[D,4] x [G,4] (x, y, w, h), float64 -> [D,G]; column g of a crowd ground truth is inter / area(dt)"""
    dt, gt = np.asarray(dt, dtype=np.float64), np.asarray(gt, dtype=np.float64)
    iw = np.minimum(dt[:, None, 0] + dt[:, None, 2], gt[None, :, 0] + gt[None, :, 2]) - np.maximum(dt[:, None, 0], gt[None, :, 0])
    ih = np.minimum(dt[:, None, 1] + dt[:, None, 3], gt[None, :, 1] + gt[None, :, 3]) - np.maximum(dt[:, None, 1], gt[None, :, 1])
    inter = np.clip(iw, 0.0, None) * np.clip(ih, 0.0, None)
    a_dt, a_gt = (dt[:, 2] * dt[:, 3])[:, None], (gt[:, 2] * gt[:, 3])[None, :]
    union = np.where(np.asarray(crowd, dtype=bool)[None, :], a_dt, a_dt + a_gt - inter)
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.where(union > 0, inter / union, 0.0)


def _last_argmax(vals):
    """per row: (max, index of the LAST entry equal to it) -- later candidates of equal IoU replace earlier ones"""
    n = vals.shape[1]
    idx = n - 1 - np.argmax(vals[:, ::-1], axis=1)
    return vals[np.arange(vals.shape[0]), idx], idx


def match_image(iou, gt_ignore, gt_crowd, gt_ids, dt_area, area_rng, thrs=IOU_THRS):
    """Greedy matching of one (image, category, area range).  iou [D,G] with detections in score order; returns
    dt_match [T,D] (matched annotation id, 0 = none), dt_ignore [T,D], gt_ignore in evaluation order [G]."""
    n_dt, n_gt = iou.shape
    n_thr = len(thrs)
    order = np.argsort(gt_ignore.astype(np.uint8), kind='mergesort')                  # regular ground truth first
    iou, g_ign, g_crowd, g_ids = iou[:, order], gt_ignore[order], gt_crowd[order], gt_ids[order]
    n_reg = int((~g_ign).sum())
    dt_match = np.zeros((n_thr, n_dt), dtype=np.int64)
    dt_ign = np.zeros((n_thr, n_dt), dtype=bool)
    taken = np.zeros((n_thr, n_gt), dtype=bool)
    floor = np.minimum(thrs, 1.0 - 1e-10)[:, None]
    rows = np.arange(n_thr)
    if n_gt:
        for d in range(n_dt):
            ok = (~taken | g_crowd[None, :]) & (iou[d][None, :] >= floor)
            vals = np.where(ok, iou[d][None, :], -1.0)
            pick = np.full(n_thr, -1, dtype=np.int64)
            if n_reg:
                best, idx = _last_argmax(vals[:, :n_reg])
                pick = np.where(best >= 0, idx, pick)
            if n_reg < n_gt:
                best, idx = _last_argmax(vals[:, n_reg:])
                pick = np.where((pick < 0) & (best >= 0), idx + n_reg, pick)
            hit = pick >= 0
            g = pick[hit]
            dt_match[rows[hit], d] = g_ids[g]
            dt_ign[rows[hit], d] = g_ign[g]
            taken[rows[hit], g] = True
    outside = (dt_area < area_rng[0]) | (dt_area > area_rng[1])
    dt_ign |= (dt_match == 0) & outside[None, :]
    return dt_match, dt_ign, g_ign


def evaluate_bbox(gt, dets, img_ids=None):
    """gt: FauxCoco.  dets: [N,7] rows (image id, x, y, w, h, score, category) -- train_detector.py:val_batch.
    Returns the 12 COCO summary numbers (STAT_NAMES order; stats[1] = AP at IoU .5 is what the driver's LR schedule
    watches); -1 where no ground truth falls into a bin."""
    dets = np.asarray(dets, dtype=np.float64).reshape(-1, 7)
    img_ids = sorted(set(range(gt.num_images) if img_ids is None else [int(i) for i in img_ids]))
    n_thr, n_rec, n_cat, n_area, n_md = len(IOU_THRS), len(REC_THRS), len(gt.cat_ids), len(AREA_RNG), len(MAX_DETS)
    precision = -np.ones((n_thr, n_rec, n_cat, n_area, n_md))
    recall = -np.ones((n_thr, n_cat, n_area, n_md))
    d_img, d_cat = dets[:, 0].astype(np.int64), dets[:, 6].astype(np.int64)
    by_img = {}
    for i in np.argsort(d_img, kind='mergesort'):
        by_img.setdefault(int(d_img[i]), []).append(i)
    cat_pos = {c: k for k, c in enumerate(gt.cat_ids)}
    # per category: list over images of (scores[D], dt_match[A][T,D], dt_ign[A][T,D], gt_ign[A][G])
    per_cat = [[] for _ in gt.cat_ids]
    for img in img_ids:
        rows = np.asarray(by_img.get(img, []), dtype=np.int64)
        g_cls = gt.cls[img]
        for c in sorted(set(g_cls.tolist()) | set(d_cat[rows].tolist())):
            if c not in cat_pos:
                continue
            gsel = np.nonzero(g_cls == c)[0]
            dsel = rows[d_cat[rows] == c]
            dsel = dsel[np.argsort(-dets[dsel, 5], kind='mergesort')][:MAX_DETS[-1]]
            boxes = dets[dsel, 1:5]
            iou = box_iou_xywh(boxes, gt.xywh[img][gsel], gt.crowd[img][gsel])
            dt_area = boxes[:, 2] * boxes[:, 3]
            entry = [dets[dsel, 5]]
            for rng in AREA_RNG:
                g_area = gt.area[img][gsel]
                ign = gt.crowd[img][gsel] | (g_area < rng[0]) | (g_area > rng[1])
                entry.append(match_image(iou, ign, gt.crowd[img][gsel], gt.ann_id[img][gsel], dt_area, rng))
            per_cat[cat_pos[c]].append(entry)
    for k, entries in enumerate(per_cat):
        if not entries:
            continue
        for a in range(n_area):
            for m, max_det in enumerate(MAX_DETS):
                scores = np.concatenate([e[0][:max_det] for e in entries])
                order = np.argsort(-scores, kind='mergesort')
                dtm = np.concatenate([e[1 + a][0][:, :max_det] for e in entries], 1)[:, order]
                dti = np.concatenate([e[1 + a][1][:, :max_det] for e in entries], 1)[:, order]
                n_pos = int(sum((~e[1 + a][2]).sum() for e in entries))
                if n_pos == 0:
                    continue
                tp = np.cumsum((dtm != 0) & ~dti, axis=1, dtype=np.float64)
                fp = np.cumsum((dtm == 0) & ~dti, axis=1, dtype=np.float64)
                n_dt = tp.shape[1]
                for t in range(n_thr):
                    rc = tp[t] / n_pos
                    pr = tp[t] / (fp[t] + tp[t] + np.spacing(1))
                    recall[t, k, a, m] = rc[-1] if n_dt else 0.0
                    pr = np.maximum.accumulate(pr[::-1])[::-1]                       # monotone from the right
                    at = np.searchsorted(rc, REC_THRS, side='left')
                    q = np.zeros(n_rec)
                    q[at < n_dt] = pr[at[at < n_dt]]
                    precision[t, :, k, a, m] = q

    def mean_valid(x):
        x = x[x > -1]
        return float(x.mean()) if x.size else -1.0

    t50, t75 = int(np.argmin(np.abs(IOU_THRS - 0.5))), int(np.argmin(np.abs(IOU_THRS - 0.75)))
    stats = [mean_valid(precision[:, :, :, 0, 2]), mean_valid(precision[t50, :, :, 0, 2]), mean_valid(precision[t75, :, :, 0, 2]),
             mean_valid(precision[:, :, :, 1, 2]), mean_valid(precision[:, :, :, 2, 2]), mean_valid(precision[:, :, :, 3, 2]),
             mean_valid(recall[:, :, 0, 0]), mean_valid(recall[:, :, 0, 1]), mean_valid(recall[:, :, 0, 2]),
             mean_valid(recall[:, :, 1, 2]), mean_valid(recall[:, :, 2, 2]), mean_valid(recall[:, :, 3, 2])]
    return np.asarray(stats)


def detection_rows(result, first_image, box_scale):
    """Rows (image index in the split, x, y, w, h, score, class) of one validation blob's detector Result -- the VG
    branch of the reference's val_batch (train_detector.py:184-205): boxes back to BOX_SCALE coordinates, widths with
    the +1 convention, blob-local image indices shifted by the blob's first image."""
    if result is None or result.boxes_assigned is None:
        return np.zeros((0, 7))
    boxes = result.boxes_assigned.detach().cpu().numpy().astype(np.float64) * box_scale
    boxes[:, 2:4] = boxes[:, 2:4] - boxes[:, 0:2] + 1
    return np.column_stack((result.im_inds.detach().cpu().numpy().astype(np.int64) + first_image, boxes,
                            result.obj_scores.detach().cpu().numpy(), result.obj_preds.detach().cpu().numpy()))


def summarize(stats):
    """the twelve lines cocoeval prints"""
    spec = [('Average Precision', '0.50:0.95', 'all', 100), ('Average Precision', '0.50', 'all', 100),
            ('Average Precision', '0.75', 'all', 100), ('Average Precision', '0.50:0.95', 'small', 100),
            ('Average Precision', '0.50:0.95', 'medium', 100), ('Average Precision', '0.50:0.95', 'large', 100),
            ('Average Recall', '0.50:0.95', 'all', 1), ('Average Recall', '0.50:0.95', 'all', 10),
            ('Average Recall', '0.50:0.95', 'all', 100), ('Average Recall', '0.50:0.95', 'small', 100),
            ('Average Recall', '0.50:0.95', 'medium', 100), ('Average Recall', '0.50:0.95', 'large', 100)]
    return '\n'.join(' {:<18} @[ IoU={:<9} | area={:>6s} | maxDets={:>3d} ] = {:0.3f}'.format(t, i, a, m, s)
                     for (t, i, a, m), s in zip(spec, stats))
