"""lib/evaluation/det_map.py -- the COCO bbox protocol behind the detector driver's validation mAP (reference
models/train_detector.py:158-181).  pycocotools is not available, so the vectorised implementation is checked against
hand-computed cases and against a separate, deliberately literal loop restatement of the published matching and
accumulation rules written here (parity unpinned, see the module header)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'neural-motifs_amd'))

from lib.evaluation import det_map as DM  # noqa: E402


def xyxy_to_det(img, box, score, cat):
    x1, y1, x2, y2 = box
    return [img, x1, y1, x2 - x1 + 1, y2 - y1 + 1, score, cat]


def test_hand_computed_single_category():
    # two ground-truth boxes; detections: 0.9 exact hit, 0.8 miss, 0.7 exact hit
    gt = DM.FauxCoco([[3, 3]], [[[10, 10, 59, 59], [100, 100, 199, 179]]], num_classes=5, first_ann_id=1)
    dets = [xyxy_to_det(0, [10, 10, 59, 59], 0.9, 3), xyxy_to_det(0, [300, 300, 340, 340], 0.8, 3),
            xyxy_to_det(0, [100, 100, 199, 179], 0.7, 3)]
    stats = DM.evaluate_bbox(gt, dets)
    # recall .5 at precision 1 for the 51 points 0..0.50, recall 1 at precision 2/3 for the other 50; same at every IoU
    want = (51 * 1.0 + 50 * (2.0 / 3.0)) / 101
    assert abs(stats[1] - want) < 1e-9 and abs(stats[0] - want) < 1e-9 and abs(stats[2] - want) < 1e-9
    assert abs(stats[6] - 0.5) < 1e-12 and abs(stats[7] - 1.0) < 1e-12 and abs(stats[8] - 1.0) < 1e-12     # AR@1, @10, @100
    # areas: 50x50 = 2500 (medium), 100x80 = 8000 (medium): nothing small or large
    assert stats[3] == -1 and stats[5] == -1 and abs(stats[4] - want) < 1e-9
    assert 'Average Precision' in DM.summarize(stats) and len(DM.summarize(stats).splitlines()) == 12


def test_iou_threshold_sweep_and_duplicates():
    # one ground truth 100x100; a detection shifted by 10 px: IoU = 90*100 / (2*10000 - 9000) = 0.8182 -> counts up to .80
    gt = DM.FauxCoco([[1]], [[[0, 0, 99, 99]]], num_classes=3, first_ann_id=1)
    dets = [xyxy_to_det(0, [10, 0, 109, 99], 0.9, 1), xyxy_to_det(0, [0, 0, 99, 99], 0.5, 1)]
    stats = DM.evaluate_bbox(gt, dets)
    # thresholds .50-.80 (7 of 10): first detection is the TP, AP 1; thresholds .85-.95: first is FP, second TP: precision .5
    assert abs(stats[1] - 1.0) < 1e-12 and abs(stats[2] - 1.0) < 1e-12
    assert abs(stats[0] - (7 * 1.0 + 3 * 0.5) / 10) < 1e-9
    np.testing.assert_allclose(DM.box_iou_xywh([[10, 0, 100, 100]], [[0, 0, 100, 100]], [False]), [[9000.0 / 11000.0]])
    np.testing.assert_allclose(DM.box_iou_xywh([[10, 0, 100, 100]], [[0, 0, 100, 100]], [True]), [[0.9]])


def test_perfect_detections_and_the_annotation_id_zero_behaviour():
    rs = np.random.RandomState(0)
    classes, boxes = [], []
    for _ in range(6):
        n = rs.randint(1, 6)
        xy = rs.randint(0, 400, (n, 2))
        wh = rs.randint(8, 150, (n, 2))
        boxes.append(np.concatenate((xy, xy + wh), 1))
        classes.append(rs.randint(1, 4, n))
    dets = [xyxy_to_det(i, b, 0.5 + 0.01 * j, c) for i, (cs, bs) in enumerate(zip(classes, boxes))
            for j, (c, b) in enumerate(zip(cs, bs))]
    textbook = DM.evaluate_bbox(DM.FauxCoco(classes, boxes, 4, first_ann_id=1), dets)
    assert all(abs(s - 1.0) < 1e-12 or s == -1 for s in textbook[:6]) and abs(textbook[8] - 1.0) < 1e-12
    reference_like = DM.evaluate_bbox(DM.FauxCoco(classes, boxes, 4), dets)       # ids from 0, as visual_genome.py:114
    assert reference_like[1] < 1.0                        # the detection matched to annotation 0 is scored as a miss
    assert reference_like[1] > 0.8


# ---- a literal restatement of the published loops (slow; the checker) ------------------------------------------
def loop_eval(gt, dets, max_det=100):
    dets = np.asarray(dets, dtype=np.float64)
    T, R, K, A = len(DM.IOU_THRS), len(DM.REC_THRS), len(gt.cat_ids), len(DM.AREA_RNG)
    precision = -np.ones((T, R, K, A))
    for k, cat in enumerate(gt.cat_ids):
        for a, rng in enumerate(DM.AREA_RNG):
            scores, matches, ignores, n_pos = [], [], [], 0
            for img in range(gt.num_images):
                gi = [j for j in range(len(gt.cls[img])) if gt.cls[img][j] == cat]
                di = [j for j in range(len(dets)) if int(dets[j, 0]) == img and int(dets[j, 6]) == cat]
                if not gi and not di:
                    continue
                di = sorted(di, key=lambda j: -dets[j, 5])[:max_det]          # sorted() is stable, like mergesort
                g_ign = [bool(gt.area[img][j] < rng[0] or gt.area[img][j] > rng[1]) for j in gi]
                order = sorted(range(len(gi)), key=lambda j: g_ign[j])
                gi, g_ign = [gi[j] for j in order], [g_ign[j] for j in order]
                iou = DM.box_iou_xywh(dets[di, 1:5], gt.xywh[img][gi], [False] * len(gi)) if gi and di else np.zeros((len(di), len(gi)))
                gtm = np.zeros((T, len(gi)))
                dtm = np.zeros((T, len(di)))
                dtig = np.zeros((T, len(di)), dtype=bool)
                for t, thr in enumerate(DM.IOU_THRS):
                    for d in range(len(di)):
                        best, m = min(thr, 1 - 1e-10), -1
                        for g in range(len(gi)):
                            if gtm[t, g] > 0:
                                continue
                            if m > -1 and not g_ign[m] and g_ign[g]:
                                break
                            if iou[d, g] < best:
                                continue
                            best, m = iou[d, g], g
                        if m == -1:
                            continue
                        dtig[t, d] = g_ign[m]
                        dtm[t, d] = gt.ann_id[img][gi[m]]
                        gtm[t, m] = d + 1
                area = dets[di, 3] * dets[di, 4]
                out = (area < rng[0]) | (area > rng[1])
                dtig |= (dtm == 0) & out[None, :]
                scores.append(dets[di, 5]); matches.append(dtm); ignores.append(dtig)
                n_pos += sum(1 for x in g_ign if not x)
            if not scores or n_pos == 0:
                continue
            s = np.concatenate(scores)
            o = np.argsort(-s, kind='mergesort')
            dtm, dtig = np.concatenate(matches, 1)[:, o], np.concatenate(ignores, 1)[:, o]
            for t in range(T):
                tp = fp = 0.0
                rc, pr = [], []
                for d in range(dtm.shape[1]):
                    if not dtig[t, d]:
                        tp += dtm[t, d] != 0
                        fp += dtm[t, d] == 0
                    rc.append(tp / n_pos); pr.append(tp / (tp + fp + np.spacing(1)))
                for i in range(len(pr) - 1, 0, -1):
                    if pr[i] > pr[i - 1]:
                        pr[i - 1] = pr[i]
                q = np.zeros(R)
                for ri, r in enumerate(DM.REC_THRS):
                    pi = int(np.searchsorted(rc, r, side='left'))
                    if pi >= len(pr):
                        break
                    q[ri] = pr[pi]
                precision[t, :, k, a] = q

    def mv(x):
        x = x[x > -1]
        return float(x.mean()) if x.size else -1.0
    return [mv(precision[:, :, :, 0]), mv(precision[0, :, :, 0]), mv(precision[5, :, :, 0]), mv(precision[:, :, :, 1]),
            mv(precision[:, :, :, 2]), mv(precision[:, :, :, 3])]


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_vectorised_matches_the_literal_loops(seed):
    rs = np.random.RandomState(seed)
    n_img, n_cls = 5, 4
    classes, boxes, dets = [], [], []
    for i in range(n_img):
        n = rs.randint(0, 7)
        xy = rs.randint(0, 300, (n, 2))
        wh = rs.randint(5, 160, (n, 2))
        b = np.concatenate((xy, xy + wh), 1)
        classes.append(rs.randint(1, n_cls, n))
        boxes.append(b)
        for j in range(n):                                 # jittered copies (some duplicated), plus clutter
            for _ in range(rs.randint(0, 3)):
                jit = rs.randint(-12, 13, 4)
                cat = classes[-1][j] if rs.rand() < 0.8 else rs.randint(1, n_cls)
                dets.append(xyxy_to_det(i, np.maximum(b[j] + jit, 0) + [0, 0, 1, 1], round(rs.rand(), 2), cat))
        for _ in range(rs.randint(0, 4)):
            xy0 = rs.randint(0, 300, 2)
            dets.append(xyxy_to_det(i, list(xy0) + list(xy0 + rs.randint(5, 120, 2)), round(rs.rand(), 2), rs.randint(0, n_cls)))
    gt = DM.FauxCoco(classes, boxes, n_cls)
    fast = DM.evaluate_bbox(gt, dets)
    slow = loop_eval(gt, dets)
    np.testing.assert_allclose(fast[:6], slow, rtol=0, atol=1e-12)


def test_no_detections_and_no_ground_truth():
    gt = DM.FauxCoco([[1], []], [[[0, 0, 9, 9]], np.zeros((0, 4))], num_classes=3)
    assert DM.evaluate_bbox(gt, np.zeros((0, 7)))[1] == -1 or DM.evaluate_bbox(gt, np.zeros((0, 7)))[1] == 0.0
    only_bg = DM.evaluate_bbox(gt, [xyxy_to_det(1, [0, 0, 9, 9], 0.9, 0)])        # background detections are dropped
    assert only_bg[1] in (-1, 0.0)


def test_detection_rows_and_dataset_ground_truth():
    """the glue of the driver's val_epoch: Result -> rows, SyntheticVG.coco -> ground truth; feeding the ground truth
    back as detections (IM_SCALE coordinates, as the detector emits them) must score 1 apart from annotation 0"""
    import torch
    from config import BOX_SCALE, IM_SCALE
    from dataloaders.synthetic import SyntheticVG
    from lib.object_detector import Result
    ds = SyntheticVG(num_images=4, seed=3, n_boxes=5, n_rels=4)
    gt = ds.coco
    assert gt.num_images == 4 and gt.cat_ids[0] == 1 and gt.ann_id[1][0] == 5
    rows = []
    for first in (0, 2):                                    # two blobs of two images
        boxes = np.concatenate([ds.gt_boxes[first + i] for i in range(2)]) * (IM_SCALE / BOX_SCALE)
        res = Result(boxes_assigned=torch.from_numpy(boxes).float(),
                     im_inds=torch.tensor([0] * 5 + [1] * 5),
                     obj_scores=torch.linspace(0.9, 0.5, 10),
                     obj_preds=torch.from_numpy(np.concatenate([ds.gt_classes[first + i] for i in range(2)])))
        rows.append(DM.detection_rows(res, first, BOX_SCALE / IM_SCALE))
    rows.append(DM.detection_rows(Result(), 4, 1.0))
    dets = np.concatenate(rows, 0)
    assert dets.shape == (20, 7) and set(dets[:, 0].astype(int)) == {0, 1, 2, 3}
    np.testing.assert_allclose(dets[:5, 1:5], gt.xywh[0], atol=1e-3)
    stats = DM.evaluate_bbox(gt, dets, range(4))
    assert 0.9 < stats[1] <= 1.0


def test_hand_computed_three_images_two_categories():
    """a 3-image, 2-category fixture whose twelve COCO statistics are derived by hand below from the published
    protocol (cocoeval.py: greedy matching in score order per IoU threshold, non-ignored ground truth first, a matched ground
    truth is skipped by later detections, unmatched detections outside the area range are ignored; precision made monotone from
    the right and sampled at 101 recall points, 0 where the recall is never reached; categories without a non-ignored ground
    truth are left out of the mean).

    ground truth   img0: A cat1 100x100 (large), B cat2 30x30 (small)    img1: C cat1 60x60 (medium)    img2: D cat2 50x50 (medium)
    cat1 detections  d1 .9 img0 = A;  d2 .8 img1 = C shifted 6 px (IoU 3240/3960 = .818: a hit up to .80);  d3 .7 img2 50x50, no
                     cat1 ground truth there;  d4 .6 img1 = C exactly
    cat2 detections  e1 .95 img2 = D;  e2 .5 img0 = B shifted 5 px (IoU 750/1050 = .714: a hit up to .70)"""
    gt = DM.FauxCoco([[1, 2], [1], [2]],
                     [[[0, 0, 99, 99], [200, 200, 229, 229]], [[50, 50, 109, 109]], [[0, 0, 49, 49]]], num_classes=3, first_ann_id=1)
    dets = [xyxy_to_det(0, [0, 0, 99, 99], 0.9, 1), xyxy_to_det(1, [56, 50, 115, 109], 0.8, 1), xyxy_to_det(2, [10, 10, 59, 59], 0.7, 1),
            xyxy_to_det(1, [50, 50, 109, 109], 0.6, 1),
            xyxy_to_det(2, [0, 0, 49, 49], 0.95, 2), xyxy_to_det(0, [205, 200, 234, 229], 0.5, 2)]
    np.testing.assert_allclose(DM.box_iou_xywh([[56, 50, 60, 60]], [[50, 50, 60, 60]], [False]), [[3240.0 / 3960.0]])
    np.testing.assert_allclose(DM.box_iou_xywh([[205, 200, 30, 30]], [[200, 200, 30, 30]], [False]), [[750.0 / 1050.0]])
    s = DM.evaluate_bbox(gt, dets)
    half = 51.0 / 101.0            # precision 1 up to recall .5 (51 of the 101 points), recall never above .5
    # cat1, thresholds .50-.80 (7): d1 TP, d2 TP, d3 FP, d4 FP (C is taken) -> precision 1 up to recall 1: AP 1
    #       thresholds .85-.95 (3): d1 TP (r .5, p 1), d2 FP, d3 FP, d4 TP (r 1, p 2/4) -> 51 points at 1, 50 at .5: 76/101
    ap1 = (7 * 1.0 + 3 * (76.0 / 101.0)) / 10
    # cat2, thresholds .50-.70 (5): e1 TP, e2 TP: AP 1;  .75-.95 (5): e1 TP, e2 FP, recall stops at .5: 51/101
    ap2 = (5 * 1.0 + 5 * half) / 10
    want = {
        'AP': (ap1 + ap2) / 2, 'AP50': 1.0, 'AP75': (1.0 + half) / 2,
        # small (B only; cat1 has no small ground truth and drops out): e1 sits on D (ignored) and is ignored; e2 hits B up to .70,
        # above that it is an unmatched small detection = FP with no TP at all -> 0
        'APs': 0.5,
        # medium (C, D): cat1 -- d1 on A (ignored) is ignored, d3 50x50 is a medium FP; .50-.80: d2 TP first -> 1;
        # .85-.95: d2 FP, d3 FP, d4 TP -> precision 1/3 at every recall -> (7 + 3/3) / 10 = .8;  cat2 -- e1 TP, e2 ignored -> 1
        'APm': (0.8 + 1.0) / 2,
        # large (A): d1 TP; d2 / d4 on C (ignored) or unmatched-and-medium are ignored, d3 medium is ignored; cat2 drops out
        'APl': 1.0,
        # one detection per image and category: cat1 keeps d1, d2, d3 -> recall 1 up to .80, .5 above: .85; cat2 keeps both: .75
        'AR1': (0.85 + 0.75) / 2,
        'AR10': (1.0 + 0.75) / 2, 'AR100': (1.0 + 0.75) / 2,     # d4 recovers C above .80
        'ARs': 0.5, 'ARm': 1.0, 'ARl': 1.0}
    for name, got in zip(DM.STAT_NAMES, s):
        assert abs(got - want[name]) < 1e-9, (name, got, want[name])
