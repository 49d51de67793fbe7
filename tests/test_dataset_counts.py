"""lib/get_dataset_counts.py (FrequencyBias statistics) against a literal loop restatement of the reference
(lib/get_dataset_counts.py:12-67; the reference module itself cannot be imported: it constructs VG(...) as a default
argument at import time), and the epoch reshuffle of the rank sampler."""
import numpy as np

from dataloaders.synthetic import SyntheticVG
from dataloaders.visual_genome import _RankSampler
from lib.fpn.box_intersections_cpu.bbox import bbox_overlaps
from lib.get_dataset_counts import box_filter, get_counts
from lib.sparse_targets import FrequencyBias


def loop_counts(ds, must_overlap):
    fg = np.zeros((ds.num_classes, ds.num_classes, ds.num_predicates), dtype=np.int64)
    bg = np.zeros((ds.num_classes, ds.num_classes), dtype=np.int64)
    for i in range(len(ds)):
        cls, rels, boxes = ds.gt_classes[i], ds.relationships[i], ds.gt_boxes[i]
        for s, o, p in rels:
            fg[cls[s], cls[o], p] += 1
        n = len(boxes)
        ov = bbox_overlaps(boxes.astype(np.float64), boxes.astype(np.float64)) > 0
        pairs = [(a, b) for a in range(n) for b in range(n) if a != b and (ov[a, b] or not must_overlap)]
        if must_overlap and not pairs:
            pairs = [(a, b) for a in range(n) for b in range(n) if a != b]
        for a, b in pairs:
            bg[cls[a], cls[b]] += 1
    return fg, bg


def test_counts_match_the_loop_restatement():
    ds = SyntheticVG(num_images=12, seed=3, n_boxes=[6, 9, 2, 14], n_rels=7)
    for must_overlap in (True, False):
        fg, bg = get_counts(ds, must_overlap=must_overlap)
        rfg, rbg = loop_counts(ds, must_overlap)
        np.testing.assert_array_equal(fg, rfg)
        np.testing.assert_array_equal(bg, rbg)
    assert fg.sum() == sum(len(r) for r in ds.relationships)


def test_box_filter_falls_back_to_all_pairs_when_nothing_overlaps():
    boxes = np.array([[0, 0, 10, 10], [100, 100, 120, 120], [300, 5, 320, 30]], dtype=np.float32)
    assert box_filter(boxes, must_overlap=True).shape == (6, 2)
    boxes[1] = [5, 5, 50, 50]
    np.testing.assert_array_equal(box_filter(boxes, must_overlap=True), [[0, 1], [1, 0]])


def test_frequency_bias_uses_the_given_counts():
    ds = SyntheticVG(num_images=6, seed=1, n_boxes=8, n_rels=10)
    fg, bg = get_counts(ds)
    fb = FrequencyBias(fg_matrix=fg, bg_matrix=bg, num_objs=ds.num_classes, num_rels=ds.num_predicates)
    full = fg.copy()
    full[:, :, 0] = bg + 1
    want = np.log(full / full.sum(2)[:, :, None] + 1e-3).reshape(-1, ds.num_predicates)
    np.testing.assert_allclose(fb.obj_baseline.weight.detach().numpy(), want.astype(np.float32), rtol=1e-6)


def test_rank_sampler_reshuffles_per_epoch_and_keeps_ranks_disjoint():
    s0, s1 = _RankSampler(40, 4, 0, 2, True, seed=7), _RankSampler(40, 4, 1, 2, True, seed=7)
    e0 = list(s0)
    assert not set(e0) & set(s1)
    s0.set_epoch(1)
    assert list(s0) != e0
