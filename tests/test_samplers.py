"""Host samplers: property tests on CPU tensors (pure numpy/torch host code)."""
import numpy as np
import torch

from lib.fpn.proposal_assignments.proposal_assignments_gtbox import proposal_assignments_gtbox
from lib.fpn.proposal_assignments.rel_assignments import rel_assignments


def _gt(rs, n_im=3, n_box=9, n_rel=12):
    boxes, classes, rels = [], [], []
    for i in range(n_im):
        xy = rs.uniform(0, 400, (n_box, 2))
        b = np.concatenate((xy, xy + rs.uniform(30, 180, (n_box, 2))), 1)
        boxes.append(b)
        classes.append(np.column_stack((np.full(n_box, i), rs.randint(1, 151, n_box))))
        pairs = np.array([(a, c) for a in range(n_box) for c in range(n_box) if a != c])
        sel = pairs[rs.choice(len(pairs), n_rel, replace=False)]
        rels.append(np.column_stack((np.full(n_rel, i), sel, rs.randint(1, 51, n_rel))))
    return (torch.from_numpy(np.concatenate(boxes)).float(), torch.from_numpy(np.concatenate(classes)).long(),
            torch.from_numpy(np.concatenate(rels)).long())


def test_gtbox_sampler_properties():
    rs = np.random.RandomState(0)
    gt_boxes, gt_classes, gt_rels = _gt(rs)
    rois = torch.cat((gt_classes[:, :1].float(), gt_boxes), 1)
    _, labels, rel = proposal_assignments_gtbox(rois, gt_boxes, gt_classes, gt_rels, 0, rs=np.random.RandomState(1))
    rel = rel.numpy()
    assert torch.equal(labels, gt_classes[:, 1])
    n = rois.shape[0]
    key = rel[:, 0] * n * n + rel[:, 1] * n + rel[:, 2]
    assert np.all(np.diff(key) > 0)                                   # sorted, no duplicate pair
    assert np.all(rel[:, 1] != rel[:, 2])
    im = gt_classes[:, 0].numpy()
    assert np.all(im[rel[:, 1]] == rel[:, 0]) and np.all(im[rel[:, 2]] == rel[:, 0])
    # every annotated relation is present with its predicate; everything else is background
    offs = {i: int(np.where(im == i)[0][0]) for i in range(3)}
    want = {(int(r[0]), int(r[1]) + offs[int(r[0])], int(r[2]) + offs[int(r[0])]): int(r[3]) for r in gt_rels.numpy()}
    got = {(int(r[0]), int(r[1]), int(r[2])): int(r[3]) for r in rel}
    for k, v in want.items():
        assert got[k] == v
    assert all(v == 0 for k, v in got.items() if k not in want)
    assert len(got) == 3 * 9 * 8                                      # all ordered pairs (fewer than 256 per image)
    # reproducible with the same RandomState seed
    _, _, rel2 = proposal_assignments_gtbox(rois, gt_boxes, gt_classes, gt_rels, 0, rs=np.random.RandomState(1))
    assert np.array_equal(rel, rel2.numpy())


def test_sgdet_rel_assignments_properties():
    rs = np.random.RandomState(3)
    gt_boxes, gt_classes, gt_rels = _gt(rs, n_im=2, n_box=8, n_rel=10)
    # detections: jittered copies of the GT boxes (label kept) + random background boxes
    det_boxes, det_labels, det_im = [], [], []
    for i in range(2):
        sel = gt_classes[:, 0] == i
        jit = gt_boxes[sel] + torch.from_numpy(rs.uniform(-6, 6, (8, 4))).float()
        extra_xy = torch.from_numpy(rs.uniform(0, 400, (10, 2))).float()
        extra = torch.cat((extra_xy, extra_xy + 90), 1)
        det_boxes.append(torch.cat((jit, extra)))
        det_labels.append(torch.cat((gt_classes[sel, 1], torch.zeros(10, dtype=torch.long))))
        det_im.append(torch.full((18,), i, dtype=torch.long))
    det_boxes, det_labels, det_im = torch.cat(det_boxes), torch.cat(det_labels), torch.cat(det_im)
    rel = rel_assignments(det_im, det_boxes, det_labels, gt_boxes, gt_classes, gt_rels, 0, num_sample_per_gt=1,
                          filter_non_overlap=True, rs=np.random.RandomState(7)).numpy()
    assert rel.shape[1] == 4 and rel.dtype == np.int64
    for i in range(2):
        rows = rel[rel[:, 0] == i]
        assert 1 <= rows.shape[0] <= 64
        assert (rows[:, 3] > 0).sum() <= 16                          # REL_FG_FRACTION * 64
        assert np.all(det_im.numpy()[rows[:, 1]] == i) and np.all(det_im.numpy()[rows[:, 2]] == i)
        assert np.all(det_labels.numpy()[rows[:, 1]] > 0) and np.all(det_labels.numpy()[rows[:, 2]] > 0)
        order = rows[:, 1] * 1000 + rows[:, 2]
        assert np.all(np.diff(order) >= 0)
    # foreground rows reproduce an annotated predicate between the matched GT boxes
    offs = {0: 0, 1: 18}
    ann = {(int(r[0]), int(r[1]), int(r[2])): int(r[3]) for r in gt_rels.numpy()}
    fg = rel[rel[:, 3] > 0]
    assert fg.shape[0] > 0
    for im, s, o, p in fg:
        assert ann[(int(im), int(s) - offs[int(im)], int(o) - offs[int(im)])] == int(p)


# ------------------------------------------------------------------------------------------------------------------
# Draw-order equality with the REFERENCE's own samplers (tests/golden/rel_samplers.npz, written by
# tests/golden/make_golden_samplers.py from lib/fpn/proposal_assignments/{proposal_assignments_gtbox,rel_assignments}.py
# with numpy's global RNG seeded per case).  np.random.seed(s) and RandomState(s) are the same MT19937 stream.
# ------------------------------------------------------------------------------------------------------------------
def test_gtbox_sampler_equals_the_reference_draw_for_draw(golden):
    import torch
    from lib.fpn.proposal_assignments.proposal_assignments_gtbox import proposal_assignments_gtbox
    g = golden('rel_samplers')
    cases = sorted({k.split('_')[0] for k in g if k.startswith('gt')})
    assert len(cases) == 5
    for c in cases:
        rois, classes, rels = (torch.from_numpy(g[c + '_' + n]) for n in ('rois', 'classes', 'rels'))
        _, labels, rel_labels = proposal_assignments_gtbox(rois, rois[:, 1:], classes, rels, 0,
                                                           rs=np.random.RandomState(int(g[c + '_seed'])))
        np.testing.assert_array_equal(labels.numpy(), g[c + '_labels'])
        np.testing.assert_array_equal(rel_labels.numpy(), g[c + '_rel_labels'])


def test_rel_assignments_equals_the_reference_draw_for_draw(golden):
    import torch
    from lib.fpn.proposal_assignments.rel_assignments import rel_assignments
    g = golden('rel_samplers')
    cases = sorted({k.split('_')[0] for k in g if k.startswith('ra')})
    assert len(cases) == 5
    for c in cases:
        a = {n: torch.from_numpy(g['%s_%s' % (c, n)]) for n in ('im_inds', 'boxes', 'labels', 'gt_boxes', 'gt_classes', 'gt_rels')}
        got = rel_assignments(a['im_inds'], a['boxes'], a['labels'], a['gt_boxes'], a['gt_classes'], a['gt_rels'], 0,
                              filter_non_overlap=bool(g[c + '_fno']), num_sample_per_gt=int(g[c + '_per_gt']),
                              rs=np.random.RandomState(int(g[c + '_seed'])))
        np.testing.assert_array_equal(got.numpy(), g[c + '_rel_labels'])


def test_pinned_ring_staging_logic(monkeypatch):
    """lib/pytorch_misc._PinnedRing (the staging memory of every small host->device upload of a step): slices are
    64-byte aligned and disjoint inside a half, values and dtypes survive, a full half switches to the other one after
    recording events on the streams that used it, and a half is only reused after its events were waited for."""
    import torch
    from lib import pytorch_misc as pm

    log = []

    class FakeEvent(object):
        def record(self, stream):
            log.append(('record', stream))

        def synchronize(self):
            log.append(('sync',))

    monkeypatch.setattr(torch.cuda, 'Event', FakeEvent)
    monkeypatch.setattr(torch.cuda, 'current_stream', lambda device=None: 'stream0')
    ring = object.__new__(pm._PinnedRing)
    ring.half = 256
    ring.buf = torch.zeros(512, dtype=torch.uint8)
    ring.active, ring.pos = 0, 0
    ring.streams, ring.events = [set(), set()], [[], []]
    import threading
    ring.lock = threading.Lock()
    a = ring.stage(torch.arange(5, dtype=torch.int64), 'cpu')               # 40 B -> 64
    b = ring.stage(torch.tensor([True, False, True]), 'cpu')                # 3 B -> 64
    c = ring.stage(torch.arange(6, dtype=torch.float32).view(2, 3), 'cpu')  # 24 B -> 64
    assert a.tolist() == [0, 1, 2, 3, 4] and b.tolist() == [True, False, True] and c.shape == (2, 3) and c[1, 2] == 5
    assert ring.pos == 192 and ring.active == 0 and not log
    assert a.data_ptr() % 8 == 0 and b.data_ptr() - a.data_ptr() == 64 and c.data_ptr() - b.data_ptr() == 64
    d = ring.stage(torch.arange(20, dtype=torch.int64), 'cpu')              # 160 B: does not fit -> other half
    assert ring.active == 1 and ring.pos == 192 and log == [('record', 'stream0')]
    assert d.tolist() == list(range(20)) and a.tolist() == [0, 1, 2, 3, 4]   # the first half is untouched
    e = ring.stage(torch.arange(16, dtype=torch.int64), 'cpu')              # 128 B: back to half 0, after waiting for it
    assert ring.active == 0 and ring.pos == 128 and log == [('record', 'stream0'), ('record', 'stream0'), ('sync',)]
    assert e.tolist() == list(range(16))
    assert ring.stage(torch.zeros(0, 4), 'cpu').shape == (0, 4)
