"""Host logic of the SGDet step that round 3 restructured to avoid device->host reads (CPU, through tests/cpu_shim.py):
   * filter_det_device (all foreground classes, one stable sort, device count) == filter_det (the reference's flow:
     classes above the score threshold, nonzero(), sort) == the oracle's restatement, on random score tables that include
     classes entirely below the threshold, ties, fewer rois than max_per_img and images with no detection at all;
   * apply_nms with all images enqueued before ONE read of the counts == the per-image loop;
   * rel_assignments with host mirrors and one packed copy == the same call on plain tensors."""
import numpy as np
import pytest
import torch


@pytest.fixture(scope='module')
def shim():
    import cpu_shim
    return cpu_shim.install()


def _boxes(rs, n, C):
    x1 = rs.uniform(0, 500, (n, C)); y1 = rs.uniform(0, 500, (n, C))
    w = rs.uniform(5, 90, (n, C)); h = rs.uniform(5, 90, (n, C))
    return torch.from_numpy(np.stack([x1, y1, x1 + w, y1 + h], 2).astype(np.float32))


@pytest.mark.parametrize('seed,n,C,max_per_img,kind', [(0, 40, 12, 10, 'mixed'), (1, 7, 6, 64, 'few rois'), (2, 60, 151, 64, 'vg size'),
                                                      (3, 30, 9, 8, 'ties'), (4, 25, 8, 16, 'nothing'), (5, 33, 10, 5, 'dead classes')])
def test_filter_det_device_equals_the_reference_flow(shim, seed, n, C, max_per_img, kind):
    from lib.object_detector import filter_det, filter_det_device
    rs = np.random.RandomState(seed)
    logits = rs.randn(n, C).astype(np.float32) * 3
    if kind == 'dead classes':
        logits[:, 3:7] -= 30.0                       # classes whose best score stays far below the threshold
    scores = torch.softmax(torch.from_numpy(logits), 1)
    if kind == 'ties':
        scores[5] = scores[4]
        scores[11] = scores[4]
    if kind == 'nothing':
        scores = torch.full((n, C), 1e-5)
        scores[:, 0] = 1.0 - 1e-5 * (C - 1)
    boxes = _boxes(rs, n, C)
    if kind == 'ties':
        boxes[5] = boxes[4]
    thresh = 0.01
    ref = filter_det(scores, boxes, start_ind=100, max_per_img=max_per_img, thresh=thresh)
    inds, sc, lab, cnt = filter_det_device(scores, boxes, start_ind=100, max_per_img=max_per_img, thresh=thresh)
    k = int(cnt)
    if ref is None:
        assert k == 0
        return
    assert k == ref[0].shape[0] and k <= max_per_img
    assert torch.equal(inds[:k], ref[0]) and torch.equal(lab[:k], ref[2])
    assert torch.equal(sc[:k], ref[1])
    assert bool((sc[:k] > thresh).all()) and (k == sc.shape[0] or k == max_per_img or float(sc[k]) <= thresh)


def test_filter_det_device_matches_the_oracle(shim):
    from lib.object_detector import filter_det_device
    from oracle import boxes as OB
    rs = np.random.RandomState(11)
    n, C = 50, 20
    scores = torch.softmax(torch.from_numpy(rs.randn(n, C).astype(np.float32) * 2.5), 1)
    boxes = _boxes(rs, n, C)
    inds, sc, lab, cnt = filter_det_device(scores, boxes, start_ind=0, max_per_img=30, thresh=0.01)
    k = int(cnt)
    ref = OB.filter_det(scores, boxes, start_ind=0, max_per_img=30, thresh=0.01)
    assert torch.equal(inds[:k], ref[0]) and torch.equal(lab[:k], ref[2]) and torch.equal(sc[:k], ref[1])


def test_apply_nms_batches_the_count_read(shim, monkeypatch):
    from lib.fpn.nms.functions import nms as N
    rs = np.random.RandomState(3)
    per = [37, 1, 64, 129]
    boxes = torch.cat([_boxes(rs, m, 1)[:, 0] for m in per], 0)
    scores = torch.from_numpy(rs.rand(sum(per)).astype(np.float32))
    reads = []
    orig_cpu = torch.Tensor.cpu
    monkeypatch.setattr(torch.Tensor, 'cpu', lambda self, *a, **k: (reads.append(tuple(self.shape)), orig_cpu(self, *a, **k))[1])
    inds, im_per = N.apply_nms(scores, boxes, pre_nms_topn=100, post_nms_topn=20, boxes_per_im=per, nms_thresh=0.5)
    counts = [r for r in reads if len(r) == 1]        # ([m, 4] reads are the CPU shim's own kernel emulation fetching its operands)
    assert counts == [(len(per),)], 'one device->host read of the result sizes for all images, got %s' % (reads,)
    monkeypatch.undo()
    s, want, want_per = 0, [], []
    for m in per:                                   # the reference's flow, image by image
        keep = N.apply_nms(scores[s:s + m], boxes[s:s + m], pre_nms_topn=100, post_nms_topn=20, nms_thresh=0.5)
        want.append(keep + s)
        want_per.append(keep.shape[0])
        s += m
    assert im_per == want_per and torch.equal(inds, torch.cat(want))


def test_rel_assignments_reads_the_device_once_and_uses_the_mirrors(shim, monkeypatch):
    from lib.fpn.proposal_assignments.rel_assignments import rel_assignments
    from lib.pytorch_misc import set_host
    rs = np.random.RandomState(5)
    n_im, per = 3, 9
    im = np.repeat(np.arange(n_im), per).astype(np.int64)
    b = _boxes(rs, n_im * per, 1)[:, 0]
    gt_b = b[::2].clone() + 1.0
    gt_c = np.column_stack((im[::2], rs.randint(1, 20, gt_b.shape[0]))).astype(np.int64)
    labels = torch.zeros(n_im * per, dtype=torch.int64)
    labels[::2] = torch.from_numpy(gt_c[:, 1])
    rels = []
    for i in range(n_im):
        k = int((gt_c[:, 0] == i).sum())
        for _ in range(4):
            a_, b_ = rs.choice(k, 2, replace=False)
            rels.append((i, a_, b_, rs.randint(1, 50)))
    gt_r = np.asarray(rels, dtype=np.int64)
    plain = rel_assignments(torch.from_numpy(im), b, labels, gt_b, torch.from_numpy(gt_c), torch.from_numpy(gt_r), 0,
                            filter_non_overlap=True, num_sample_per_gt=1, rs=np.random.RandomState(8))
    reads = []
    orig_cpu = torch.Tensor.cpu
    monkeypatch.setattr(torch.Tensor, 'cpu', lambda self, *a, **k: (reads.append(tuple(self.shape)), orig_cpu(self, *a, **k))[1])
    mirrored = rel_assignments(set_host(torch.from_numpy(im), im), b, labels, set_host(gt_b.clone(), gt_b.numpy()),
                               set_host(torch.from_numpy(gt_c), gt_c), set_host(torch.from_numpy(gt_r), gt_r), 0,
                               filter_non_overlap=True, num_sample_per_gt=1, rs=np.random.RandomState(8))
    assert reads == [(n_im * per, 5)], 'one packed copy (boxes + labels), got %s' % (reads,)
    assert torch.equal(plain, mirrored)
