"""Detector-training host logic pinned to goldens produced by the reference's own functions
(tests/golden/make_golden.py): anchor_target_layer, the RoI sampling core of proposal_assignments_det, bbox_loss."""
import numpy as np
import torch


def test_anchor_target_layer_matches_reference(golden):
    from lib.fpn.anchor_targets import anchor_target_layer
    g = golden('det_train')
    for case in range(3):
        np.random.seed(int(g['at%d_seed' % case]))
        anchors, inds, targets, labels = anchor_target_layer(g['at%d_gt' % case], (592, 592))
        np.testing.assert_array_equal(inds, g['at%d_inds' % case])
        np.testing.assert_array_equal(labels, g['at%d_labels' % case])
        np.testing.assert_array_equal(anchors, g['at%d_anchors' % case])
        np.testing.assert_array_equal(targets, g['at%d_targets' % case])
        # the injectable RNG draws the same sequence as the global one
        a2, i2, t2, l2 = anchor_target_layer(g['at%d_gt' % case], (592, 592),
                                             rs=np.random.RandomState(int(g['at%d_seed' % case])))
        np.testing.assert_array_equal(i2, inds)
        assert labels.shape[0] <= 256 and (labels == 1).sum() <= 128


def test_roi_sampling_core_matches_reference(golden):
    from lib.fpn.proposal_assignments.proposal_assignments_det import _sel_inds
    g = golden('det_train')
    for case in range(3):
        np.random.seed(int(g['sel%d_seed' % case]))
        keep, num_fg = _sel_inds(g['sel%d_overlaps' % case].copy(), 0.5, 64, 256)
        np.testing.assert_array_equal(keep, g['sel%d_keep' % case])
        assert num_fg == int(g['sel%d_numfg' % case])


def test_bbox_loss_matches_reference(golden):
    from lib.fpn.box_utils import bbox_loss
    g = golden('det_train')
    loss = bbox_loss(torch.from_numpy(g['bl_prior']), torch.from_numpy(g['bl_deltas']), torch.from_numpy(g['bl_gt']))
    np.testing.assert_allclose(float(loss), float(g['bl_loss']), rtol=1e-6)


def test_proposal_assignments_det_properties():
    """whole function on CPU tensors: per image <= 256 RoIs, <= 64 foreground first, background labels 0, targets =
    the best-overlapping GT box, GT boxes themselves are candidates (IoU 1 -> foreground)"""
    from lib.fpn.proposal_assignments.proposal_assignments_det import proposal_assignments_det
    from lib.fpn.box_utils import bbox_overlaps
    rs = np.random.RandomState(3)
    gt = torch.tensor([[10., 10., 100., 120.], [200., 50., 400., 300.], [30., 300., 90., 380.], [5., 5., 500., 500.]])
    gtc = torch.tensor([[0, 7], [0, 3], [1, 9], [1, 2]])
    props = []
    for im in range(2):
        x1 = rs.uniform(0, 500, 400); y1 = rs.uniform(0, 500, 400)
        props.append(np.column_stack([np.full(400, im), x1, y1, x1 + rs.uniform(10, 90, 400), y1 + rs.uniform(10, 90, 400)]))
    rois_in = torch.from_numpy(np.concatenate(props).astype(np.float32))
    rois, labels, targets = proposal_assignments_det(rois_in, gt, gtc, 0, fg_thresh=0.5, rs=np.random.RandomState(5))
    assert rois.shape[1] == 5 and rois.shape[0] == labels.shape[0] == targets.shape[0]
    for im in range(2):
        m = rois[:, 0] == im
        assert 0 < int(m.sum()) <= 256
        lab = labels[m]
        nfg = int((lab > 0).sum())
        assert nfg <= 64 and (lab[:nfg] > 0).all() and (lab[nfg:] == 0).all()       # foreground first
        g_im = gt[gtc[:, 0] == im]
        iou = bbox_overlaps(rois[m][:, 1:], g_im)
        best = iou.max(1)[0]
        assert (best[:nfg] >= 0.5).all() and (best[nfg:] < 0.5).all()
        assert torch.equal(targets[m], g_im[iou.argmax(1)])


def test_blob_det_mode_carries_the_anchor_targets():
    """dataloaders/blob.py in 'det' training mode (reference blob.py:91-102,139-142): per image the sampled anchors,
    their matched GT boxes and labels; train_anchor_inds = (img, h, w, A) of the labelled anchors"""
    from dataloaders.synthetic import SyntheticVG, make_blob
    from lib.fpn.anchor_targets import anchor_target_layer
    ds = SyntheticVG(num_images=2, seed=5, n_boxes=7, n_rels=3)
    np.random.seed(42)
    blob = make_blob(ds, [0, 1], is_train=True, mode='det')
    assert blob.train_anchor_labels.shape[1] == 5 and blob.train_anchors.shape[1] == 8
    assert torch.equal(blob.train_anchor_inds, blob.train_anchor_labels[:, :4])
    assert blob.train_anchor_labels.shape[0] == blob.train_anchors.shape[0] <= 2 * 256
    assert set(blob.train_anchor_labels[:, 4].tolist()) <= {0, 1}
    np.random.seed(42)                                        # same draws -> same targets, image by image
    off = 0
    for i in range(2):
        d = ds[i]
        a, inds, t, l = anchor_target_layer(d['gt_boxes'].astype(np.float32) * d['scale'], (592, 592))
        n = inds.shape[0]
        got = blob.train_anchor_labels[off:off + n].numpy()
        assert (got[:, 0] == i).all()
        np.testing.assert_array_equal(got[:, 1:4], inds)
        np.testing.assert_array_equal(got[:, 4], l)
        np.testing.assert_array_equal(blob.train_anchors[off:off + n].numpy(), np.hstack((a, t)).astype(np.float32))
        off += n
    assert off == blob.train_anchor_labels.shape[0]
    assert len(blob[0]) == 8                                  # (..., proposals, train_anchor_inds)
    # 'rel' mode keeps the empty placeholder
    assert make_blob(ds, [0, 1], is_train=True, mode='rel').train_anchor_inds.shape == (0, 4)


def test_detector_losses_match_the_oracle_restatement():
    """lib/detector_loss.py (product, pure host glue) vs the loss part of oracle.model.detector_train_losses on the same
    random predictions -- two independent statements of models/train_detector.py:100-140"""
    from lib.detector_loss import detector_losses
    from lib.object_detector import Result
    from oracle import model as OM
    g = torch.Generator().manual_seed(0)
    n, C, k = 40, 11, 30
    scores = torch.randn(n, C, generator=g)
    deltas = torch.randn(n, C, 4, generator=g) * 0.3
    labels = torch.randint(0, C, (n,), generator=g)
    labels[::3] = 0
    x1y1 = torch.rand(n, 2, generator=g) * 300
    priors = torch.cat((x1y1, x1y1 + 20 + torch.rand(n, 2, generator=g) * 200), 1)
    g1 = torch.rand(n, 2, generator=g) * 300
    targets = torch.cat((g1, g1 + 20 + torch.rand(n, 2, generator=g) * 200), 1)
    rpn_scores = torch.randn(k, 2, generator=g)
    rpn_deltas = torch.randn(k, 4, generator=g) * 0.3
    a1 = torch.rand(k, 2, generator=g) * 300
    anchors = torch.cat((a1, a1 + 16 + torch.rand(k, 2, generator=g) * 100, a1 + 3, a1 + 40 + torch.rand(k, 2, generator=g) * 100), 1)
    tal = torch.cat((torch.zeros(k, 4, dtype=torch.long), torch.randint(0, 2, (k, 1), generator=g)), 1)
    res = Result(od_obj_dists=scores, od_box_deltas=deltas, od_obj_labels=labels, od_box_priors=priors,
                 od_box_targets=targets, rpn_scores=rpn_scores, rpn_box_deltas=rpn_deltas)
    got = detector_losses(res, tal, anchors)
    # the oracle's arithmetic on the same tensors
    valid = (labels != 0).nonzero().squeeze(1)
    fg, bg = valid.numel(), n - valid.numel()
    twod = valid * C + labels[valid]
    box = OM._bbox_loss(priors[valid], deltas.reshape(-1, 4)[twod], targets[valid]) * (2 * 4.0 * fg / (fg + bg + 1e-4))
    pos = (tal[:, -1] == 1).nonzero().squeeze(1)
    rbox = OM._bbox_loss(anchors[:, :4][pos], rpn_deltas[pos], anchors[:, 4:][pos]) * (2 * 2.0 * pos.numel() / (k + 1e-4))
    np.testing.assert_allclose(float(got['box_loss']), float(box), rtol=1e-6)
    np.testing.assert_allclose(float(got['rpn_box_loss']), float(rbox), rtol=1e-6)
    np.testing.assert_allclose(float(got['class_loss']), float(torch.nn.functional.cross_entropy(scores, labels)), rtol=1e-6)
    np.testing.assert_allclose(float(got['total']), float(got['class_loss'] + got['box_loss'] + got['rpn_class_loss'] + got['rpn_box_loss']), rtol=1e-6)
