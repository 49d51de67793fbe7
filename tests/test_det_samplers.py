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
