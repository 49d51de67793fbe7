"""
oracle/model.py -- TEST INFRASTRUCTURE ONLY.

Functional CPU (torch fp32) restatement of the reference's ObjectDetector (VGG16 trunk,
`gtbox` and `refinerels` eval paths) and RelModel / LinearizedContext forward, written over a
plain state-dict whose keys are the reference's own, so a product model's
``state_dict()`` can be fed in unchanged.

  trunk / heads        lib/object_detector.py:78-138, :300-303, :623-633
  RPN head             lib/object_detector.py:488-612
  context ordering     lib/rel_model.py:31-61, :139-169; lib/pytorch_misc.py:278-287, :365-384
  LinearizedContext    lib/rel_model.py:171-296
  RelModel.forward     lib/rel_model.py:403-547
  union features       lib/get_union_boxes.py:15-93
  FrequencyBias        lib/sparse_targets.py:32-37
  filter_dets          lib/surgery.py:21-59

Third-party arithmetic (conv, GEMM, softmax, BN) comes from today's PyTorch CPU kernels; the
reference pinned PyTorch 0.3/cuDNN for it (README.md:21) and ships no vectors -> tolerance 1e-4.

Randomness is *injected*: every dropout / sampling site draws from the ``HostRNG`` passed in,
in the reference's call order, so the HIP path can reproduce the exact masks.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import boxes as B
from . import lstm as L
from . import native

VGG_CONVS = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)
VGG_POOLS_AFTER = (2, 7, 14, 21)
BATCHNORM_MOMENTUM = 0.01


class HostRNG(object):
    """Seeded host-side source of dropout masks (numpy MT19937), consumed in call order."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)

    def keep_mask(self, shape, keep_prob):
        """float32 {0,1} mask with P(1)=keep_prob."""
        return torch.from_numpy((self.rs.random_sample(tuple(shape)) < keep_prob).astype(np.float32))


def dropout(x, p, training, rng):
    if not training or p == 0.0:
        return x
    return x * rng.keep_mask(x.shape, 1.0 - p) / (1.0 - p)


# --------------------------------------------------------------------------- trunk + heads
# --------------------------------------------------------------------------- kink accounting (tests only)
# A ReLU input within rounding of 0 (or two pool candidates within rounding of each other) may fall on either side in two
# correct fp32 evaluations; the forward value is unaffected (it is ~0 / equal either way) but the BACKWARD mask differs,
# which changes one row of the weight gradient below the unit by O(1) of that row.  To compare gradients at a true
# relative tolerance the parity tests hand the product's own masks to the oracle: TAPS = {'force': {name: mask}} makes
# the named ReLU / max-pool use that mask, and records under TAPS['flips'][name] how many units differ from the oracle's
# own decision and how far the largest of them is from the kink (relative to the tensor's largest magnitude) -- the
# tests assert that every such unit is a genuine near-kink case.  TAPS = None (the default): plain F.relu / F.max_pool2d.
TAPS = None


def _relu(x, name):
    if TAPS is None:
        return F.relu(x)
    forced = TAPS.get('force', {}).get(name)
    own = x > 0
    TAPS.setdefault('mask', {})[name] = own
    if forced is None:
        return F.relu(x)
    forced = forced.reshape(x.shape)
    diff = own != forced
    n = int(diff.sum())
    far = float(x.detach().abs()[diff].max() / x.detach().abs().max()) if n else 0.0
    TAPS.setdefault('flips', {})[name] = (n, far, int(x.numel()))
    return x * forced.to(x.dtype)


def _relu6(x, name):
    """F.relu6 whose BACKWARD mask (0 < x < 6) can be forced; the forward value is the clamp either way"""
    if TAPS is None:
        return F.relu6(x)
    forced = TAPS.get('force', {}).get(name)
    own = (x > 0) & (x < 6)
    if forced is None:
        return F.relu6(x)
    forced = forced.reshape(x.shape)
    diff = own != forced
    n = int(diff.sum())
    xd = x.detach()
    far = float(torch.minimum(xd.abs(), (xd - 6).abs())[diff].max() / xd.abs().max()) if n else 0.0
    TAPS.setdefault('flips', {})[name] = (n, far, int(x.numel()))
    lin = x * forced.to(x.dtype)
    return lin + (F.relu6(xd) - lin.detach())


def _max_pool_2x2(x, name):
    """F.max_pool2d(x, 2, 2) on NCHW; a forced table [N,Ho,Wo,C] (dy * 2 + dx) decides the routing"""
    forced = None if TAPS is None else TAPS.get('force', {}).get(name)
    if forced is None:
        return F.max_pool2d(x, 2, 2)
    N, C, H, W = x.shape
    Ho, Wo = H // 2, W // 2
    arg = forced.reshape(N, Ho, Wo, C).permute(0, 3, 1, 2).long()
    yy = 2 * torch.arange(Ho).view(1, 1, Ho, 1) + arg // 2
    xx = 2 * torch.arange(Wo).view(1, 1, 1, Wo) + arg % 2
    out = x.flatten(2).gather(2, (yy * W + xx).flatten(2)).view(N, C, Ho, Wo)
    gap = (F.max_pool2d(x.detach(), 2, 2) - out.detach()).abs()
    TAPS.setdefault('flips', {})[name] = (int((gap > 0).sum()), float(gap.max() / x.detach().abs().max()), int(out.numel()))
    return out


def _max_pool_3x3s2p1(x, name):
    """F.max_pool2d(x, 3, 2, 1) on NCHW; with a forced arg-max table [N,Ho,Wo,C] (values ky*3+kx, the product's
    mh_bn_pool_fwd output) the routing follows the table instead of the oracle's own maxima"""
    forced = None if TAPS is None else TAPS.get('force', {}).get(name)
    if forced is None:
        return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    N, C, H, W = x.shape
    Ho, Wo = H // 2, W // 2
    arg = forced.reshape(N, Ho, Wo, C).permute(0, 3, 1, 2).long()
    yo = torch.arange(Ho).view(1, 1, Ho, 1)
    xo = torch.arange(Wo).view(1, 1, 1, Wo)
    yy, xx = 2 * yo - 1 + arg // 3, 2 * xo - 1 + arg % 3
    assert bool(((yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)).all()), 'forced pool index outside the image'
    out = x.flatten(2).gather(2, (yy * W + xx).flatten(2)).view(N, C, Ho, Wo)
    own = F.max_pool2d(x.detach(), kernel_size=3, stride=2, padding=1)
    gap = (own - out.detach()).abs()
    TAPS.setdefault('flips', {})[name] = (int((gap > 0).sum()), float(gap.max() / x.detach().abs().max()), int(out.numel()))
    return out


def vgg_features(sd, x, prefix='detector.features.'):
    for idx in VGG_CONVS:
        x = _relu(F.conv2d(x, sd[prefix + '%d.weight' % idx], sd[prefix + '%d.bias' % idx], padding=1), prefix + '%d' % idx)
        if idx in VGG_POOLS_AFTER:
            x = _max_pool_2x2(x, prefix + 'pool%d' % idx)
    return x


# --------------------------------------------------------------------------- ResNet-101 detector branch
RESNET_LAYERS = (('layer1', 3, 1), ('layer2', 4, 2), ('layer3', 23, 2))


def _bn(sd, x, prefix, training, momentum=0.1):
    """nn.BatchNorm2d (torchvision default momentum 0.1, eps 1e-5); in train mode batch statistics are used AND the
    running statistics in `sd` are updated in place -- the reference runs the frozen detector in train() mode
    (models/train_rels.py:101)"""
    return F.batch_norm(x, sd[prefix + 'running_mean'], sd[prefix + 'running_var'], sd[prefix + 'weight'],
                        sd[prefix + 'bias'], training, momentum, 1e-5)


def resnet_bottleneck(sd, x, p, stride, training):
    """Bottleneck.forward, lib/resnet.py:25-46 (stride on the 3x3; `p` = state-dict prefix of the block)"""
    out = F.relu(_bn(sd, F.conv2d(x, sd[p + 'conv1.weight']), p + 'bn1.', training))
    out = F.relu(_bn(sd, F.conv2d(out, sd[p + 'conv2.weight'], None, stride=stride, padding=1), p + 'bn2.', training))
    out = _bn(sd, F.conv2d(out, sd[p + 'conv3.weight']), p + 'bn3.', training)
    if p + 'downsample.0.weight' in sd:
        x = _bn(sd, F.conv2d(x, sd[p + 'downsample.0.weight'], None, stride=stride), p + 'downsample.1.', training)
    return F.relu(out + x)


def resnet_features(sd, x, training, prefix='detector.features.', taps=None):
    """ObjectDetector.feature_map, ResNet branch (lib/object_detector.py:119-127) over torchvision resnet101 minus
    layer4 (load_resnet :615-620).  `taps` (dict) receives the input of every block ('layer3.22' -> tensor) and the
    stem output ('stem')."""
    x = F.conv2d(x, sd[prefix + 'conv1.weight'], None, stride=2, padding=3)
    x = F.relu(_bn(sd, x, prefix + 'bn1.', training))
    x = F.max_pool2d(x, 3, 2, 1)
    if taps is not None:
        taps['stem'] = x
    for name, blocks, stride in RESNET_LAYERS:
        for b in range(blocks):
            if taps is not None:
                taps['%s.%d' % (name, b)] = x
            x = resnet_bottleneck(sd, x, '%s%s.%d.' % (prefix, name, b), stride if b == 0 else 1, training)
    return x


def resnet_compress(sd, fmap, training, prefix='detector.compress.'):
    """Conv2d(1024,256,1) -> ReLU -> BatchNorm2d(256)  (lib/object_detector.py:84-88)"""
    y = F.relu(F.conv2d(fmap, sd[prefix + '0.weight'], sd[prefix + '0.bias']))
    return _bn(sd, y, prefix + '2.', training)


def resnet_l4_head(sd, x, prefix, training, relu_end=False):
    """the relation model's RoI feature extractor in the ResNet configuration,
    nn.Sequential(resnet_l4(relu_end=False), nn.AvgPool2d(7), Flattener()) (lib/rel_model.py:360-365, lib/resnet.py:126-133):
    torchvision's layer4 with the stride removed from block 0, the last block without its final ReLU, then the 7x7 mean.
    x [n,1024,7,7] -> [n,2048]; `prefix` e.g. 'roi_fmap.0.'"""
    m = BATCHNORM_MOMENTUM      # lib/resnet.py:14-19 (the reference's own Bottleneck), not torchvision's 0.1
    for b in range(3):
        p = '%s%d.' % (prefix, b)
        last = (b == 2) and not relu_end
        # ReLU sites are named after the BatchNorm in front of them (`<prefix><block>.bn1` ...): forceable through TAPS
        out = _relu(_bn(sd, F.conv2d(x, sd[p + 'conv1.weight']), p + 'bn1.', training, m), p + 'bn1')
        out = _relu(_bn(sd, F.conv2d(out, sd[p + 'conv2.weight'], None, stride=1, padding=1), p + 'bn2.', training, m), p + 'bn2')
        out = _bn(sd, F.conv2d(out, sd[p + 'conv3.weight']), p + 'bn3.', training, m)
        if p + 'downsample.0.weight' in sd:
            x = _bn(sd, F.conv2d(x, sd[p + 'downsample.0.weight']), p + 'downsample.1.', training, m)
        x = out + x if last else _relu(out + x, p + 'bn3')
    return x.mean((2, 3))


_ALPHA_PRIME = -1.7580993408473766      # -selu_scale * selu_alpha: the value a dropped unit takes (torch.nn.AlphaDropout)


def alpha_dropout(x, p, training, rng):
    """torch.nn.functional.alpha_dropout with the mask drawn from `rng` (reference: nn.AlphaDropout(p=0.05),
    lib/object_detector.py:92,95): y = a (x m + alpha' (1 - m)) + b, a = ((1-p)(1 + p alpha'^2))^-1/2, b = -a alpha' p"""
    if not training or p == 0.0 or rng is None:
        return x
    m = rng.keep_mask(x.shape, 1.0 - p).to(x.dtype)
    a = ((1.0 - p) * (1.0 + p * _ALPHA_PRIME ** 2)) ** -0.5
    return a * (x * m + _ALPHA_PRIME * (1.0 - m)) + (-a * _ALPHA_PRIME * p)


def resnet_roi_head(sd, pooled, prefix='detector.roi_fmap.', training=False, rng=None):
    """Linear -> SELU -> AlphaDropout -> Linear -> SELU -> AlphaDropout (:89-96); the dropout masks are drawn from `rng` in
    train mode (identity without one: eval mode)"""
    y = alpha_dropout(F.selu(F.linear(pooled, sd[prefix + '0.weight'], sd[prefix + '0.bias'])), 0.05, training, rng)
    return alpha_dropout(F.selu(F.linear(y, sd[prefix + '3.weight'], sd[prefix + '3.bias'])), 0.05, training, rng)


class _RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, ph, pw, scale):
        ctx.save_for_backward(rois)
        ctx.meta = (tuple(feat.shape), scale)
        # (an fp64 evaluation -- tests measuring the fp32 rounding floor -- interpolates the fp32-rounded map: the reference op is fp32)
        return torch.from_numpy(native.roi_align_fwd(feat.detach().float().numpy(), rois.float().numpy(), ph, pw, scale)).to(feat.dtype)

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        shape, scale = ctx.meta
        return torch.from_numpy(native.roi_align_bwd(g.contiguous().numpy(), rois.numpy(), shape, scale)), \
            None, None, None, None


def roi_align(feat, rois, ph=7, pw=7, spatial_scale=1.0 / 16):
    return _RoIAlignFn.apply(feat, rois.detach(), ph, pw, spatial_scale)


def vgg_classifier(sd, x, prefix, training, rng, use_dropout=True, use_relu=True):
    """load_vgg(...).classifier (object_detector.py:623-633): fc6,ReLU,Dropout,fc7[,ReLU[,Dropout]]"""
    x = _relu(F.linear(x, sd[prefix + '0.weight'], sd[prefix + '0.bias']), prefix + '0')
    x = dropout(x, 0.5, training, rng)
    x = F.linear(x, sd[prefix + '3.weight'], sd[prefix + '3.bias'])
    if use_relu:
        x = _relu(x, prefix + '3')
        if use_dropout:
            x = dropout(x, 0.5, training, rng)
    return x


def rpn_head(sd, fmap, prefix='detector.rpn_head.'):
    """RPNHead.forward (object_detector.py:521-531) -> [B,h,w,A,6]"""
    x = F.conv2d(fmap, sd[prefix + 'conv.0.weight'], sd[prefix + 'conv.0.bias'], padding=1)
    x = _relu6(x, prefix + 'conv.0')
    x = F.conv2d(x, sd[prefix + 'conv.2.weight'], sd[prefix + 'conv.2.bias'])
    b, nc, h, w = x.shape
    x = x.view(b, nc, -1).transpose(1, 2).contiguous().view(b, h, w, nc)
    return x.view(b, h, w, nc // 6, 6)


def detector_forward(sd, cfg, x, im_sizes, image_offset, gt_boxes, gt_classes, training, rng,
                     rel_labels=None):
    """
    ObjectDetector.forward for mode 'gtbox' (sgcls / predcls) and eval 'refinerels' (sgdet).
    Returns a dict with the Result fields the RelModel reads.
    `rel_labels`: the sampled relation rows (host sampler output) for gtbox training.
    """
    resnet = cfg.get('use_resnet', False)
    fmap = resnet_features(sd, x, training) if resnet else vgg_features(sd, x)
    res = {'fmap': fmap}
    if cfg['mode'] in ('sgcls', 'predcls'):
        im_inds = gt_classes[:, 0] - image_offset
        rois = torch.cat((im_inds.to(gt_boxes.dtype)[:, None], gt_boxes), 1)
        if resnet:      # lib/object_detector.py:129-138 with the compress conv in front of RoIAlign (:84-96); the AlphaDropout of
            # the SELU head draws from `rng` only with cfg['resnet_alpha_dropout'] (the parity tests of the relation model run the
            # frozen detector's dropout off on both sides)
            obj_fmap = resnet_roi_head(sd, roi_align(resnet_compress(sd, fmap, training), rois).view(rois.size(0), -1),
                                       training=training and cfg.get('resnet_alpha_dropout', False), rng=rng)
        else:
            obj_fmap = vgg_classifier(sd, roi_align(fmap, rois).view(rois.size(0), -1),
                                      'detector.roi_fmap.', training, rng)
        od_obj_dists = F.linear(obj_fmap, sd['detector.score_fc.weight'], sd['detector.score_fc.bias'])
        res.update(im_inds=rois[:, 0].long() + image_offset, rm_obj_dists=od_obj_dists,
                   od_obj_dists=od_obj_dists, rm_box_priors=rois[:, 1:], rm_obj_labels=gt_classes[:, 1],
                   rel_labels=rel_labels, boxes_all=None, obj_fmap=obj_fmap)
        return res
    # sgdet eval: RPN -> proposals -> RoI head -> per-class NMS
    assert not training, "oracle covers sgdet eval only"
    feats = rpn_head(sd, fmap)
    rois = B.roi_proposals(feats, sd['detector.rpn_head.anchors'], im_sizes, nms_thresh=0.7,
                           pre_nms_topn=6000, post_nms_topn=1000)
    obj_fmap = vgg_classifier(sd, roi_align(fmap, rois).view(rois.size(0), -1),
                              'detector.roi_fmap.', training, rng)
    od_obj_dists = F.linear(obj_fmap, sd['detector.score_fc.weight'], sd['detector.score_fc.bias'])
    od_box_deltas = F.linear(obj_fmap, sd['detector.bbox_fc.weight'], sd['detector.bbox_fc.bias']).view(
        -1, od_obj_dists.size(1), 4)
    out = B.nms_boxes(od_obj_dists, rois, od_box_deltas, im_sizes, max_per_img=cfg.get('max_per_img', 64),
                      thresh=cfg.get('thresh', 0.01))
    if out is None:
        return None
    nms_inds, nms_scores, nms_preds, nms_boxes_assign, nms_boxes, nms_imgs = out
    res.update(im_inds=nms_imgs + image_offset, rm_obj_dists=od_obj_dists[nms_inds],
               od_obj_dists=od_obj_dists, rm_box_priors=nms_boxes[:, 0], rm_obj_labels=None,
               rel_labels=None, boxes_all=nms_boxes, obj_fmap=obj_fmap[nms_inds], rois=rois,
               nms_inds=nms_inds, obj_scores=nms_scores, obj_preds=nms_preds)
    return res


def sgdet_gt_matching(box_priors, im_inds, gt_boxes, gt_classes):
    """Labels of detections in SGDet training (lib/object_detector.py:319-326): the class of the GT box of the SAME image
    with the largest IoU, background (0) when that IoU is below 0.5."""
    ov = B.bbox_overlaps(box_priors, gt_boxes).clone()
    ov[im_inds[:, None] != gt_classes[None, :, 0]] = 0.0
    max_ov, arg = ov.max(1)
    labels = gt_classes[:, 1][arg].clone()
    labels[max_ov < 0.5] = 0
    return labels


# --------------------------------------------------------------------------- detector pre-training step (§8f rank 1)
def _bbox_loss(prior_boxes, deltas, gt_boxes, eps=1e-4):
    """lib/fpn/box_utils.py:8-25"""
    pc, gc = B.center_size(prior_boxes), B.center_size(gt_boxes)
    targets = torch.cat(((gc[:, :2] - pc[:, :2]) / pc[:, 2:], torch.log(gc[:, 2:]) - torch.log(pc[:, 2:])), 1)
    return F.smooth_l1_loss(deltas, targets, reduction='sum') / (eps + pc.size(0))


def detector_train_losses(sd, x, rois, labels, bbox_targets, train_anchor_labels, train_anchors, rng,
                          fg_fraction=0.25, rpn_fg_fraction=0.5):
    """ObjectDetector.forward in mode 'rpntrain', training (lib/object_detector.py:224-361) followed by the four
    losses of models/train_detector.py:100-140.  `rois`, `labels`, `bbox_targets` are the outputs of the host sampler
    (proposal_assignments_det, pinned separately by goldens); everything differentiable is restated here."""
    fmap = vgg_features(sd, x)
    feats = rpn_head(sd, fmap)
    tai = train_anchor_labels[:, :-1]
    regions = feats[tai[:, 0], tai[:, 1], tai[:, 2], tai[:, 3]]
    rpn_scores, rpn_box_deltas = regions[:, :2], regions[:, 2:]
    obj_fmap = vgg_classifier(sd, roi_align(fmap, rois).view(rois.size(0), -1), 'detector.roi_fmap.', True, rng)
    scores = F.linear(obj_fmap, sd['detector.score_fc.weight'], sd['detector.score_fc.bias'])
    box_deltas = F.linear(obj_fmap, sd['detector.bbox_fc.weight'], sd['detector.bbox_fc.bias']).view(
        -1, scores.size(1), 4)
    valid = (labels != 0).nonzero().squeeze(1)
    fg, bg = valid.size(0), labels.size(0) - valid.size(0)
    out = {'class_loss': F.cross_entropy(scores, labels)}
    twod = valid * box_deltas.size(1) + labels[valid]
    out['box_loss'] = _bbox_loss(rois[:, 1:][valid], box_deltas.reshape(-1, 4)[twod], bbox_targets[valid]) * (
        2 * (1. / fg_fraction) * fg / (fg + bg + 1e-4))
    al = train_anchor_labels[:, -1]
    pos = (al == 1).nonzero().squeeze(1)
    out['rpn_class_loss'] = F.cross_entropy(rpn_scores, al)
    out['rpn_box_loss'] = _bbox_loss(train_anchors[:, :4][pos], rpn_box_deltas[pos], train_anchors[:, 4:][pos]) * (
        2 * (1. / rpn_fg_fraction) * pos.size(0) / (al.size(0) + 1e-4))
    out['total'] = out['class_loss'] + out['box_loss'] + out['rpn_class_loss'] + out['rpn_box_loss']
    out.update(scores=scores, box_deltas=box_deltas, rpn_scores=rpn_scores, rpn_box_deltas=rpn_box_deltas, fmap=fmap)
    return out


# --------------------------------------------------------------------------- context ordering
def transpose_packed_sequence_inds(lengths):
    """lib/pytorch_misc.py:365-384"""
    new_inds, new_lens = [], []
    cum_add = np.cumsum([0] + list(lengths))
    max_len = lengths[0]
    length_pointer = len(lengths) - 1
    for i in range(max_len):
        while length_pointer > 0 and lengths[length_pointer] <= i:
            length_pointer -= 1
        new_inds.append(cum_add[:(length_pointer + 1)].copy())
        cum_add[:(length_pointer + 1)] += 1
        new_lens.append(length_pointer + 1)
    return np.concatenate(new_inds, 0), new_lens


def sort_by_score(im_inds, scores):
    """lib/rel_model.py:31-61 (descending sort with (value desc, index asc) tie rule)"""
    num_im = int(im_inds[-1]) + 1
    rois_per_image = scores.new_zeros(num_im)
    lengths = []
    for i, s, e in B.enumerate_by_image(im_inds.numpy()):
        rois_per_image[i] = 2 * (s - e) * num_im + i
        lengths.append(e - s)
    lengths = sorted(lengths, reverse=True)
    inds, ls_transposed = transpose_packed_sequence_inds(lengths)
    inds = torch.from_numpy(np.asarray(inds, dtype=np.int64))
    roi_order = scores - 2 * rois_per_image[im_inds]
    _, perm = torch.sort(roi_order, dim=0, descending=True, stable=True)
    perm = perm[inds]
    _, inv_perm = torch.sort(perm)
    return perm, inv_perm, ls_transposed


def sort_rois(order, batch_idx, confidence, box_priors):
    """LinearizedContext.sort_rois (rel_model.py:139-162); 'random' is not reproducible -> unsupported."""
    cxcywh = B.center_size(box_priors)
    if order == 'size':
        sizes = cxcywh[:, 2] * cxcywh[:, 3]
        scores = sizes / (sizes.max() + 1)
    elif order == 'confidence':
        scores = confidence
    elif order == 'leftright':
        centers = cxcywh[:, 0]
        scores = centers / (centers.max() + 1)
    else:
        raise ValueError(order)
    return sort_by_score(batch_idx, scores)


def _lstm_mask(rng, L_, Bsz, H, p, training):
    """alternating_highway_lstm.py:279-287: bernoulli(1-p)/(1-p) in train, ones in eval."""
    if not training:
        return torch.ones(L_, Bsz, H)
    return rng.keep_mask((L_, Bsz, H), 1.0 - p) / (1.0 - p)


def context_forward(sd, cfg, obj_fmaps, obj_logits, im_inds, obj_labels, box_priors, boxes_per_cls,
                    training, rng, prefix='context.'):
    """LinearizedContext.forward (rel_model.py:236-296), including the nl_obj == 0 (decoder_lin / one-hot, :259-284)
    and nl_edge == 0 (edge_ctx None) baseline branches; the SGDet-eval NMS of the nl_obj == 0 branch (:266-281) is
    restated with the greedy per-class NMS of oracle/boxes.py."""
    H = cfg['hidden_dim']
    num_classes = sd[prefix + 'obj_embed.weight'].shape[0]
    obj_embed = F.softmax(obj_logits, dim=1) @ sd[prefix + 'obj_embed.weight']
    cs = B.center_size(box_priors)
    pe = F.batch_norm(cs, sd[prefix + 'pos_embed.0.running_mean'], sd[prefix + 'pos_embed.0.running_var'],
                      sd[prefix + 'pos_embed.0.weight'], sd[prefix + 'pos_embed.0.bias'],
                      training=training, momentum=BATCHNORM_MOMENTUM / 10.0, eps=1e-5)
    pe = _relu(F.linear(pe, sd[prefix + 'pos_embed.1.weight'], sd[prefix + 'pos_embed.1.bias']), prefix + 'pos_embed.1')
    pos_embed = dropout(pe, 0.1, training, rng)
    obj_pre_rep = torch.cat((obj_fmaps, obj_embed, pos_embed), 1)

    if cfg['nl_obj'] > 0:
        # ---- obj_ctx (rel_model.py:197-234)
        confidence = F.softmax(obj_logits, dim=1).detach()[:, 1:].max(1)[0]
        perm, inv_perm, ls_transposed = sort_rois(cfg['order'], im_inds, confidence, box_priors)
        obj_inp_rep = obj_pre_rep[perm].contiguous()
        mask = _lstm_mask(rng, cfg['nl_obj'], int(ls_transposed[0]), H, cfg['rec_dropout'], training)
        encoder_rep = L.alternating_highway_lstm(obj_inp_rep, ls_transposed, sd[prefix + 'obj_ctx_rnn.weight'],
                                                 sd[prefix + 'obj_ctx_rnn.bias'], H, cfg['nl_obj'], training, mask)
        if cfg['mode'] != 'predcls':
            dec_in = torch.cat((obj_inp_rep, encoder_rep), 1) if cfg['pass_in_obj_feats_to_decoder'] else encoder_rep
            dec_p = {k[len(prefix + 'decoder_rnn.'):]: v for k, v in sd.items() if k.startswith(prefix + 'decoder_rnn.')}
            dmask = None
            if cfg['rec_dropout'] > 0.0:
                # decoder_rnn.py:13-37 -- drawn in eval mode too (it is simply not applied there)
                dmask = rng.keep_mask((int(ls_transposed[0]), H), 1.0 - cfg['rec_dropout']) / (1.0 - cfg['rec_dropout'])
            obj_dists2, obj_preds = L.decoder_forward(
                dec_p, dec_in, ls_transposed, H, training,
                labels=obj_labels[perm] if obj_labels is not None else None,
                boxes_for_nms=boxes_per_cls[perm] if boxes_per_cls is not None else None,
                dropout_mask=dmask)
            obj_preds = obj_preds[inv_perm]
            obj_dists2 = obj_dists2[inv_perm]
        else:
            obj_preds = obj_labels
            obj_dists2 = torch.full((obj_preds.size(0), num_classes), -1000.0)
            obj_dists2[torch.arange(obj_preds.size(0)), obj_preds] = 1000.0
        obj_ctx = encoder_rep[inv_perm]
    else:
        if cfg['mode'] == 'predcls':
            obj_dists2 = torch.full((obj_labels.size(0), num_classes), -1000.0)          # to_onehot, pytorch_misc.py
            obj_dists2[torch.arange(obj_labels.size(0)), obj_labels] = 1000.0
        else:
            obj_dists2 = F.linear(obj_pre_rep, sd[prefix + 'decoder_lin.weight'], sd[prefix + 'decoder_lin.bias'])
        if cfg['mode'] == 'sgdet' and not training:
            probs = F.softmax(obj_dists2, 1).detach()
            nms_mask = torch.zeros_like(probs)
            for c_i in range(1, num_classes):
                keep = B.apply_nms(probs[:, c_i], boxes_per_cls[:, c_i], pre_nms_topn=probs.size(0),
                                   post_nms_topn=probs.size(0), nms_thresh=0.3)
                nms_mask[:, c_i][keep] = 1
            obj_preds = (nms_mask * probs)[:, 1:].max(1)[1] + 1
        else:
            obj_preds = obj_labels if obj_labels is not None else obj_dists2[:, 1:].max(1)[1] + 1
        obj_ctx = obj_pre_rep

    if cfg['nl_edge'] == 0:
        return obj_dists2, obj_preds, None

    # ---- edge_ctx (rel_model.py:171-195)
    edge_in_feats = torch.cat((obj_fmaps, obj_ctx), 1) if cfg['pass_in_obj_feats_to_edge'] else obj_ctx
    obj_embed2 = sd[prefix + 'obj_embed2.weight'][obj_preds]
    inp_feats = torch.cat((obj_embed2, edge_in_feats), 1)
    d2 = obj_dists2.detach()
    confidence = F.softmax(d2, dim=1).view(-1)[obj_preds + torch.arange(obj_preds.size(0)) * num_classes]
    perm, inv_perm, ls_transposed = sort_rois(cfg['order'], im_inds, confidence, box_priors)
    mask = _lstm_mask(rng, cfg['nl_edge'], int(ls_transposed[0]), H, cfg['rec_dropout'], training)
    edge_reps = L.alternating_highway_lstm(inp_feats[perm], ls_transposed, sd[prefix + 'edge_ctx_rnn.weight'],
                                           sd[prefix + 'edge_ctx_rnn.bias'], H, cfg['nl_edge'], training, mask)
    edge_ctx = edge_reps[inv_perm]
    return obj_dists2, obj_preds, edge_ctx


# --------------------------------------------------------------------------- union features
def union_boxes_feats(sd, fmap, rois, union_inds, training, prefix='union_boxes.', pooling_size=7):
    """UnionBoxesAndFeats.forward (get_union_boxes.py:42-53); updates BN running stats in `sd`."""
    im_inds = rois[:, 0][union_inds[:, 0]]
    union_rois = torch.cat((im_inds[:, None],
                            torch.min(rois[:, 1:3][union_inds[:, 0]], rois[:, 1:3][union_inds[:, 1]]),
                            torch.max(rois[:, 3:5][union_inds[:, 0]], rois[:, 3:5][union_inds[:, 1]])), 1)
    union_pools = roi_align(fmap, union_rois, pooling_size, pooling_size, 1.0 / 16)
    pair_rois = torch.cat((rois[:, 1:][union_inds[:, 0]], rois[:, 1:][union_inds[:, 1]]), 1).float().numpy()
    rects = torch.from_numpy(native.draw_union_boxes(pair_rois, pooling_size * 4 - 1) - np.float32(0.5)).to(fmap.dtype)
    x = F.conv2d(rects, sd[prefix + 'conv.0.weight'], sd[prefix + 'conv.0.bias'], stride=2, padding=3)
    x = _relu(x, prefix + 'conv.0')
    x = F.batch_norm(x, sd[prefix + 'conv.2.running_mean'], sd[prefix + 'conv.2.running_var'],
                     sd[prefix + 'conv.2.weight'], sd[prefix + 'conv.2.bias'], training=training,
                     momentum=BATCHNORM_MOMENTUM, eps=1e-5)
    x = _max_pool_3x3s2p1(x, prefix + 'pool')
    x = F.conv2d(x, sd[prefix + 'conv.4.weight'], sd[prefix + 'conv.4.bias'], stride=1, padding=1)
    x = _relu(x, prefix + 'conv.4')
    x = F.batch_norm(x, sd[prefix + 'conv.6.running_mean'], sd[prefix + 'conv.6.running_var'],
                     sd[prefix + 'conv.6.weight'], sd[prefix + 'conv.6.bias'], training=training,
                     momentum=BATCHNORM_MOMENTUM, eps=1e-5)
    return union_pools + x


def get_rel_inds(cfg, rel_labels, im_inds, box_priors, training):
    """RelModel.get_rel_inds (rel_model.py:416-437)"""
    if training:
        return rel_labels[:, :3].clone()
    rel_cands = im_inds[:, None] == im_inds[None]
    rel_cands.view(-1)[torch.arange(im_inds.size(0)) * (im_inds.size(0) + 1)] = False
    if cfg['mode'] == 'sgdet' and cfg.get('require_overlap', True):
        rel_cands = rel_cands & (B.bbox_overlaps(box_priors, box_priors) > 0)
    rel_cands = rel_cands.nonzero()
    if rel_cands.numel() == 0:
        rel_cands = im_inds.new_zeros(1, 2)
    return torch.cat((im_inds[rel_cands[:, 0]][:, None], rel_cands), 1)


def filter_dets(boxes, obj_scores, obj_classes, rel_inds, pred_scores):
    """lib/surgery.py:21-59 (stable descending sort)"""
    obj_scores0 = obj_scores[rel_inds[:, 0]]
    obj_scores1 = obj_scores[rel_inds[:, 1]]
    pred_scores_max, _ = pred_scores[:, 1:].max(1)
    rel_scores_argmaxed = pred_scores_max * obj_scores0 * obj_scores1
    _, rel_scores_idx = torch.sort(rel_scores_argmaxed.view(-1), dim=0, descending=True, stable=True)
    return (boxes.numpy(), obj_classes.numpy(), obj_scores.numpy(), rel_inds[rel_scores_idx].numpy(),
            pred_scores[rel_scores_idx].numpy())


def relmodel_forward(sd, cfg, x, im_sizes, image_offset, gt_boxes, gt_classes, training, rng,
                     rel_labels=None, det_override=None):
    """
    RelModel.forward (rel_model.py:450-547).  `sd` maps the reference's state-dict keys to CPU
    fp32 tensors (BN running stats are updated in place in training mode, as nn.BatchNorm does).
    Training returns a dict of Result fields; eval returns the filter_dets 5-tuple.
    `det_override`: the detector stage's outputs (fmap, im_inds, rm_box_priors, rm_obj_dists, rm_obj_labels, rel_labels,
    boxes_all) taken from elsewhere -- stage-wise parity of SGDet TRAINING, where the detections themselves are compared
    stage by stage (tests/test_gpu_sgdet.py) and the relation model is then run on identical detections.
    """
    with torch.no_grad():   # train_rels.py:50-52 freezes the detector; rel_model.py:491 detaches fmap
        det = det_override if det_override is not None else detector_forward(
            sd, cfg, x, im_sizes, image_offset, gt_boxes, gt_classes, training, rng, rel_labels)
    if det is None:
        return None
    im_inds = det['im_inds'] - image_offset
    boxes = det['rm_box_priors']
    rel_inds = get_rel_inds(cfg, det['rel_labels'], im_inds, boxes, training)
    rois = torch.cat((im_inds[:, None].to(boxes.dtype), boxes), 1)
    fmap = det['fmap'].detach()
    if cfg.get('use_resnet', False):    # the repaired model: the object branch owns a copy of the layer4 stack
        obj_fmap = resnet_l4_head(sd, roi_align(fmap, rois), 'roi_fmap_obj.0.', training)
    else:
        obj_fmap = vgg_classifier(sd, roi_align(fmap, rois).view(rois.size(0), -1), 'roi_fmap_obj.', training, rng)
    use_labels = training or cfg['mode'] == 'predcls'
    rm_obj_dists, obj_preds, edge_ctx = context_forward(
        sd, cfg, obj_fmap, det['rm_obj_dists'].detach(), im_inds,
        det['rm_obj_labels'] if use_labels else None, boxes, det['boxes_all'], training, rng)
    P = cfg['pooling_dim']
    if edge_ctx is None:            # nl_edge == 0: rel_model.py:500-503 (edge_rep = self.post_emb(result.obj_preds))
        edge_rep = sd['post_emb.weight'][obj_preds].view(-1, 2, P)
    else:
        edge_rep = F.linear(edge_ctx, sd['post_lstm.weight'], sd['post_lstm.bias']).view(-1, 2, P)
    subj_rep, obj_rep = edge_rep[:, 0], edge_rep[:, 1]
    prod_rep = subj_rep[rel_inds[:, 1]] * obj_rep[rel_inds[:, 2]]
    if cfg.get('use_vision', True):
        ub = union_boxes_feats(sd, fmap, rois, rel_inds[:, 1:], training)
        if cfg.get('use_resnet', False):
            vr = resnet_l4_head(sd, ub, 'roi_fmap.0.', training)
        else:
            vr = vgg_classifier(sd, ub.view(ub.size(0), -1), 'roi_fmap.1.', training, rng,
                                use_dropout=False, use_relu=False)
        if cfg['limit_vision']:
            prod_rep = torch.cat((prod_rep[:, :2048] * vr[:, :2048], prod_rep[:, 2048:]), 1)
        else:
            prod_rep = prod_rep * vr
    if cfg['use_tanh']:
        prod_rep = torch.tanh(prod_rep)
    rel_dists = F.linear(prod_rep, sd['rel_compress.weight'], sd['rel_compress.bias'])
    if cfg['use_bias']:
        num_objs = sd['context.obj_embed.weight'].shape[0]
        rel_dists = rel_dists + sd['freq_bias.obj_baseline.weight'][
            obj_preds[rel_inds[:, 1]] * num_objs + obj_preds[rel_inds[:, 2]]]
    if training:
        return dict(rm_obj_dists=rm_obj_dists, rm_obj_labels=det['rm_obj_labels'], rel_dists=rel_dists,
                    rel_labels=det['rel_labels'], obj_preds=obj_preds, obj_fmap=obj_fmap,
                    od_obj_dists=det['od_obj_dists'], fmap=fmap, rel_inds=rel_inds)
    num_classes = rm_obj_dists.size(1)
    twod_inds = torch.arange(obj_preds.size(0)) * num_classes + obj_preds
    obj_scores = F.softmax(rm_obj_dists, dim=1).view(-1)[twod_inds]
    if cfg['mode'] == 'sgdet':
        bboxes = det['boxes_all'].view(-1, 4)[twod_inds].view(det['boxes_all'].size(0), 4)
    else:
        bboxes = det['rm_box_priors']
    rel_rep = F.softmax(rel_dists, dim=1)
    out = filter_dets(bboxes, obj_scores, obj_preds, rel_inds[:, 1:], rel_rep)
    if cfg.get('return_logits', False):
        return out, dict(rel_dists=rel_dists, rm_obj_dists=rm_obj_dists, rel_inds=rel_inds)
    return out


# --------------------------------------------------------------------------- message-passing baseline (§8f rank 4)
def _gru_cell(sd, prefix, x, h):
    """nn.GRUCell: r, z, n gate order"""
    H = h.size(1)
    gi = F.linear(x, sd[prefix + 'weight_ih'], sd[prefix + 'bias_ih'])
    gh = F.linear(h, sd[prefix + 'weight_hh'], sd[prefix + 'bias_hh'])
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def stanford_message_pass(sd, rel_rep, obj_rep, rel_inds, size=512):
    """RelModelStanford.message_pass (lib/rel_model_stanford.py:59-108)"""
    def gate(name, a, b):
        return torch.sigmoid(F.linear(torch.cat((a, b), 1), sd[name + '.0.weight'], sd[name + '.0.bias']))
    vert = [_gru_cell(sd, 'node_gru.', obj_rep, torch.zeros(obj_rep.size(0), size))]
    edge = [_gru_cell(sd, 'edge_gru.', rel_rep, torch.zeros(rel_rep.size(0), size))]
    for i in range(3):
        sv, ov = vert[i][rel_inds[:, 0]], vert[i][rel_inds[:, 1]]
        ws = gate('sub_vert_w_fc', sv, edge[i]) * sv
        wo = gate('obj_vert_w_fc', ov, edge[i]) * ov
        edge.append(_gru_cell(sd, 'edge_gru.', ws + wo, edge[i]))
        pre_out = gate('out_edge_w_fc', sv, edge[i]) * edge[i]
        pre_in = gate('in_edge_w_fc', ov, edge[i]) * edge[i]
        ctx = torch.zeros_like(vert[i])
        ctx.index_add_(0, rel_inds[:, 0], pre_out)
        ctx.index_add_(0, rel_inds[:, 1], pre_in)
        vert.append(_gru_cell(sd, 'node_gru.', ctx, vert[i]))
    return (F.linear(vert[-1], sd['obj_fc.weight'], sd['obj_fc.bias']),
            F.linear(edge[-1], sd['rel_fc.weight'], sd['rel_fc.bias']))


def stanford_forward_train(sd, cfg, x, im_sizes, image_offset, gt_boxes, gt_classes, rng, rel_labels):
    """RelModelStanford.forward in training (gtbox modes): detector -> union features -> unary projections ->
    message passing (lib/rel_model_stanford.py:110-156)"""
    det = detector_forward(sd, cfg, x, im_sizes, image_offset, gt_boxes, gt_classes, True, rng, rel_labels=rel_labels)
    fmap = det['fmap']
    im_inds = det['im_inds'] - image_offset
    boxes = det['rm_box_priors']
    rel_inds = get_rel_inds(cfg, det['rel_labels'], im_inds, boxes, True)
    rois = torch.cat((im_inds[:, None].to(boxes.dtype), boxes), 1)
    ub = union_boxes_feats(sd, fmap, rois, rel_inds[:, 1:], True)
    vr = vgg_classifier(sd, ub.view(ub.size(0), -1), 'roi_fmap.1.', True, rng, use_dropout=False, use_relu=False)
    obj_fmap = vgg_classifier(sd, roi_align(fmap, rois).view(rois.size(0), -1), 'roi_fmap_obj.', True, rng)
    rel_rep = F.relu(F.linear(vr, sd['edge_unary.weight'], sd['edge_unary.bias']))
    obj_rep = F.linear(obj_fmap, sd['obj_unary.weight'], sd['obj_unary.bias'])
    obj_dists, rel_dists = stanford_message_pass(sd, rel_rep, obj_rep, rel_inds[:, 1:])
    return dict(rm_obj_dists=obj_dists, rel_dists=rel_dists, rel_inds=rel_inds, rm_obj_labels=det['rm_obj_labels'])
