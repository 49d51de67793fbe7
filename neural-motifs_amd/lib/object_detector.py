"""
Faster-R-CNN style object detector of the scene-graph pipeline -- the reference's `ObjectDetector` API
(lib/object_detector.py:50-422) on the gfx950 kernels: VGG16 trunk (NHWC implicit-GEMM convs), RPN head +
proposal decode + on-device NMS, RoIAlign (coalesced NHWC gather) + fc6/fc7 on the MFMA GEMM, class/box heads,
batched per-class NMS (`filter_det`).

Modes, constructor arguments, `forward` positional arguments, `Result` fields and state-dict keys are the
reference's.  `use_resnet=True` builds the (reference-deprecated) ResNet-101 detector: lib/resnet.py.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.nn import functional as F

from config import ANCHOR_SIZE, ANCHOR_RATIOS, ANCHOR_SCALES
from lib import _hip
from lib import hip_ops
from lib.fpn.box_utils import bbox_preds, center_size, bbox_overlaps
from lib.fpn.generate_anchors import generate_anchors
from lib.fpn.nms.functions.nms import apply_nms, nms_mask_per_class
from lib.fpn.proposal_assignments.proposal_assignments_gtbox import proposal_assignments_gtbox
from lib.fpn.proposal_assignments.proposal_assignments_det import proposal_assignments_det
from lib.fpn.roi_align.functions.roi_align import RoIAlignFunction
from lib.hip_ops import (AlphaDropout, Conv3x3, Dropout, FCStack, Linear, ReLU, VGG16Features, EPI_NONE, EPI_RELU6, _is_nhwc,
                         _Conv3x3Fn, linear)
from lib.pytorch_misc import enumerate_by_image, gather_nd, h2d, has_host, host_np, set_host


class Result(object):
    """container for the detector / relation-model outputs (od: object detector, rm: rel model)"""

    def __init__(self, od_obj_dists=None, rm_obj_dists=None, obj_scores=None, obj_preds=None, obj_fmap=None,
                 od_box_deltas=None, rm_box_deltas=None, od_box_targets=None, rm_box_targets=None,
                 od_box_priors=None, rm_box_priors=None, boxes_assigned=None, boxes_all=None, od_obj_labels=None,
                 rm_obj_labels=None, rpn_scores=None, rpn_box_deltas=None, rel_labels=None, im_inds=None, fmap=None,
                 rel_dists=None, rel_inds=None, rel_rep=None):
        self.__dict__.update(locals())
        del self.__dict__['self']

    def is_none(self):
        return all(v is None for k, v in self.__dict__.items() if k != 'self')


def gather_res(outputs, target_device, dim=0):
    """concatenate the fields of per-replica Results (reference :40-47); with one process per GPU each rank
    holds exactly one Result, so this is only used by tests / tools"""
    out = outputs[0]
    args = {f: torch.cat([getattr(o, f) for o in outputs], dim) for f, v in out.__dict__.items() if v is not None}
    return type(out)(**args)


def load_vgg(use_dropout=True, use_relu=True, use_linear=True, pretrained=False):
    """VGG16 `features` (without the last max-pool) and `classifier` (without the class layer) with torchvision's
    child indices (reference :623-633).  There are no model-zoo weights offline: weights are He-initialised and are
    expected to come from a detector checkpoint (`optimistic_restore`)."""
    model = nn.Module()
    model.features = VGG16Features()
    cls = [('0', Linear(512 * 7 * 7, 4096)), ('1', ReLU()), ('2', Dropout(0.5)),
           ('3', Linear(4096, 4096)), ('4', ReLU()), ('5', Dropout(0.5))]
    if not use_dropout:
        cls = [c for c in cls if c[0] != '5']
        if not use_relu:
            cls = [c for c in cls if c[0] != '4']
            if not use_linear:
                cls = [c for c in cls if c[0] != '3']
    stack = FCStack()
    for name, mod in cls:
        stack.add_module(name, mod)
    for m in stack.children():
        if isinstance(m, Linear):
            nn.init.normal_(m.weight, 0, 0.01)
            nn.init.constant_(m.bias, 0)
    model.classifier = stack
    return model


def load_resnet():
    """resnet101 minus layer4 / avgpool / fc (lib/object_detector.py:615-620); random init offline"""
    from lib.resnet import ResNet101Trunk
    return ResNet101Trunk()


class ResNetCompress(nn.Sequential):
    """Conv2d(1024, 256, 1) -> ReLU -> BatchNorm2d(256)  (lib/object_detector.py:84-88; child indices 0 and 2 carry the
    parameters).  Runs NHWC: the 1x1 conv is one MFMA GEMM with fused bias + ReLU, BN on the HIP kernels."""

    def __init__(self):
        from lib.resnet import _BN
        super(ResNetCompress, self).__init__(Conv1x1(1024, 256), nn.ReLU(inplace=True), _BN(256))

    def forward(self, fmap):                     # logical NCHW, channels_last memory
        from lib import _hip as H
        if torch.is_grad_enabled() and (fmap.requires_grad or any(p.requires_grad for p in self.parameters())):
            # detector pre-training: the same three layers through autograd (product with fused bias + ReLU, train-mode BN)
            from lib.hip_ops import linear
            x = fmap.permute(0, 2, 3, 1).contiguous()
            B, Hh, Ww, C = x.shape
            conv, bn = self[0], self[2]
            y = linear(x.view(-1, C), conv.weight.view(conv.weight.shape[0], C), conv.bias, relu=True).view(B, Hh, Ww, -1)
            return bn(y).permute(0, 3, 1, 2)
        with torch.no_grad():
            x = fmap.permute(0, 2, 3, 1).contiguous()
            B, Hh, Ww, C = x.shape
            conv, bn = self[0], self[2]
            y = H.gemm(x.view(-1, C), conv.weight.detach().view(conv.weight.shape[0], C), False, True,
                       bias=conv.bias.detach(), epilogue=1).view(B, Hh, Ww, -1)
            y = bn(y)
        return y.permute(0, 3, 1, 2)


class Conv1x1(nn.Module):
    """parameter holder with nn.Conv2d's names/shapes (weight [Cout,Cin,1,1], bias [Cout])"""

    def __init__(self, cin, cout):
        super(Conv1x1, self).__init__()
        ref = nn.Conv2d(cin, cout, kernel_size=1)
        self.weight, self.bias = ref.weight, ref.bias


class ObjectDetector(nn.Module):
    MODES = ('rpntrain', 'gtbox', 'refinerels', 'proposals')

    def __init__(self, classes, mode='rpntrain', num_gpus=1, nms_filter_duplicates=True, max_per_img=64,
                 use_resnet=False, thresh=0.05):
        super(ObjectDetector, self).__init__()
        if mode not in self.MODES:
            raise ValueError("invalid mode")
        self.mode = mode
        self.classes = classes
        self.num_gpus = num_gpus
        self.pooling_size = 7
        self.nms_filter_duplicates = nms_filter_duplicates
        self.max_per_img = max_per_img
        self.use_resnet = use_resnet
        self.thresh = thresh
        if not self.use_resnet:
            vgg_model = load_vgg()
            self.features = vgg_model.features
            self.roi_fmap = vgg_model.classifier
            rpn_input_dim, output_dim = 512, 4096
        else:   # "Deprecated" in the reference (lib/object_detector.py:83-99) but part of its surface
            self.features = load_resnet()
            self.compress = ResNetCompress()
            self.roi_fmap = nn.Sequential(
                Linear(256 * 7 * 7, 2048), nn.SELU(inplace=True), AlphaDropout(p=0.05),
                Linear(2048, 2048), nn.SELU(inplace=True), AlphaDropout(p=0.05))
            rpn_input_dim, output_dim = 1024, 2048
        self.score_fc = Linear(output_dim, self.num_classes)
        self.bbox_fc = Linear(output_dim, self.num_classes * 4)
        self.rpn_head = RPNHead(dim=512, input_dim=rpn_input_dim)

    @property
    def num_classes(self):
        return len(self.classes)

    def feature_map(self, x):
        """[B,3,S,S] image -> [B,512,S/16,S/16] feature map (channels_last memory)"""
        return self.features(x)

    def obj_feature_map(self, features, rois):
        pooled = RoIAlignFunction(self.pooling_size, self.pooling_size, spatial_scale=1 / 16)(
            self.compress(features) if self.use_resnet else features, rois)
        return self.roi_fmap(pooled.view(rois.size(0), -1))

    # ---------------------------------------------------------------------------------- box sources
    def rpn_boxes(self, fmap, im_sizes, image_offset, gt_boxes=None, gt_classes=None, gt_rels=None,
                  train_anchor_inds=None, proposals=None):
        rpn_feats = self.rpn_head(fmap)
        big = self.training and self.mode == 'rpntrain'
        rois = self.rpn_head.roi_proposals(rpn_feats, im_sizes, nms_thresh=0.7,
                                           pre_nms_topn=12000 if big else 6000,
                                           post_nms_topn=2000 if big else 1000)
        rpn_scores = rpn_box_deltas = labels = bbox_targets = rel_labels = None
        if self.training:
            if gt_boxes is None or gt_classes is None or train_anchor_inds is None:
                raise ValueError("Must supply GT boxes, GT classes, trainanchors when in train mode")
            rpn_scores, rpn_box_deltas = self.rpn_head.anchor_preds(rpn_feats, train_anchor_inds, image_offset)
            if gt_rels is not None and self.mode == 'rpntrain':
                raise ValueError("Training the object detector and the relationship model with detection"
                                 "at the same time isn't supported")
            if self.mode != 'refinerels':          # detector pre-training: sample <= 256 RoIs/img, <= 25 % foreground
                rois, labels, bbox_targets = proposal_assignments_det(
                    rois, gt_boxes.detach(), gt_classes.detach(), image_offset, fg_thresh=0.5,
                    rs=getattr(self, 'sampler_rs', None))
        return rois, labels, bbox_targets, rpn_scores, rpn_box_deltas, rel_labels

    def gt_boxes(self, fmap, im_sizes, image_offset, gt_boxes=None, gt_classes=None, gt_rels=None,
                 train_anchor_inds=None, proposals=None):
        assert gt_boxes is not None
        mirrored = has_host(gt_classes) and has_host(gt_boxes)
        if mirrored:        # GT arrays of a Blob carry their host values: the sampler / packing order need no D2H copy
            rois_np = np.ascontiguousarray(np.column_stack(((host_np(gt_classes)[:, 0] - image_offset).astype(np.float32),
                                                            host_np(gt_boxes).astype(np.float32))))
            if gt_boxes.is_cuda and not gt_boxes.requires_grad:
                rois = set_host(h2d(rois_np, gt_boxes.device), rois_np)       # one upload instead of subtract + cast + concatenate (round 6)
            else:
                rois = set_host(torch.cat(((gt_classes[:, 0] - image_offset).float()[:, None], gt_boxes), 1), rois_np)
        else:
            im_inds = gt_classes[:, 0] - image_offset
            rois = torch.cat((im_inds.float()[:, None], gt_boxes), 1)
        if gt_rels is not None and self.training:
            rois, labels, rel_labels = proposal_assignments_gtbox(
                rois, gt_boxes, gt_classes, gt_rels, image_offset, fg_thresh=0.5, rs=getattr(self, 'sampler_rs', None))
        else:
            labels, rel_labels = gt_classes[:, 1], None
        if mirrored:
            set_host(labels, host_np(gt_classes)[:, 1])
        return rois, labels, None, None, None, rel_labels

    def proposal_boxes(self, fmap, im_sizes, image_offset, gt_boxes=None, gt_classes=None, gt_rels=None,
                       train_anchor_inds=None, proposals=None):
        assert proposals is not None
        rois = filter_roi_proposals(proposals[:, 2:].detach().contiguous(), proposals[:, 1].detach().contiguous(),
                                    np.array([2000] * len(im_sizes)), nms_thresh=0.7, pre_nms_topn=6000,
                                    post_nms_topn=1000)
        if self.training:
            # the reference's branch (object_detector.py:247-252) returns sampled + all RoIs with labels for the sampled
            # ones only, which its own loss (train_detector.py:108-111) cannot consume: no behaviour to reproduce
            raise NotImplementedError('detector training on precomputed proposals is not runnable in the reference either')
        return rois, None, None, None, None, None

    def get_boxes(self, *args, **kwargs):
        if self.mode == 'gtbox':
            return self.gt_boxes(*args, **kwargs)
        if self.mode == 'proposals':
            assert kwargs['proposals'] is not None
            return self.proposal_boxes(*args, **kwargs)
        return self.rpn_boxes(*args, **kwargs)

    # ---------------------------------------------------------------------------------- forward
    def forward(self, x, im_sizes, image_offset, gt_boxes=None, gt_classes=None, gt_rels=None, proposals=None,
                train_anchor_inds=None, return_fmap=False):
        fmap = self.feature_map(x)
        rois, obj_labels, bbox_targets, rpn_scores, rpn_box_deltas, rel_labels = self.get_boxes(
            fmap, im_sizes, image_offset, gt_boxes, gt_classes, gt_rels, train_anchor_inds, proposals=proposals)
        obj_fmap = self.obj_feature_map(fmap, rois)
        od_obj_dists = self.score_fc(obj_fmap)
        od_box_deltas = self.bbox_fc(obj_fmap).view(-1, len(self.classes), 4) if self.mode != 'gtbox' else None
        od_box_priors = rois[:, 1:]

        if (not self.training and not self.mode == 'gtbox') or self.mode in ('proposals', 'refinerels'):
            nms_out = self.nms_boxes(od_obj_dists, rois, od_box_deltas, im_sizes)
            if nms_out is None:
                return Result()
            nms_inds, nms_scores, nms_preds, nms_boxes_assign, nms_boxes, nms_imgs = nms_out
            im_inds = nms_imgs + image_offset
            if has_host(nms_imgs):
                set_host(im_inds, host_np(nms_imgs) + image_offset)
            obj_dists = od_obj_dists[nms_inds]
            obj_fmap = obj_fmap[nms_inds]
            box_deltas = od_box_deltas[nms_inds]
            box_priors = nms_boxes[:, 0]
            if self.training and not self.mode == 'gtbox':
                pred_to_gtbox = bbox_overlaps(box_priors.detach().contiguous(), gt_boxes).detach()
                pred_to_gtbox[im_inds[:, None] != gt_classes[None, :, 0]] = 0.0
                max_overlaps, argmax_overlaps = pred_to_gtbox.max(1)
                rm_obj_labels = gt_classes[:, 1][argmax_overlaps].clone()
                rm_obj_labels[max_overlaps < 0.5] = 0
            else:
                rm_obj_labels = None
        else:
            if has_host(rois) and rois.is_cuda:
                im_np = np.ascontiguousarray(host_np(rois)[:, 0].astype(np.int64) + image_offset)
                im_inds = set_host(h2d(im_np, rois.device), im_np)           # (was cast + copy + add on the device)
            else:
                im_inds = rois[:, 0].long().contiguous() + image_offset
                if has_host(rois):
                    set_host(im_inds, host_np(rois)[:, 0].astype(np.int64) + image_offset)
            nms_scores = nms_preds = nms_boxes_assign = nms_boxes = None
            box_priors = rois[:, 1:]
            if has_host(rois):
                set_host(box_priors, host_np(rois)[:, 1:])      # the context's packing order is box geometry: computed on the host
            rm_obj_labels = obj_labels
            box_deltas = od_box_deltas
            obj_dists = od_obj_dists

        return Result(
            od_obj_dists=od_obj_dists, rm_obj_dists=obj_dists, obj_scores=nms_scores, obj_preds=nms_preds,
            obj_fmap=obj_fmap, od_box_deltas=od_box_deltas, rm_box_deltas=box_deltas, od_box_targets=bbox_targets,
            rm_box_targets=bbox_targets, od_box_priors=od_box_priors, rm_box_priors=box_priors,
            boxes_assigned=nms_boxes_assign, boxes_all=nms_boxes, od_obj_labels=obj_labels,
            rm_obj_labels=rm_obj_labels, rpn_scores=rpn_scores, rpn_box_deltas=rpn_box_deltas, rel_labels=rel_labels,
            im_inds=im_inds, fmap=fmap if return_fmap else None)

    def nms_boxes(self, obj_dists, rois, box_deltas, im_sizes):
        """class-specific box decode + per-image detection filter (reference :363-408).  The filter of every image is
        enqueued without reading anything back (filter_det_device); the detection counts of all images come back in ONE
        copy."""
        boxes = bbox_preds(rois[:, None, 1:].expand_as(box_deltas).contiguous().view(-1, 4),
                           box_deltas.reshape(-1, 4)).view(*box_deltas.size()).detach().clone()
        inds = rois[:, 0].long().contiguous()
        if has_host(rois):
            set_host(inds, host_np(rois)[:, 0].astype(np.int64))
        elif getattr(rois, '_host_im', None) is not None:
            set_host(inds, rois._host_im)
        pending = []
        for i, s, e in enumerate_by_image(inds):
            h, w = im_sizes[i, :2]
            boxes[s:e, :, 0].clamp_(min=0, max=float(w) - 1)
            boxes[s:e, :, 1].clamp_(min=0, max=float(h) - 1)
            boxes[s:e, :, 2].clamp_(min=0, max=float(w) - 1)
            boxes[s:e, :, 3].clamp_(min=0, max=float(h) - 1)
            sc = F.softmax(obj_dists[s:e].detach(), 1)
            if self.nms_filter_duplicates:
                pending.append((i,) + filter_det_device(sc, boxes[s:e], start_ind=s, max_per_img=self.max_per_img, thresh=self.thresh))
            else:
                d = filter_det(sc, boxes[s:e], start_ind=s, nms_filter_duplicates=False, max_per_img=self.max_per_img,
                               thresh=self.thresh)
                if d is not None:
                    pending.append((i,) + d + (None,))
        counts = None
        if any(p[4] is not None for p in pending):
            counts = torch.stack([p[4] for p in pending if p[4] is not None]).cpu().tolist()     # the only host sync
        dets, im_host, ci = [], [], 0
        for i, d_inds, d_scores, d_labels, cnt in pending:
            if cnt is not None:
                k = int(counts[ci])
                ci += 1
                if k == 0:
                    continue
                d_inds, d_scores, d_labels = d_inds[:k], d_scores[:k], d_labels[:k]
            dets.append((d_inds, d_scores, d_labels))
            im_host += [i] * int(d_inds.shape[0])
        if len(dets) == 0:
            print("nothing was detected", flush=True)
            return None
        nms_inds, nms_scores, nms_labels = [torch.cat(x, 0) for x in zip(*dets)]
        twod_inds = nms_inds * boxes.size(1) + nms_labels
        nms_boxes_assign = boxes.view(-1, 4)[twod_inds]
        nms_boxes = torch.cat((rois[:, 1:][nms_inds][:, None], boxes[nms_inds][:, 1:]), 1)
        nms_imgs = set_host(inds[nms_inds], np.asarray(im_host, dtype=np.int64))     # the image of every detection, known on the host
        return nms_inds, nms_scores, nms_labels, nms_boxes_assign, nms_boxes, nms_imgs

    def __getitem__(self, batch):
        """`detector[blob]` (reference :410-422).  Data parallelism is one process per GPU (lib/dist.py); inside a
        process there is exactly one replica."""
        batch.scatter()
        if self.num_gpus != 1:
            raise RuntimeError('in-process multi-GPU replication is replaced by one process per GPU: launch with '
                               'torchrun and keep num_gpus=1 per rank')
        return self(*batch[0])


def filter_det_device(scores, boxes, start_ind=0, max_per_img=100, thresh=0.001, pre_nms_topn=6000, post_nms_topn=300,
                      nms_thresh=0.3):
    """filter_det (nms_filter_duplicates=True) without a device->host read: returns (roi ids, scores, labels) of the
    top `max_per_img` candidates in the reference's order and the NUMBER of them that are detections as a device scalar.
    Same result as filter_det: per-class suppression runs over ALL foreground classes instead of the ones whose best score
    beats `thresh` (a class below the threshold cannot contribute -- every entry it keeps scores <= thresh and is cut by the
    final `> thresh`; where it wins a roi's maximum, the reference has 0 there and drops the roi as well), and
    nonzero() + sort becomes one stable sort over all rois (zeros sort last, ties keep roi order)."""
    n, C = scores.shape
    if n > pre_nms_topn:
        raise NotImplementedError('more rois per image than pre_nms_topn')
    class_ids = torch.arange(1, C, device=scores.device)
    nms_mask = nms_mask_per_class(scores, boxes, class_ids, nms_thresh, post_nms_topn)
    scores_pre, labels_pre = (nms_mask * scores).max(1)
    vs, order = torch.sort(scores_pre, dim=0, descending=True, stable=True)
    count = (vs > thresh).sum().clamp(max=max_per_img)
    top = order[:max_per_img]
    return top + start_ind, vs[:max_per_img], labels_pre[top], count


def filter_det(scores, boxes, start_ind=0, max_per_img=100, thresh=0.001, pre_nms_topn=6000, post_nms_topn=300,
               nms_thresh=0.3, nms_filter_duplicates=True):
    """detections of ONE image (reference :425-485): per-class NMS over the classes whose best score beats `thresh`
    (all classes in one batched launch), one label per roi, top `max_per_img` by score."""
    valid_cls = (scores[:, 1:].max(0)[0] > thresh).nonzero() + 1
    if valid_cls.numel() == 0:
        return None
    if scores.size(0) > pre_nms_topn:
        raise NotImplementedError('more rois per image than pre_nms_topn')
    nms_mask = nms_mask_per_class(scores, boxes, valid_cls.view(-1), nms_thresh, post_nms_topn)
    dists_all = nms_mask * scores
    if nms_filter_duplicates:
        scores_pre, labels_pre = dists_all.max(1)
        inds_all = scores_pre.nonzero().view(-1)
        labels_all, scores_all = labels_pre[inds_all], scores_pre[inds_all]
    else:
        nz = nms_mask.nonzero()
        inds_all, labels_all = nz[:, 0], nz[:, 1]
        scores_all = scores.reshape(-1)[inds_all * scores.size(1) + labels_all]
    vs, idx = torch.sort(scores_all, dim=0, descending=True, stable=True)
    idx = idx[vs > thresh]
    if max_per_img < idx.size(0):
        idx = idx[:max_per_img]
    return inds_all[idx] + start_ind, scores_all[idx], labels_all[idx]


class _Conv1x1(nn.Module):
    """1x1 convolution = GEMM over NHWC pixels; nn.Conv2d parameter shapes ([Cout,Cin,1,1])"""

    def __init__(self, cin, cout):
        super(_Conv1x1, self).__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 1, 1))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)

    def forward_nhwc(self, x_nhwc):
        B, H, W, C = x_nhwc.shape
        if torch.is_grad_enabled() and (self.weight.requires_grad or x_nhwc.requires_grad):
            y = linear(x_nhwc.reshape(-1, C), self.weight.view(self.weight.size(0), C), self.bias)
        else:
            y = _hip.gemm(x_nhwc.reshape(-1, C), self.weight.detach().view(self.weight.size(0), C), False, True,
                          bias=self.bias.detach())
        return y.view(B, H, W, -1)


class RPNHead(nn.Module):
    """3x3 conv + ReLU6 + 1x1 conv -> (2 class logits + 4 box deltas) per anchor (reference :488-597)"""

    def __init__(self, dim=512, input_dim=1024):
        super(RPNHead, self).__init__()
        self.anchor_target_dim = 6
        self.stride = 16
        self.conv = nn.Sequential()
        self.conv.add_module('0', Conv3x3(input_dim, dim))
        self.conv.add_module('1', nn.ReLU6(inplace=True))
        self.conv.add_module('2', _Conv1x1(dim, self.anchor_target_dim * self._A))
        ans_np = generate_anchors(base_size=ANCHOR_SIZE, feat_stride=self.stride, anchor_scales=ANCHOR_SCALES,
                                  anchor_ratios=ANCHOR_RATIOS)
        self.register_buffer('anchors', torch.FloatTensor(ans_np))

    @property
    def _A(self):
        return len(ANCHOR_RATIOS) * len(ANCHOR_SCALES)

    def forward(self, fmap):
        """[B,C,h,w] feature map -> [B,h,w,A,6] (NHWC output == the reference's _reshape_channels).  Frozen on the
        relation-model path (forward-only kernels); trainable in detector pre-training (autograd Functions)."""
        trainable = torch.is_grad_enabled() and (fmap.requires_grad or any(p.requires_grad for p in self.parameters()))
        if trainable:
            x = fmap.permute(0, 2, 3, 1).contiguous()
            x = _Conv3x3Fn.apply(x, self.conv[0].weight, self.conv[0].bias, EPI_RELU6)
            hip_ops._tap_act('detector.rpn_head.conv.0', x, EPI_RELU6)         # test hook, no-op in production
            x = self.conv[2].forward_nhwc(x)
        else:
            with torch.no_grad():
                x = fmap.permute(0, 2, 3, 1) if _is_nhwc(fmap) else _hip.nchw_to_nhwc(fmap.contiguous())
                x = self.conv[0].forward_nhwc(x.contiguous(), EPI_RELU6)
                x = self.conv[2].forward_nhwc(x)
        return x.view(x.size(0), x.size(1), x.size(2), self._A, self.anchor_target_dim)

    def anchor_preds(self, preds, train_anchor_inds, image_offset):
        assert train_anchor_inds.size(1) == 4
        tai = train_anchor_inds.detach().clone()
        tai[:, 0] -= image_offset
        train_regions = gather_nd(preds, tai)
        return train_regions[:, :2], train_regions[:, 2:]

    def roi_proposals(self, fmap, im_sizes, nms_thresh=0.7, pre_nms_topn=12000, post_nms_topn=2000):
        """[B,h,w,A,6] RPN output -> rois [n,5] (reference :560-597)"""
        class_fmap = fmap[:, :, :, :, :2].contiguous()
        class_preds = F.softmax(class_fmap, 4)[..., 1].detach().contiguous().clone()
        box_fmap = fmap[:, :, :, :, 2:].detach().contiguous()
        anchor_stacked = torch.cat([self.anchors[None]] * fmap.size(0), 0)
        box_preds = bbox_preds(anchor_stacked.view(-1, 4), box_fmap.view(-1, 4)).view(*box_fmap.size()).clone()
        for i, (h, w, scale) in enumerate(im_sizes):
            h_end, w_end = int(h) // self.stride, int(w) // self.stride
            if h_end < class_preds.size(1):
                class_preds[i, h_end:] = -0.01
            if w_end < class_preds.size(2):
                class_preds[i, :, w_end:] = -0.01
            box_preds[i, :, :, :, 0].clamp_(min=0, max=float(w) - 1)
            box_preds[i, :, :, :, 1].clamp_(min=0, max=float(h) - 1)
            box_preds[i, :, :, :, 2].clamp_(min=0, max=float(w) - 1)
            box_preds[i, :, :, :, 3].clamp_(min=0, max=float(h) - 1)
        sizes = center_size(box_preds.view(-1, 4))
        class_preds.view(-1)[(sizes[:, 2] < 4) | (sizes[:, 3] < 4)] = -0.01
        return filter_roi_proposals(box_preds.view(-1, 4), class_preds.view(-1),
                                    boxes_per_im=np.array([np.prod(box_preds.size()[1:-1])] * fmap.size(0)),
                                    nms_thresh=nms_thresh, pre_nms_topn=pre_nms_topn, post_nms_topn=post_nms_topn)


def filter_roi_proposals(box_preds, class_preds, boxes_per_im, nms_thresh=0.7, pre_nms_topn=12000,
                         post_nms_topn=2000):
    inds, im_per = apply_nms(class_preds, box_preds, pre_nms_topn=pre_nms_topn, post_nms_topn=post_nms_topn,
                             boxes_per_im=boxes_per_im, nms_thresh=nms_thresh)
    img_inds = torch.cat([torch.full((n,), float(val), device=box_preds.device) for val, n in enumerate(im_per)], 0)
    rois = torch.cat((img_inds[:, None], box_preds[inds]), 1)
    rois._host_im = np.repeat(np.arange(len(im_per), dtype=np.int64), im_per)     # image of every roi, known on the host
    return rois
