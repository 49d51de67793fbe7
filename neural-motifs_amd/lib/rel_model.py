"""
MotifNet relation model -- the reference's `RelModel` / `LinearizedContext` API (lib/rel_model.py:66-560) on the
gfx950 kernels: object/edge context through the stacked alternating highway LSTM, label decoder, O(N^2) relation
head (union-box features -> fc6/fc7 on the MFMA GEMM), frequency bias, eval post-processing.

Constructor arguments, forward arguments, Result fields and state-dict keys follow the reference.
Behaviours kept on purpose (SURVEY.md §7 "quirks"): 'leftright' order is descending centre-x; the "BiLSTM" is
alternating-direction stacking; limit_vision multiplies only the first 2048 dims; ties in every sort break by index.
"""
import math

import os

import numpy as np
import torch
import torch.nn as nn
from torch.nn import functional as F
from torch.nn.utils.rnn import PackedSequence

from config import BATCHNORM_MOMENTUM
from lib.fpn.box_utils import bbox_overlaps, center_size
from lib.fpn.nms.functions.nms import apply_nms
from lib.fpn.proposal_assignments.rel_assignments import rel_assignments
from lib.fpn.roi_align.functions.roi_align import RoIAlignFunction
from lib.get_union_boxes import UnionBoxesAndFeats
from lib import _hip
from lib import rng as rng_mod
from lib.hip_ops import Dropout, Flattener, Linear, ReLU, linear
from lib.lstm.decoder_rnn import DecoderRNN
from lib.lstm.highway_lstm_cuda.alternating_highway_lstm import AlternatingHighwayLSTM
from lib.object_detector import ObjectDetector, gather_res, load_vgg
from lib.pytorch_misc import (transpose_packed_sequence_inds, to_onehot, arange, enumerate_by_image, has_host, host_np,
                              set_host, h2d)
from lib.sparse_targets import FrequencyBias
from lib.surgery import filter_dets
from lib.word_vectors import obj_edge_vectors

MODES = ('sgdet', 'sgcls', 'predcls')
# SGDet training, two-stream mode: MOTIFS_SGDET_CONTEXT_FIRST=1 enqueues the object-context branch BEFORE the host-side relation
# sampling (rel_assignments: 3.5 ms during which the GPU queue runs dry, profiles/r04_cfg3_trace_gaps_final.txt).  Measured in
# round 5 (gpurun r05_c3, cfg3, one box): 142.0 / 139.7 img/s with it, 146.6 without -- the SGDet step is bound by the HOST's
# enqueue time (~1000 launches per step), and the order of two pieces of host work does not change their sum.  Off.
SGDET_CONTEXT_FIRST = os.environ.get('MOTIFS_SGDET_CONTEXT_FIRST', '0') == '1'


class _AheadStage(object):
    """the detector stage of one batch, started by RelModel.detect_ahead() and collected by the forward() of the same `x`"""
    __slots__ = ('x', 'future')

    def __init__(self, x, future):
        self.x, self.future = x, future


_HOST_GEOMETRY = os.environ.get('MOTIFS_HOST_GEOMETRY', '1') != '0'
# the relation tail's gather-gather-multiply(-multiply) as one autograd node over csrc/exact_ops.hip pair_product_* (round 6);
# MOTIFS_PAIR_PRODUCT=0: the framework's index / multiply ops (A/B)
_PAIR_PRODUCT = os.environ.get('MOTIFS_PAIR_PRODUCT', '1') != '0'


def _pair_lists(i1, i2, n):
    """which rows have box i as their subject (side 0) / object (side 1): order [2, R] (stable: ascending row) and ptr [2, n + 1]
    (offsets into the flattened order array), int32, from the HOST copies of the pair indices"""
    R = i1.shape[0]
    order = np.empty((2, R), dtype=np.int32)
    ptr = np.empty((2, n + 1), dtype=np.int32)
    for side, idx in enumerate((i1, i2)):
        order[side] = np.argsort(idx, kind='stable').astype(np.int32)
        ptr[side] = side * R + np.searchsorted(idx[order[side]], np.arange(n + 1)).astype(np.int32)
    return order, ptr


class _FreqBiasAddFn(torch.autograd.Function):
    """rel_dists + FrequencyBias[obj_preds[i1], obj_preds[i2]] (reference lib/rel_model.py:528-531) as one node: one launch forward
    (was two label gathers, a stack, the key arithmetic, the embedding gather and the add), and the table's gradient from a
    leader-per-key kernel in ascending row order instead of the framework's device sort (csrc/exact_ops.hip: freq_bias_*)."""

    @staticmethod
    def forward(ctx, logits, table, obj_preds, rel_inds, num_objs):
        i1, i2 = rel_inds[:, 1].contiguous(), rel_inds[:, 2].contiguous()
        out, keys = _hip.freq_bias_add(logits.contiguous(), table, obj_preds.contiguous(), i1, i2, num_objs)
        ctx.table_rows = table.shape[0]
        ctx.save_for_backward(keys)
        return out

    @staticmethod
    def backward(ctx, g):
        keys, = ctx.saved_tensors
        d_table = _hip.freq_bias_bwd(g.contiguous(), keys, ctx.table_rows) if ctx.needs_input_grad[1] else None
        return g, d_table, None, None, None


class _PairProductFn(torch.autograd.Function):
    """prod[r] = edge[i1[r], 0] * edge[i2[r], 1] * vis[r] (reference lib/rel_model.py:500-512) -- one launch forward, two backward;
    the backward's sums over the rows that share a box run over lists made on the host from the mirror of the pair indices
    (deterministic order, no atomics, no device sort)."""

    @staticmethod
    def forward(ctx, edge, vis, rel_inds):
        i1, i2 = rel_inds[:, 1].contiguous(), rel_inds[:, 2].contiguous()
        edge_c = edge.contiguous()
        vis_c = vis.contiguous() if vis is not None else None
        out = _hip.pair_product_fwd(edge_c, i1, i2, vis_c)
        lists = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:      # (an evaluation forward has no use for the row lists)
            host = host_np(rel_inds)
            order, ptr = _pair_lists(host[:, 1], host[:, 2], edge.shape[0])
            lists = h2d(np.concatenate((order.reshape(-1), ptr.reshape(-1))), edge.device)
        ctx.n_rows = i1.shape[0]
        ctx.has_vis = vis is not None
        ctx.save_for_backward(edge_c, vis_c, i1, i2, lists)
        return out

    @staticmethod
    def backward(ctx, g):
        edge, vis, i1, i2, lists = ctx.saved_tensors
        R = ctx.n_rows
        d_edge, d_vis = _hip.pair_product_bwd(edge, i1, i2, vis, g.contiguous(), lists[:2 * R], lists[2 * R:])
        return d_edge, (d_vis if ctx.has_vis else None), None


def _cols_from(t, c0):
    """t[:, c0:] with its host mirror (a slice is a new tensor object: the mirror would be lost)"""
    out = t[:, c0:]
    if has_host(t):
        set_host(out, host_np(t)[:, c0:])
    return out


def _packing_plan(im_host):
    """host part of the LSTM packing order: per-image sort key offsets, and the time-major (TxB) gather of the image-sorted
    rois (reference :31-61, lib/pytorch_misc.py:365-384)"""
    num_im = int(im_host[-1]) + 1
    im_key = np.zeros(num_im, dtype=np.float32)
    lengths = []
    for i, s, e in enumerate_by_image(im_host):
        im_key[i] = 2 * (s - e) * num_im + i
        lengths.append(e - s)
    lengths = sorted(lengths, reverse=True)
    inds, ls_transposed = transpose_packed_sequence_inds(lengths)
    return im_key, np.asarray(inds, dtype=np.int64), ls_transposed


def _sort_by_score(im_inds, scores):
    """Permutation that orders rois for the LSTMs (reference :31-61): inside an image by descending score, images by
    decreasing object count, then time-major (TxB packed).  Returns (perm, inv_perm, batch size per timestep)."""
    im_host = host_np(im_inds)           # the mirror of GT-derived indices (no D2H synchronisation), else a copy
    im_key, inds, ls_transposed = _packing_plan(im_host)
    inds = h2d(inds, im_inds.device)
    roi_order = scores - 2 * h2d(im_key, scores.device)[im_inds]
    _, perm = torch.sort(roi_order, dim=0, descending=True, stable=True)
    perm = perm[inds]
    _, inv_perm = torch.sort(perm)
    return perm, inv_perm, ls_transposed


def _sort_by_score_host(im_inds, scores_host, device):
    """_sort_by_score for scores that are known on the HOST (box geometry of ground-truth boxes: their Blob mirror): the
    same fp32 arithmetic and the same stable sorts on CPU tensors, ONE upload of (perm | inv_perm) -- none of the ~22 small
    launches of the device version (two sorts, gathers, key arithmetic) on the context branch's critical path."""
    im_host = host_np(im_inds)
    im_key, inds, ls_transposed = _packing_plan(im_host)
    roi_order = scores_host - 2 * torch.from_numpy(im_key)[torch.from_numpy(np.ascontiguousarray(im_host, dtype=np.int64))]
    _, perm = torch.sort(roi_order, dim=0, descending=True, stable=True)
    perm = perm[torch.from_numpy(inds)]
    _, inv_perm = torch.sort(perm)
    both = h2d(torch.stack((perm, inv_perm)).numpy(), device)
    perm_d, inv_d = both[0], both[1]
    set_host(perm_d, perm.numpy())
    set_host(inv_d, inv_perm.numpy())
    return perm_d, inv_d, ls_transposed


class _LateBackward(torch.autograd.Function):
    """Identity whose backward runs the backward pass of a sub-graph that was built EARLIER in the forward.

    Why: autograd executes ready nodes in reverse creation order.  The two-stream forward enqueues the union-box branch
    first (a few chip-filling kernels on the main stream) and the context branch second (~200 small launches on the side
    stream), so backward starts with the context branch: the host spends ~4 ms enqueuing its small kernels while the main
    stream -- whose big fc6 / fc7 gradient GEMMs autograd has not reached yet -- sits idle (kernel trace of round 3:
    main stream idle 3.9 ms per step during the context backward, tools/trace_gaps.py).  Wrapping the union-box
    branch's OUTPUT in this node, created after the context branch, makes it the first thing backward runs: it launches the
    branch's whole backward (nested autograd call, gradients accumulate into the parameters as usual), and the context
    branch's launches then overlap with those GEMMs.  The branch's inputs (frozen feature map, boxes) need no gradient."""

    @staticmethod
    def forward(ctx, leaf, holder):
        ctx.holder = holder
        return leaf.view_as(leaf)

    @staticmethod
    def backward(ctx, g):
        # the holder keeps its entry: a second backward (retain_graph) reaches the same inner graph and gets autograd's own
        # error or result, not an IndexError.  Gradients of the inner graph's parameters are ACCUMULATED into .grad as a side
        # effect -- torch.autograd.grad(loss, params) does not see them; only loss.backward() is supported in this mode
        torch.autograd.backward(ctx.holder[0], g)
        return None, None


class LinearizedContext(nn.Module):
    """object context -> label decoder -> edge context"""

    def __init__(self, classes, rel_classes, mode='sgdet', embed_dim=200, hidden_dim=256, obj_dim=2048, nl_obj=2,
                 nl_edge=2, dropout_rate=0.2, order='confidence', pass_in_obj_feats_to_decoder=True,
                 pass_in_obj_feats_to_edge=True):
        super(LinearizedContext, self).__init__()
        self.classes = classes
        self.rel_classes = rel_classes
        assert mode in MODES
        self.mode = mode
        self.nl_obj, self.nl_edge = nl_obj, nl_edge
        self.embed_dim, self.hidden_dim, self.obj_dim = embed_dim, hidden_dim, obj_dim
        self.dropout_rate = dropout_rate
        self.pass_in_obj_feats_to_decoder = pass_in_obj_feats_to_decoder
        self.pass_in_obj_feats_to_edge = pass_in_obj_feats_to_edge
        assert order in ('size', 'confidence', 'random', 'leftright')
        self.order = order

        embed_vecs = obj_edge_vectors(self.classes, wv_dim=self.embed_dim)
        self.obj_embed = nn.Embedding(self.num_classes, self.embed_dim)
        self.obj_embed.weight.data = embed_vecs.clone()
        self.obj_embed2 = nn.Embedding(self.num_classes, self.embed_dim)
        self.obj_embed2.weight.data = embed_vecs.clone()

        self.pos_embed = nn.Sequential(
            nn.BatchNorm1d(4, momentum=BATCHNORM_MOMENTUM / 10.0),
            Linear(4, 128),
            ReLU(),
            Dropout(0.1),
        )
        if self.nl_obj > 0:
            self.obj_ctx_rnn = AlternatingHighwayLSTM(input_size=self.obj_dim + self.embed_dim + 128,
                                                      hidden_size=self.hidden_dim, num_layers=self.nl_obj,
                                                      recurrent_dropout_probability=dropout_rate)
            decoder_inputs_dim = self.hidden_dim
            if self.pass_in_obj_feats_to_decoder:
                decoder_inputs_dim += self.obj_dim + self.embed_dim
            self.decoder_rnn = DecoderRNN(self.classes, embed_dim=self.embed_dim, inputs_dim=decoder_inputs_dim,
                                          hidden_dim=self.hidden_dim, recurrent_dropout_probability=dropout_rate)
        else:
            self.decoder_lin = Linear(self.obj_dim + self.embed_dim + 128, self.num_classes)
        if self.nl_edge > 0:
            input_dim = self.embed_dim
            if self.nl_obj > 0:
                input_dim += self.hidden_dim
            if self.pass_in_obj_feats_to_edge:
                input_dim += self.obj_dim
            self.edge_ctx_rnn = AlternatingHighwayLSTM(input_size=input_dim, hidden_size=self.hidden_dim,
                                                       num_layers=self.nl_edge,
                                                       recurrent_dropout_probability=dropout_rate)

    @property
    def num_classes(self):
        return len(self.classes)

    @property
    def num_rels(self):
        return len(self.rel_classes)

    def sort_rois(self, batch_idx, confidence, box_priors):
        """`confidence` may be a callable: it is evaluated only for order == 'confidence' (three to five launches saved per
        call in the other orders).  Orders that depend on the box geometry alone ('size', 'leftright') are computed on the
        host when the boxes carry their Blob mirror (GT-box modes), once per forward: the object and the edge context use
        the same order."""
        geometric = self.order in ('size', 'leftright')
        if geometric and has_host(box_priors) and has_host(batch_idx):
            cache = getattr(self, '_order_cache', None)
            if cache is not None and cache[0] is box_priors and cache[1] is batch_idx:
                return cache[2]
            cxcywh = center_size(torch.from_numpy(np.ascontiguousarray(host_np(box_priors), dtype=np.float32)))
            if self.order == 'size':
                key = cxcywh[:, 2] * cxcywh[:, 3]
            else:
                key = cxcywh[:, 0]
            out = _sort_by_score_host(batch_idx, key / (key.max() + 1), box_priors.device)
            self._order_cache = (box_priors, batch_idx, out)
            return out
        cxcywh = center_size(box_priors)
        if self.order == 'size':
            sizes = cxcywh[:, 2] * cxcywh[:, 3]
            scores = sizes / (sizes.max() + 1)
        elif self.order == 'confidence':
            scores = confidence() if callable(confidence) else confidence
        elif self.order == 'random':
            scores = h2d(np.random.rand(batch_idx.size(0)).astype(np.float32), batch_idx.device)
        elif self.order == 'leftright':
            centers = cxcywh[:, 0]
            scores = centers / (centers.max() + 1)
        else:
            raise ValueError("invalid mode {}".format(self.order))
        return _sort_by_score(batch_idx, scores)

    def edge_ctx(self, obj_feats, obj_dists, im_inds, obj_preds, box_priors=None):
        obj_embed2 = self.obj_embed2(obj_preds)
        inp_feats = torch.cat((obj_embed2, obj_feats), 1)
        confidence = lambda: F.softmax(obj_dists, dim=1).detach().view(-1)[obj_preds.detach() + arange(obj_preds) * self.num_classes]
        perm, inv_perm, ls_transposed = self.sort_rois(im_inds, confidence, box_priors)
        edge_input_packed = PackedSequence(inp_feats[perm], torch.tensor(ls_transposed))
        edge_reps = self.edge_ctx_rnn(edge_input_packed)[0][0]
        return edge_reps[inv_perm]

    def obj_ctx(self, obj_feats, obj_dists, im_inds, obj_labels=None, box_priors=None, boxes_per_cls=None):
        confidence = lambda: F.softmax(obj_dists, dim=1).detach()[:, 1:].max(1)[0]
        perm, inv_perm, ls_transposed = self.sort_rois(im_inds, confidence, box_priors)
        obj_inp_rep = obj_feats[perm].contiguous()
        bs = torch.tensor(ls_transposed)
        encoder_rep = self.obj_ctx_rnn(PackedSequence(obj_inp_rep, bs))[0][0]
        if self.mode != 'predcls':
            dec_in = torch.cat((obj_inp_rep, encoder_rep), 1) if self.pass_in_obj_feats_to_decoder else encoder_rep
            # teacher forcing needs to know whether any label is background: from the labels' host mirror when they
            # came from a Blob (no synchronisation), else the decoder asks the device
            has_bg = bool((host_np(obj_labels) == 0).any()) if obj_labels is not None and has_host(obj_labels) else None
            obj_dists, obj_preds = self.decoder_rnn(
                PackedSequence(dec_in, bs),
                labels=obj_labels[perm] if obj_labels is not None else None, labels_have_background=has_bg,
                boxes_for_nms=boxes_per_cls[perm] if boxes_per_cls is not None else None)
            obj_preds = obj_preds[inv_perm]
            obj_dists = obj_dists[inv_perm]
        else:
            assert obj_labels is not None
            obj_preds = obj_labels
            obj_dists = to_onehot(obj_preds.detach(), self.num_classes)
        return obj_dists, obj_preds, encoder_rep[inv_perm]

    def forward(self, obj_fmaps, obj_logits, im_inds, obj_labels=None, box_priors=None, boxes_per_cls=None):
        obj_embed = linear(F.softmax(obj_logits, dim=1), self.obj_embed.weight.t())        # probs @ E
        pos_embed = self.pos_embed(center_size(box_priors))
        obj_pre_rep = torch.cat((obj_fmaps, obj_embed, pos_embed), 1)

        if self.nl_obj > 0:
            obj_dists2, obj_preds, obj_ctx = self.obj_ctx(obj_pre_rep, obj_logits, im_inds, obj_labels, box_priors,
                                                          boxes_per_cls)
        else:
            if self.mode == 'predcls':
                obj_dists2 = to_onehot(obj_labels.detach(), self.num_classes)
            else:
                obj_dists2 = self.decoder_lin(obj_pre_rep)
            if self.mode == 'sgdet' and not self.training:
                probs = F.softmax(obj_dists2, 1).detach()
                nms_mask = torch.zeros_like(probs)
                for c_i in range(1, obj_dists2.size(1)):
                    keep = apply_nms(probs[:, c_i], boxes_per_cls.detach()[:, c_i], pre_nms_topn=probs.size(0),
                                     post_nms_topn=probs.size(0), nms_thresh=0.3)
                    nms_mask[:, c_i][keep] = 1
                obj_preds = (nms_mask * probs)[:, 1:].max(1)[1] + 1
            else:
                obj_preds = obj_labels if obj_labels is not None else obj_dists2[:, 1:].max(1)[1] + 1
            obj_ctx = obj_pre_rep

        edge_ctx = None
        if self.nl_edge > 0:
            edge_ctx = self.edge_ctx(torch.cat((obj_fmaps, obj_ctx), 1) if self.pass_in_obj_feats_to_edge else obj_ctx,
                                     obj_dists=obj_dists2.detach(), im_inds=im_inds, obj_preds=obj_preds,
                                     box_priors=box_priors)
        self._order_cache = None          # the batch's tensors are not kept on the module between forwards
        return obj_dists2, obj_preds, edge_ctx


class RelModel(nn.Module):
    def __init__(self, classes, rel_classes, mode='sgdet', num_gpus=1, use_vision=True, require_overlap_det=True,
                 embed_dim=200, hidden_dim=256, pooling_dim=2048, nl_obj=1, nl_edge=2, use_resnet=False,
                 order='confidence', thresh=0.01, use_proposals=False, pass_in_obj_feats_to_decoder=True,
                 pass_in_obj_feats_to_edge=True, rec_dropout=0.0, use_bias=True, use_tanh=True, limit_vision=True,
                 max_per_img=64, freq_counts=None, resnet_obj_fmap=None):
        """Arguments as in the reference (lib/rel_model.py:303-308) plus two additions: `max_per_img` (the reference
        hard-codes 64, :345; BASELINE cfg5 needs 80) and `freq_counts=(fg_matrix, bg_matrix)` to inject the
        predicate statistics instead of scanning the dataset at construction time."""
        super(RelModel, self).__init__()
        self.classes = classes
        self.rel_classes = rel_classes
        self.num_gpus = num_gpus
        assert mode in MODES
        self.mode = mode
        self.use_resnet = use_resnet
        if use_resnet and resnet_obj_fmap != 'layer4':
            # The reference cannot run this configuration: with use_resnet it never creates `roi_fmap_obj`
            # (lib/rel_model.py:360-365) but obj_feature_map uses it unconditionally (:448) -> AttributeError on the
            # first forward.  `resnet_obj_fmap='layer4'` is the documented repair --
            # the object branch gets its own copy of the layer4 stack, exactly as the VGG branch has its own fc6 / fc7 copy.
            raise NotImplementedError('RelModel(use_resnet=True) is broken in the reference itself '
                                      "(rel_model.py:360-365 vs :448); pass resnet_obj_fmap='layer4' for the repaired model")
        if use_resnet and pooling_dim != 2048:
            raise ValueError('the ResNet relation head produces 2048 features per pair: pooling_dim must be 2048')
        self.pooling_size = 7
        self.embed_dim = embed_dim
        self.hidden_dim = hidden_dim
        self.obj_dim = 2048 if use_resnet else 4096          # lib/rel_model.py:330
        self.pooling_dim = pooling_dim
        self.use_bias = use_bias
        self.use_vision = use_vision
        self.use_tanh = use_tanh
        self.limit_vision = limit_vision
        self.require_overlap = require_overlap_det and self.mode == 'sgdet'
        self.sampler_rs = None            # optional numpy RandomState for the relation sampler (reproducible runs)
        # Run the object/edge-context branch (RoI fc6/fc7 on ~100 rows, ~100 latency-bound LSTM/decoder step kernels
        # that occupy at most half of the CUs) on a second HIP stream, concurrently with the union-box relation head
        # (a handful of chip-filling MFMA GEMMs).  The two branches only meet at the subject/object product.
        self.overlap_streams = os.environ.get('MOTIFS_OVERLAP', '1') != '0'     # context branch on a second HIP stream
        self._side_stream = None
        # Backward ORDER of the two branches (see _LateBackward): '0' (default) = autograd's own order (context branch first);
        # 'auto' = the union-box branch's backward is issued first whenever the two-stream forward is used with a frozen
        # trunk; 'force' = also on one stream / on the CPU (tests).  Measured in round 4 on one box, alternating, unprofiled
        # (profiles/r04_variance.jsonl): 18.75 ms per step with it, 18.80 without, both +-0.1 ms -- no gain, so it is off
        self.late_vr_backward = os.environ.get('MOTIFS_LATE_VR', '0')
        # detector stage one batch ahead (detect_ahead): worker thread, its stream, the stages in flight keyed by id(x)
        self._ahead_pool, self._ahead_tls, self._ahead = None, None, {}

        self.detector = ObjectDetector(
            classes=classes,
            mode=('proposals' if use_proposals else 'refinerels') if mode == 'sgdet' else 'gtbox',
            use_resnet=use_resnet, thresh=thresh, max_per_img=max_per_img)

        self.context = LinearizedContext(self.classes, self.rel_classes, mode=self.mode, embed_dim=self.embed_dim,
                                         hidden_dim=self.hidden_dim, obj_dim=self.obj_dim, nl_obj=nl_obj,
                                         nl_edge=nl_edge, dropout_rate=rec_dropout, order=order,
                                         pass_in_obj_feats_to_decoder=pass_in_obj_feats_to_decoder,
                                         pass_in_obj_feats_to_edge=pass_in_obj_feats_to_edge)

        self.union_boxes = UnionBoxesAndFeats(pooling_size=self.pooling_size, stride=16, dim=1024 if use_resnet else 512)

        if use_resnet:
            from lib.resnet import Layer4Stack, AvgPoolNHWC
            self.roi_fmap = nn.Sequential(Layer4Stack(relu_end=False), AvgPoolNHWC(), Flattener())
            self.roi_fmap_obj = nn.Sequential(Layer4Stack(relu_end=False), AvgPoolNHWC(), Flattener())   # the repair
        else:
            roi_fmap = [Flattener(),
                        load_vgg(use_dropout=False, use_relu=False, use_linear=pooling_dim == 4096).classifier]
            if pooling_dim != 4096:
                roi_fmap.append(Linear(4096, pooling_dim))
            self.roi_fmap = nn.Sequential(*roi_fmap)
            self.roi_fmap_obj = load_vgg().classifier

        self.post_lstm = Linear(self.hidden_dim, self.pooling_dim * 2)
        self.post_lstm.weight.data.normal_(0, 10.0 * math.sqrt(1.0 / self.hidden_dim))
        self.post_lstm.bias.data.zero_()
        if nl_edge == 0:
            self.post_emb = nn.Embedding(self.num_classes, self.pooling_dim * 2)
            self.post_emb.weight.data.normal_(0, math.sqrt(1.0))
        self.rel_compress = Linear(self.pooling_dim, self.num_rels, bias=True)
        nn.init.xavier_normal_(self.rel_compress.weight, gain=1.0)
        if self.use_bias:
            fg, bg = freq_counts if freq_counts is not None else (None, None)
            self.freq_bias = FrequencyBias(fg_matrix=fg, bg_matrix=bg, num_objs=self.num_classes,
                                           num_rels=self.num_rels)

    @property
    def num_classes(self):
        return len(self.classes)

    @property
    def num_rels(self):
        return len(self.rel_classes)

    def _late_ok(self, fmap):
        # the re-ordered backward hands no gradient to the feature map: only with a frozen trunk (the relation drivers)
        return self.training and torch.is_grad_enabled() and not fmap.requires_grad

    def visual_rep(self, features, rois, pair_inds):
        assert pair_inds.size(1) == 2
        uboxes = self.union_boxes(features, rois, pair_inds)
        return self.roi_fmap(uboxes)

    def _side(self, device):
        """the context branch's stream.  High priority: that branch is a long chain of small, latency-bound launches; its
        workgroups should not queue behind the union-box branch's chip-filling GEMM / conv tiles.  MOTIFS_SIDE_PRIORITY=0
        restores the default priority (A/B)"""
        if self._side_stream is None:
            prio = -1 if os.environ.get('MOTIFS_SIDE_PRIORITY', '-1') != '0' else 0
            self._side_stream = torch.cuda.Stream(device=device, priority=prio)
        return self._side_stream

    def get_rel_inds(self, rel_labels, im_inds, box_priors):
        """candidate (image, subject, object) rows: the sampled labels in training, every ordered pair of distinct
        boxes of an image in eval (overlapping ones only for sgdet)"""
        if self.training:
            out = rel_labels[:, :3].detach().clone()
            if has_host(rel_labels):
                set_host(out, host_np(rel_labels)[:, :3])
            return out
        if not self.require_overlap and has_host(im_inds):
            # GT-box evaluation: every ordered pair of distinct boxes of an image, enumerated on the host from the
            # mirrored image indices in the order nonzero() gives (row-major) -- no device->host synchronisation
            im = host_np(im_inds)
            cand = im[:, None] == im[None, :]
            np.fill_diagonal(cand, False)
            ij = np.column_stack(np.nonzero(cand)).astype(np.int64)
            if ij.shape[0] == 0:
                ij = np.zeros((1, 2), dtype=np.int64)
            cand_np = np.ascontiguousarray(np.column_stack((im[ij[:, 0]].astype(np.int64), ij)))
            return set_host(h2d(cand_np, im_inds.device), cand_np)
        rel_cands = im_inds[:, None] == im_inds[None]
        rel_cands.fill_diagonal_(False)
        if self.require_overlap:
            rel_cands = rel_cands & (bbox_overlaps(box_priors.detach().contiguous(),
                                                   box_priors.detach().contiguous()) > 0)
        rel_cands = rel_cands.nonzero()
        if rel_cands.numel() == 0:
            rel_cands = im_inds.new_zeros(1, 2)
        return torch.cat((im_inds[rel_cands[:, 0]][:, None], rel_cands), 1)

    def obj_feature_map(self, features, rois):
        pooled = RoIAlignFunction(self.pooling_size, self.pooling_size, spatial_scale=1 / 16)(features, rois)
        return self.roi_fmap_obj(pooled if self.use_resnet else pooled.view(rois.size(0), -1))

    # ---- the detector stage: everything up to the sampled relation labels -------------------------------------------------
    def _detect(self, x, im_sizes, image_offset, gt_boxes=None, gt_classes=None, gt_rels=None, proposals=None,
                train_anchor_inds=None, ahead=False):
        """boxes, labels, RoI logits and the feature map of a batch (reference :466-475).  ahead: the call is a detect_ahead stage
        on the worker thread -- the detector is frozen there, so the stage needs no ordering against a deferred optimizer step;
        the forward that collects it waits on the MAIN stream (forward(): _hip.wait_param_update after _take_ahead)"""
        self.detector.sampler_rs = self.sampler_rs
        # a FusedClipSGD step deferred to its own stream (lib/optim.py: overlap_next_forward) may still be updating the trainable
        # parameters: the frozen detector stage runs beside it, everything after it waits
        detector_trains = x.is_cuda and any(p.requires_grad for p in self.detector.parameters())
        if detector_trains:
            _hip.wait_param_update()
        result = self.detector(x, im_sizes, image_offset, gt_boxes, gt_classes, gt_rels, proposals,
                               train_anchor_inds, return_fmap=True)
        if x.is_cuda and not detector_trains and not ahead:
            _hip.wait_param_update()
        if result.is_none():
            return result, None
        im_inds = result.im_inds - image_offset
        if has_host(result.im_inds):
            set_host(im_inds, host_np(result.im_inds) - image_offset)
        return result, im_inds

    def _sample_relations(self, result, im_inds, image_offset, gt_boxes, gt_classes, gt_rels):
        """SGDet training: relation rows sampled against the ground truth on the HOST (reference :480-487)"""
        if self.training and result.rel_labels is None:
            assert self.mode == 'sgdet'
            # index / ground-truth tensors are passed as they are (no .detach(): a new tensor object would drop the host mirror)
            result.rel_labels = rel_assignments(im_inds, result.rm_box_priors.detach(), result.rm_obj_labels.detach(),
                                                gt_boxes, gt_classes, gt_rels, image_offset,
                                                filter_non_overlap=True, num_sample_per_gt=1, rs=self.sampler_rs)

    def detect_ahead(self, x, im_sizes, image_offset, gt_boxes=None, gt_classes=None, gt_rels=None, proposals=None,
                     train_anchor_inds=None, return_fmap=False):
        """Start the detector stage of a LATER forward(x, ...) now, on a worker thread and its own HIP stream.

        Why: with a frozen detector (the relation drivers: reference models/train_rels.py:75-77) that stage of batch i+1 depends
        on nothing batch i's step changes, and it is where the host has to WAIT for the device -- the proposal counts after the
        RPN's NMS, the kept detections after the per-class NMS, the boxes that rel_assignments matches to the ground truth on
        the host.  Issued in line, each wait finds the queue empty behind it: the SGDet step had the device idle for 9.6 of
        its 41 ms (profiles/r04_cfg3_trace_gaps_final.txt).  Ahead, the waits of batch i+1 are served while the relation
        stage / backward / optimizer of batch i are queued on the main stream.  Measured (gpurun r05_c16, b = 6): in line
        160 img/s, one batch ahead 183 (the main thread still waits for the stage it has just asked for), two ahead 217-223,
        three 214; a second worker 194-200 (two threads contending for the interpreter); SGDet evaluation 66 -> 77 img/s.

        Call it with the arguments of the forward it belongs to, any time before that forward; the forward recognises its `x`
        (same tensor object) and takes the stage.  Results are those of the in-line order (tests/test_gpu_sgdet.py); only
        the interleaving of random draws between the two stages differs when the detector's dropout is active.  Refused
        (returns False, the forward then runs the stage in line) when the detector has trainable parameters or the seeded host
        mask stream of the parity tests is in use."""
        if any(p.requires_grad for p in self.detector.parameters()) or rng_mod._host_rng is not None:
            return False
        if id(x) in self._ahead:
            return True
        if self._ahead_pool is None:
            import sys
            import threading
            from concurrent.futures import ThreadPoolExecutor
            # ONE worker: stages run in the order they were asked for, so the relation sampler's random stream
            # (self.sampler_rs) is consumed in batch order, as in line.  MOTIFS_AHEAD_WORKERS=2 (measurements): two stages
            # at a time, each worker on its own stream
            self._ahead_pool = ThreadPoolExecutor(max_workers=max(1, int(os.environ.get('MOTIFS_AHEAD_WORKERS', '1'))),
                                                  thread_name_prefix='detect_ahead')
            self._ahead_tls = threading.local()
            # the worker comes back from a device wait needing the interpreter lock the main thread holds while it enqueues:
            # bound that hand-over (default 5 ms) well below the waits it is there to hide
            sys.setswitchinterval(min(sys.getswitchinterval(), 2e-4))
        args = (x, im_sizes, image_offset, gt_boxes, gt_classes, gt_rels, proposals, train_anchor_inds)
        grad, training = torch.is_grad_enabled(), self.training
        ready = None
        if x.is_cuda:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(x.device))          # the inputs' uploads are ordered on the caller's stream

        def run():
            assert self.training == training, 'train() / eval() was switched between detect_ahead() and its forward()'
            with torch.set_grad_enabled(grad):
                if ready is None:
                    result, im_inds = self._detect(*args, ahead=True)
                    if im_inds is not None:
                        self._sample_relations(result, im_inds, image_offset, gt_boxes, gt_classes, gt_rels)
                    return result, im_inds, None
                torch.cuda.set_device(x.device)
                stream = getattr(self._ahead_tls, 'stream', None)      # one stream per worker thread
                if stream is None:
                    # default priority: measured (gpurun r05_c16, cfg3, two batches ahead) 223 img/s against 217 with a
                    # high-priority stream (MOTIFS_AHEAD_PRIORITY=-1) -- the stage is two batches early, nothing waits for it
                    prio = -1 if os.environ.get('MOTIFS_AHEAD_PRIORITY', '0') == '-1' else 0
                    stream = self._ahead_tls.stream = torch.cuda.Stream(device=x.device, priority=prio)
                with torch.cuda.stream(stream):
                    stream.wait_event(ready)
                    for t in args:
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(stream)
                    result, im_inds = self._detect(*args, ahead=True)
                    if im_inds is not None:
                        self._sample_relations(result, im_inds, image_offset, gt_boxes, gt_classes, gt_rels)
                    done = torch.cuda.Event()
                    done.record(stream)
                return result, im_inds, done

        self._ahead[id(x)] = _AheadStage(x, self._ahead_pool.submit(run))
        return True

    def __getstate__(self):
        # copy.deepcopy / pickle of the module: the worker pool and the stages in flight belong to THIS object
        state = self.__dict__.copy()
        state['_ahead_pool'], state['_ahead_tls'], state['_ahead'] = None, None, {}
        return state

    def detect_ahead_blob(self, batch):
        """detect_ahead for a dataloader blob (the argument of `model[blob]`)"""
        batch.scatter()
        return self.detect_ahead(*batch[0])

    def ahead_pending(self):
        return len(self._ahead)

    def ahead_drain(self):
        """wait for every stage in flight (they stay collectable): the host-side work of those stages is over afterwards"""
        for st in list(self._ahead.values()):
            st.future.exception()

    def ahead_discard(self):
        """drop the stages that no forward will collect (a loop left early); their results are waited for, then released"""
        self.ahead_drain()
        self._ahead.clear()

    def _take_ahead(self, x):
        st = self._ahead.pop(id(x), None)
        if st is None or st.x is not x:
            return None
        result, im_inds, done = st.future.result()         # re-raises what the stage raised
        if done is not None:
            main = torch.cuda.current_stream(x.device)
            main.wait_event(done)
            # the stage's tensors live in the worker stream's pool: tell the allocator that this stream reads them too
            seen = [im_inds] + list(vars(result).values())
            for t in seen:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(main)
        return result, im_inds

    def forward(self, x, im_sizes, image_offset, gt_boxes=None, gt_classes=None, gt_rels=None, proposals=None,
                train_anchor_inds=None, return_fmap=False):
        ahead = self._take_ahead(x) if self._ahead else None
        if ahead is not None and x.is_cuda:
            # the stage ran on the worker's stream: the wait for a deferred optimizer step (the relation stage below reads the
            # trainable weights) belongs to THIS stream and was not taken there
            _hip.wait_param_update()
        result, im_inds = ahead if ahead is not None else self._detect(x, im_sizes, image_offset, gt_boxes, gt_classes, gt_rels,
                                                                        proposals, train_anchor_inds)
        if result.is_none():
            return ValueError("heck")            # the reference returns (not raises) this, :474-475

        boxes = result.rm_box_priors
        self.last_detector_obj_dists = result.rm_obj_dists.detach()   # the detector's logits of the kept boxes (the
        if _HOST_GEOMETRY and x.is_cuda and has_host(im_inds) and has_host(boxes) and not boxes.requires_grad:
            # GT-box modes: [image, x1, y1, x2, y2] assembled on the host from the two mirrors -- one upload instead of a cast + a
            # concatenation, and the union-box geometry below (lib/get_union_boxes.py) needs no launch at all
            rois_np = np.ascontiguousarray(np.column_stack((host_np(im_inds).astype(np.float32), host_np(boxes).astype(np.float32))))
            rois = set_host(h2d(rois_np, boxes.device), rois_np)
        else:
            rois = torch.cat((im_inds[:, None].float(), boxes), 1)    # field is overwritten by the context's below)
        fmap = result.fmap.detach()

        def context_branch():
            result.obj_fmap = self.obj_feature_map(fmap, rois)
            result.rm_obj_dists, result.obj_preds, edge_ctx = self.context(
                result.obj_fmap, result.rm_obj_dists.detach(), im_inds,
                result.rm_obj_labels if self.training or self.mode == 'predcls' else None,
                boxes if not boxes.requires_grad else boxes.detach(),      # (a detached copy would drop the host mirror)
                result.boxes_all)
            er = self.post_emb(result.obj_preds) if edge_ctx is None else self.post_lstm(edge_ctx)
            return er.view(er.size(0), 2, self.pooling_dim)

        # Two-stream overlap only with device-side randomness: the seeded host mask stream used by the parity tests
        # fixes the draw ORDER (context before vision, as in the reference), which sequential issue preserves.
        overlap = (self.overlap_streams and self.use_vision and x.is_cuda and rng_mod._host_rng is None)
        early_edge_rep = None
        if overlap and self.training and result.rel_labels is None and SGDET_CONTEXT_FIRST:
            # SGDet training samples its relations on the HOST below; the context branch needs none of that, so its kernels go
            # to the side stream first and run under the sampling
            main, side = torch.cuda.current_stream(), self._side(x.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for t in (fmap, rois, im_inds, boxes, result.rm_obj_dists):
                    t.record_stream(side)
                early_edge_rep = context_branch()
        self._sample_relations(result, im_inds, image_offset, gt_boxes, gt_classes, gt_rels)

        rel_inds = self.get_rel_inds(result.rel_labels, im_inds, boxes)
        if self.training and getattr(self, 'rows_hook', None) is not None:
            # multi-GPU: both loss terms' row counts are known here, long before the losses: lib.dist.RowWeights launches
            # its (tiny, asynchronous) all-reduce now instead of blocking in front of backward
            self.rows_hook(int(result.rm_obj_labels.shape[0]), int(result.rel_labels.shape[0]))
        vr = None
        if overlap:
            main, side = torch.cuda.current_stream(), self._side(x.device)
            late = self._late_ok(fmap) and self.late_vr_backward in ('auto', '1', 'force')
            if early_edge_rep is None:
                side.wait_stream(main)                               # fmap / rois / labels are ready
                vr = self.visual_rep(fmap, rois, _cols_from(rel_inds, 1))     # big kernels first: the GPU is busy while
                with torch.cuda.stream(side):                         # the host enqueues the small ones
                    for t in (fmap, rois, im_inds, boxes, result.rm_obj_dists):
                        t.record_stream(side)
                    edge_rep = context_branch()
            else:
                vr = self.visual_rep(fmap, rois, _cols_from(rel_inds, 1))
                edge_rep = early_edge_rep
            marks = getattr(self, 'stream_marks', None)      # measurement hook (bench.py): how long the main stream waits here
            if marks is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(main)
            main.wait_stream(side)
            if marks is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(main)
                marks.append((e0, e1))
            for t in (edge_rep, result.obj_fmap, result.rm_obj_dists, result.obj_preds):
                if torch.is_tensor(t):
                    t.record_stream(main)
            if late and vr.requires_grad:
                vr = _LateBackward.apply(vr.detach().requires_grad_(True), [vr])
        else:
            edge_rep = context_branch()
            if self.use_vision and self.late_vr_backward == 'force' and self._late_ok(fmap):
                vr_inner = self.visual_rep(fmap, rois, _cols_from(rel_inds, 1))
                vr = _LateBackward.apply(vr_inner.detach().requires_grad_(True), [vr_inner]) if vr_inner.requires_grad else vr_inner
        if self.use_vision and vr is None:
            vr = self.visual_rep(fmap, rois, _cols_from(rel_inds, 1))
        fused = (_PAIR_PRODUCT and x.is_cuda and has_host(rel_inds) and edge_rep.dtype == torch.float32 and self.pooling_dim % 4 == 0
                 and not (self.use_vision and self.limit_vision))
        if fused:
            prod_rep = _PairProductFn.apply(edge_rep, vr if self.use_vision else None, rel_inds)
        else:
            subj_rep, obj_rep = edge_rep[:, 0], edge_rep[:, 1]
            prod_rep = subj_rep[rel_inds[:, 1]] * obj_rep[rel_inds[:, 2]]
            if self.use_vision:
                if self.limit_vision:
                    prod_rep = torch.cat((prod_rep[:, :2048] * vr[:, :2048], prod_rep[:, 2048:]), 1)
                else:
                    prod_rep = prod_rep * vr
        if self.use_tanh:
            prod_rep = torch.tanh(prod_rep)

        result.rel_dists = self.rel_compress(prod_rep)
        if self.use_bias:
            fb = self.freq_bias.obj_baseline.weight
            if (_PAIR_PRODUCT and x.is_cuda and result.rel_dists.dtype == torch.float32 and result.obj_preds.dtype == torch.int64
                    and fb.is_contiguous() and 0 < rel_inds.shape[0] <= _hip.FREQ_BIAS_MAX_ROWS):
                result.rel_dists = _FreqBiasAddFn.apply(result.rel_dists, fb, result.obj_preds, rel_inds, self.freq_bias.num_objs)
            else:
                result.rel_dists = result.rel_dists + self.freq_bias.index_with_labels(torch.stack((
                    result.obj_preds[rel_inds[:, 1]], result.obj_preds[rel_inds[:, 2]]), 1))
        if self.training:
            return result

        twod_inds = arange(result.obj_preds) * self.num_classes + result.obj_preds
        result.obj_scores = F.softmax(result.rm_obj_dists, dim=1).view(-1)[twod_inds]
        if self.mode == 'sgdet':
            bboxes = result.boxes_all.view(-1, 4)[twod_inds].view(result.boxes_all.size(0), 4)
        else:
            bboxes = result.rm_box_priors
        rel_rep = F.softmax(result.rel_dists, dim=1)
        self.last_eval_result = result            # raw logits of the last eval forward (tests / debugging)
        to_numpy = not getattr(self, 'eval_on_device', False)
        out = filter_dets(bboxes, result.obj_scores, result.obj_preds, rel_inds[:, 1:], rel_rep, to_numpy=to_numpy)
        if to_numpy and x.is_cuda:
            _hip.check_faults()                   # the D2H copies above synchronised: a timed-out LSTM launch is visible now
        return out

    def __getitem__(self, batch):
        """`detector[blob]` (reference :549-560); one replica per process, see lib/dist.py for the multi-GPU path"""
        batch.scatter()
        if self.num_gpus != 1:
            raise RuntimeError('in-process multi-GPU replication is replaced by one process per GPU: launch with '
                               'torchrun and keep num_gpus=1 per rank')
        return self(*batch[0])
