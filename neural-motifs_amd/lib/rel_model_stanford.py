"""
Iterative message passing baseline (Xu et al., "Scene graph generation by iterative message passing") with the
reference's interface -- lib/rel_model_stanford.py:20-218: a `RelModel` without the LSTM context whose object /
relation logits come from three rounds of GRU message passing between object nodes and relation edges
(an adjacent model variant that reuses every kernel built for the hot path).

Dense work runs on the HIP GEMM (`lib.hip_ops.Linear`): the unary projections, the two projections of each GRU cell,
the four gate heads and the node <- edge aggregation (incidence matrix x edge messages, as the reference writes it);
gate nonlinearities are elementwise device ops.  Parameter names / shapes are nn.GRUCell's and nn.Linear's
(`edge_gru.weight_ih [3H,H]`, `sub_vert_w_fc.0.weight [1,2H]`, ...), so reference checkpoints load.
"""
import math

import torch
import torch.nn as nn
from torch.nn import functional as F

from lib.fpn.proposal_assignments.rel_assignments import rel_assignments
from lib.hip_ops import Linear, linear
from lib.object_detector import filter_det
from lib.pytorch_misc import arange
from lib.rel_model import RelModel
from lib.surgery import filter_dets

MODES = ('sgdet', 'sgcls', 'predcls')
SIZE = 512


class GRUCell(nn.Module):
    """nn.GRUCell (parameters weight_ih [3H,in], weight_hh [3H,H], bias_ih, bias_hh; gate order r, z, n) on the HIP GEMM"""

    def __init__(self, input_size, hidden_size):
        super(GRUCell, self).__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.weight_ih = nn.Parameter(torch.empty(3 * hidden_size, input_size))
        self.weight_hh = nn.Parameter(torch.empty(3 * hidden_size, hidden_size))
        self.bias_ih = nn.Parameter(torch.empty(3 * hidden_size))
        self.bias_hh = nn.Parameter(torch.empty(3 * hidden_size))
        stdv = 1.0 / math.sqrt(hidden_size)
        for p in self.parameters():
            nn.init.uniform_(p, -stdv, stdv)

    def forward(self, x, hx):
        H = self.hidden_size
        gi = linear(x, self.weight_ih, self.bias_ih)
        gh = linear(hx, self.weight_hh, self.bias_hh)
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        return (1 - z) * n + z * hx


def _gate(dim):
    return nn.Sequential(Linear(dim, 1), nn.Sigmoid())


class RelModelStanford(RelModel):
    def __init__(self, classes, rel_classes, mode='sgdet', num_gpus=1, require_overlap_det=True, use_resnet=False,
                 use_proposals=False, **kwargs):
        super(RelModelStanford, self).__init__(classes, rel_classes, mode=mode, num_gpus=num_gpus,
                                               require_overlap_det=require_overlap_det, use_resnet=use_resnet,
                                               nl_obj=0, nl_edge=0, use_proposals=use_proposals, thresh=0.01,
                                               pooling_dim=4096)
        del self.context
        del self.post_lstm
        del self.post_emb
        self.rel_fc = Linear(SIZE, self.num_rels)
        self.obj_fc = Linear(SIZE, self.num_classes)
        self.obj_unary = Linear(self.obj_dim, SIZE)
        self.edge_unary = Linear(4096, SIZE)
        self.edge_gru = GRUCell(input_size=SIZE, hidden_size=SIZE)
        self.node_gru = GRUCell(input_size=SIZE, hidden_size=SIZE)
        self.n_iter = 3
        self.sub_vert_w_fc = _gate(SIZE * 2)
        self.obj_vert_w_fc = _gate(SIZE * 2)
        self.out_edge_w_fc = _gate(SIZE * 2)
        self.in_edge_w_fc = _gate(SIZE * 2)

    def message_pass(self, rel_rep, obj_rep, rel_inds):
        """rel_rep [num_rel,512], obj_rep [num_obj,512], rel_inds [num_rel,2] -> (object logits, relation logits)"""
        n_rel, n_obj = rel_rep.size(0), obj_rep.size(0)
        numer = torch.arange(0, n_rel, device=rel_inds.device)
        objs_to_outrels = rel_rep.new_zeros(n_obj, n_rel)
        objs_to_outrels.view(-1)[rel_inds[:, 0] * n_rel + numer] = 1
        objs_to_inrels = rel_rep.new_zeros(n_obj, n_rel)
        objs_to_inrels.view(-1)[rel_inds[:, 1] * n_rel + numer] = 1
        hx_rel = rel_rep.new_zeros(n_rel, SIZE)
        hx_obj = obj_rep.new_zeros(n_obj, SIZE)
        vert_factor = [self.node_gru(obj_rep, hx_obj)]
        edge_factor = [self.edge_gru(rel_rep, hx_rel)]
        for i in range(3):
            sub_vert = vert_factor[i][rel_inds[:, 0]]
            obj_vert = vert_factor[i][rel_inds[:, 1]]
            weighted_sub = self.sub_vert_w_fc(torch.cat((sub_vert, edge_factor[i]), 1)) * sub_vert
            weighted_obj = self.obj_vert_w_fc(torch.cat((obj_vert, edge_factor[i]), 1)) * obj_vert
            edge_factor.append(self.edge_gru(weighted_sub + weighted_obj, edge_factor[i]))
            pre_out = self.out_edge_w_fc(torch.cat((sub_vert, edge_factor[i]), 1)) * edge_factor[i]
            pre_in = self.in_edge_w_fc(torch.cat((obj_vert, edge_factor[i]), 1)) * edge_factor[i]
            # node <- edges: incidence x messages (weight = the message matrix as the [K=n_rel, N=512] operand)
            vert_ctx = linear(objs_to_outrels, pre_out.t()) + linear(objs_to_inrels, pre_in.t())
            vert_factor.append(self.node_gru(vert_ctx, vert_factor[i]))
        return self.obj_fc(vert_factor[-1]), self.rel_fc(edge_factor[-1])

    def forward(self, x, im_sizes, image_offset, gt_boxes=None, gt_classes=None, gt_rels=None, proposals=None,
                train_anchor_inds=None, return_fmap=False):
        self.detector.sampler_rs = self.sampler_rs
        result = self.detector(x, im_sizes, image_offset, gt_boxes, gt_classes, gt_rels, proposals,
                               train_anchor_inds, return_fmap=True)
        if result.is_none():
            return ValueError("heck")
        im_inds = result.im_inds - image_offset
        boxes = result.rm_box_priors
        if self.training and result.rel_labels is None:
            assert self.mode == 'sgdet'
            result.rel_labels = rel_assignments(im_inds.detach(), boxes.detach(), result.rm_obj_labels.detach(),
                                                gt_boxes.detach(), gt_classes.detach(), gt_rels.detach(), image_offset,
                                                filter_non_overlap=True, num_sample_per_gt=1, rs=self.sampler_rs)
        rel_inds = self.get_rel_inds(result.rel_labels, im_inds, boxes)
        rois = torch.cat((im_inds[:, None].float(), boxes), 1)
        visual_rep = self.visual_rep(result.fmap, rois, rel_inds[:, 1:])
        result.obj_fmap = self.obj_feature_map(result.fmap.detach(), rois)
        result.rm_obj_dists, result.rel_dists = self.message_pass(
            self.edge_unary(visual_rep, relu=True), self.obj_unary(result.obj_fmap), rel_inds[:, 1:])
        if self.training:
            return result

        if self.mode == 'predcls':
            result.obj_scores = result.rm_obj_dists.new_ones(gt_classes.size(0))
            result.obj_preds = gt_classes[:, 1]
        elif self.mode == 'sgdet':
            order, obj_scores, obj_preds = filter_det(F.softmax(result.rm_obj_dists, 1), result.boxes_all, start_ind=0,
                                                      max_per_img=100, thresh=0.00, pre_nms_topn=6000,
                                                      post_nms_topn=300, nms_thresh=0.3, nms_filter_duplicates=True)
            idx, perm = torch.sort(order, dim=0, stable=True)
            result.obj_preds = rel_inds.new_ones(result.rm_obj_dists.size(0))
            result.obj_scores = result.rm_obj_dists.new_zeros(result.rm_obj_dists.size(0))
            result.obj_scores[idx] = obj_scores[perm]
            result.obj_preds[idx] = obj_preds[perm]
        else:
            scores_nz = F.softmax(result.rm_obj_dists, 1).detach().clone()
            scores_nz[:, 0] = 0.0
            sorted_scores, score_ord = scores_nz[:, 1:].sort(dim=1, descending=True, stable=True)
            result.obj_preds = score_ord[:, 0] + 1
            result.obj_scores = sorted_scores[:, 0]
        twod_inds = arange(result.obj_preds) * self.num_classes + result.obj_preds
        if self.mode == 'sgdet':
            bboxes = result.boxes_all.view(-1, 4)[twod_inds].view(result.boxes_all.size(0), 4)
        else:
            bboxes = result.rm_box_priors
        rel_rep = F.softmax(result.rel_dists, 1)
        return filter_dets(bboxes, result.obj_scores, result.obj_preds, rel_inds[:, 1:], rel_rep,
                           to_numpy=not getattr(self, 'eval_on_device', False))
