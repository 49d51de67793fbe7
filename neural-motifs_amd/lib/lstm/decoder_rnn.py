"""
Highway-LSTM label decoder (reference lib/lstm/decoder_rnn.py:40-251) on the gfx950 kernels.

Same constructor, parameters (obj_embed [152,100], input_linearity, state_linearity, out) and
`forward(PackedSequence, labels=None, boxes_for_nms=None) -> (out_dists, out_commitments)`.

How the per-timestep Python loop of the reference is restructured:
  * x_t = [enc_t || emb(prev_t)], so  input_linearity(x_t) = W_enc enc_t + b  +  W_emb emb(prev_t).
    The first term is ONE MFMA GEMM over all packed rows; the second is a row gather from the 152 x 6H table
    E_proj = obj_embed.weight @ W_emb^T (one tiny GEMM per forward).  No per-step input GEMM remains.
  * the recurrence itself is one fused GEMV+gate kernel per step (mh_hwlstm_cell_fwd) inside a single autograd
    Function whose backward replays the steps in reverse (mh_hwlstm_cell_bwd + mh_gemv_rows) and computes the
    recurrent weight gradient with ONE GEMM over all steps.
  * teacher forcing makes every `prev_t` known up front; when a label is background (0 -> the step's own arg-max
    is fed back, :205-213) or in eval (greedy, :214-227) a no-grad sequential pass first resolves the fed-back
    labels, then the differentiable pass runs with them as constants (arg-max has no gradient).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.rnn import PackedSequence

from lib import _hip
from lib import rng
from lib.fpn.box_utils import nms_overlaps
from lib.hip_ops import Linear, linear
from lib.pytorch_misc import h2d
from lib.lstm.highway_lstm_cuda.alternating_highway_lstm import block_orthogonal
from lib.word_vectors import obj_edge_vectors


def get_dropout_mask(dropout_probability, shape, device):
    """binary keep mask scaled by 1/(1-p) (reference :13-37)"""
    return rng.keep_mask(shape, 1.0 - dropout_probability, device) / (1.0 - dropout_probability)


def _step_bounds(batch_sizes):
    ends = np.cumsum(batch_sizes)
    return [(int(e - n), int(e), int(n)) for e, n in zip(ends, batch_sizes)]


def _prev_state_rows(batch_sizes):
    """index of the previous timestep's row for every packed row (-1 for the first timestep)"""
    idx, start_prev = [], None
    for s, e, n in _step_bounds(batch_sizes):
        idx.append(np.full(n, -1, np.int64) if start_prev is None else start_prev + np.arange(n))
        start_prev = s
    return np.concatenate(idx)


class _DecoderRecurrenceFn(torch.autograd.Function):
    """h_all [N,H] from pre_i_all [N,6H] (input projection incl. bias and fed-back embedding) over a packed,
    time-major batch.  state weight is nn.Linear layout [5H,H] == the kernel's K-contiguous wh_t."""

    @staticmethod
    def forward(ctx, pre_i_all, w_state, b_state, dropout_mask, batch_sizes):
        N, H = pre_i_all.shape[0], w_state.shape[1]
        pre_i_all = pre_i_all.contiguous()
        w_state = w_state.contiguous()
        ctx.seq = _hip.hwcell_seq_supported(H, int(batch_sizes[0]))
        if ctx.seq:
            # the whole recurrence in one launch (mh_hwcell_seq_fwd); the state buffers carry B zero rows in front
            B = int(batch_sizes[0])
            mask = None if dropout_mask is None else dropout_mask.contiguous()
            h_buf, c_buf, gates_all = _hip.hwcell_seq_fwd(pre_i_all, [int(v) for v in batch_sizes], w_state, b_state,
                                                          mask)
            ctx.batch_sizes = [int(v) for v in batch_sizes]
            ctx.has_mask = dropout_mask is not None
            ctx.save_for_backward(w_state, h_buf, c_buf, gates_all, mask if mask is not None else h_buf[:B])
            return h_buf[B:]
        h_all = pre_i_all.new_empty(N, H)
        c_all = pre_i_all.new_empty(N, H)
        gates_all = pre_i_all.new_empty(N, 6 * H)
        B = int(batch_sizes[0])
        zeros = pre_i_all.new_zeros(B, H)
        h_prev = c_prev = zeros
        for s, e, n in _step_bounds(batch_sizes):
            h, c, g = _hip.hwlstm_cell_fwd(pre_i_all[s:e], h_prev[:n].contiguous(), c_prev[:n].contiguous(), w_state,
                                           b_state, None if dropout_mask is None else dropout_mask[:n].contiguous(),
                                           True)
            h_all[s:e], c_all[s:e], gates_all[s:e] = h, c, g
            h_prev, c_prev = h, c
        ctx.batch_sizes = [int(v) for v in batch_sizes]
        ctx.has_mask = dropout_mask is not None
        ctx.save_for_backward(w_state, h_all, c_all, gates_all, dropout_mask if dropout_mask is not None else zeros)
        return h_all

    @staticmethod
    def backward(ctx, dh_all):
        w_state, h_all, c_all, gates_all, mask = ctx.saved_tensors
        bs = ctx.batch_sizes
        dh_all = dh_all.contiguous()
        w_state_t = w_state.t().contiguous()              # [H,5H]: row k contiguous over the gate columns
        if ctx.seq:
            B = bs[0]
            d_pre = _hip.hwcell_seq_bwd(dh_all, bs, c_all, gates_all, mask if ctx.has_mask else None, w_state_t)
            h_all = h_all[B:]
            return (d_pre,) + _DecoderRecurrenceFn._param_grads(ctx, d_pre, h_all, bs) + (None, None)
        N, H = h_all.shape
        d_pre = dh_all.new_zeros(N, 6 * H)
        steps = _step_bounds(bs)
        dh_rec = dc_rec = None                             # gradients flowing from step t+1 (first n_{t+1} rows)
        for t in range(len(steps) - 1, -1, -1):
            s, e, n = steps[t]
            d_h = dh_all[s:e].clone()
            d_c = dh_all.new_zeros(n, H)
            if dh_rec is not None:
                m = dh_rec.shape[0]
                d_h[:m] += dh_rec
                d_c[:m] = dc_rec
            if t > 0:
                ps = steps[t - 1][0]
                c_prev = c_all[ps:ps + n].contiguous()
            else:
                c_prev = dh_all.new_zeros(n, H)
            dg, dc_in = _hip.hwlstm_cell_bwd(d_h, d_c, c_prev, c_all[s:e].contiguous(), gates_all[s:e].contiguous(),
                                             mask[:n].contiguous() if ctx.has_mask else None)
            d_pre[s:e] = dg
            if t > 0:
                dh_rec = _hip.gemv_rows(dg[:, :5 * H], w_state_t)
                dc_rec = dc_in
        return (d_pre,) + _DecoderRecurrenceFn._param_grads(ctx, d_pre, h_all, bs) + (None, None)

    @staticmethod
    def _param_grads(ctx, d_pre, h_all, bs):
        H = h_all.shape[1]
        gw = gb = None
        if ctx.needs_input_grad[1]:
            prev_rows = _prev_state_rows(bs)
            valid = h2d((prev_rows >= 0), h_all.device)
            h_prev_all = h_all.index_select(0, h2d(np.maximum(prev_rows, 0), h_all.device))
            h_prev_all = h_prev_all * valid[:, None].to(h_all.dtype)
            gw = _hip.gemm(d_pre[:, :5 * H], h_prev_all, True, False)         # [5H, H]
        if ctx.needs_input_grad[2]:
            gb = d_pre[:, :5 * H].sum(0)
        return gw, gb


class DecoderRNN(torch.nn.Module):
    def __init__(self, classes, embed_dim, inputs_dim, hidden_dim, recurrent_dropout_probability=0.2,
                 use_highway=True, use_input_projection_bias=True):
        super(DecoderRNN, self).__init__()
        self.classes = classes
        embed_vecs = obj_edge_vectors(['start'] + self.classes, wv_dim=100)       # [152,100] (reference :56-58)
        self.obj_embed = nn.Embedding(len(self.classes), embed_dim)
        self.obj_embed.weight.data = embed_vecs
        self.hidden_size = hidden_dim
        self.inputs_dim = inputs_dim
        self.nms_thresh = 0.3
        self.recurrent_dropout_probability = recurrent_dropout_probability
        self.use_highway = use_highway
        # reference :68-81: the plain LSTM cell (use_highway=False) has four gate blocks, the highway cell six / five
        ng_in, ng_state = (6, 5) if use_highway else (4, 4)
        self.input_linearity = Linear(self.input_size, ng_in * self.hidden_size, bias=use_input_projection_bias)
        self.state_linearity = Linear(self.hidden_size, ng_state * self.hidden_size, bias=True)
        self.out = Linear(self.hidden_size, len(self.classes))
        self.reset_parameters()

    @property
    def input_size(self):
        return self.inputs_dim + self.obj_embed.weight.size(1)

    def reset_parameters(self):
        block_orthogonal(self.input_linearity.weight.data, [self.hidden_size, self.input_size])
        block_orthogonal(self.state_linearity.weight.data, [self.hidden_size, self.hidden_size])
        self.state_linearity.bias.data.fill_(0.0)
        self.state_linearity.bias.data[self.hidden_size:2 * self.hidden_size].fill_(1.0)

    # ------------------------------------------------------------------------------------------
    _OPEN_GATE = 40.0       # sigmoid(40) == 1.0f

    def _cell_params(self):
        """(w_in [6H,in], b_in [6H] or None, w_state [5H,H], b_state [5H]) as the highway-cell kernels read them.  The plain
        LSTM cell of use_highway=False (reference :96-131 without :122-127) IS the highway cell with its highway gate held
        open: the four gate blocks are padded with a highway-gate block whose pre-activation is the constant 40 (sigmoid = 1.0f
        exactly, so (1 - gate) * projection == 0) and a zero projection block -- same kernels, the reference's parameter shapes
        ([4H,in], [4H,H]) and state-dict keys; gradients reach the four real blocks through the concatenation."""
        w_in, b_in = self.input_linearity.weight, self.input_linearity.bias
        w_state, b_state = self.state_linearity.weight, self.state_linearity.bias
        if self.use_highway:
            return w_in, b_in, w_state, b_state
        H = self.hidden_size
        w_in6 = torch.cat((w_in, w_in.new_zeros(2 * H, w_in.shape[1])), 0)
        b4 = b_in if b_in is not None else w_in.new_zeros(4 * H)
        b_in6 = torch.cat((b4, b4.new_full((H,), self._OPEN_GATE), b4.new_zeros(H)), 0)
        w_state5 = torch.cat((w_state, w_state.new_zeros(H, H)), 0)
        b_state5 = torch.cat((b_state, b_state.new_zeros(H)), 0)
        return w_in6, b_in6, w_state5, b_state5

    def _projections(self, sequence_tensor):
        D = self.inputs_dim
        w_in, b_in, w_state, b_state = self._cell_params()
        enc_proj = linear(sequence_tensor, w_in[:, :D], b_in)          # [N,6H]
        emb_proj = linear(self.obj_embed.weight, w_in[:, D:], None)                         # [152,6H]
        # the cell's state parameters travel with the call (with use_highway=False they are non-leaf concatenations: kept on
        # the module they would hold the step's graph between steps and break copy.deepcopy of the model)
        return enc_proj, emb_proj, (w_state, b_state)

    def _greedy_feedback(self, enc_proj, emb_proj, cell, batch_sizes, labels, dropout_mask):
        """Sequential no-grad pass that resolves which label index is fed back at every row
        (train: the GT label, or the step's non-bg arg-max where the label is 0; eval: the arg-max)."""
        H = self.hidden_size
        if enc_proj.is_cuda and _hip.hwcell_seq_supported(H, int(batch_sizes[0])):
            with torch.no_grad():                     # one persistent launch (mh_decoder_greedy)
                h_all, logits, fed, commits = _hip.decoder_greedy(
                    enc_proj.contiguous(), emb_proj.contiguous(), batch_sizes, cell[0].detach().contiguous(),
                    cell[1].detach(), dropout_mask, self.out.weight.contiguous(), self.out.bias,
                    None if labels is None else labels.contiguous())
            self._greedy_states = (h_all, logits)
            return fed, commits
        self._greedy_states = None
        with torch.no_grad():
            B = int(batch_sizes[0])
            h_prev = c_prev = enc_proj.new_zeros(B, H)
            prev = torch.zeros(B, dtype=torch.long, device=enc_proj.device)      # 'start'
            fed, commits = [], []
            w_state, b_state = cell[0].detach().contiguous(), cell[1].detach()
            for s, e, n in _step_bounds(batch_sizes):
                fed.append(prev[:n])
                pre_i = enc_proj[s:e] + emb_proj.index_select(0, prev[:n])
                h, c, _ = _hip.hwlstm_cell_fwd(pre_i.contiguous(), h_prev[:n].contiguous(), c_prev[:n].contiguous(),
                                               w_state, b_state,
                                               None if dropout_mask is None else dropout_mask[:n].contiguous(), False)
                pred = _hip.gemv_rows(h, self.out.weight, self.out.bias)
                best = pred[:, 1:].max(1)[1] + 1
                if labels is not None:
                    lab = labels[s:e].clone()
                    lab = torch.where(lab == 0, best, lab)
                else:
                    lab = best
                commits.append(lab)
                prev = lab + 1
                h_prev, c_prev = h, c
        return torch.cat(fed, 0), torch.cat(commits, 0)

    def forward(self, inputs, initial_state=None, labels=None, boxes_for_nms=None, labels_have_background=None):
        if not isinstance(inputs, PackedSequence):
            raise ValueError('inputs must be PackedSequence but got %s' % (type(inputs)))
        if initial_state is not None:
            raise NotImplementedError('initial_state is ignored by the reference too')
        sequence_tensor = inputs.data
        batch_sizes = [int(v) for v in inputs.batch_sizes]
        B = batch_sizes[0]
        dropout_mask = None
        if self.recurrent_dropout_probability > 0.0:
            m = get_dropout_mask(self.recurrent_dropout_probability, (B, self.hidden_size), sequence_tensor.device)
            dropout_mask = m if self.training else None            # reference :126-130 applies it in train only
        enc_proj, emb_proj, cell = self._projections(sequence_tensor)

        if self.training:
            if labels is None:
                raise ValueError('training needs labels (teacher forcing)')
            if labels_have_background is None:                      # unknown on the host: ask the device (synchronises)
                labels_have_background = bool((labels == 0).any())
            if labels_have_background:
                fed, commits = self._greedy_feedback(enc_proj.detach(), emb_proj.detach(), cell, batch_sizes, labels,
                                                     dropout_mask)
            else:
                # prev label of row r at step t is the label of the same sequence at step t-1
                prev_rows = h2d(_prev_state_rows(batch_sizes), labels.device)
                fed = torch.where(prev_rows >= 0, labels[prev_rows.clamp(min=0)] + 1, torch.zeros_like(labels))
                commits = labels.clone()
        else:
            for n in batch_sizes:
                assert n == 1, 'eval decodes one image at a time (reference :215)'
            fed, commits = self._greedy_feedback(enc_proj.detach(), emb_proj.detach(), cell, batch_sizes, None, None)
            if getattr(self, '_greedy_states', None) is not None and not torch.is_grad_enabled():
                # the fused launch already produced the states and the class logits of every row: no second pass
                out_dists = self._greedy_states[1]
                self._greedy_states = None
                if boxes_for_nms is not None:
                    commits = self._nms_commitments(out_dists, boxes_for_nms)
                return out_dists, commits

        self._greedy_states = None
        pre_i_all = enc_proj + emb_proj.index_select(0, fed)
        h_all = _DecoderRecurrenceFn.apply(pre_i_all, cell[0], cell[1],
                                           dropout_mask, batch_sizes)
        out_dists = self.out(h_all)

        if boxes_for_nms is not None and not self.training:
            commits = self._nms_commitments(out_dists, boxes_for_nms)
        return out_dists, commits

    def _nms_commitments(self, out_dists, boxes_for_nms):
        """class-wise greedy suppression of the sampled labels in sgdet eval (reference :230-247): one kernel on the
        device (mh_decoder_nms_commit); the host loop below is the reference's and serves CPU tensors (tests)."""
        if out_dists.is_cuda and _hip.decoder_nms_commit_fits(out_dists.size(0), out_dists.size(1)):
            return _hip.decoder_nms_commit(F.softmax(out_dists.detach(), 1).contiguous(),
                                           boxes_for_nms.detach().float().contiguous(), self.nms_thresh)
        is_overlap = nms_overlaps(boxes_for_nms.detach()).view(
            boxes_for_nms.size(0), boxes_for_nms.size(0), boxes_for_nms.size(1)).cpu().numpy() >= self.nms_thresh
        sampled = F.softmax(out_dists.detach(), 1).cpu().numpy().copy()
        sampled[:, 0] = 0
        commits = np.zeros(out_dists.size(0), dtype=np.int64)
        for _ in range(commits.shape[0]):
            box_ind, cls_ind = np.unravel_index(sampled.argmax(), sampled.shape)
            commits[int(box_ind)] = int(cls_ind)
            sampled[is_overlap[box_ind, :, cls_ind], cls_ind] = 0.0
            sampled[box_ind] = -1.0
        return h2d(commits, out_dists.device)
