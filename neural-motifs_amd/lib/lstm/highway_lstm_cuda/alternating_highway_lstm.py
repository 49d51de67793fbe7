"""
Stacked alternating-direction highway LSTM -- the reference's import path
(lib/lstm/highway_lstm_cuda/alternating_highway_lstm.py) served by the gfx950 implementation.

API parity: `AlternatingHighwayLSTM(input_size, hidden_size, num_layers, recurrent_dropout_probability)` with the
flat 1-D `weight` / `bias` parameters (per layer Wx[in_l,6H] row-major, then Wh[H,5H]; bias [5H] per layer, forget
gate bias 1) and `forward(PackedSequence) -> (PackedSequence, None)` (reference :165-303).
The compute is mh_hwlstm_fwd / mh_hwlstm_bwd (csrc/lstm.hip) behind one autograd Function.
"""
import itertools

import numpy as np
import torch
from torch.nn import Parameter
from torch.nn.utils.rnn import PackedSequence

from lib import _hip
from lib import rng
from lib.pytorch_misc import h2d


def block_orthogonal(tensor, split_sizes, gain=1.0):
    """Initialise `tensor` block-wise with (semi-)orthogonal blocks of size `split_sizes`
    (reference :12-59; used for the per-gate blocks of the fused projection matrices)."""
    sizes = list(tensor.size())
    if any(a % b != 0 for a, b in zip(sizes, split_sizes)):
        raise ValueError("tensor dimensions must be divisible by their respective split_sizes. "
                         "Found size: {} and split_sizes: {}".format(sizes, split_sizes))
    assert len(sizes) == 2
    starts = [range(0, n, s) for n, s in zip(sizes, split_sizes)]
    with torch.no_grad():
        for r0, c0 in itertools.product(*starts):
            rows, cols = split_sizes
            side = max(rows, cols)
            blk = torch.empty(side, side, dtype=tensor.dtype)
            torch.nn.init.orthogonal_(blk, gain=gain)
            tensor[r0:r0 + rows, c0:c0 + cols] = blk[:rows, :cols].to(tensor.device)
    return tensor


def packed_layout(batch_sizes):
    """For a time-major packed sequence: (T, B, lengths per sequence, flat indices t*B+b of every packed row)."""
    bs = [int(v) for v in batch_sizes]
    T, B = len(bs), bs[0]
    lengths = [sum(1 for v in bs if v > b) for b in range(B)]
    idx = np.concatenate([t * B + np.arange(n) for t, n in enumerate(bs)]).astype(np.int64)
    return T, B, lengths, idx


class _HighwayLSTMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_padded, weight, bias, dropout, lengths, hidden_size, num_layers, training):
        x_padded = x_padded.contiguous()
        keep_gates = bool(training) or any(ctx.needs_input_grad[:3])      # the backward needs the gate activations
        h_data, c_data, gates = _hip.hwlstm_fwd(x_padded, lengths, weight.contiguous(), bias.contiguous(),
                                                dropout, hidden_size, num_layers, keep_gates)
        ctx.lengths, ctx.H, ctx.L = list(lengths), hidden_size, num_layers
        if keep_gates:
            ctx.save_for_backward(x_padded, weight, dropout, h_data, c_data, gates)
        return h_data[-1, 1:]

    @staticmethod
    def backward(ctx, grad_out):
        x_padded, weight, dropout, h_data, c_data, gates = ctx.saved_tensors
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        xg, wg, bg = _hip.hwlstm_bwd(grad_out.contiguous(), x_padded, ctx.lengths, weight.contiguous(), dropout,
                                     ctx.H, ctx.L, h_data, c_data, gates, need_weight_grad=need_w)
        return xg, wg, bg, None, None, None, None, None


class AlternatingHighwayLSTM(torch.nn.Module):
    def __init__(self, input_size, hidden_size, num_layers=1, recurrent_dropout_probability=0):
        super(AlternatingHighwayLSTM, self).__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.recurrent_dropout_probability = recurrent_dropout_probability
        total_w = 0
        for layer in range(num_layers):
            in_l = input_size if layer == 0 else hidden_size
            total_w += 6 * hidden_size * in_l + 5 * hidden_size * hidden_size
        self.weight = Parameter(torch.zeros(total_w))
        self.bias = Parameter(torch.zeros(5 * hidden_size * num_layers))
        self.reset_parameters()

    def reset_parameters(self):
        H = self.hidden_size
        with torch.no_grad():
            self.bias.zero_()
            w = 0
            for layer in range(self.num_layers):
                in_l = self.input_size if layer == 0 else H
                wx = block_orthogonal(torch.zeros(in_l, 6 * H), [in_l, H])
                self.weight[w:w + wx.numel()] = wx.reshape(-1)
                w += wx.numel()
                wh = block_orthogonal(torch.zeros(H, 5 * H), [H, H])
                self.weight[w:w + wh.numel()] = wh.reshape(-1)
                w += wh.numel()
                self.bias[5 * H * layer + H:5 * H * layer + 2 * H] = 1.0     # forget-gate bias

    def dropout_mask(self, batch_size, device):
        """[L,B,H] variational mask shared over time (reference :279-287)."""
        p = self.recurrent_dropout_probability
        shape = (self.num_layers, batch_size, self.hidden_size)
        if not self.training or p == 0:
            return torch.ones(shape, device=device) if not self.training else \
                rng.keep_mask(shape, 1.0, device)         # p == 0 still consumes a draw, like bernoulli_(1)
        return rng.keep_mask(shape, 1.0 - p, device) / (1.0 - p)

    def forward(self, inputs, initial_state=None):
        if not isinstance(inputs, PackedSequence):
            raise ValueError('inputs must be PackedSequence but got %s' % type(inputs))
        data, batch_sizes = inputs.data, inputs.batch_sizes
        T, B, lengths, idx = packed_layout(batch_sizes)
        idx_dev = h2d(idx, data.device)
        padded = data.new_zeros(T * B, data.shape[1]).index_copy(0, idx_dev, data).view(T, B, -1)
        mask = self.dropout_mask(B, data.device)
        out = _HighwayLSTMFn.apply(padded, self.weight, self.bias, mask, lengths, self.hidden_size,
                                   self.num_layers, self.training)
        out_packed = out.reshape(T * B, self.hidden_size).index_select(0, idx_dev)
        return PackedSequence(out_packed, batch_sizes), None
