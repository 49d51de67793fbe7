"""
oracle/lstm.py -- TEST INFRASTRUCTURE ONLY.

CPU fp32 (torch) restatement of the reference's stacked alternating-direction
highway LSTM and of its label decoder.

  highway_lstm_forward / highway_lstm_backward follow
      lib/lstm/highway_lstm_cuda/src/highway_lstm_kernel.cu:377-496 (fwd host loop),
      :108-160 (elementWise_fp), :162-375 (bwd host loop), :46-104 (elementWise_bp)
  AlternatingHighwayLSTM packing / parameter layout follow
      lib/lstm/highway_lstm_cuda/alternating_highway_lstm.py:195-303
  decoder_forward follows lib/lstm/decoder_rnn.py:96-251

The forward is written with differentiable torch ops, so torch.autograd is an
independent check of the hand-written backward restatement (tests/test_oracle_lstm.py).
PINNED (round 3): the reference ships no CPU path for these kernels, so its own highway_lstm_kernel.cu is compiled
for the CPU (oracle/build_ref_cuda.py, oracle/cuda_cpu/: a launch / thread / cuBLAS shim around the unmodified file) and
its forward + backward outputs on seeded inputs are committed as tests/golden/cuda_ref.npz; this restatement equals them
to 2e-6 (tests/test_oracle_ref_cuda.py).  The decoder is pinned against the reference's own Python DecoderRNN
(tests/golden/decoder_*.npz).
"""
import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------
# parameter layout helpers (alternating_highway_lstm.py:213-257)
# ---------------------------------------------------------------------------
def layer_offsets(input_size, hidden_size, num_layers):
    """[(wx_start, wh_start, in_size)] per layer inside the flat weight vector."""
    H = hidden_size
    offs, w = [], 0
    for layer in range(num_layers):
        in_size = input_size if layer == 0 else H
        offs.append((w, w + 6 * H * in_size, in_size))
        w += 6 * H * in_size + 5 * H * H
    return offs, w


def _sigmoid(x):
    # highway_lstm_kernel.cu:32-34   1.f / (1.f + expf(-in))
    return 1.0 / (1.0 + torch.exp(-x))


def _covered_schedule(lengths, T, B, forward_dir):
    """numCovered per timestep, in the order the reference visits the timesteps.
    highway_lstm_kernel.cu:410-424 (fwd) -- lengths sorted descending."""
    sched = []
    if forward_dir:
        n = B
        for t in range(T):
            while lengths[n - 1] <= t:
                n -= 1
            sched.append((t, n))
    else:
        n = 0
        for t in range(T - 1, -1, -1):
            while n < B and lengths[n] > t:
                n += 1
            sched.append((t, n))
    return sched


def highway_lstm_forward(x, lengths, weight, bias, dropout, hidden_size, num_layers, training,
                         return_state=False):
    """
    x        [T,B,in] zero padded, sequences sorted by decreasing length
    lengths  python list / int array [B]
    weight   flat [sum_l 6H*in_l + 5H*H];  bias flat [L*5H]
    dropout  [L,B,H] (already scaled), shared across time
    returns  output [T,B,H] = last layer's states (zeros where t >= length)
    """
    T, B, in0 = x.shape
    H = hidden_size
    offs, _ = layer_offsets(in0, H, num_layers)
    lengths = [int(v) for v in lengths]
    # accumulators are python lists of per-slot tensors so autograd sees no in-place writes
    h_slots = [[x.new_zeros(B, H) for _ in range(T + 1)] for _ in range(num_layers)]
    c_slots = [[x.new_zeros(B, H) for _ in range(T + 1)] for _ in range(num_layers)]
    gates_saved = [[None] * T for _ in range(num_layers)]

    for layer in range(num_layers):
        forward_dir = (layer % 2 == 0)
        wx0, wh0, in_size = offs[layer]
        Wx = weight[wx0:wx0 + 6 * H * in_size].view(in_size, 6 * H)
        Wh = weight[wh0:wh0 + 5 * H * H].view(H, 5 * H)
        b = bias[5 * H * layer:5 * H * (layer + 1)]
        for t, n in _covered_schedule(lengths, T, B, forward_dir):
            prev = t if forward_dir else (t + 2) % (T + 1)
            inp = x[t] if layer == 0 else h_slots[layer - 1][t + 1]
            if n == 0:
                continue
            tmp_i = inp[:n] @ Wx
            tmp_h = h_slots[layer][prev][:n] @ Wh
            g = (tmp_i[:, :5 * H] + tmp_h) + b[None]
            in_gate = _sigmoid(g[:, 0 * H:1 * H])
            forget_gate = _sigmoid(g[:, 1 * H:2 * H])
            act_gate = torch.tanh(g[:, 2 * H:3 * H])
            out_gate = _sigmoid(g[:, 3 * H:4 * H])
            r_gate = _sigmoid(g[:, 4 * H:5 * H])
            lin_gate = tmp_i[:, 5 * H:6 * H]
            c_in = c_slots[layer][prev][:n]
            c_new = (forget_gate * c_in) + (in_gate * act_gate)
            val = out_gate * torch.tanh(c_new)
            # `val * r_gate + (1. - r_gate) * lin_gate` : the literal 1. makes the sum double
            val = ((val * r_gate).double() + (1.0 - r_gate.double()) * lin_gate.double()).to(val.dtype)      # .float() in fp32
            val = val * dropout[layer][:n]
            pad = x.new_zeros(B - n, H)
            h_slots[layer][t + 1] = torch.cat((val, pad), 0) if n < B else val
            c_slots[layer][t + 1] = torch.cat((c_new, pad), 0) if n < B else c_new
            if training:
                gates_saved[layer][t] = (in_gate, forget_gate, act_gate, out_gate, r_gate, lin_gate, n)
    out = torch.stack(h_slots[-1][1:], 0)
    if return_state:
        return out, h_slots, c_slots, gates_saved
    return out


def highway_lstm_backward(grad_out, x, lengths, weight, dropout, hidden_size, num_layers,
                          h_slots, c_slots, gates_saved):
    """Hand restatement of highway_lstm_backward_ongpu; returns (x_grad, weight_grad, bias_grad)."""
    T, B, in0 = x.shape
    H = hidden_size
    offs, wtot = layer_offsets(in0, H, num_layers)
    lengths = [int(v) for v in lengths]
    x_grad = torch.zeros_like(x)
    w_grad = torch.zeros(wtot, dtype=x.dtype)
    b_grad = torch.zeros(5 * H * num_layers, dtype=x.dtype)
    h_grad = torch.zeros(num_layers, T + 1, B, H)
    c_grad = torch.zeros(num_layers, T + 1, B, H)
    h_out_grad = torch.zeros(num_layers, T, B, H)   # gradient handed to the layer below

    for layer in range(num_layers - 1, -1, -1):
        forward_dir = (layer % 2 == 0)
        wx0, wh0, in_size = offs[layer]
        Wx = weight[wx0:wx0 + 6 * H * in_size].view(in_size, 6 * H)
        Wh = weight[wh0:wh0 + 5 * H * H].view(H, 5 * H)
        # the backward pass walks time in the opposite order of the forward pass
        sched = list(reversed(_covered_schedule(lengths, T, B, forward_dir)))
        for t, n in sched:
            if forward_dir:
                prev_grad_index, prev_index = (t + 2) % (T + 1), t
            else:
                prev_grad_index, prev_index = t, (t + 2) % (T + 1)
            if n == 0:
                continue
            g_in = grad_out[t] if layer == num_layers - 1 else h_out_grad[layer, t]
            in_gate, forget_gate, act_gate, out_gate, r_gate, lin_gate, n_saved = gates_saved[layer][t]
            assert n_saved == n
            c_out = c_slots[layer][t + 1][:n]
            c_in = c_slots[layer][prev_index][:n]
            d_h = (g_in[:n] + h_grad[layer, prev_grad_index, :n]) * dropout[layer][:n]
            d_out = d_h * r_gate
            tanh_c = torch.tanh(c_out)
            d_c = d_out * out_gate * (1.0 - tanh_c * tanh_c) + c_grad[layer, prev_grad_index, :n]
            h_prime = out_gate * tanh_c
            d_in = d_c * act_gate * in_gate * (1.0 - in_gate)
            d_forget = d_c * c_in * forget_gate * (1.0 - forget_gate)
            d_act = d_c * in_gate * (1.0 - act_gate * act_gate)
            d_outg = d_out * tanh_c * out_gate * (1.0 - out_gate)
            d_r = d_h * (h_prime - lin_gate) * r_gate * (1.0 - r_gate)
            d_lin = d_h * (1 - r_gate)
            hg = torch.cat((d_in, d_forget, d_act, d_outg, d_r), 1)          # [n,5H]
            ig = torch.cat((hg, d_lin), 1)                                   # [n,6H]
            c_grad[layer, t + 1, :n] = forget_gate * d_c
            inp_grad = ig @ Wx.t()
            if layer == 0:
                x_grad[t, :n] = inp_grad
                inp = x[t]
            else:
                h_out_grad[layer - 1, t, :n] = inp_grad
                inp = h_slots[layer - 1][t + 1]
            h_grad[layer, t + 1, :n] = hg @ Wh.t()
            w_grad[wx0:wx0 + 6 * H * in_size] += (inp[:n].t() @ ig).reshape(-1)
            w_grad[wh0:wh0 + 5 * H * H] += (h_slots[layer][prev_index][:n].t() @ hg).reshape(-1)
            b_grad[5 * H * layer:5 * H * (layer + 1)] += hg.sum(0)
    return x_grad, w_grad, b_grad


def pad_packed(data, batch_sizes):
    """PackedSequence (time-major rows) -> [T,B,D] zero padded + lengths."""
    T, B = len(batch_sizes), int(batch_sizes[0])
    rows = []
    s = 0
    for bs in batch_sizes:
        bs = int(bs)
        chunk = data[s:s + bs]
        if bs < B:
            chunk = torch.cat((chunk, data.new_zeros(B - bs, data.shape[1])), 0)
        rows.append(chunk)
        s += bs
    lengths = [sum(1 for bs in batch_sizes if int(bs) > b) for b in range(B)]
    return torch.stack(rows, 0), lengths


def pack_padded(padded, batch_sizes):
    return torch.cat([padded[t, :int(bs)] for t, bs in enumerate(batch_sizes)], 0)


def alternating_highway_lstm(packed_data, batch_sizes, weight, bias, hidden_size, num_layers,
                             training, dropout_mask=None):
    """AlternatingHighwayLSTM.forward (alternating_highway_lstm.py:259-303) on a packed input.
    dropout_mask [L,B,H] must be supplied in training mode (it is *injected*, not sampled, so
    that oracle and HIP path see the same mask); eval mode uses ones."""
    x, lengths = pad_packed(packed_data, batch_sizes)
    B = x.shape[1]
    if dropout_mask is None:
        dropout_mask = x.new_ones(num_layers, B, hidden_size)
    out = highway_lstm_forward(x, lengths, weight, bias, dropout_mask, hidden_size, num_layers,
                               training)
    return pack_padded(out, batch_sizes)


# ---------------------------------------------------------------------------
# decoder (lib/lstm/decoder_rnn.py)
# ---------------------------------------------------------------------------
def decoder_lstm_equations(p, timestep_input, previous_state, previous_memory, H,
                           dropout_mask, training):
    """decoder_rnn.py:96-131.  p: dict with input_linearity.{weight,bias}, state_linearity.{weight,bias}."""
    pi = F.linear(timestep_input, p['input_linearity.weight'], p['input_linearity.bias'])
    ps = F.linear(previous_state, p['state_linearity.weight'], p['state_linearity.bias'])
    input_gate = torch.sigmoid(pi[:, 0 * H:1 * H] + ps[:, 0 * H:1 * H])
    forget_gate = torch.sigmoid(pi[:, 1 * H:2 * H] + ps[:, 1 * H:2 * H])
    memory_init = torch.tanh(pi[:, 2 * H:3 * H] + ps[:, 2 * H:3 * H])
    output_gate = torch.sigmoid(pi[:, 3 * H:4 * H] + ps[:, 3 * H:4 * H])
    memory = input_gate * memory_init + forget_gate * previous_memory
    timestep_output = output_gate * torch.tanh(memory)
    if p['input_linearity.weight'].shape[0] == 6 * H:      # use_highway (the reference default, :122-127); four blocks: plain LSTM cell
        highway_gate = torch.sigmoid(pi[:, 4 * H:5 * H] + ps[:, 4 * H:5 * H])
        highway_input_projection = pi[:, 5 * H:6 * H]
        timestep_output = highway_gate * timestep_output + (1 - highway_gate) * highway_input_projection
    if dropout_mask is not None and training:
        timestep_output = timestep_output * dropout_mask
    return timestep_output, memory


def nms_overlaps(boxes):
    """lib/fpn/box_utils.py:134-154  boxes [N,nc,4] -> [N,N,nc] IoU per class channel."""
    N, nc = boxes.shape[0], boxes.shape[1]
    max_xy = torch.min(boxes[:, None, :, 2:].expand(N, N, nc, 2), boxes[None, :, :, 2:].expand(N, N, nc, 2))
    min_xy = torch.max(boxes[:, None, :, :2].expand(N, N, nc, 2), boxes[None, :, :, :2].expand(N, N, nc, 2))
    inter = torch.clamp((max_xy - min_xy + 1.0), min=0)
    inters = inter[:, :, :, 0] * inter[:, :, :, 1]
    boxes_flat = boxes.reshape(-1, 4)
    areas_flat = (boxes_flat[:, 2] - boxes_flat[:, 0] + 1.0) * (boxes_flat[:, 3] - boxes_flat[:, 1] + 1.0)
    areas = areas_flat.view(N, nc)
    union = -inters + areas[None] + areas[:, None]
    return inters / union


def decoder_forward(p, sequence_tensor, batch_lengths, H, training, labels=None,
                    boxes_for_nms=None, dropout_mask=None, nms_thresh=0.3):
    """
    DecoderRNN.forward (decoder_rnn.py:133-251).
    p: decoder parameters {obj_embed.weight [152,100], input_linearity.*, state_linearity.*, out.*}
    dropout_mask: [B,H] already scaled (injected), or None.
    returns (out_dists [sum T_b,151], out_commitments [sum T_b] int64)
    """
    batch_size = int(batch_lengths[0])
    previous_memory = sequence_tensor.new_zeros(batch_size, H)
    previous_state = sequence_tensor.new_zeros(batch_size, H)
    emb_w = p['obj_embed.weight']
    previous_embed = emb_w[0, None].expand(batch_size, emb_w.shape[1])
    out_dists, out_commitments = [], []
    end_ind = 0
    for i, l_batch in enumerate(batch_lengths):
        l_batch = int(l_batch)
        start_ind, end_ind = end_ind, end_ind + l_batch
        if previous_memory.shape[0] != l_batch:
            previous_memory = previous_memory[:l_batch]
            previous_state = previous_state[:l_batch]
            previous_embed = previous_embed[:l_batch]
            if dropout_mask is not None:
                dropout_mask = dropout_mask[:l_batch]
        timestep_input = torch.cat((sequence_tensor[start_ind:end_ind], previous_embed), 1)
        previous_state, previous_memory = decoder_lstm_equations(
            p, timestep_input, previous_state, previous_memory, H, dropout_mask, training)
        pred_dist = F.linear(previous_state, p['out.weight'], p['out.bias'])
        out_dists.append(pred_dist)
        if training:
            labels_to_embed = labels[start_ind:end_ind].clone()
            nonzero_pred = pred_dist[:, 1:].max(1)[1] + 1
            is_bg = (labels_to_embed == 0).nonzero()
            if is_bg.numel() > 0:
                labels_to_embed[is_bg.squeeze(1)] = nonzero_pred[is_bg.squeeze(1)]
            out_commitments.append(labels_to_embed)
            previous_embed = emb_w[labels_to_embed + 1]
        else:
            assert l_batch == 1
            out_dist_sample = F.softmax(pred_dist, dim=1)
            best_ind = out_dist_sample[:, 1:].max(1)[1] + 1
            out_commitments.append(best_ind)
            previous_embed = emb_w[best_ind + 1]

    if boxes_for_nms is not None and not training:
        is_overlap = nms_overlaps(boxes_for_nms).view(
            boxes_for_nms.size(0), boxes_for_nms.size(0), boxes_for_nms.size(1)).numpy() >= nms_thresh
        out_dists_sampled = F.softmax(torch.cat(out_dists, 0), 1).detach().numpy().copy()
        out_dists_sampled[:, 0] = 0
        commit = torch.zeros(len(out_commitments), dtype=torch.int64)
        for i in range(commit.size(0)):
            box_ind, cls_ind = np.unravel_index(out_dists_sampled.argmax(), out_dists_sampled.shape)
            commit[int(box_ind)] = int(cls_ind)
            out_dists_sampled[is_overlap[box_ind, :, cls_ind], cls_ind] = 0.0
            out_dists_sampled[box_ind] = -1.0
        out_commitments = commit
    else:
        out_commitments = torch.cat(out_commitments, 0)
    return torch.cat(out_dists, 0), out_commitments
