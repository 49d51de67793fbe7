"""Oracle LSTM: hand-restated backward vs autograd of the restated forward; decoder vs the
reference's own DecoderRNN outputs (tests/golden/decoder.npz)."""
import numpy as np
import pytest
import torch

from oracle import lstm as L


def _make(T_lengths, in_size, H, nl, seed=0, dropout_p=0.0):
    g = torch.Generator().manual_seed(seed)
    lengths = sorted(T_lengths, reverse=True)
    T, B = lengths[0], len(lengths)
    x = torch.randn(T, B, in_size, generator=g)
    for b, l in enumerate(lengths):
        x[l:, b] = 0
    _, wtot = L.layer_offsets(in_size, H, nl)
    weight = torch.randn(wtot, generator=g) * 0.2
    bias = torch.randn(5 * H * nl, generator=g) * 0.1
    drop = torch.ones(nl, B, H)
    if dropout_p > 0:
        drop = (torch.rand(nl, B, H, generator=g) > dropout_p).float() / (1 - dropout_p)
    return x, lengths, weight, bias, drop


@pytest.mark.parametrize('lengths,nl,p', [([5, 3, 3, 1], 1, 0.0), ([6, 6, 2], 2, 0.3), ([4, 2, 1], 3, 0.2),
                                          ([7], 4, 0.0)])
def test_manual_backward_matches_autograd(lengths, nl, p):
    in_size, H = 10, 6
    x, lengths, weight, bias, drop = _make(lengths, in_size, H, nl, seed=nl, dropout_p=p)
    x64, w64, b64 = x.double().requires_grad_(), weight.double().requires_grad_(), bias.double().requires_grad_()
    out64 = L.highway_lstm_forward(x64, lengths, w64, b64, drop.double(), H, nl, True)
    gout = torch.randn(out64.shape, generator=torch.Generator().manual_seed(5)).double()
    for b, l in enumerate(lengths):
        gout[l:, b] = 0
    out64.backward(gout)
    out, h_slots, c_slots, gates = L.highway_lstm_forward(x, lengths, weight, bias, drop, H, nl, True,
                                                         return_state=True)
    np.testing.assert_allclose(out.numpy(), out64.detach().numpy(), atol=2e-5)
    xg, wg, bg = L.highway_lstm_backward(gout.float(), x, lengths, weight, drop, H, nl, h_slots, c_slots, gates)
    np.testing.assert_allclose(xg.numpy(), x64.grad.numpy(), atol=5e-5)
    np.testing.assert_allclose(wg.numpy(), w64.grad.numpy(), atol=5e-5)
    np.testing.assert_allclose(bg.numpy(), b64.grad.numpy(), atol=5e-5)


def test_padding_rows_are_zero_and_directions_alternate():
    x, lengths, weight, bias, drop = _make([5, 2], 4, 3, 2, seed=9)
    out = L.highway_lstm_forward(x, lengths, weight, bias, drop, 3, 2, False)
    assert torch.all(out[2:, 1] == 0)
    # layer 1 runs right-to-left: changing the LAST input changes the FIRST output of a 2-layer stack
    x2 = x.clone()
    x2[4, 0] += 1.0
    out2 = L.highway_lstm_forward(x2, lengths, weight, bias, drop, 3, 2, False)
    assert not torch.allclose(out[0, 0], out2[0, 0])
    # ... while a 1-layer (left-to-right only) stack is causal
    _, w1tot = L.layer_offsets(4, 3, 1)
    o1 = L.highway_lstm_forward(x, lengths, weight[:w1tot], bias[:15], drop[:1], 3, 1, False)
    o2 = L.highway_lstm_forward(x2, lengths, weight[:w1tot], bias[:15], drop[:1], 3, 1, False)
    assert torch.allclose(o1[:4, 0], o2[:4, 0]) and not torch.allclose(o1[4, 0], o2[4, 0])


def test_pack_roundtrip():
    bs = [3, 3, 2, 1]
    data = torch.arange(9 * 2, dtype=torch.float32).view(9, 2)
    padded, lengths = L.pad_packed(data, bs)
    assert lengths == [4, 3, 2]
    assert torch.equal(L.pack_padded(padded, bs), data)


def _dec_params(g, tag):
    pre = tag + '_param_'
    return {k[len(pre):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(pre)}


@pytest.mark.parametrize('tag', ['p0', 'p2'])
def test_decoder_matches_reference_python(golden, tag):
    g = golden('decoder')
    p = _dec_params(g, tag)
    H = p['state_linearity.weight'].shape[1]
    mask = torch.from_numpy(g[tag + '_train_mask']) if (tag + '_train_mask') in g else None
    dists, commits = L.decoder_forward(p, torch.from_numpy(g[tag + '_train_seq']), g[tag + '_train_lengths'].tolist(),
                                       H, True, labels=torch.from_numpy(g[tag + '_train_labels']), dropout_mask=mask)
    np.testing.assert_allclose(dists.numpy(), g[tag + '_train_dists'], atol=1e-5)
    np.testing.assert_array_equal(commits.numpy(), g[tag + '_train_commits'])
    seq1 = torch.from_numpy(g[tag + '_eval_seq'])
    dists, commits = L.decoder_forward(p, seq1, [1] * seq1.shape[0], H, False)
    np.testing.assert_allclose(dists.numpy(), g[tag + '_eval_dists'], atol=1e-5)
    np.testing.assert_array_equal(commits.numpy(), g[tag + '_eval_commits'])
    dists, commits = L.decoder_forward(p, seq1, [1] * seq1.shape[0], H, False,
                                       boxes_for_nms=torch.from_numpy(g[tag + '_evalnms_boxes']))
    np.testing.assert_allclose(dists.numpy(), g[tag + '_evalnms_dists'], atol=1e-5)
    np.testing.assert_array_equal(commits.numpy(), g[tag + '_evalnms_commits'])


def test_reference_lstm_parameter_layout(golden):
    """alternating_highway_lstm.py:233-257: per layer Wx[in,6H] then Wh[H,5H]; forget bias = 1;
    each [in,H] / [H,H] block is (semi-)orthogonal."""
    g = golden('ahlstm_layout')
    in_size, H, nl = g['dims'].tolist()
    offs, wtot = L.layer_offsets(in_size, H, nl)
    assert wtot == g['weight'].shape[0]
    bias = g['bias'].reshape(nl, 5, H)
    assert np.all(bias[:, 1] == 1) and np.all(bias[:, [0, 2, 3, 4]] == 0)
    for (wx0, wh0, ins) in offs:
        Wx = g['weight'][wx0:wx0 + 6 * H * ins].reshape(ins, 6 * H)
        Wh = g['weight'][wh0:wh0 + 5 * H * H].reshape(H, 5 * H)
        for k in range(6):
            blk = Wx[:, k * H:(k + 1) * H]
            np.testing.assert_allclose(blk.T @ blk, np.eye(H), atol=1e-5)
        for k in range(5):
            blk = Wh[:, k * H:(k + 1) * H]
            np.testing.assert_allclose(blk.T @ blk, np.eye(H), atol=1e-5)
