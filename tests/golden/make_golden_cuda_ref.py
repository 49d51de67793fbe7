#!/usr/bin/env python3
"""Generates tests/golden/cuda_ref.npz: inputs and outputs of the REFERENCE'S OWN CUDA kernel files, compiled for the CPU
from where they lie under /root/reference (oracle/build_ref_cuda.py) -- NMS (nms_kernel.cu), RoIAlign forward / backward
(roi_align_kernel.cu) and the stacked highway LSTM forward / backward (highway_lstm_kernel.cu).  Only the build container
can run this; the vectors travel, the objects need not.   python tests/golden/make_golden_cuda_ref.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref_cuda, ref_cuda as R          # noqa: E402


def rand_boxes(rs, n, hi=560.0):
    xy = rs.uniform(0, hi, (n, 2))
    wh = rs.uniform(4, 300, (n, 2))
    return np.concatenate([xy, np.minimum(xy + wh, 591.0)], 1).astype(np.float32)


def main():
    build_ref_cuda.build()
    out = {}
    rs = np.random.RandomState(2024)
    # ---- NMS: sizes around the 64-box block edge, duplicates, a pile of near-duplicates, the threshold boundary
    cases = []
    for i, (n, thr) in enumerate([(1, 0.5), (2, 0.3), (63, 0.3), (64, 0.7), (65, 0.5), (129, 0.3), (1000, 0.3), (2500, 0.7)]):
        b = rand_boxes(rs, n)
        if n >= 64:
            b[5] = b[3]
            b[40:44] = b[40] + np.float32(0.5)
            b[50:60] = b[50] + rs.uniform(-2, 2, (10, 4)).astype(np.float32)
        out['nms%d_boxes' % i], out['nms%d_thresh' % i] = b, np.float32(thr)
        out['nms%d_keep' % i] = R.nms(b, float(np.float32(thr)))
        cases.append(i)
    b = np.array([[0, 0, 9, 9], [0, 5, 9, 14], [100, 100, 119, 119], [100, 110, 119, 129]], np.float32)   # IoU exactly 1/3
    for j, thr in enumerate((np.float32(1.0) / np.float32(3.0), np.float32(0.3333), np.float32(0.33334))):
        out['nmsb%d_boxes' % j], out['nmsb%d_thresh' % j], out['nmsb%d_keep' % j] = b, thr, R.nms(b, float(thr))
    out['nms_cases'] = np.array(cases)
    # ---- RoIAlign: border spill, border straddle, degenerate box, image index out of range, boxes outside the map
    B, C = 3, 24
    feat = rs.randn(B, C, 37, 37).astype(np.float32)
    n = 48
    rois = np.concatenate([rs.randint(0, B, (n, 1)).astype(np.float32), rand_boxes(rs, n)], 1)
    rois[0, 1:] = [0, 0, 591, 591]
    rois[1, 1:] = [575.5, 10, 576.5, 400]
    rois[2, 1:] = [16, 16, 16, 16]
    rois[3, 0] = 9
    rois[4, 1:] = [-300, -300, -100, -50]
    rois[5:15, 1:] = rois[5, 1:] + rs.uniform(-3, 3, (10, 4)).astype(np.float32)
    rois[5:15, 0] = rois[5, 0]
    g = rs.randn(n, C, 7, 7).astype(np.float32)
    out['roi_feat'], out['roi_rois'], out['roi_grad'] = feat, rois, g
    out['roi_out'] = R.roi_align_fwd(feat, rois)
    out['roi_gfeat'] = R.roi_align_bwd(g, rois, feat.shape)
    # ---- highway LSTM: 3 layers (two directions + a repeat), ragged lengths, recurrent dropout masks
    H, L, insz = 16, 3, 24
    lengths = np.array([6, 5, 5, 2], dtype=np.int32)
    T, Bsz = 6, 4
    x = rs.randn(T, Bsz, insz).astype(np.float32)
    for bi, ln in enumerate(lengths):
        x[ln:, bi] = 0
    wtot = sum(6 * H * (insz if l == 0 else H) + 5 * H * H for l in range(L))
    w = (rs.randn(wtot) * 0.2).astype(np.float32)
    bias = (rs.randn(5 * H * L) * 0.1).astype(np.float32)
    drop = ((rs.rand(L, Bsz, H) > 0.2) / 0.8).astype(np.float32)
    h, c, gates = R.highway_lstm_forward(x, lengths, w, bias, drop, H, L, True)
    go = rs.randn(T, Bsz, H).astype(np.float32)
    for bi, ln in enumerate(lengths):
        go[ln:, bi] = 0
    gx, gw, gb = R.highway_lstm_backward(go, x, lengths, w, drop, H, L, h, c, gates)
    h_eval, _, _ = R.highway_lstm_forward(x, lengths, w, bias, np.ones_like(drop), H, L, False)
    out.update(lstm_x=x, lstm_lengths=lengths, lstm_w=w, lstm_bias=bias, lstm_drop=drop, lstm_h=h, lstm_c=c, lstm_gates=gates,
               lstm_gout=go, lstm_gx=gx, lstm_gw=gw, lstm_gb=gb, lstm_h_eval=h_eval, lstm_dims=np.array([H, L, insz]))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cuda_ref.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
