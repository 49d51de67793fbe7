"""
RPN anchor grid (reference lib/fpn/generate_anchors.py:39-126): A = len(ratios) * len(scales) anchors per cell of the
IM_SCALE/feat_stride grid, un-rounded (the reference dropped the rounding, :110).  Returns float64 [h, w, A, 4].
"""
import numpy as np

from config import IM_SCALE


def _centre(box):
    w, h = box[2] - box[0] + 1, box[3] - box[1] + 1
    return w, h, box[0] + 0.5 * (w - 1), box[1] + 0.5 * (h - 1)


def _boxes_around(ws, hs, cx, cy):
    ws, hs = np.asarray(ws, dtype=np.float64)[:, None], np.asarray(hs, dtype=np.float64)[:, None]
    return np.hstack((cx - 0.5 * (ws - 1), cy - 0.5 * (hs - 1), cx + 0.5 * (ws - 1), cy + 0.5 * (hs - 1)))


def generate_base_anchors(base_size=16, ratios=(0.5, 1, 2), scales=2 ** np.arange(3, 6)):
    ratios, scales = np.asarray(ratios, dtype=np.float64), np.asarray(scales, dtype=np.float64)
    w, h, cx, cy = _centre(np.array([0, 0, base_size - 1, base_size - 1], dtype=np.float64))
    ws = np.sqrt(w * h / ratios)
    per_ratio = _boxes_around(ws, ws * ratios, cx, cy)
    out = []
    for box in per_ratio:
        w, h, cx, cy = _centre(box)
        out.append(_boxes_around(w * scales, h * scales, cx, cy))
    return np.vstack(out)


def generate_anchors(base_size=16, feat_stride=16, anchor_scales=(8, 16, 32), anchor_ratios=(0.5, 1, 2)):
    base = generate_base_anchors(base_size=base_size, ratios=anchor_ratios, scales=anchor_scales)
    shift = np.arange(0, IM_SCALE // feat_stride) * feat_stride
    sx, sy = np.meshgrid(shift, shift)
    shifts = np.stack([sx, sy, sx, sy], -1)
    return shifts[:, :, None] + base[None, None]
