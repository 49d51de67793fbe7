"""
float64 pairwise IoU / intersection ratio on the host (reference: Cython lib/fpn/box_intersections_cpu/bbox.pyx,
used by the evaluator and the host samplers).  Vectorised numpy with the same arithmetic per element:
inclusive pixel boxes (+1), zero where the boxes do not overlap.
"""
import numpy as np


def _iw_ih(boxes, query_boxes):
    b = np.ascontiguousarray(boxes, dtype=np.float64)[:, None, :]
    q = np.ascontiguousarray(query_boxes, dtype=np.float64)[None, :, :]
    iw = np.minimum(b[..., 2], q[..., 2]) - np.maximum(b[..., 0], q[..., 0]) + 1
    ih = np.minimum(b[..., 3], q[..., 3]) - np.maximum(b[..., 1], q[..., 1]) + 1
    return b, q, iw, ih


def bbox_overlaps(boxes, query_boxes):
    """(N,4) x (K,4) -> (N,K) IoU"""
    b, q, iw, ih = _iw_ih(boxes, query_boxes)
    box_area = (q[..., 2] - q[..., 0] + 1) * (q[..., 3] - q[..., 1] + 1)
    ua = (b[..., 2] - b[..., 0] + 1) * (b[..., 3] - b[..., 1] + 1) + box_area - iw * ih
    ok = (iw > 0) & (ih > 0)
    out = np.zeros(ok.shape, dtype=np.float64)
    np.divide(iw * ih, ua, out=out, where=ok)
    return out


def bbox_intersections(boxes, query_boxes):
    """(N,4) x (K,4) -> (N,K) fraction of each query box covered"""
    b, q, iw, ih = _iw_ih(boxes, query_boxes)
    box_area = (q[..., 2] - q[..., 0] + 1) * (q[..., 3] - q[..., 1] + 1)
    ok = (iw > 0) & (ih > 0)
    out = np.zeros(ok.shape, dtype=np.float64)
    np.divide(iw * ih, np.broadcast_to(box_area, ok.shape), out=out, where=ok)
    return out
