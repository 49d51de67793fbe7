"""
ctypes front-end for oracle/native_ops.c (TEST INFRASTRUCTURE ONLY).

numpy in, numpy out.  Builds the shared object on first use if it is missing
(gcc only; no GPU needed).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'liboracle_native.so')
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    src = os.path.join(_HERE, 'native_ops.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-ffp-contract=off',
                               '-fno-fast-math', '-std=c11', '-o', _SO, src, '-lm'])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_nms.restype = ctypes.c_int
        _lib.orc_nms_bitmask.restype = ctypes.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def nms(boxes_sorted, thresh, bitmask=False):
    """Greedy NMS over score-sorted boxes [N,4]; returns kept positions (int32, ascending)."""
    b = _f32(boxes_sorted).reshape(-1, 4)
    n = b.shape[0]
    keep = np.zeros(max(n, 1), dtype=np.int32)
    fn = lib().orc_nms_bitmask if bitmask else lib().orc_nms
    k = fn(b.ctypes.data_as(_f32p), ctypes.c_int(n), ctypes.c_float(thresh),
           keep.ctypes.data_as(_i32p))
    return keep[:k].copy()


def roi_align_fwd(feat, rois, ph=7, pw=7, spatial_scale=1.0 / 16):
    feat = _f32(feat)
    rois = _f32(rois).reshape(-1, 5)
    B, C, H, W = feat.shape
    N = rois.shape[0]
    out = np.zeros((N, C, ph, pw), dtype=np.float32)
    lib().orc_roi_align_fwd(feat.ctypes.data_as(_f32p), B, C, H, W, rois.ctypes.data_as(_f32p),
                            N, ph, pw, ctypes.c_float(spatial_scale), out.ctypes.data_as(_f32p))
    return out


def roi_align_bwd(grad_out, rois, feat_shape, spatial_scale=1.0 / 16):
    g = _f32(grad_out)
    rois = _f32(rois).reshape(-1, 5)
    B, C, H, W = feat_shape
    N, _, ph, pw = g.shape
    gf = np.zeros((B, C, H, W), dtype=np.float32)
    lib().orc_roi_align_bwd(g.ctypes.data_as(_f32p), B, C, H, W, rois.ctypes.data_as(_f32p),
                            N, ph, pw, ctypes.c_float(spatial_scale), gf.ctypes.data_as(_f32p))
    return gf


def draw_union_boxes(box_pairs, pooling_size):
    bp = _f32(box_pairs).reshape(-1, 8)
    N = bp.shape[0]
    P = int(pooling_size)
    out = np.zeros((N, 2, P, P), dtype=np.float32)
    lib().orc_draw_union_boxes(bp.ctypes.data_as(_f32p), N, ctypes.c_uint(P),
                               out.ctypes.data_as(_f32p))
    return out


def bbox_overlaps(boxes, query):
    a, q = _f64(boxes).reshape(-1, 4), _f64(query).reshape(-1, 4)
    out = np.zeros((a.shape[0], q.shape[0]), dtype=np.float64)
    lib().orc_bbox_overlaps(a.ctypes.data_as(_f64p), a.shape[0], q.ctypes.data_as(_f64p),
                            q.shape[0], out.ctypes.data_as(_f64p))
    return out


def bbox_intersections(boxes, query):
    a, q = _f64(boxes).reshape(-1, 4), _f64(query).reshape(-1, 4)
    out = np.zeros((a.shape[0], q.shape[0]), dtype=np.float64)
    lib().orc_bbox_intersections(a.ctypes.data_as(_f64p), a.shape[0], q.ctypes.data_as(_f64p),
                                 q.shape[0], out.ctypes.data_as(_f64p))
    return out
