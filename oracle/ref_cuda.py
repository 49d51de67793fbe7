"""TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/_ref/ref_*.so = the reference's own CUDA kernel files compiled for
the CPU (oracle/build_ref_cuda.py + oracle/cuda_cpu/cuda_on_cpu.h), together with restatements of the few lines of Python /
C glue that sit between the reference's autograd Functions and those kernels (cited per function).  Only the build
container can build the objects; tests compare against goldens generated from them (tests/golden/make_golden_cuda_ref.py
-> tests/golden/cuda_ref.npz) and, when the objects are present, against the objects directly."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)
_libs = {}


def available():
    return all(os.path.exists(os.path.join(_HERE, '_ref', 'ref_%s.so' % n))
               for n in ('nms_kernel', 'roi_align_kernel', 'highway_lstm_kernel'))


def _lib(name):
    if name not in _libs:
        _libs[name] = ctypes.CDLL(os.path.join(_HERE, '_ref', 'ref_%s.so' % name))
    return _libs[name]


def _p(a):
    return a.ctypes.data_as(_f32p)


def nms(boxes_sorted, thresh):
    """ApplyNMSGPU (nms_kernel.cu:96-131: nms_kernel mask + host sweep) on score-sorted boxes [N,4]; kept positions"""
    b = np.ascontiguousarray(boxes_sorted, dtype=np.float32).reshape(-1, 4)
    n = b.shape[0]
    keep = np.zeros(max(n, 1), dtype=np.int32)
    L = _lib('nms_kernel')
    L.ApplyNMSGPU.restype = ctypes.c_int
    k = L.ApplyNMSGPU(keep.ctypes.data_as(_i32p), _p(b), ctypes.c_int(n), ctypes.c_float(thresh), ctypes.c_int(0)) if n else 0
    return keep[:k].copy()


def _normalise_rois(rois, H, W, spatial_scale):
    """RoIAlignFunction.forward (functions/roi_align.py:20-32): height = (H - 1) / spatial_scale in Python double, the fp32
    tensor divided in place by that scalar (THC converts the scalar to fp32 first)"""
    r = np.array(rois, dtype=np.float32).reshape(-1, 5).copy()
    height, width = np.float32((H - 1) / float(spatial_scale)), np.float32((W - 1) / float(spatial_scale))
    r[:, 1] /= width
    r[:, 2] /= height
    r[:, 3] /= width
    r[:, 4] /= height
    return r


def roi_align_fwd(feat, rois, ph=7, pw=7, spatial_scale=1.0 / 16):
    """roi_align_forward_cuda (roi_align_cuda.c:7-41) -> ROIAlignForwardLaucher: zero-filled output, extrapolation 0"""
    feat = np.ascontiguousarray(feat, dtype=np.float32)
    B, C, H, W = feat.shape
    r = _normalise_rois(rois, H, W, spatial_scale)
    out = np.zeros((r.shape[0], C, ph, pw), dtype=np.float32)
    if r.shape[0]:
        _lib('roi_align_kernel').ROIAlignForwardLaucher(_p(feat), _p(r), r.shape[0], B, H, W, ph, pw, C, ctypes.c_float(0.0),
                                                        _p(out), None)
    return out


def roi_align_bwd(grad_out, rois, feat_shape, spatial_scale=1.0 / 16):
    """roi_align_backward_cuda (roi_align_cuda.c:43-76) -> ROIAlignBackwardLaucher into a zero-filled gradient"""
    g = np.ascontiguousarray(grad_out, dtype=np.float32)
    B, C, H, W = feat_shape
    r = _normalise_rois(rois, H, W, spatial_scale)
    gf = np.zeros((B, C, H, W), dtype=np.float32)
    if r.shape[0]:
        _lib('roi_align_kernel').ROIAlignBackwardLaucher(_p(g), _p(r), r.shape[0], B, H, W, g.shape[2], g.shape[3], C, _p(gf), None)
    return gf


def highway_lstm_forward(x, lengths, weight, bias, dropout, H, L, training):
    """_AlternatingHighwayLSTMFunction.forward (alternating_highway_lstm.py:71-104) + AlternatingHighwayLSTM.forward's
    buffer set-up (:279-292): zero accumulators [L, T+1, B, H], gates [L, T, B, 6H]; x padded [T, B, in]"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    T, B, insz = x.shape
    lengths = np.ascontiguousarray(lengths, dtype=np.int32)
    h = np.zeros((L, T + 1, B, H), dtype=np.float32)
    c = np.zeros((L, T + 1, B, H), dtype=np.float32)
    gates = np.zeros((L, T, B, 6 * H), dtype=np.float32)
    tmp_i = np.zeros((B, 6 * H), dtype=np.float32)
    tmp_h = np.zeros((B, 5 * H), dtype=np.float32)
    w = np.ascontiguousarray(weight, dtype=np.float32)
    b = np.ascontiguousarray(bias, dtype=np.float32)
    d = np.ascontiguousarray(dropout, dtype=np.float32)
    _lib('highway_lstm_kernel').highway_lstm_forward_ongpu(
        insz, H, B, L, T, _p(x), lengths.ctypes.data_as(_i32p), _p(h), _p(c), _p(tmp_i), _p(tmp_h), _p(w), _p(b), _p(d),
        _p(gates) if training else None, 1 if training else 0, None, None)
    return h, c, gates


def highway_lstm_backward(grad_out, x, lengths, weight, dropout, H, L, h, c, gates):
    """_AlternatingHighwayLSTMFunction.backward (alternating_highway_lstm.py:106-160): zeroed gradient buffers, grad_hy = 0"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    T, B, insz = x.shape
    lengths = np.ascontiguousarray(lengths, dtype=np.int32)
    g = np.ascontiguousarray(grad_out, dtype=np.float32)
    w = np.ascontiguousarray(weight, dtype=np.float32)
    d = np.ascontiguousarray(dropout, dtype=np.float32)
    gx = np.zeros_like(x)
    gh = np.zeros_like(h)
    gc = np.zeros_like(c)
    gw = np.zeros_like(w)
    gb = np.zeros(5 * H * L, dtype=np.float32)
    ti = np.zeros((B, 6 * H), dtype=np.float32)
    th = np.zeros((B, 5 * H), dtype=np.float32)
    ghy = np.zeros((L, T, B, H), dtype=np.float32)
    _lib('highway_lstm_kernel').highway_lstm_backward_ongpu(
        insz, H, B, L, T, _p(g), lengths.ctypes.data_as(_i32p), _p(gh), _p(gc), _p(x), _p(h), _p(c), _p(w), _p(gates), _p(d),
        _p(th), _p(ti), _p(ghy), _p(gx), _p(gw), _p(gb), 1, 1, None, None)
    return gx, gw, gb
