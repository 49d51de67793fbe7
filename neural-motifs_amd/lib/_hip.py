"""
ctypes binding of libmotifs_hip.so (the C ABI declared in include/motifs_hip.h).

This is the ONLY gateway from Python to the hot-path kernels.  There is no CPU or eager-PyTorch
fallback: if the shared object is missing, or a tensor is not a contiguous fp32/int32 CUDA(HIP)
tensor, the call raises.  Build the library with ``python neural-motifs_amd/csrc/build.py``
(or ``__graft_entry__.build()``).
"""
import ctypes
import threading
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MOTIFS_HIP_LIB selects another BUILD of the same library (csrc/build.py with MH_OUT=...: e.g. the MH_SPLIT_RN=1 or
# MH_MFMA_SPLIT=0 variants); it is never a fallback -- a missing file raises like the default path does
SO_PATH = os.environ.get('MOTIFS_HIP_LIB') or os.path.join(os.path.dirname(_HERE), 'csrc', 'libmotifs_hip.so')

# every symbol include/motifs_hip.h declares (checked by tests/test_cabi.py against the header)
SYMBOLS = (
    'mh_version', 'mh_mfma_split', 'mh_split_rne', 'mh_split_f16', 'mh_last_error',
    'mh_nms_ws_bytes', 'mh_nms', 'mh_nms_batched_ws_bytes', 'mh_nms_batched',
    'mh_roi_align_fwd', 'mh_roi_align_bwd', 'mh_roi_align_bwd_det', 'mh_draw_union_boxes', 'mh_bbox_overlaps', 'mh_triplet_match',
    'mh_pair_product_fwd', 'mh_pair_product_bwd', 'mh_freq_bias_add', 'mh_freq_bias_bwd', 'mh_ce_pair_fwd', 'mh_ce_pair_bwd',
    'mh_gemm_ws_bytes', 'mh_gemm_auto_splitk', 'mh_gemm_f32',
    'mh_planes_bytes', 'mh_make_planes', 'mh_make_planes_both', 'mh_gemm_planes_ws_bytes', 'mh_gemm_planes_auto_splitk', 'mh_gemm_planes',
    'mh_act_planes_bytes', 'mh_act_planes', 'mh_image_maxbits', 'mh_plconv_packed_bytes', 'mh_plconv_pack_weight', 'mh_plconv3x3_ws_bytes',
    'mh_plconv3x3', 'mh_plconv3x3_to_image', 'mh_plconv3x3_pool_to_image', 'mh_stem_to_image', 'mh_conv_first_nchw_max', 'mh_debug_plconv_shape', 'mh_debug_plconv_splitk', 'mh_debug_plconv_flags', 'mh_decoder_nms_commit_max_bytes',
    'mh_debug_pl_shape', 'mh_debug_pl_order', 'mh_debug_pl_item', 'mh_gemm_ws_bytes_v2', 'mh_gemm_auto_splitk_v2', 'mh_gemm_f32_v2',
    'mh_gemm_small_max_counters', 'mh_gemm_small_f32', 'mh_debug_small_plan',
    'mh_conv3x3_packed_floats', 'mh_conv3x3_pack_weight', 'mh_conv3x3_ws_bytes', 'mh_conv3x3_schedule', 'mh_conv3x3_nhwc',
    'mh_conv3x3_wgrad_ws_bytes', 'mh_conv3x3_wgrad', 'mh_conv_first_nchw', 'mh_maxpool2x2_nhwc',
    'mh_maxpool2x2_bwd_nhwc', 'mh_act_bwd',
    'mh_im2col_nhwc', 'mh_nchw_to_nhwc', 'mh_nhwc_to_nchw',
    'mh_hwlstm_fwd_ws_bytes', 'mh_hwlstm_fwd', 'mh_hwlstm_bwd_ws_bytes', 'mh_hwlstm_bwd',
    'mh_hwlstm_cell_fwd', 'mh_hwlstm_cell_bwd', 'mh_gemv_rows',
    'mh_hwcell_seq_ws_bytes', 'mh_hwcell_seq_fwd', 'mh_hwcell_seq_bwd',
    'mh_decoder_greedy_ws_bytes', 'mh_decoder_greedy', 'mh_decoder_nms_commit',
    'mh_fault_pending', 'mh_fault_clear', 'mh_debug_lstm_barrier_fault',
    'mh_opt_chunk_elems', 'mh_opt_skipped_steps', 'mh_opt_skipped_clear', 'mh_opt_build_chunks', 'mh_multi_sumsq', 'mh_multi_sgd_step',
    'mh_bn_ws_bytes', 'mh_bn_stats', 'mh_bn_pool_fwd', 'mh_bn_residual_nchw', 'mh_bn_apply_nhwc', 'mh_nchw_to_nhwc_small', 'mh_bn_bwd',
    'mh_tower_conv1_out_size', 'mh_tower_conv1_padded_bytes', 'mh_tower_conv1_wgrad_ws_bytes', 'mh_tower_conv1_pad', 'mh_tower_conv1_fwd',
    'mh_tower_conv1_wgrad',
)

_lib = None


class HipKernelError(RuntimeError):
    pass


def lib():
    """Load the shared object (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise HipKernelError(
                'libmotifs_hip.so not found at %s -- build it with '
                '`python neural-motifs_amd/csrc/build.py`; there is no fallback path' % SO_PATH)
        L = ctypes.CDLL(SO_PATH)
        for name in SYMBOLS:
            getattr(L, name)          # AttributeError if the library is stale
        L.mh_last_error.restype = ctypes.c_char_p
        for name in ('mh_nms_ws_bytes', 'mh_nms_batched_ws_bytes', 'mh_gemm_ws_bytes', 'mh_planes_bytes',
                     'mh_gemm_planes_ws_bytes', 'mh_gemm_ws_bytes_v2', 'mh_act_planes_bytes', 'mh_plconv_packed_bytes',
                     'mh_plconv3x3_ws_bytes', 'mh_conv3x3_ws_bytes', 'mh_bn_ws_bytes',
                     'mh_conv3x3_packed_floats', 'mh_hwcell_seq_ws_bytes', 'mh_conv3x3_wgrad_ws_bytes',
                     'mh_hwlstm_fwd_ws_bytes', 'mh_hwlstm_bwd_ws_bytes',
                     'mh_decoder_greedy_ws_bytes', 'mh_decoder_nms_commit_max_bytes', 'mh_tower_conv1_padded_bytes',
                     'mh_tower_conv1_wgrad_ws_bytes'):
            getattr(L, name).restype = ctypes.c_size_t
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        msg = lib().mh_last_error()
        drop_zeroed_workspaces()
        raise HipKernelError('%s failed with status %d: %s' % (what, rc, msg.decode() if msg else ''))


def check_faults():
    """Raise if a persistent kernel (grid-barrier LSTM layers / decoder recurrence) reported a device-side fault since
    the last mh_fault_clear().  A host read of a pinned word: no synchronisation, so it sees faults of launches that
    have COMPLETED -- call it after a natural sync point (loss.item(), the eval tuple's D2H copy, optimizer step)."""
    n = lib().mh_fault_pending()
    if n:
        drop_zeroed_workspaces()
        raise HipKernelError('a persistent LSTM launch timed out in its grid barrier on %d device(s): the results of '
                             'this step are invalid (outputs were NaN-poisoned); call lib().mh_fault_clear() to re-arm' % n)


def check_skipped_steps():
    """Raise if the fused optimizer skipped a step on the device because the gradient norm was not finite (the kernel
    leaves weights and momentum untouched in that case: csrc/optim.hip).  Host read of a pinned counter, no sync."""
    n = lib().mh_opt_skipped_steps()
    if n:
        raise HipKernelError('%d optimizer step(s) were skipped on the device: the global gradient norm was NaN / inf '
                             '(weights and momentum were left untouched); lib().mh_opt_skipped_clear() re-arms' % n)


# Parameters updated through raw pointers (mh_multi_sgd_step) never bump torch's `_version`; anything that caches a
# derived copy of a parameter (packed conv weights, im2col weight matrices) keys it on version_of(p) instead.
_raw_updates = {}


def note_raw_update(p):
    k = p.data_ptr()
    _raw_updates[k] = _raw_updates.get(k, 0) + 1


# A FusedClipSGD step enqueued on the optimizer's own stream (lib/optim.py: overlap_next_forward) leaves this event; whoever
# touches a trainable parameter or a gradient next makes ITS stream wait for it first (RelModel.forward after the frozen
# detector stage, FusedClipSGD.zero_grad / synchronize).  None: nothing pending.
_pending_param_update = None


def set_pending_param_update(event):
    global _pending_param_update
    _pending_param_update = event


def wait_param_update():
    """make the current stream wait for a deferred optimizer step, if one is in flight (no host synchronisation).  The event is
    consumed only by the thread that runs the step: a helper thread (RelModel.detect_ahead's worker) orders ITS stream behind
    the update but leaves the event for the main stream, whose relation stage reads the trainable weights."""
    global _pending_param_update
    ev = _pending_param_update
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)
        if threading.current_thread() is threading.main_thread():
            _pending_param_update = None


# ---- values cached across calls that are BUILT by kernels on one stream and READ from others (the weight plane images of
# hip_ops / resnet, packed conv weights): the detector stage may run on a worker thread with its own stream
# (RelModel.detect_ahead) while the main stream runs the same frozen layers, and a cache entry is made by whichever of the two
# touches it first.  `cache_lock` keeps the two threads from building an entry twice; `built_here()` marks an entry with an
# event on the building stream and `use_built()` makes any OTHER stream wait for it once before its kernels read the entry
# (eval_rels started two ahead stages before the first in-line batch, with cold caches).
cache_lock = threading.RLock()


class _Built(object):
    __slots__ = ('event', 'stream', 'synced')


def built_here(device):
    """marker of a cache entry whose build kernels were just enqueued on the current stream of `device` (None off the GPU)"""
    if getattr(device, 'type', None) != 'cuda':
        return None
    b = _Built()
    b.stream = torch.cuda.current_stream(device)
    b.event = torch.cuda.Event()
    b.event.record(b.stream)
    b.synced = set()
    return b


def use_built(b):
    """the current stream is about to read the cache entry marked `b`: order it behind the entry's build"""
    if b is None:
        return
    cur = torch.cuda.current_stream(b.stream.device)
    if cur != b.stream and cur.cuda_stream not in b.synced:
        cur.wait_event(b.event)
        b.synced.add(cur.cuda_stream)


def version_of(p):
    """(torch version counter, raw-pointer update count, storage address) of a parameter"""
    return (p._version, _raw_updates.get(p.data_ptr(), 0), p.data_ptr())


def ptr(t):
    """device pointer of a contiguous CUDA tensor (or NULL for None)"""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise HipKernelError('expected a CUDA (HIP) tensor, got a %s tensor: the hot path has no CPU fallback'
                             % t.device.type)
    if not t.is_contiguous():
        raise HipKernelError('tensor must be contiguous')
    return ctypes.c_void_p(t.data_ptr())


def f32(t):
    if t is not None and t.dtype != torch.float32:
        raise HipKernelError('expected float32, got %s' % t.dtype)
    return ptr(t)


def i32(t):
    if t is not None and t.dtype != torch.int32:
        raise HipKernelError('expected int32, got %s' % t.dtype)
    return ptr(t)


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_ws_cache = {}


def workspace(nbytes, device, tag='default'):
    """Grow-only scratch buffer per (device, tag).  All users are ordered on the current stream."""
    nbytes = max(int(nbytes), 256)
    # one buffer per (device, purpose, stream): kernels enqueued on different HIP streams may run concurrently
    key = (str(device), tag, torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# include/motifs_hip.h MH_EPI_WS_ZEROED: the workspace's counter words are zero on entry and stay zero (MOTIFS_PLCONV_WS_ZEROED=0: the
# library clears them per launch, the A/B arm)
EPI_WS_ZEROED = 0x100 if os.environ.get('MOTIFS_PLCONV_WS_ZEROED', '1') != '0' else 0


def zeroed_workspace(nbytes, device, tag):
    """like workspace(), but zero-filled when it is (re)allocated: for kernels that need zeroed scratch on entry and leave it
    zeroed (the split-K arrival counters of mh_gemm_small_f32) -- no memset per call"""
    nbytes = max(int(nbytes), 256)
    key = (str(device), tag, torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
        _zeroed_keys.add(key)
    return buf


_zeroed_keys = set()


def drop_zeroed_workspaces():
    """Forget every zero-on-allocation workspace: the next call allocates a fresh zeroed one.  Called when a launch reports an
    error or a persistent kernel a fault -- a product that stopped half-way leaves arrival counters non-zero, and every later
    product on that stream would skip or repeat its reduction."""
    for key in list(_zeroed_keys):
        _ws_cache.pop(key, None)
    _zeroed_keys.clear()


c_int = ctypes.c_int
c_ll = ctypes.c_longlong
c_float = ctypes.c_float
c_size_t = ctypes.c_size_t


# ----------------------------------------------------------------------------------------------- GEMM
def gemm(a, b, trans_a=False, trans_b=False, bias=None, epilogue=0, out=None, accumulate=False, splitk=0):
    """C = epi(op(a) @ op(b) + bias) on the FP32 MFMA GEMM.  a, b: 2-D fp32 CUDA tensors whose last dim is
    contiguous (row stride may exceed the width)."""
    L = lib()
    for t in (a, b):
        if t.dim() != 2 or t.stride(1) != 1 or not t.is_cuda or t.dtype != torch.float32:
            raise HipKernelError('gemm operands must be 2-D fp32 CUDA tensors with unit inner stride')
    M = a.shape[1] if trans_a else a.shape[0]
    K = a.shape[0] if trans_a else a.shape[1]
    Kb = b.shape[1] if trans_b else b.shape[0]
    N = b.shape[0] if trans_b else b.shape[1]
    if K != Kb:
        raise HipKernelError('gemm inner dimensions differ: %d vs %d' % (K, Kb))
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
        accumulate = False
    elif out.shape != (M, N) or out.stride(1) != 1:
        raise HipKernelError('bad output tensor for gemm')
    if M == 0 or N == 0:
        return out
    if K == 0:
        if not accumulate:
            out.zero_()
        return out
    if splitk <= 0 and is_small_product(M, N, K):
        return _gemm_small(a, b, trans_a, trans_b, bias, epilogue, out, accumulate)
    if splitk <= 0:
        splitk = L.mh_gemm_auto_splitk(M, N, K)
    wsb = L.mh_gemm_ws_bytes(M, N, K, splitk)
    ws = workspace(wsb, a.device, 'gemm') if wsb else None
    rc = L.mh_gemm_f32(c_int(int(trans_a)), c_int(int(trans_b)), M, N, K,
                       ctypes.c_void_p(a.data_ptr()), c_int(a.stride(0)),
                       ctypes.c_void_p(b.data_ptr()), c_int(b.stride(0)),
                       ctypes.c_void_p(out.data_ptr()), c_int(out.stride(0)),
                       f32(bias), c_int(epilogue), c_int(int(accumulate)), c_int(splitk),
                       ptr(ws), c_size_t(ws.numel() if ws is not None else 0), stream())
    _check(rc, 'mh_gemm_f32')
    return out


class PlaneImage(object):
    """An operand of the plane GEMM (csrc/pl_tile.h): `rows` rows of K elements as f16 (h1 | h2) cells + row maxima, in
    one uint8 device buffer.  Made once per operand VALUE and orientation (make_planes), consumed by gemm_planes."""
    __slots__ = ('buf', 'rows', 'K')

    def __init__(self, buf, rows, K):
        self.buf, self.rows, self.K = buf, int(rows), int(K)


def make_planes(x, k_contiguous=True):
    """plane image of the operand held by the 2-D fp32 tensor x: k_contiguous -> operand rows = x rows (x is [rows, K]);
    otherwise operand rows = x COLUMNS (x is [K, rows]: the transposition happens in this pass)"""
    L = lib()
    if x.dim() != 2 or x.stride(1) != 1 or not x.is_cuda or x.dtype != torch.float32:
        raise HipKernelError('make_planes needs a 2-D fp32 CUDA tensor with unit inner stride')
    rows, K = (x.shape[0], x.shape[1]) if k_contiguous else (x.shape[1], x.shape[0])
    if rows == 0 or K == 0:
        raise HipKernelError('make_planes: empty operand')
    buf = torch.empty(L.mh_planes_bytes(c_ll(rows), c_ll(K)), dtype=torch.uint8, device=x.device)
    rc = L.mh_make_planes(ctypes.c_void_p(x.data_ptr()), c_int(int(k_contiguous)), c_ll(rows), c_ll(K), c_ll(x.stride(0)),
                          ctypes.c_void_p(buf.data_ptr()), stream())
    _check(rc, 'mh_make_planes')
    return PlaneImage(buf, rows, K)


def make_planes_both(x):
    """(image with operand rows = x rows, image with operand rows = x columns) of a 2-D fp32 tensor, one HBM read"""
    L = lib()
    if x.dim() != 2 or x.stride(1) != 1 or not x.is_cuda or x.dtype != torch.float32:
        raise HipKernelError('make_planes_both needs a 2-D fp32 CUDA tensor with unit inner stride')
    R, C = x.shape
    if R == 0 or C == 0:
        raise HipKernelError('make_planes_both: empty operand')
    br = torch.empty(L.mh_planes_bytes(c_ll(R), c_ll(C)), dtype=torch.uint8, device=x.device)
    bc = torch.empty(L.mh_planes_bytes(c_ll(C), c_ll(R)), dtype=torch.uint8, device=x.device)
    rc = L.mh_make_planes_both(ctypes.c_void_p(x.data_ptr()), c_ll(R), c_ll(C), c_ll(x.stride(0)), ctypes.c_void_p(br.data_ptr()),
                               ctypes.c_void_p(bc.data_ptr()), stream())
    _check(rc, 'mh_make_planes_both')
    return PlaneImage(br, R, C), PlaneImage(bc, C, R)


def gemm_planes(a, b, bias=None, epilogue=0, out=None, accumulate=False, splitk=0):
    """C[a.rows, b.rows] = epi(A . B^T + bias) (+ C) from two plane images with the same K"""
    L = lib()
    if a.K != b.K:
        raise HipKernelError('gemm_planes inner dimensions differ: %d vs %d' % (a.K, b.K))
    M, N, K = a.rows, b.rows, a.K
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.buf.device)
        accumulate = False
    elif out.shape != (M, N) or out.stride(1) != 1:
        raise HipKernelError('bad output tensor for gemm_planes')
    wsb = L.mh_gemm_planes_ws_bytes(M, N, K, splitk)
    ws = workspace(wsb, a.buf.device, 'gemm') if wsb else None
    rc = L.mh_gemm_planes(M, N, K, ctypes.c_void_p(a.buf.data_ptr()), ctypes.c_void_p(b.buf.data_ptr()),
                          ctypes.c_void_p(out.data_ptr()), c_int(out.stride(0)), f32(bias), c_int(epilogue),
                          c_int(int(accumulate)), c_int(splitk), ptr(ws), c_size_t(ws.numel() if ws is not None else 0), stream())
    _check(rc, 'mh_gemm_planes')
    return out


def is_small_product(M, N, K):
    """the dispatch rule of mh_gemm_f32 (csrc/pl_gemm.hip): products below 20 GFLOP or with K < 512 are not worth plane images"""
    return 2.0 * M * N * K < 20e9 or K < 512


def gemm_inloop(a, b, trans_a=False, trans_b=False, bias=None, epilogue=0, out=None, accumulate=False):
    """see _gemm_small (the name is round 2's: the product whose operands are split inside the K loop)"""
    return _gemm_small(a, b, trans_a, trans_b, bias, epilogue, out, accumulate)


def _gemm_small(a, b, trans_a=False, trans_b=False, bias=None, epilogue=0, out=None, accumulate=False):
    """the small-product engine (csrc/gemm.hip, mh_gemm_small_f32): fp32 operands read once and split into three bf16 terms
    inside the K loop, no pass for row maxima, the split-K reduction fused into the same launch -- ONE launch per product.
    For everything that is not worth plane images: the ~40 small products of a step and the skinny product (<= 128 rows)
    against a big weight matrix that changes every step and is read exactly once (the trainable object fc6)."""
    L = lib()
    M = a.shape[1] if trans_a else a.shape[0]
    K = a.shape[0] if trans_a else a.shape[1]
    N = b.shape[0] if trans_b else b.shape[1]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
        accumulate = False
    elif tuple(out.shape) != (M, N) or out.stride(1) != 1 or out.dtype != torch.float32:
        raise HipKernelError('bad output tensor for gemm_inloop')
    if M == 0 or N == 0:
        return out
    splitk = 0                     # the library plans the K split and whether its reduction is fused (csrc/gemm.hip: plan_small)
    wsb = L.mh_gemm_ws_bytes_v2(M, N, K, splitk)
    ws = workspace(wsb, a.device, 'gemm') if wsb else None
    nctr = L.mh_gemm_small_max_counters()
    ctr = zeroed_workspace(4 * nctr, a.device, 'gemm_counters')
    rc = L.mh_gemm_small_f32(c_int(int(trans_a)), c_int(int(trans_b)), M, N, K, ctypes.c_void_p(a.data_ptr()), c_int(a.stride(0)),
                             ctypes.c_void_p(b.data_ptr()), c_int(b.stride(0)), ctypes.c_void_p(out.data_ptr()), c_int(out.stride(0)),
                             f32(bias), c_int(epilogue), c_int(int(accumulate)), c_int(splitk), ptr(ws),
                             c_size_t(ws.numel() if ws is not None else 0), ptr(ctr), c_int(nctr), stream())
    _check(rc, 'mh_gemm_small_f32')
    return out


# ----------------------------------------------------------------------------------------------- NMS
def nms(boxes_sorted, thresh):
    """boxes_sorted [n,4] fp32 (score-descending).  Returns (keep int32 [n], num_keep int32 [1]) on device."""
    L = lib()
    n = boxes_sorted.shape[0]
    keep = torch.empty(max(n, 1), dtype=torch.int32, device=boxes_sorted.device)
    num = torch.zeros(1, dtype=torch.int32, device=boxes_sorted.device)
    ws = workspace(L.mh_nms_ws_bytes(n), boxes_sorted.device, 'nms')
    rc = L.mh_nms(f32(boxes_sorted), n, c_float(thresh), i32(keep), i32(num), ptr(ws), c_size_t(ws.numel()),
                  stream())
    _check(rc, 'mh_nms')
    return keep, num


def nms_batched(boxes_sorted, seg_offsets, max_seg, thresh):
    """boxes_sorted [total,4]; seg_offsets int32 [nseg+1] (device).  Returns keep [total] (segment-relative
    positions, packed at each segment's offset) and num_keep [nseg]."""
    L = lib()
    total = boxes_sorted.shape[0]
    nseg = seg_offsets.numel() - 1
    keep = torch.empty(max(total, 1), dtype=torch.int32, device=boxes_sorted.device)
    num = torch.zeros(max(nseg, 1), dtype=torch.int32, device=boxes_sorted.device)
    ws = workspace(L.mh_nms_batched_ws_bytes(total, nseg, max_seg), boxes_sorted.device, 'nms')
    rc = L.mh_nms_batched(f32(boxes_sorted), i32(seg_offsets), nseg, total, int(max_seg), c_float(thresh),
                          i32(keep), i32(num), ptr(ws), c_size_t(ws.numel()), stream())
    _check(rc, 'mh_nms_batched')
    return keep, num


# ----------------------------------------------------------------------------------------------- RoIAlign
def roi_align_fwd(feat, rois, ph, pw, spatial_scale, nhwc):
    """feat: [B,C,H,W] (nhwc=False) or [B,H,W,C] (nhwc=True) contiguous.  -> [n,C,ph,pw]"""
    if nhwc:
        B, H, W, C = feat.shape
    else:
        B, C, H, W = feat.shape
    n = rois.shape[0]
    out = torch.empty(n, C, ph, pw, dtype=torch.float32, device=feat.device)
    rc = lib().mh_roi_align_fwd(f32(feat), B, C, H, W, c_int(int(nhwc)), f32(rois), n, ph, pw,
                                c_float(spatial_scale), f32(out), stream())
    _check(rc, 'mh_roi_align_fwd')
    return out


def roi_align_bwd(grad_out, rois, B, C, H, W, spatial_scale, nhwc):
    n, _, ph, pw = grad_out.shape
    shape = (B, H, W, C) if nhwc else (B, C, H, W)
    gf = torch.empty(shape, dtype=torch.float32, device=grad_out.device)
    # deterministic gather by default (bit-reproducible gradients into the trunk); MOTIFS_ROIALIGN_BWD=atomic selects the
    # reference-style atomicAdd scatter
    det = os.environ.get('MOTIFS_ROIALIGN_BWD', 'gather') != 'atomic' and C <= 1024
    fn = lib().mh_roi_align_bwd_det if det else lib().mh_roi_align_bwd
    rc = fn(f32(grad_out), B, C, H, W, c_int(int(nhwc)), f32(rois), n, ph, pw, c_float(spatial_scale), f32(gf), stream())
    _check(rc, 'mh_roi_align_bwd_det' if det else 'mh_roi_align_bwd')
    return gf


def draw_union_boxes(box_pairs, P, offset=0.0, channels_last=False):
    n = box_pairs.shape[0]
    shape = (n, P, P, 2) if channels_last else (n, 2, P, P)
    out = torch.empty(shape, dtype=torch.float32, device=box_pairs.device)
    rc = lib().mh_draw_union_boxes(f32(box_pairs), n, P, c_float(offset), c_int(int(channels_last)), f32(out),
                                   stream())
    _check(rc, 'mh_draw_union_boxes')
    return out


def _i64(t):
    if t is not None and t.dtype != torch.int64:
        raise HipKernelError('expected int64, got %s' % t.dtype)
    return ptr(t)


def pair_product_fwd(edge, i1, i2, vis=None):
    """edge [n,2,D] (subject / object representation of every box), i1 / i2 [R] int64, vis [R,D] or None ->
    edge[i1, 0] * edge[i2, 1] (* vis), one launch (csrc/exact_ops.hip: pair_product_fwd_kernel)"""
    n, D, R = edge.shape[0], edge.shape[2], i1.shape[0]
    out = torch.empty(R, D, dtype=torch.float32, device=edge.device)
    _check(lib().mh_pair_product_fwd(f32(edge), n, D, _i64(i1), _i64(i2), R, f32(vis), f32(out), stream()), 'mh_pair_product_fwd')
    return out


def pair_product_bwd(edge, i1, i2, vis, grad_out, order, ptr_):
    """-> (d_edge [n,2,D], d_vis [R,D] or None); order [2,R] / ptr_ [2,n+1] int32: the rows of every box per side (host-made,
    see lib/rel_model.py: _PairProductFn), summed in list order"""
    n, D, R = edge.shape[0], edge.shape[2], i1.shape[0]
    d_edge = torch.empty_like(edge)
    d_vis = torch.empty(R, D, dtype=torch.float32, device=edge.device) if vis is not None else None
    _check(lib().mh_pair_product_bwd(f32(edge), n, D, _i64(i1), _i64(i2), R, f32(vis), f32(grad_out), i32(order), i32(ptr_), f32(d_edge),
                                     f32(d_vis), stream()), 'mh_pair_product_bwd')
    return d_edge, d_vis


FREQ_BIAS_MAX_ROWS = 8192


def freq_bias_add(logits, table, labels, i1, i2, num_objs):
    """logits [R,P] + table[labels[i1] * num_objs + labels[i2]] -> (out [R,P], keys [R] int64)"""
    R, P = logits.shape
    out = torch.empty_like(logits)
    keys = torch.empty(R, dtype=torch.int64, device=logits.device)
    _check(lib().mh_freq_bias_add(f32(logits), f32(table), _i64(labels), _i64(i1), _i64(i2), R, P, int(num_objs), f32(out), _i64(keys),
                                  stream()), 'mh_freq_bias_add')
    return out, keys


def freq_bias_bwd(grad_out, keys, table_rows):
    """-> d_table [table_rows, P]: the gradient rows of every key summed in ascending row order"""
    R, P = grad_out.shape
    d_table = torch.empty(table_rows, P, dtype=torch.float32, device=grad_out.device)
    _check(lib().mh_freq_bias_bwd(f32(grad_out), _i64(keys), R, P, c_ll(table_rows), f32(d_table), stream()), 'mh_freq_bias_bwd')
    return d_table


def _labels_arg(labels):
    """(pointer, element stride) of a 1-D int64 label view (e.g. rel_labels[:, -1]: stride 4)"""
    if labels.dtype != torch.int64 or labels.dim() != 1 or not labels.is_cuda:
        raise HipKernelError('labels: expected a 1-D int64 CUDA (HIP) tensor')
    stride = labels.stride(0) if labels.numel() > 1 else 1
    if stride < 1:
        raise HipKernelError('labels: expected a positive element stride')
    return ctypes.c_void_p(labels.data_ptr()), c_ll(stride)          # a strided view is fine: the kernels index labels[r * stride]


def ce_pair_fwd(logits_a, labels_a, logits_b, labels_b):
    """the two mean cross-entropy losses -> (losses [2], lse [Ra + Rb])"""
    Ra, Ca = logits_a.shape
    Rb, Cb = logits_b.shape
    dev = logits_a.device
    lse = torch.empty(Ra + Rb, dtype=torch.float32, device=dev)
    rowloss = torch.empty(Ra + Rb, dtype=torch.float32, device=dev)
    losses = torch.empty(2, dtype=torch.float32, device=dev)
    pa, sa = _labels_arg(labels_a)
    pb, sb = _labels_arg(labels_b)
    _check(lib().mh_ce_pair_fwd(f32(logits_a), pa, sa, Ra, Ca, f32(logits_b), pb, sb, Rb, Cb, f32(lse), f32(rowloss), f32(losses), stream()),
           'mh_ce_pair_fwd')
    return losses, lse


def ce_pair_bwd(logits_a, labels_a, logits_b, labels_b, lse, upstream, need_a=True, need_b=True):
    """-> (grad_a, grad_b) of the two mean losses; upstream [2] fp32 on the device"""
    Ra, Ca = logits_a.shape
    Rb, Cb = logits_b.shape
    ga = torch.empty_like(logits_a) if need_a else None
    gb = torch.empty_like(logits_b) if need_b else None
    pa, sa = _labels_arg(labels_a)
    pb, sb = _labels_arg(labels_b)
    _check(lib().mh_ce_pair_bwd(f32(logits_a), pa, sa, Ra, Ca, f32(logits_b), pb, sb, Rb, Cb, f32(lse), f32(upstream), f32(ga), f32(gb), stream()),
           'mh_ce_pair_bwd')
    return ga, gb


def bbox_overlaps(a, b):
    out = torch.empty(a.shape[0], b.shape[0], dtype=torch.float32, device=a.device)
    rc = lib().mh_bbox_overlaps(f32(a), a.shape[0], f32(b), b.shape[0], f32(out), stream())
    _check(rc, 'mh_bbox_overlaps')
    return out


def triplet_match(gt_triplets, gt_boxes, pred_triplets, pred_boxes, iou_thresh=0.5):
    """(first_match [G] int32, nmatch [P] int32): see mh_triplet_match"""
    G, P = gt_triplets.shape[0], pred_triplets.shape[0]
    dev = gt_boxes.device
    first = torch.empty(G, dtype=torch.int32, device=dev)
    nmatch = torch.empty(P, dtype=torch.int32, device=dev)
    gt_t, pr_t = gt_triplets.to(torch.int32).contiguous(), pred_triplets.to(torch.int32).contiguous()
    gt_b, pr_b = gt_boxes.float().contiguous(), pred_boxes.float().contiguous()
    _check(lib().mh_triplet_match(ptr(gt_t), f32(gt_b), G, ptr(pr_t), f32(pr_b), P, ctypes.c_double(iou_thresh),
                                  ptr(first), ptr(nmatch), stream()), 'mh_triplet_match')
    return first, nmatch


# ----------------------------------------------------------------------------------------------- conv stack
def conv3x3_pack_weight(w, flip_transpose=False):
    """packed weights (opaque fp32 container [9, N, *]) of the conv that consumes them: N = Cout outputs, or for the
    dgrad conv (flip_transpose) N = Cin outputs"""
    Cout, Cin = w.shape[0], w.shape[1]
    N, K = (Cin, Cout) if flip_transpose else (Cout, Cin)
    wt = torch.empty(9, N, lib().mh_conv3x3_packed_floats(N, K) // (9 * N), dtype=torch.float32, device=w.device)
    rc = lib().mh_conv3x3_pack_weight(f32(w), Cout, Cin, c_int(int(flip_transpose)), f32(wt), stream())
    _check(rc, 'mh_conv3x3_pack_weight')
    return wt


def conv3x3_nhwc(x, wt, bias, epilogue):
    """x [B,H,W,Cin], wt = conv3x3_pack_weight(...) [9,Cout,*] -> [B,H,W,Cout]"""
    B, H, W, Cin = x.shape
    Cout = wt.shape[1]
    out = torch.empty(B, H, W, Cout, dtype=torch.float32, device=x.device)
    wsb = lib().mh_conv3x3_ws_bytes(B, H, W, Cin, Cout)
    ws = workspace(wsb, x.device, 'conv') if wsb else None
    rc = lib().mh_conv3x3_nhwc(f32(x), B, H, W, Cin, f32(wt), Cout, f32(bias), c_int(epilogue), f32(out),
                               ptr(ws), c_size_t(ws.numel() if ws is not None else 0), stream())
    _check(rc, 'mh_conv3x3_nhwc')
    return out


def conv3x3_schedule(B, H, W, Cin, Cout):
    """dict of the tile schedule mh_conv3x3_nhwc uses for this shape (host arithmetic only)"""
    out = (ctypes.c_int * 8)()
    _check(lib().mh_conv3x3_schedule(B, H, W, Cin, Cout, out), 'mh_conv3x3_schedule')
    keys = ('bm', 'bn', 'tiles_m', 'tiles_n', 'splitk', 'body_mtiles', 'tail_tiles', 'tail_slices')
    return dict(zip(keys, list(out)))


def conv3x3_wgrad(x, gy):
    """dW [Cout, 9*Cin] (tap-major, then cin) of the 3x3/1/1 conv from x [B,H,W,Cin], gy [B,H,W,Cout]; None when this
    build has no implicit-GEMM wgrad kernel (f32-MFMA build: the caller uses im2col + gemm)"""
    if os.environ.get('MOTIFS_WGRAD') == 'im2col':          # A/B switch: the patch-matrix path
        return None
    B, H, W, Cin = x.shape
    Cout = gy.shape[3]
    dw = torch.empty(Cout, 9 * Cin, dtype=torch.float32, device=x.device)
    ws = workspace(lib().mh_conv3x3_wgrad_ws_bytes(B, H, W, Cin, Cout), x.device, 'wgrad')
    rc = lib().mh_conv3x3_wgrad(f32(x), f32(gy), B, H, W, Cin, Cout, f32(dw), ptr(ws), c_size_t(ws.numel()), stream())
    if rc == -2:
        return None
    _check(rc, 'mh_conv3x3_wgrad')
    return dw


class ActImage(object):
    """activation plane image (csrc/pl_conv.hip) of an NHWC tensor [B,H,W,C]: uint8 buffer = cells + per-image maxima"""
    __slots__ = ('buf', 'B', 'H', 'W', 'C')

    def __init__(self, buf, B, H, W, C):
        self.buf, self.B, self.H, self.W, self.C = buf, int(B), int(H), int(W), int(C)


def act_planes(x_nhwc, maxbits, pool=False):
    """fp32 NHWC tensor + its per-image |x| maxima (int32 [B], fp32 bit patterns, from the producing kernel) -> ActImage
    of the tensor, or of its 2x2/2 max-pool"""
    L = lib()
    B, H, W, C = x_nhwc.shape
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    buf = torch.empty(L.mh_act_planes_bytes(B, Ho, Wo, C), dtype=torch.uint8, device=x_nhwc.device)
    rc = L.mh_act_planes(f32(x_nhwc), i32(maxbits), B, H, W, C, c_int(int(pool)), ctypes.c_void_p(buf.data_ptr()), stream())
    _check(rc, 'mh_act_planes')
    return ActImage(buf, B, Ho, Wo, C)


def image_maxbits(x_nhwc):
    """int32 [B]: the largest |x| of every image of an NHWC tensor as fp32 bit patterns (input of act_planes)"""
    B = x_nhwc.shape[0]
    bits = torch.empty(B, dtype=torch.int32, device=x_nhwc.device)
    _check(lib().mh_image_maxbits(f32(x_nhwc), B, c_ll(x_nhwc.numel() // B), i32(bits), stream()), 'mh_image_maxbits')
    return bits


def plconv_many_images_ok(B, H, W, cin, cout):
    """can mh_plconv3x3 run a 3x3 conv over B small maps on the ring engine?  (Cout >= 128: the ring shapes; 32-bit image offsets)"""
    M = B * H * W
    return (cin % 16 == 0 and cout % 4 == 0 and cout >= 128 and B <= 65535 and M < (1 << 25)
            and (M * cin * 4 + (2 * (W + 1) + 256) * 64) < 0x7ff00000 and M * cout * 4 < 0x7ff00000)


def plconv_pack_weight(w, flip_transpose=False):
    L = lib()
    Cout, Cin = w.shape[0], w.shape[1]
    N, K = (Cin, Cout) if flip_transpose else (Cout, Cin)
    buf = torch.empty(L.mh_plconv_packed_bytes(N, K), dtype=torch.uint8, device=w.device)
    rc = L.mh_plconv_pack_weight(f32(w), Cout, Cin, c_int(int(flip_transpose)), ctypes.c_void_p(buf.data_ptr()), stream())
    _check(rc, 'mh_plconv_pack_weight')
    return buf


def plconv3x3(img, packed, cout, bias, epilogue, out_maxbits=None):
    """3x3 / 1 / 1 conv of an ActImage with packed plane weights -> fp32 NHWC [B,H,W,cout]; out_maxbits (int32 [B], zeroed by
    the caller) receives the per-image maxima of the output"""
    L = lib()
    out = torch.empty(img.B, img.H, img.W, cout, dtype=torch.float32, device=img.buf.device)
    wsb = L.mh_plconv3x3_ws_bytes(img.B, img.H, img.W, img.C, cout)
    ws = zeroed_workspace(wsb, out.device, 'plconv') if wsb else None        # counters first, zero once: no memset per launch (MH_EPI_WS_ZEROED)
    rc = L.mh_plconv3x3(ctypes.c_void_p(img.buf.data_ptr()), img.B, img.H, img.W, img.C, ctypes.c_void_p(packed.data_ptr()), cout,
                        f32(bias), c_int(epilogue | EPI_WS_ZEROED), f32(out), i32(out_maxbits), ptr(ws),
                        c_size_t(ws.numel() if ws is not None else 0), stream())
    _check(rc, 'mh_plconv3x3')
    return out


def plconv3x3_to_image(img, in_true_maxbits, packed, cout, bias, epilogue, out_maxbits):
    """the same conv, output written AS the next layer's ActImage (no fp32 tensor); in_true_maxbits / out_maxbits: int32 [B]
    true per-image maxima of input / output (out zeroed by the caller)"""
    L = lib()
    buf = torch.empty(L.mh_act_planes_bytes(img.B, img.H, img.W, cout), dtype=torch.uint8, device=img.buf.device)
    wsb = L.mh_plconv3x3_ws_bytes(img.B, img.H, img.W, img.C, cout)
    ws = zeroed_workspace(wsb, buf.device, 'plconv') if wsb else None
    rc = L.mh_plconv3x3_to_image(ctypes.c_void_p(img.buf.data_ptr()), i32(in_true_maxbits), img.B, img.H, img.W, img.C,
                                 ctypes.c_void_p(packed.data_ptr()), cout, f32(bias), c_int(epilogue | EPI_WS_ZEROED), ctypes.c_void_p(buf.data_ptr()),
                                 i32(out_maxbits), ptr(ws), c_size_t(ws.numel() if ws is not None else 0), stream())
    _check(rc, 'mh_plconv3x3_to_image')
    return ActImage(buf, img.B, img.H, img.W, cout)


def plconv3x3_pool_to_image(img, in_true_maxbits, packed, cout, bias, epilogue, out_maxbits):
    """the same conv through the 2x2 / 2 max-pool behind it: the ActImage of the POOLED output [B, H/2, W/2, cout], pooled in the
    kernel's epilogue (no fp32 tensor, no converter pass); H and W even"""
    L = lib()
    if img.H % 2 or img.W % 2:
        raise HipKernelError('plconv3x3_pool_to_image needs even map sizes, got %d x %d' % (img.H, img.W))
    buf = torch.empty(L.mh_act_planes_bytes(img.B, img.H // 2, img.W // 2, cout), dtype=torch.uint8, device=img.buf.device)
    wsb = L.mh_plconv3x3_ws_bytes(img.B, img.H, img.W, img.C, cout)
    ws = zeroed_workspace(wsb, buf.device, 'plconv') if wsb else None
    rc = L.mh_plconv3x3_pool_to_image(ctypes.c_void_p(img.buf.data_ptr()), i32(in_true_maxbits), img.B, img.H, img.W, img.C,
                                      ctypes.c_void_p(packed.data_ptr()), cout, f32(bias), c_int(epilogue | EPI_WS_ZEROED), ctypes.c_void_p(buf.data_ptr()),
                                      i32(out_maxbits), ptr(ws), c_size_t(ws.numel() if ws is not None else 0), stream())
    _check(rc, 'mh_plconv3x3_pool_to_image')
    return ActImage(buf, img.B, img.H // 2, img.W // 2, cout)


def stem_to_image(x, w, bias, epilogue, out_maxbits):
    """conv1_1: NCHW image -> ActImage of its [B,H,W,Cout] output (bias + activation fused)"""
    L = lib()
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    buf = torch.empty(L.mh_act_planes_bytes(B, H, W, Cout), dtype=torch.uint8, device=x.device)
    rc = L.mh_stem_to_image(f32(x), B, Cin, H, W, f32(w), Cout, f32(bias), c_int(epilogue), ctypes.c_void_p(buf.data_ptr()),
                            i32(out_maxbits), stream())
    _check(rc, 'mh_stem_to_image')
    return ActImage(buf, B, H, W, Cout)


def conv_first_nchw_max(x, w, bias, epilogue, maxbits):
    L = lib()
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    out = torch.empty(B, H, W, Cout, dtype=torch.float32, device=x.device)
    rc = L.mh_conv_first_nchw_max(f32(x), B, Cin, H, W, f32(w), Cout, f32(bias), c_int(epilogue), f32(out), i32(maxbits), stream())
    _check(rc, 'mh_conv_first_nchw_max')
    return out


def conv_first_nchw(x, w, bias, epilogue):
    """x [B,Cin,H,W] NCHW, w [Cout,Cin,3,3] -> NHWC [B,H,W,Cout]"""
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    out = torch.empty(B, H, W, Cout, dtype=torch.float32, device=x.device)
    rc = lib().mh_conv_first_nchw(f32(x), B, Cin, H, W, f32(w), Cout, f32(bias), c_int(epilogue), f32(out), stream())
    _check(rc, 'mh_conv_first_nchw')
    return out


def maxpool2x2_nhwc(x):
    B, H, W, C = x.shape
    out = torch.empty(B, H // 2, W // 2, C, dtype=torch.float32, device=x.device)
    rc = lib().mh_maxpool2x2_nhwc(f32(x), B, H, W, C, f32(out), stream())
    _check(rc, 'mh_maxpool2x2_nhwc')
    return out


def maxpool2x2_bwd_nhwc(x, gy):
    B, H, W, C = x.shape
    gx = torch.empty_like(x)
    _check(lib().mh_maxpool2x2_bwd_nhwc(f32(x), f32(gy), B, H, W, C, f32(gx), stream()), 'mh_maxpool2x2_bwd_nhwc')
    return gx


def act_bwd(g, y, epilogue):
    """gradient through a fused ReLU (1) / ReLU6 (2) epilogue; y = the activated output"""
    out = torch.empty_like(g)
    _check(lib().mh_act_bwd(f32(g), f32(y), c_ll(g.numel()), c_int(epilogue), f32(out), stream()), 'mh_act_bwd')
    return out


def im2col_nhwc(x, kh, kw, stride, pad, ldo=None):
    B, H, W, C = x.shape
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    K = kh * kw * C
    ldo = K if ldo is None else ldo
    out = torch.empty(B * Ho * Wo, ldo, dtype=torch.float32, device=x.device)
    rc = lib().mh_im2col_nhwc(f32(x), B, H, W, C, kh, kw, stride, pad, f32(out), ldo, stream())
    _check(rc, 'mh_im2col_nhwc')
    return out, Ho, Wo


def nchw_to_nhwc(x):
    B, C, H, W = x.shape
    out = torch.empty(B, H, W, C, dtype=torch.float32, device=x.device)
    _check(lib().mh_nchw_to_nhwc(f32(x), B, C, H, W, f32(out), stream()), 'mh_nchw_to_nhwc')
    return out


def nhwc_to_nchw(x):
    B, H, W, C = x.shape
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
    _check(lib().mh_nhwc_to_nchw(f32(x), B, C, H, W, f32(out), stream()), 'mh_nhwc_to_nchw')
    return out


# ----------------------------------------------------------------------------------------------- LSTM
def _lengths_array(lengths):
    arr = (ctypes.c_int * len(lengths))(*[int(v) for v in lengths])
    return arr


def _check_lstm_params(in_size, H, L_, weight, bias):
    """the flat parameter vectors must have exactly the size the layer geometry implies (alternating_highway_lstm.py:
    213-229): the kernels index them with (in_size, H, L) and would otherwise read past the end"""
    want_w = sum(6 * H * (in_size if l == 0 else H) + 5 * H * H for l in range(L_))
    if weight.numel() != want_w or (bias is not None and bias.numel() != 5 * H * L_):
        raise HipKernelError('highway LSTM parameters do not match the input: %d weights / %s biases given, input size %d, '
                             'hidden %d, %d layers need %d / %d' % (weight.numel(), None if bias is None else bias.numel(),
                                                                   in_size, H, L_, want_w, 5 * H * L_))


def hwlstm_fwd(x, lengths, weight, bias, dropout, H, L_, training):
    """x [T,B,in]; returns (h_data, c_data [L,T+1,B,H], gates [L,T,B,6H] or None)"""
    T, B, in_size = x.shape
    _check_lstm_params(in_size, H, L_, weight, bias)
    dev = x.device
    h_data = torch.zeros(L_, T + 1, B, H, dtype=torch.float32, device=dev)
    c_data = torch.zeros(L_, T + 1, B, H, dtype=torch.float32, device=dev)
    gates = torch.empty(L_, T, B, 6 * H, dtype=torch.float32, device=dev) if training else None
    wsb = lib().mh_hwlstm_fwd_ws_bytes(in_size, H, B, L_, T)
    ws = workspace(wsb, dev, 'lstm')
    rc = lib().mh_hwlstm_fwd(in_size, H, B, L_, T, f32(x), _lengths_array(lengths), f32(h_data), f32(c_data),
                             f32(weight), f32(bias), f32(dropout), f32(gates), c_int(int(training)),
                             ptr(ws), c_size_t(ws.numel()), stream())
    _check(rc, 'mh_hwlstm_fwd')
    return h_data, c_data, gates


def hwlstm_bwd(out_grad, x, lengths, weight, dropout, H, L_, h_data, c_data, gates, need_weight_grad=True):
    T, B, in_size = x.shape
    _check_lstm_params(in_size, H, L_, weight, None)
    dev = x.device
    x_grad = torch.empty_like(x)
    w_grad = torch.empty_like(weight) if need_weight_grad else None          # every region is written by mh_hwlstm_bwd (no zero fill)
    b_grad = torch.zeros(5 * H * L_, dtype=torch.float32, device=dev) if need_weight_grad else None
    wsb = lib().mh_hwlstm_bwd_ws_bytes(in_size, H, B, L_, T)
    ws = workspace(wsb, dev, 'lstm')
    rc = lib().mh_hwlstm_bwd(in_size, H, B, L_, T, f32(out_grad), _lengths_array(lengths), f32(x), f32(h_data),
                             f32(c_data), f32(weight), f32(gates), f32(dropout), f32(x_grad), f32(w_grad),
                             f32(b_grad), c_int(int(need_weight_grad)), ptr(ws), c_size_t(ws.numel()), stream())
    _check(rc, 'mh_hwlstm_bwd')
    return x_grad, w_grad, b_grad


def hwlstm_cell_fwd(pre_i, h_prev, c_prev, wh_t, bias_h, dropout, want_gates):
    n, H = h_prev.shape
    h_out = torch.empty_like(h_prev)
    c_out = torch.empty_like(c_prev)
    gates = torch.empty(n, 6 * H, dtype=torch.float32, device=h_prev.device) if want_gates else None
    rc = lib().mh_hwlstm_cell_fwd(n, H, f32(pre_i), f32(h_prev), f32(c_prev), f32(wh_t), f32(bias_h), f32(dropout),
                                  f32(h_out), f32(c_out), f32(gates), stream())
    _check(rc, 'mh_hwlstm_cell_fwd')
    return h_out, c_out, gates


def hwlstm_cell_bwd(d_h, d_c_out, c_prev, c_out, gates, dropout):
    n, H = d_h.shape
    d_gates = torch.empty(n, 6 * H, dtype=torch.float32, device=d_h.device)
    d_c_in = torch.empty(n, H, dtype=torch.float32, device=d_h.device)
    rc = lib().mh_hwlstm_cell_bwd(n, H, f32(d_h), f32(d_c_out), f32(c_prev), f32(c_out), f32(gates), f32(dropout),
                                  f32(d_gates), f32(d_c_in), stream())
    _check(rc, 'mh_hwlstm_cell_bwd')
    return d_gates, d_c_in


def hwcell_seq_supported(H, B):
    """shapes the single-launch packed recurrence (mh_hwcell_seq_*) accepts"""
    return H <= 512 and H % 4 == 0 and B <= 32 and not os.environ.get('MOTIFS_STEP_DECODER')


def hwcell_seq_fwd(pre_i_all, batch_sizes, w_state, b_state, dropout):
    """packed recurrence in one launch; returns (h_buf, c_buf [B+N,H] with the B zero initial rows first, gates)"""
    N, H = pre_i_all.shape[0], w_state.shape[1]
    B, T, dev = int(batch_sizes[0]), len(batch_sizes), pre_i_all.device
    h_buf = torch.zeros(B + N, H, dtype=torch.float32, device=dev)
    c_buf = torch.zeros(B + N, H, dtype=torch.float32, device=dev)
    gates = torch.empty(N, 6 * H, dtype=torch.float32, device=dev)
    ws = workspace(lib().mh_hwcell_seq_ws_bytes(), dev, 'lstm_seq')
    rc = lib().mh_hwcell_seq_fwd(H, B, T, _lengths_array(batch_sizes), f32(pre_i_all), f32(w_state), f32(b_state),
                                 f32(dropout), f32(h_buf), f32(c_buf), f32(gates), ptr(ws), c_size_t(ws.numel()),
                                 stream())
    _check(rc, 'mh_hwcell_seq_fwd')
    return h_buf, c_buf, gates


def hwcell_seq_bwd(dh_all, batch_sizes, c_buf, gates, dropout, w_state_t):
    N, H = dh_all.shape
    B, T, dev = int(batch_sizes[0]), len(batch_sizes), dh_all.device
    d_pre = torch.empty(N, 6 * H, dtype=torch.float32, device=dev)
    hg = torch.empty(B + N, H, dtype=torch.float32, device=dev)
    cg = torch.empty(B + N, H, dtype=torch.float32, device=dev)
    ws = workspace(lib().mh_hwcell_seq_ws_bytes(), dev, 'lstm_seq')
    rc = lib().mh_hwcell_seq_bwd(H, B, T, _lengths_array(batch_sizes), f32(dh_all), f32(c_buf), f32(gates),
                                 f32(dropout), f32(w_state_t), f32(d_pre), f32(hg), f32(cg), ptr(ws),
                                 c_size_t(ws.numel()), stream())
    _check(rc, 'mh_hwcell_seq_bwd')
    return d_pre


def decoder_greedy(enc_proj, emb_proj, batch_sizes, w_state, b_state, dropout, w_out, b_out, labels=None):
    """the label decoder's greedy pass in one launch (mh_decoder_greedy).  Returns (h_all [N,H], logits [N,C],
    fed [N] int64 embedding rows, commits [N] int64 labels)"""
    N, H, C = enc_proj.shape[0], w_state.shape[1], w_out.shape[0]
    B, T, dev = int(batch_sizes[0]), len(batch_sizes), enc_proj.device
    h_buf = torch.zeros(B + N, H, dtype=torch.float32, device=dev)
    c_buf = torch.zeros(B + N, H, dtype=torch.float32, device=dev)
    logits = torch.empty(N, C, dtype=torch.float32, device=dev)
    fed = torch.empty(N, dtype=torch.int64, device=dev)
    commits = torch.empty(N, dtype=torch.int64, device=dev)
    if labels is not None and (labels.dtype != torch.int64 or not labels.is_contiguous()):
        raise HipKernelError('labels must be a contiguous int64 tensor')
    ws = workspace(lib().mh_decoder_greedy_ws_bytes(N), dev, 'lstm_seq')
    rc = lib().mh_decoder_greedy(H, B, T, _lengths_array(batch_sizes), C, f32(enc_proj), f32(emb_proj), f32(w_state),
                                 f32(b_state), f32(dropout), f32(w_out), f32(b_out), ptr(labels), f32(h_buf), f32(c_buf),
                                 f32(logits), ptr(fed), ptr(commits), ptr(ws), c_size_t(ws.numel()), stream())
    _check(rc, 'mh_decoder_greedy')
    return h_buf[B:], logits, fed, commits


def decoder_nms_commit_fits(n, c):
    """whether the [n, c] score table fits the single-workgroup suppression kernel's LDS on the current device"""
    return n * c * 4 <= lib().mh_decoder_nms_commit_max_bytes()


def decoder_nms_commit(probs, boxes, thresh):
    """probs [N,C] softmax, boxes [N,C,4] -> commits [N] int64 (class-wise greedy suppression on the device)"""
    N, C = probs.shape
    commits = torch.zeros(N, dtype=torch.int64, device=probs.device)
    _check(lib().mh_decoder_nms_commit(f32(probs), f32(boxes), N, C, c_float(thresh), ptr(commits), stream()),
           'mh_decoder_nms_commit')
    return commits


def gemv_rows(v, wt, bias=None):
    """out[n,R] = v[n,K] @ wt[R,K]^T (+ bias): small-n GEMV, one wave per 4 output rows"""
    n, K = v.shape
    R = wt.shape[0]
    out = torch.empty(n, R, dtype=torch.float32, device=v.device)
    rc = lib().mh_gemv_rows(n, R, K, ctypes.c_void_p(v.data_ptr()), c_int(v.stride(0)),
                            ctypes.c_void_p(wt.data_ptr()), c_int(wt.stride(0)), f32(bias), f32(out), R, stream())
    _check(rc, 'mh_gemv_rows')
    return out


# ----------------------------------------------------------------------------------------------- BN / pool tower
c_ll = ctypes.c_longlong


def bn_stats(x2d, eps, momentum, running_mean=None, running_var=None):
    """x2d [M,C] -> (mean, invstd); running stats updated in place when given"""
    M, C = x2d.shape
    mean = torch.empty(C, dtype=torch.float32, device=x2d.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x2d.device)
    ws = workspace(lib().mh_bn_ws_bytes(c_ll(M), C), x2d.device, 'bn')
    _check(lib().mh_bn_stats(f32(x2d), c_ll(M), C, c_float(eps), c_float(momentum), f32(mean), f32(invstd),
                             f32(running_mean), f32(running_var), ptr(ws), c_size_t(ws.numel()), stream()), 'mh_bn_stats')
    return mean, invstd


def bn_pool_fwd(x, mean, invstd, gamma, beta):
    N, H, W, C = x.shape
    z = torch.empty(N, H // 2, W // 2, C, dtype=torch.float32, device=x.device)
    arg = torch.empty(N, H // 2, W // 2, C, dtype=torch.uint8, device=x.device)
    _check(lib().mh_bn_pool_fwd(f32(x), c_ll(N), H, W, C, f32(mean), f32(invstd), f32(gamma), f32(beta), f32(z),
                                ptr(arg), stream()), 'mh_bn_pool_fwd')
    return z, arg


def bn_residual_nchw(x, mean, invstd, gamma, beta, residual):
    N, Hh, Ww, C = x.shape
    out = torch.empty(N, C, Hh, Ww, dtype=torch.float32, device=x.device)
    _check(lib().mh_bn_residual_nchw(f32(x), c_ll(N), Hh * Ww, C, f32(mean), f32(invstd), f32(gamma), f32(beta),
                                     f32(residual), f32(out), stream()), 'mh_bn_residual_nchw')
    return out


def bn_apply_nhwc(x, mean, invstd, gamma, beta, residual=None, relu=False):
    """act(BN(x) + residual) on an NHWC tensor (last dim = channels)"""
    C = x.shape[-1]
    out = torch.empty_like(x)
    _check(lib().mh_bn_apply_nhwc(f32(x), c_ll(x.numel() // C), C, f32(mean), f32(invstd), f32(gamma), f32(beta),
                                  f32(residual), c_int(int(relu)), f32(out), stream()), 'mh_bn_apply_nhwc')
    return out


def nchw_to_nhwc_small(x):
    N, C, Hh, Ww = x.shape
    out = torch.empty(N, Hh, Ww, C, dtype=torch.float32, device=x.device)
    _check(lib().mh_nchw_to_nhwc_small(f32(x), c_ll(N), Hh * Ww, C, f32(out), stream()), 'mh_nchw_to_nhwc_small')
    return out


def bn_bwd(x, g, argmax, mean, invstd, gamma, relu_mask):
    """x [N,H,W,C]; g dense [N,H,W,C] (argmax None) or pooled [N,H/2,W/2,C] -> (dx, dgamma, dbeta)"""
    N, H, W, C = x.shape
    dx = torch.empty_like(x)
    dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    Mg = g.numel() // C
    ws = workspace(lib().mh_bn_ws_bytes(c_ll(Mg), C), x.device, 'bn')
    _check(lib().mh_bn_bwd(f32(x), f32(g), ptr(argmax), c_ll(N), H, W, C, f32(mean), f32(invstd), f32(gamma),
                           c_int(int(argmax is not None)), c_int(int(relu_mask)), f32(dx), f32(dgamma), f32(dbeta),
                           ptr(ws), c_size_t(ws.numel()), stream()), 'mh_bn_bwd')
    return dx, dgamma, dbeta


# ----------------------------------------------------------------------------------------------- tower conv 1, direct
def tower_conv1_supported(rects_nhwc, weight):
    """the direct kernels cover the reference's layer (7x7, stride 2, padding 3, 2 input channels) for C0 % 256 == 0 and an
    even output size <= 16 (S = 27 -> 14)"""
    C0, Ci, kh, kw = weight.shape
    S = rects_nhwc.shape[1]
    if not (Ci == 2 and kh == 7 and kw == 7 and rects_nhwc.shape[2] == S and rects_nhwc.shape[3] == 2 and C0 % 256 == 0
            and rects_nhwc.shape[0] > 0 and rects_nhwc.dtype == torch.float32):
        return False
    Ho = (S + 6 - 7) // 2 + 1
    return (S + 6 - 7) % 2 == 0 and Ho % 2 == 0 and Ho <= 16


def tower_conv1_pad(rects_nhwc):
    """[N,S,S,2] fp32 -> the zero-padded copy [N,S+6,S+6,2] both direct kernels read"""
    N, S = rects_nhwc.shape[0], rects_nhwc.shape[1]
    xp = torch.empty(N, S + 6, S + 6, 2, dtype=torch.float32, device=rects_nhwc.device)
    _check(lib().mh_tower_conv1_pad(f32(rects_nhwc), c_ll(N), S, f32(xp), stream()), 'mh_tower_conv1_pad')
    return xp


def tower_conv1_fwd(xp, w_kc, bias):
    """relu(conv7x7/2(xp) + bias): xp from tower_conv1_pad, w_kc [98, C0] (k = (ky*7 + kx)*2 + ci) -> y [N,Ho,Wo,C0]"""
    N, S, C0 = xp.shape[0], xp.shape[1] - 6, w_kc.shape[1]
    Ho = lib().mh_tower_conv1_out_size(S)
    y = torch.empty(N, Ho, Ho, C0, dtype=torch.float32, device=xp.device)
    _check(lib().mh_tower_conv1_fwd(f32(xp), c_ll(N), S, f32(w_kc), f32(bias), C0, f32(y), stream()), 'mh_tower_conv1_fwd')
    return y


def tower_conv1_wgrad(xp, dy):
    """dy [N,Ho,Wo,C0] -> (dw_kc [98,C0], db [C0])"""
    N, S, C0 = xp.shape[0], xp.shape[1] - 6, dy.shape[3]
    L = lib()
    out = torch.empty(99, C0, dtype=torch.float32, device=xp.device)
    ws = workspace(L.mh_tower_conv1_wgrad_ws_bytes(c_ll(N), C0), xp.device, 'tower')
    _check(L.mh_tower_conv1_wgrad(f32(xp), f32(dy), c_ll(N), S, C0, f32(out), ptr(ws), c_size_t(ws.numel()), stream()),
           'mh_tower_conv1_wgrad')
    return out[:98], out[98]
