"""Two-stream race, kernel-pair level: RoIAlign (object boxes: 20 rois on a [1,37,37,512] NHWC map) runs on a side stream while
ONE kind of work runs on the main stream; its output is compared with the single-stream result.  Finds the minimal pair."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'neural-motifs_amd')):
    sys.path.insert(0, p)
import torch
from lib import _hip

torch.manual_seed(0)
dev = 'cuda'
fmap = torch.randn(1, 37, 37, 512, device=dev).relu_()
n = 20
xy = torch.rand(n, 2, device=dev) * 300
wh = torch.rand(n, 2, device=dev) * 250 + 20
rois = torch.cat((torch.zeros(n, 1, device=dev), xy, (xy + wh).clamp(max=591)), 1).contiguous()
pairs = torch.tensor([(a, b) for a in range(n) for b in range(n) if a != b], device=dev)
urois = torch.cat((torch.zeros(pairs.shape[0], 1, device=dev), torch.min(rois[pairs[:, 0], 1:3], rois[pairs[:, 1], 1:3]),
                   torch.max(rois[pairs[:, 0], 3:5], rois[pairs[:, 1], 3:5])), 1).contiguous()
ref = _hip.roi_align_fwd(fmap, rois, 7, 7, 1 / 16, True)
uref = _hip.roi_align_fwd(fmap, urois, 7, 7, 1 / 16, True)
torch.cuda.synchronize()
side = torch.cuda.Stream()
A = torch.randn(4096, 4096, device=dev)
Bm = torch.randn(4096, 4096, device=dev)
big = torch.randn(380, 25088, device=dev)
W6 = torch.randn(4096, 25088, device=dev) * 0.01
z = torch.randn(380, 7, 7, 256, device=dev)
w4 = torch.randn(512, 256, 3, 3, device=dev) * 0.01
wt4 = _hip.conv3x3_pack_weight(w4, False)
b4 = torch.zeros(512, device=dev)
rects = torch.rand(380, 27, 27, 2, device=dev)
img = torch.randn(1, 3, 592, 592, device=dev)


def main_work(kind):
    if kind == 'nothing':
        return None
    if kind == 'roi_align(union)':
        return _hip.roi_align_fwd(fmap, urois, 7, 7, 1 / 16, True)
    if kind == 'torch matmul':
        return A @ Bm
    if kind == 'gemm_inloop fc6':
        return _hip.gemm_inloop(big, W6, False, True)
    if kind == 'gemm (planes) 4096^3':
        return _hip.gemm(A, Bm, False, True)
    if kind == 'conv3x3_nhwc tower':
        return _hip.conv3x3_nhwc(z, wt4, b4, 1)
    if kind == 'im2col':
        return _hip.im2col_nhwc(rects, 7, 7, 2, 3, ldo=100)[0]
    if kind == 'torch elementwise':
        return (A * 2 + 1).tanh()
    if kind == 'torch fill/zeros':
        return [torch.zeros(1 << 20, device=dev) for _ in range(8)]
    raise ValueError(kind)


for kind in ('nothing', 'roi_align(union)', 'torch matmul', 'gemm_inloop fc6', 'gemm (planes) 4096^3', 'conv3x3_nhwc tower', 'im2col',
             'torch elementwise', 'torch fill/zeros'):
    bad_side = bad_main = 0
    worst = 0.0
    for trial in range(30):
        torch.cuda.synchronize()
        side.wait_stream(torch.cuda.current_stream())
        keep = [main_work(kind) for _ in range(3)]
        with torch.cuda.stream(side):
            outs = [_hip.roi_align_fwd(fmap, rois, 7, 7, 1 / 16, True) for _ in range(3)]
        keep2 = [main_work(kind) for _ in range(3)]
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, ref):
                bad_side += 1
                worst = max(worst, float((o - ref).abs().max()))
        if kind == 'roi_align(union)':
            for o in keep + keep2:
                if not torch.equal(o, uref):
                    bad_main += 1
    print('PAIR side=roi_align(objects) main=%-24s wrong side outputs %d of 90 (worst %.3e), wrong main outputs %d' % (kind, bad_side, worst, bad_main), flush=True)
