"""Kernel-pair race, second round: which kernels are VICTIMS of a co-resident in-loop-split (round-2) conv launch, and does extra
LDS behind the aggressor's tile image (MH_LDS_PAD) change it?  side stream = victim, main stream = aggressor."""
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
fmap_nchw = fmap.permute(0, 3, 1, 2).contiguous()
n = 20
xy = torch.rand(n, 2, device=dev) * 300
wh = torch.rand(n, 2, device=dev) * 250 + 20
rois = torch.cat((torch.zeros(n, 1, device=dev), xy, (xy + wh).clamp(max=591)), 1).contiguous()
side = torch.cuda.Stream()
A = torch.randn(2048, 2048, device=dev)
z = torch.randn(380, 7, 7, 256, device=dev)
w4 = torch.randn(512, 256, 3, 3, device=dev) * 0.01
wt4 = _hip.conv3x3_pack_weight(w4, False)
b4 = torch.zeros(512, device=dev)
big = torch.randn(380, 25088, device=dev)
W6 = torch.randn(4096, 25088, device=dev) * 0.01
rects = torch.rand(380, 27, 27, 2, device=dev)
zimg = _hip.act_planes(z, torch.full((380,), 0x40c00000, dtype=torch.int32, device=dev)) if False else None

victims = {
    'roi_align_fwd nhwc': lambda: _hip.roi_align_fwd(fmap, rois, 7, 7, 1 / 16, True),
    'roi_align_fwd nchw': lambda: _hip.roi_align_fwd(fmap_nchw, rois, 7, 7, 1 / 16, False),
    'torch tanh': lambda: (A * 0.5).tanh(),
    'im2col': lambda: _hip.im2col_nhwc(rects, 7, 7, 2, 3, ldo=100)[0],
    'maxpool2x2_nhwc': lambda: _hip.maxpool2x2_nhwc(fmap[:, :36, :36].contiguous()),
    'plane gemm 2048^3': lambda: _hip.gemm(A, A, False, True),
    'draw_union_boxes': lambda: _hip.draw_union_boxes(torch.cat((rois[:, 1:], rois.flip(0)[:, 1:]), 1).contiguous(), 27, offset=-0.5, channels_last=True),
}
aggressors = {
    'conv3x3_nhwc (in-loop split)': lambda: _hip.conv3x3_nhwc(z, wt4, b4, 1),
    'gemm_inloop fc6': lambda: _hip.gemm_inloop(big, W6, False, True),
}
only = os.environ.get('DIAG7_VICTIMS')
for an, af in aggressors.items():
    for vn, vf in victims.items():
        if only and vn not in only.split(','):
            continue
        ref = vf()
        torch.cuda.synchronize()
        bad, worst, nel = 0, 0.0, 0
        for trial in range(20):
            torch.cuda.synchronize()
            side.wait_stream(torch.cuda.current_stream())
            keep = [af() for _ in range(2)]
            with torch.cuda.stream(side):
                outs = [vf() for _ in range(3)]
            keep2 = [af() for _ in range(2)]
            torch.cuda.synchronize()
            for o in outs:
                if not torch.equal(o, ref):
                    bad += 1
                    worst = max(worst, float((o.float() - ref.float()).abs().max()))
                    nel = max(nel, int((o != ref).sum()))
        print('PAIR lib=%s pad=%s aggressor=%-30s victim=%-20s wrong %2d of 60 (worst %.3e, up to %d elements)' % (
            os.path.basename(os.path.dirname(os.environ.get('MOTIFS_HIP_LIB', 'x/default/y'))), os.environ.get('MH_LDS_PAD', '0'), an, vn, bad, worst, nel), flush=True)
