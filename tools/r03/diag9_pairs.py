"""Co-residency matrix: each VICTIM op runs on a side stream while an AGGRESSOR runs on the main stream; the victim's result is
compared bit by bit with what it returns alone.  Run once per library build (MOTIFS_HIP_LIB): the default build has no packed
FP32 VALU instructions, csrc/_variants/pk is the compiler's default code generation."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'neural-motifs_amd')):
    sys.path.insert(0, p)
import torch
from lib import _hip

tag = sys.argv[1] if len(sys.argv) > 1 else 'default'
torch.manual_seed(0)
dev = 'cuda'
fmap = torch.randn(1, 37, 37, 512, device=dev).relu_()
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
small = torch.randn(20, 4096, device=dev)
W7 = torch.randn(4096, 4096, device=dev) * 0.01
zs = torch.randn(40, 7, 7, 256, device=dev)
y0 = torch.randn(40, 14, 14, 256, device=dev).relu_()
mean, invstd, gam, bet = torch.randn(256, device=dev) * 0.1, torch.rand(256, device=dev) + 0.5, torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1
y1 = torch.randn(40, 7, 7, 512, device=dev).relu_()
m2, i2, g2, be2 = torch.randn(512, device=dev) * 0.1, torch.rand(512, device=dev) + 0.5, torch.rand(512, device=dev) + 0.5, torch.randn(512, device=dev) * 0.1
res = torch.randn(40, 512, 7, 7, device=dev)
aimg = _hip.act_planes(z[:64].contiguous(), torch.full((64,), 0x40c00000, dtype=torch.int32, device=dev))
pk4 = _hip.plconv_pack_weight(w4, False)

victims = {
    'roi_align_fwd nhwc': lambda: _hip.roi_align_fwd(fmap, rois, 7, 7, 1 / 16, True),
    'gemm_inloop fc7 (20 rows)': lambda: _hip.gemm_inloop(small, W7, False, True),
    'conv3x3_nhwc (40 patches)': lambda: _hip.conv3x3_nhwc(zs, wt4, b4, 1),
    'bn_pool_fwd': lambda: _hip.bn_pool_fwd(y0, mean, invstd, gam, bet)[0],
    'bn_residual_nchw': lambda: _hip.bn_residual_nchw(y1, m2, i2, g2, be2, res),
    'plconv3x3 (64 patches)': lambda: _hip.plconv3x3(aimg, pk4, 512, b4, 1),
    'plane gemm 2048^3': lambda: _hip.gemm(A, A, False, True),
    'torch tanh': lambda: (A * 0.5).tanh(),
    'torch matmul 2048^3': lambda: A @ A,
    'torch softmax': lambda: torch.softmax(A, 1),
}
aggressors = {
    'conv3x3_nhwc (in-loop split)': lambda: _hip.conv3x3_nhwc(z, wt4, b4, 1),
    'gemm_inloop fc6': lambda: _hip.gemm_inloop(big, W6, False, True),
    'plane gemm 2048^3': lambda: _hip.gemm(A, A, False, True),
    'torch matmul 2048^3': lambda: A @ A,
}
only = os.environ.get('DIAG9_VICTIMS')
for an, af in aggressors.items():
    for vn, vf in victims.items():
        if only and vn not in only.split(','):
            continue
        ref = vf()
        ref2 = vf()
        torch.cuda.synchronize()
        det = torch.equal(ref, ref2)
        bad, worst, nel = 0, 0.0, 0
        for trial in range(15):
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
        print('PAIR lib=%-8s aggressor=%-30s victim=%-26s wrong %2d of 45 (worst %.3e of max %.3g, up to %d elements)%s' % (
            tag, an, vn, bad, worst, float(ref.float().abs().max()), nel, '' if det else '  [victim not run-to-run deterministic alone]'), flush=True)
