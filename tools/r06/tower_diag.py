"""round 6: the matrix-core tower conv1 against the direct (VALU) kernels and float64 on the data of the tower test"""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'neural-motifs_amd')]
from lib import _hip as hip
torch.manual_seed(5)
N, C0 = 96, 256
rects = torch.rand(N, 27, 27, 2)
w = (torch.rand(C0, 2, 7, 7) * 2 - 1) / 98 ** 0.5
b = (torch.rand(C0) * 2 - 1) / 98 ** 0.5
pre = F.conv2d(rects.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=2, padding=3).permute(0, 2, 3, 1)
ref = F.relu(pre).float()
xp = hip.tower_conv1_pad(rects.cuda())
wk = w.permute(2, 3, 1, 0).reshape(98, C0).contiguous().cuda()
ys = {}
for mode in ('valu', 'mfma'):
    os.environ['MH_TOWER_CONV1'] = mode
    y = hip.tower_conv1_fwd(xp, wk, b.cuda()).cpu()
    ys[mode] = y
    err = (y.double() - F.relu(pre)).abs()
    print(mode, 'fwd max err', float(err.max()), 'rms err', float(err.pow(2).mean().sqrt()), 'max ref', float(ref.max()),
          'mask flips vs float64', int(((y > 0) != (pre > 0)).sum()), 'of', y.numel())
print('valu vs mfma mask flips', int(((ys['valu'] > 0) != (ys['mfma'] > 0)).sum()))
# gradient with the structure of a BatchNorm backward: zero mean per channel over all pixels, then the ReLU mask
g = torch.randn(N, 14, 14, C0)
g = g - g.mean((0, 1, 2), keepdim=True)
g = g * (ref > 0)
A = F.unfold(F.pad(rects.permute(0, 3, 1, 2).double(), (3, 3, 3, 3)), 7, stride=2)      # [N, 2*49, 196], k = ci*49 + ky*7 + kx
A = A.view(N, 2, 49, 196).permute(0, 3, 2, 1).reshape(N * 196, 98)                        # k = (ky*7+kx)*2 + ci
dref = A.t() @ g.view(-1, C0).double()
absref = A.t().abs() @ g.view(-1, C0).double().abs()
for mode in ('valu', 'mfma'):
    os.environ['MH_TOWER_CONV1'] = mode
    dwk, db = hip.tower_conv1_wgrad(xp, g.cuda())
    e = (dwk.cpu().double() - dref).abs()
    print(mode, 'wgrad max err', float(e.max()), '= %.2e of max|ref| %.3e' % (float(e.max() / dref.abs().max()), float(dref.abs().max())),
          ' = %.2e of sum|terms|' % float((e / absref).max()), 'bias err', float((db.cpu().double() - g.view(-1, C0).double().sum(0)).abs().max()))

# ---- the whole tower node in the three forms of its first convolution
import lib.get_union_boxes as GUB
torch.manual_seed(5)
N = 96
tower = GUB.UnionBoxesAndFeats(pooling_size=7, stride=16, dim=512).cuda().train()
rects = torch.rand(N, 27, 27, 2).cuda()
pools = torch.randn(N, 512, 7, 7).cuda()
gout = torch.randn(N, 512, 7, 7).cuda()
res = {}
for mode, env in (('gemm', 'valu'), ('valu', 'valu'), ('mfma', 'mfma'), ('mfma2', 'mfma')):
    GUB.TOWER_CONV1 = 'gemm' if mode == 'gemm' else 'direct'
    os.environ['MH_TOWER_CONV1'] = env
    for bn in (tower.conv[2], tower.conv[6]):
        bn.reset_running_stats()
    tower.zero_grad(set_to_none=True)
    c = tower.conv
    taps = {}
    GUB.TAPS = taps
    out = GUB._TowerFn.apply(rects, pools, c[0].weight, c[0].bias, c[2].weight, c[2].bias, c[4].weight, c[4].bias,
                             c[6].weight, c[6].bias, c[2], c[6], True)
    GUB.TAPS = None
    out.backward(gout)
    res[mode] = dict(out=out.detach().double().cpu(), taps=taps, **{n: p.grad.detach().double().cpu() for n, p in tower.named_parameters()})
for a, b_ in (('gemm', 'valu'), ('gemm', 'mfma'), ('valu', 'mfma'), ('mfma', 'mfma2')):
    print(a, 'vs', b_, {k: '%.2e' % float((res[a][k] - res[b_][k]).abs().max() / res[a][k].abs().max()) for k in res[a] if k != 'taps'},
          'kink decisions that differ:', {k: int((res[a]['taps'][k] != res[b_]['taps'][k]).sum()) for k in res[a]['taps']})
