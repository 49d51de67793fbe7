"""the conv3x3 launches of one bench.py step, one dispatch per distinct shape (for a rocprofv3 --pmc FETCH_SIZE WRITE_SIZE pass)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'neural-motifs_amd'))
from lib import _hip
dev = 'cuda'
SHAPES = [(6, 592, 64, 64), (6, 296, 64, 128), (6, 296, 128, 128), (6, 148, 128, 256), (6, 148, 256, 256), (6, 74, 256, 512),
          (6, 74, 512, 512), (6, 37, 512, 512), (1536, 7, 256, 512), (1536, 7, 512, 256)]
for (B, S, ci, co) in SHAPES:
    x = torch.randn(B, S, S, ci, device=dev); wt = _hip.conv3x3_pack_weight(torch.randn(co, ci, 3, 3, device=dev) * 0.05)
    bias = torch.randn(co, device=dev)
    torch.cuda.synchronize()
    _hip.conv3x3_nhwc(x, wt, bias, 1)
    torch.cuda.synchronize()
    del x, wt
