import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'neural-motifs_amd'))
from lib import _hip
dev='cuda'
T, Bb, H, ins, L_ = 20, 6, 512, 4424, 2
x = torch.randn(T, Bb, ins, device=dev)
wtot = sum(6 * H * (ins if l == 0 else H) + 5 * H * H for l in range(L_))
wgt = torch.randn(wtot, device=dev) * 0.02
bias = torch.zeros(5 * H * L_, device=dev)
drop = torch.ones(L_, Bb, H, device=dev)
lengths = [T] * Bb
go = torch.randn(T, Bb, H, device=dev)
import time
for it in range(5):
    torch.cuda.synchronize(); t0=time.time()
    h, c, g = _hip.hwlstm_fwd(x, lengths, wgt, bias, drop, H, L_, True)
    t1=time.time()
    torch.cuda.synchronize(); t2=time.time()
    _hip.hwlstm_bwd(go, x, lengths, wgt, drop, H, L_, h, c, g)
    t3=time.time()
    torch.cuda.synchronize(); t4=time.time()
    print('fwd host %.3f ms total %.3f ms | bwd host %.3f ms total %.3f ms' % ((t1-t0)*1e3,(t2-t0)*1e3,(t3-t2)*1e3,(t4-t2)*1e3))
