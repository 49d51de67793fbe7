"""What do the wrong RoIAlign outputs look like?  (aggressor: round-2 conv on the main stream, victim: roi_align_fwd nhwc on a side stream)"""
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
side = torch.cuda.Stream()
z = torch.randn(380, 7, 7, 256, device=dev)
w4 = torch.randn(512, 256, 3, 3, device=dev) * 0.01
wt4 = _hip.conv3x3_pack_weight(w4, False)
b4 = torch.zeros(512, device=dev)
ref = _hip.roi_align_fwd(fmap, rois, 7, 7, 1 / 16, True)
torch.cuda.synchronize()
shown = 0
for trial in range(20):
    torch.cuda.synchronize()
    side.wait_stream(torch.cuda.current_stream())
    keep = [_hip.conv3x3_nhwc(z, wt4, b4, 1) for _ in range(2)]
    with torch.cuda.stream(side):
        out = _hip.roi_align_fwd(fmap, rois, 7, 7, 1 / 16, True)
    keep2 = [_hip.conv3x3_nhwc(z, wt4, b4, 1) for _ in range(2)]
    torch.cuda.synchronize()
    bad = (out != ref)
    if not bool(bad.any()) or shown >= 4:
        continue
    shown += 1
    o, r, b = out.cpu(), ref.cpu(), bad.cpu()
    idx = b.nonzero()
    blocks = {}
    for (ni, ci, y, x) in idx.tolist():
        blocks.setdefault((ni, ci // 64), []).append((ci % 64, y * 7 + x))
    print('TRIAL %d: %d wrong elements in %d (roi, chunk) blocks of 160' % (trial, idx.shape[0], len(blocks)))
    for (ni, ch), lst in sorted(blocks.items())[:6]:
        chans = sorted(set(c for c, _ in lst))
        bins = sorted(set(bb for _, bb in lst))
        ci, bi = lst[0]
        gv, rv = float(o[ni, ch * 64 + ci].view(-1)[bi]), float(r[ni, ch * 64 + ci].view(-1)[bi])
        # is the wrong value some OTHER element of the reference block (a transposition / staging mix-up) or of another roi?
        blk = r[ni, ch * 64:(ch + 1) * 64].reshape(-1)
        where_blk = (blk == gv).nonzero().view(-1).tolist()[:4]
        where_all = (r.view(-1) == gv).nonzero().view(-1).tolist()[:4]
        print('   roi %2d chunk %d: %4d wrong; channels %s%s  bins %s%s; e.g. ch %d bin %d got %.6f expected %.6f; got-value found in own block at %s, anywhere in ref at %s, zero=%s'
              % (ni, ch, len(lst), chans[:10], '...' if len(chans) > 10 else '', bins[:12], '...' if len(bins) > 12 else '', ci, bi, gv, rv,
                 [(w // 49, w % 49) for w in where_blk], where_all, gv == 0.0))
