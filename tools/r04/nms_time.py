#!/usr/bin/env python3
"""Time mh_nms (mask kernel + sweep) on RPN-like proposal sets, prefetching sweep vs round 3's chained sweep.

    python tools/r04/nms_time.py            # runs itself twice (MH_NMS_SWEEP unset / =chain), one JSON line each

The variant is read once per process (csrc/exact_ops.hip sweep_chain), hence the two child processes."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def proposals(rs, n, hi=591.0):
    """anchors-like boxes: centres on a 37 x 37 grid (stride 16), a handful of shapes, jittered -- dense overlaps as after the RPN"""
    import numpy as np
    cx = rs.randint(0, 37, n) * 16 + rs.uniform(-6, 6, n)
    cy = rs.randint(0, 37, n) * 16 + rs.uniform(-6, 6, n)
    s = rs.choice([32, 64, 128, 256, 384], n) * rs.uniform(0.8, 1.25, n)
    a = rs.choice([0.5, 1.0, 2.0], n)
    w, h = s * np.sqrt(a), s / np.sqrt(a)
    return np.stack([np.clip(cx - w / 2, 0, hi), np.clip(cy - h / 2, 0, hi), np.clip(cx + w / 2, 0, hi), np.clip(cy + h / 2, 0, hi)],
                    1).astype(np.float32)


def child():
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'neural-motifs_amd'))
    from lib import _hip
    _hip.lib()
    out = {'sweep': os.environ.get('MH_NMS_SWEEP', 'prefetch')}
    for n in (1000, 6000, 12000):
        boxes = torch.as_tensor(proposals(np.random.RandomState(n), n)).cuda()
        keep, num = _hip.nms(boxes, 0.7)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            keep, num = _hip.nms(boxes, 0.7)
        e1.record()
        torch.cuda.synchronize()
        out['n%d' % n] = {'kept': int(num.item()), 'us_per_call': round(e0.elapsed_time(e1) * 1000 / 50, 1),
                          'keep_crc': int(keep[:int(num.item())].to(torch.int64).mul(torch.arange(1, int(num.item()) + 1, device='cuda')).sum().item())}
    print(json.dumps(out))


if __name__ == '__main__':
    if os.environ.get('NMS_TIME_CHILD'):
        child()
    else:
        for v in ('', 'chain'):
            env = dict(os.environ, NMS_TIME_CHILD='1')
            env.pop('MH_NMS_SWEEP', None)
            if v:
                env['MH_NMS_SWEEP'] = v
            subprocess.check_call([sys.executable, os.path.abspath(__file__)], env=env)
