"""Bisect of the two-stream evaluation race: run the overlap-vs-single-stream comparison of diag2_cfg1.py under one
configuration (environment set by the caller) and print one summary line."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'neural-motifs_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import test_gpu_configs as T
from dataloaders.synthetic import make_blob

tag = sys.argv[1]
ds, model, sd = T.build('predcls', 1234 + 100, 4)
model.cuda().eval()
model.load_state_dict(T.calibrated(sd))
worst, nbad, n = 0.0, 0, 0
for idx in (1, 2):
    blob = make_blob(ds, [idx], is_train=False)
    with torch.no_grad():
        model.overlap_streams = False
        model[blob]
        torch.cuda.synchronize()
        base_rel = model.last_eval_result.rel_dists.clone()
        for trial in range(4):
            model.overlap_streams = True
            model[blob]
            torch.cuda.synchronize()
            d = float((model.last_eval_result.rel_dists - base_rel).abs().max())
            worst, nbad, n = max(worst, d), nbad + (d > 0), n + 1
print('VARIANT %-28s %d of %d overlapped runs differ from the single-stream run, worst %.3e' % (tag, nbad, n, worst), flush=True)
