"""cfg1 (PredCls evaluation, 4 images) error survey: the numbers tests/test_gpu_configs.py asserts, printed for all four
images without stopping at the first one beyond the bound.  Run once per trunk engine:
    MOTIFS_TRUNK=planes python tools/r03/cfg1_errors.py ; MOTIFS_TRUNK=v2 python tools/r03/cfg1_errors.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'neural-motifs_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import torch
import test_gpu_configs as T

orig = T.report


def report(what, got, ref, abs_tol=None, rel_tol=None):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    scale = float(np.abs(ref).max()) if ref.size else 0.0
    err = np.abs(got - ref)
    print('%-40s trunk=%s max|ref|=%9.4f max abs err=%.3e rms err=%.3e' % (what, os.environ.get('MOTIFS_TRUNK', 'planes'), scale,
                                                                            err.max(), np.sqrt((err ** 2).mean())), flush=True)
    return float(err.max()), scale


T.report = report
ds, model, sd = T.build('predcls', 1234 + 100, 4)
model.cuda().eval()
model.load_state_dict(T.calibrated(sd))
from dataloaders.synthetic import make_blob
from oracle import model as OM
for idx in range(4):
    blob = make_blob(ds, [idx], is_train=False)
    a = blob[0]
    with torch.no_grad():
        model[blob]
        last = model.last_eval_result
        ref, rl = OM.relmodel_forward({k: v.clone() for k, v in T.calibrated(sd).items()}, dict(T.MODEL_KW, mode='predcls', return_logits=True),
                                      a[0], a[1], 0, a[3], a[4], False, OM.HostRNG(0))
        fm = OM.vgg_features({k: v for k, v in sd.items()}, a[0])
    report('cfg1 img %d trunk feature map' % idx, last.fmap.float().cpu().numpy(), fm.numpy())
    report('cfg1 img %d relation logits' % idx, last.rel_dists.cpu().numpy(), rl['rel_dists'].numpy())
