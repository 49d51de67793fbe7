"""Which intermediate differs between the two-stream and the single-stream evaluation forward?  (follow-up of diag_cfg1.py)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'neural-motifs_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import test_gpu_configs as T
from dataloaders.synthetic import make_blob

ds, model, sd = T.build('predcls', 1234 + 100, 4)
model.cuda().eval()
model.load_state_dict(T.calibrated(sd))
sites = {'union_boxes': model.union_boxes, 'roi_fmap (vr)': model.roi_fmap, 'roi_fmap_obj': model.roi_fmap_obj, 'context': model.context,
         'post_lstm': model.post_lstm, 'rel_compress': model.rel_compress, 'obj_ctx_rnn': model.context.obj_ctx_rnn,
         'edge_ctx_rnn': model.context.edge_ctx_rnn, 'decoder_rnn': model.context.decoder_rnn, 'pos_embed': model.context.pos_embed,
         'detector.roi_fmap': model.detector.roi_fmap, 'detector.features': model.detector.features}
store = {}


def flat(o):
    if torch.is_tensor(o):
        return [o]
    if hasattr(o, 'data') and torch.is_tensor(getattr(o, 'data', None)):
        return [o.data]
    if isinstance(o, (tuple, list)):
        return [t for x in o for t in flat(x)]
    return []


def hook(name):
    def fn(_m, _i, out):
        store[name] = [t.detach().clone() for t in flat(out)]
    return fn


for n, m in sites.items():
    m.register_forward_hook(hook(n))
for idx in (1, 2):
    blob = make_blob(ds, [idx], is_train=False)
    with torch.no_grad():
        model.overlap_streams = False
        model[blob]
        torch.cuda.synchronize()
        base = {k: v for k, v in store.items()}
        base_rel = model.last_eval_result.rel_dists.clone()
        for trial in range(3):
            store.clear()
            model.overlap_streams = True
            model[blob]
            torch.cuda.synchronize()
            bad = []
            for k in base:
                for i, (a, b) in enumerate(zip(base[k], store.get(k, []))):
                    if a.shape != b.shape or not torch.equal(a, b):
                        d = float((a.float() - b.float()).abs().max()) if a.shape == b.shape else -1
                        bad.append('%s[%d] max diff %.3e of %.3g' % (k, i, d, float(a.float().abs().max())))
            d = float((model.last_eval_result.rel_dists - base_rel).abs().max())
            print('img %d trial %d: relation logits differ by %.3e from the single-stream run; differing sites: %s' % (idx, trial, d, bad or 'none'), flush=True)
