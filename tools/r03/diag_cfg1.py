"""Diagnosis of the run-to-run varying relation-logit error of the cfg1 evaluation (gpurun r03_c5 / c6): determinism of two
identical forwards, one HIP stream vs two, and the relation head stage by stage on the PRODUCT's own inputs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'neural-motifs_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import torch
import test_gpu_configs as T
from dataloaders.synthetic import make_blob
from lib import _hip
from oracle import model as OM

ds, model, sd = T.build('predcls', 1234 + 100, 4)
model.cuda().eval()
csd = T.calibrated(sd)
model.load_state_dict(csd)
cfg = dict(T.MODEL_KW, mode='predcls', return_logits=True)


def err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return '%.3e (max|ref| %.3g)' % (np.abs(a - b).max(), np.abs(b).max())


for idx in (1, 2):
    blob = make_blob(ds, [idx], is_train=False)
    a = blob[0]
    with torch.no_grad():
        ref, rl = OM.relmodel_forward({k: v.clone() for k, v in csd.items()}, cfg, a[0], a[1], 0, a[3], a[4], False, OM.HostRNG(0))
        outs = {}
        for overlap in (True, True, False, False):
            model.overlap_streams = overlap
            model[blob]
            torch.cuda.synchronize()
            last = model.last_eval_result
            r = last.rel_dists.clone()
            key = 'overlap' if overlap else 'single stream'
            print('img %d %-13s relation logits vs oracle %s%s' % (idx, key, err(r.cpu().numpy(), rl['rel_dists'].numpy()),
                  '' if key not in outs else '   bitwise equal to the previous run: %s' % bool(torch.equal(outs[key], r))), flush=True)
            outs[key] = r
        # stage by stage on the product's inputs (single stream)
        fmap = last.fmap.detach()
        rois = torch.cat((last.im_inds[:, None].float(), last.rm_box_priors), 1)
        rel_inds = model.get_rel_inds(None, last.im_inds, last.rm_box_priors)
        fm_c, rois_c, ri_c = fmap.float().cpu().contiguous(), rois.cpu(), rel_inds.cpu()
        sdc = {k: v.clone() for k, v in csd.items()}
        ub_p = model.union_boxes(fmap, rois, rel_inds[:, 1:])
        ub_o = OM.union_boxes_feats(sdc, fm_c, rois_c, ri_c[:, 1:], False)
        print('img %d   union-box features (RoIAlign + tower)   %s' % (idx, err(ub_p.cpu().numpy(), ub_o.numpy())))
        vr_p = model.roi_fmap(ub_p)
        vr_o = OM.vgg_classifier(sdc, ub_o.view(ub_o.size(0), -1), 'roi_fmap.1.', False, OM.HostRNG(0), use_dropout=False, use_relu=False)
        print('img %d   visual rep (fc6 / fc7)                  %s' % (idx, err(vr_p.cpu().numpy(), vr_o.numpy())))
        vr_p2 = model.roi_fmap(ub_o.cuda().contiguous(memory_format=torch.channels_last))
        print('img %d   visual rep on the ORACLE union features  %s' % (idx, err(vr_p2.cpu().numpy(), vr_o.numpy())))
        of_p = model.obj_feature_map(fmap, rois)
        of_o = OM.vgg_classifier(sdc, OM.roi_align(fm_c, rois_c).view(rois_c.size(0), -1), 'roi_fmap_obj.', False, OM.HostRNG(0))
        print('img %d   object features (RoIAlign + fc6 / fc7)  %s' % (idx, err(of_p.cpu().numpy(), of_o.numpy())))
        # RoIAlign: NHWC kernel (rewritten this round) vs the one-thread-per-output NCHW kernel, on the real union boxes
        ur = torch.cat((rois[:, :1][rel_inds[:, 1]], torch.min(rois[:, 1:3][rel_inds[:, 1]], rois[:, 1:3][rel_inds[:, 2]]),
                        torch.max(rois[:, 3:5][rel_inds[:, 1]], rois[:, 3:5][rel_inds[:, 2]])), 1).contiguous()
        x_nhwc = fmap.permute(0, 2, 3, 1).contiguous()
        p1 = _hip.roi_align_fwd(x_nhwc, ur, 7, 7, 1.0 / 16, True)
        p2 = _hip.roi_align_fwd(fmap.float().contiguous(), ur, 7, 7, 1.0 / 16, False)
        print('img %d   RoIAlign NHWC kernel == NCHW kernel on %d union boxes: %s' % (idx, ur.shape[0], bool(torch.equal(p1, p2))))
