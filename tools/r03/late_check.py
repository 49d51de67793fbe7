"""GPU check of RelModel.late_vr_backward (lib/rel_model.py: _LateBackward) on a tiny model, a few seconds: the two-stream step
with the re-ordered backward == the one-stream plain step (logits bit-equal, gradients to 1e-6 of their maxima)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'neural-motifs_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch
import torch.nn.functional as F
from dataloaders.synthetic import SyntheticVG, make_blob
from lib.rel_model import RelModel

t0 = time.time()
torch.manual_seed(0)
ds = SyntheticVG(num_images=2, seed=5, n_boxes=6, n_rels=8, im_size=224)
model = RelModel(classes=ds.ind_to_classes, rel_classes=ds.ind_to_predicates, mode='sgcls', num_gpus=1, hidden_dim=128, pooling_dim=4096,
                 nl_obj=2, nl_edge=2, order='leftright', rec_dropout=0.0, use_bias=True, pass_in_obj_feats_to_decoder=False,
                 pass_in_obj_feats_to_edge=False, use_tanh=False, limit_vision=False)
for _, p in model.detector.named_parameters():
    p.requires_grad = False
model.cuda().train()
for m in model.modules():
    if m.__class__.__name__ in ('Dropout', 'AlphaDropout'):
        m.eval()
blob = make_blob(ds, [0, 1], is_train=True)


def step(overlap, late):
    model.overlap_streams, model.late_vr_backward = overlap, late
    model.zero_grad(set_to_none=True)
    model.sampler_rs = np.random.RandomState(9)
    res = model[blob]
    loss = F.cross_entropy(res.rm_obj_dists, res.rm_obj_labels) + F.cross_entropy(res.rel_dists, res.rel_labels[:, -1])
    loss.backward()
    torch.cuda.synchronize()
    return res.rel_dists.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


one = step(False, '0')
worst = 0.0
for trial in range(3):
    two = step(True, 'auto')
    assert torch.equal(one[0], two[0]), 'logits differ'
    assert set(one[1]) == set(two[1])
    for n, g in one[1].items():
        worst = max(worst, float((g - two[1][n]).abs().max()) / (float(g.abs().max()) + 1e-30))
assert worst <= 1e-6, worst
print('LATE_VR_OK worst relative gradient difference %.2e, %.1f s' % (worst, time.time() - t0), flush=True)
