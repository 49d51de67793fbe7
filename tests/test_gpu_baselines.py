"""GPU runs of the pieces SURVEY.md section 8f lists that round 1 only exercised on the CPU shim:
  * the baseline branches of the context module (nl_obj = 0 / nl_edge = 0, reference lib/rel_model.py:259-296, :500-503),
  * the message-passing baseline RelModelStanford (reference lib/rel_model_stanford.py:20-156),
both against the oracle on identical inputs; and the detector driver's train + VALIDATION epoch end to end
(models/train_detector.py:78-181: detections of the eval forward -> COCO-protocol box mAP -> ReduceLROnPlateau), including
the stale-weight hazard the validation forward used to have (packed conv weights cached across FusedClipSGD steps)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _close(got, ref, what, tol=1e-4):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    scale = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(got - ref).max())
    print('%-40s max|ref| = %9.4f  max abs err = %.3e  (%.2e of scale)' % (what, scale, err, err / scale))
    assert err <= tol * scale, what


# (nl_obj = 0 with nl_edge > 0 cannot run in the reference either: its edge LSTM is built for embed_dim + hidden inputs and
# is fed the 4424-d object representation, lib/rel_model.py:126-136 vs :284-295 -- the product raises, see below)
@pytest.mark.parametrize('nl_obj,nl_edge,mode', [(0, 0, 'sgcls'), (2, 0, 'sgcls'), (0, 0, 'predcls')])
def test_baseline_context_variants_on_the_gpu(nl_obj, nl_edge, mode):
    if not torch.cuda.is_available():
        pytest.fail('needs a HIP device')
    from dataloaders.synthetic import SyntheticVG, make_blob
    from lib import rng
    from lib.rel_model import RelModel
    from oracle import model as OM
    torch.manual_seed(1)
    ds = SyntheticVG(num_images=2, seed=4, n_boxes=6, n_rels=8, im_size=320)
    kw = dict(hidden_dim=128, pooling_dim=4096, nl_obj=nl_obj, nl_edge=nl_edge, order='leftright', rec_dropout=0.1,
              use_bias=True, pass_in_obj_feats_to_decoder=False, pass_in_obj_feats_to_edge=False, use_tanh=False,
              limit_vision=False)
    model = RelModel(classes=ds.ind_to_classes, rel_classes=ds.ind_to_predicates, mode=mode, num_gpus=1, **kw)
    for _, p in model.detector.named_parameters():
        p.requires_grad = False
    if hasattr(model, 'post_emb'):
        model.post_emb.weight.data.mul_(0.2)
    model.post_lstm.weight.data.mul_(0.1)                  # logits O(10): see tests/test_gpu_configs.py
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.cuda().train()
    blob = make_blob(ds, [0, 1], is_train=True)
    a = blob[0]
    model.sampler_rs = np.random.RandomState(2)
    rng.use_host_rng(13)
    res = model[blob]
    rng.use_host_rng(None)
    trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    params = {k: v.clone().requires_grad_(k in trainable) for k, v in sd.items()}
    out = OM.relmodel_forward(params, dict(kw, mode=mode), a[0], a[1], 0, a[3], a[4], True, OM.HostRNG(13),
                              rel_labels=res.rel_labels.cpu())
    np.testing.assert_array_equal(res.obj_preds.cpu().numpy(), out['obj_preds'].numpy())
    _close(res.rm_obj_dists.detach().cpu().numpy(), out['rm_obj_dists'].detach().numpy(), 'nl_obj=%d nl_edge=%d object logits' % (nl_obj, nl_edge))
    _close(res.rel_dists.detach().cpu().numpy(), out['rel_dists'].detach().numpy(), 'nl_obj=%d nl_edge=%d relation logits' % (nl_obj, nl_edge))
    loss = F.cross_entropy(res.rel_dists, res.rel_labels[:, -1])
    loss_ref = F.cross_entropy(out['rel_dists'], out['rel_labels'][:, -1])
    if mode != 'predcls':
        loss = loss + F.cross_entropy(res.rm_obj_dists, res.rm_obj_labels)
        loss_ref = loss_ref + F.cross_entropy(out['rm_obj_dists'], out['rm_obj_labels'])
    loss.backward()
    loss_ref.backward()
    _close(loss.item(), loss_ref.item(), 'loss')
    n = 0
    for name, p in model.named_parameters():
        if p.requires_grad and p.grad is not None and params[name].grad is not None:
            _close(p.grad.cpu().numpy(), params[name].grad.numpy(), 'grad ' + name[-30:], tol=2e-4)
            n += 1
    assert n >= 10


def test_mismatched_lstm_input_is_refused_not_read_out_of_bounds():
    """nl_obj = 0 with nl_edge = 2 feeds the edge LSTM a 4424+200-d input its weights were not built for (a reference
    defect): the binding must refuse instead of letting the kernels read past the parameter vector"""
    if not torch.cuda.is_available():
        pytest.fail('needs a HIP device')
    from lib import _hip
    from lib.lstm.highway_lstm_cuda.alternating_highway_lstm import AlternatingHighwayLSTM
    from torch.nn.utils.rnn import PackedSequence
    lstm = AlternatingHighwayLSTM(input_size=40, hidden_size=32, num_layers=2).cuda()
    x = PackedSequence(torch.randn(5, 48, device='cuda'), torch.tensor([2, 2, 1]))
    with pytest.raises(_hip.HipKernelError, match='do not match the input'):
        lstm(x)


def test_message_passing_baseline_on_the_gpu():
    if not torch.cuda.is_available():
        pytest.fail('needs a HIP device')
    from dataloaders.synthetic import SyntheticVG, make_blob
    from lib import rng
    from lib.rel_model_stanford import RelModelStanford
    from oracle import model as OM
    torch.manual_seed(2)
    ds = SyntheticVG(num_images=2, seed=6, n_boxes=6, n_rels=8, im_size=320)
    model = RelModelStanford(classes=ds.ind_to_classes, rel_classes=ds.ind_to_predicates, mode='sgcls', num_gpus=1)
    for _, p in model.detector.named_parameters():
        p.requires_grad = False
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.cuda().train()
    blob = make_blob(ds, [0, 1], is_train=True)
    a = blob[0]
    model.sampler_rs = np.random.RandomState(3)
    rng.use_host_rng(21)
    res = model[blob]
    rng.use_host_rng(None)
    trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    params = {k: v.clone().requires_grad_(k in trainable) for k, v in sd.items()}
    out = OM.stanford_forward_train(params, dict(mode='sgcls', require_overlap=False), a[0], a[1], 0, a[3], a[4],
                                    OM.HostRNG(21), res.rel_labels.cpu())
    _close(res.rm_obj_dists.detach().cpu().numpy(), out['rm_obj_dists'].detach().numpy(), 'message passing: object logits')
    _close(res.rel_dists.detach().cpu().numpy(), out['rel_dists'].detach().numpy(), 'message passing: relation logits')
    loss = F.cross_entropy(res.rm_obj_dists, res.rm_obj_labels) + F.cross_entropy(res.rel_dists, res.rel_labels[:, -1])
    loss_ref = F.cross_entropy(out['rm_obj_dists'], out['rm_obj_labels']) + F.cross_entropy(out['rel_dists'], res.rel_labels.cpu()[:, -1])
    loss.backward()
    loss_ref.backward()
    _close(loss.item(), loss_ref.item(), 'message passing: loss')
    n = 0
    for name, p in model.named_parameters():
        if p.requires_grad and p.grad is not None and params[name].grad is not None:
            _close(p.grad.cpu().numpy(), params[name].grad.numpy(), 'grad ' + name[-30:], tol=2e-4)
            n += 1
    assert n >= 15
    model.eval()
    with torch.no_grad():
        tup = model[make_blob(ds, [1], is_train=False)]
    assert len(tup) == 5 and tup[3].shape[1] == 2 and tup[4].shape[1] == 51


@pytest.mark.parametrize('trunk', ['vgg', 'resnet'])
def test_detector_driver_trains_and_validates(trunk):
    """models/train_detector.py end to end on the GPU: 2 training batches, then the validation epoch (eval forward with
    the freshly updated weights -> detections -> box mAP -> scheduler) -- as a subprocess, the way a user runs it; with the
    VGG16 trunk and (since round 5: lib/resnet.py trains) with `-resnet`"""
    if not torch.cuda.is_available():
        pytest.fail('needs a HIP device')
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, 'neural-motifs_amd'))
    cmd = [sys.executable, os.path.join(ROOT, 'neural-motifs_amd', 'models', 'train_detector.py'), '-b', '2', '-nepoch', '1',
           '-max_iters', '2', '-val_size', '4', '-synthetic', '12', '-p', '1', '-lr', '1e-3'] + (['-resnet'] if trunk == 'resnet' else [])
    r = subprocess.run(cmd, env=env, cwd=os.path.join(ROOT, 'neural-motifs_amd'), capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    print(tail)
    assert r.returncode == 0, tail
    assert 'overall' in r.stdout and ('Average Precision' in r.stdout or 'No detections anywhere' in r.stdout)


def test_relation_drivers_train_and_evaluate_sgdet_from_a_detector_checkpoint(tmp_path):
    """models/train_rels.py -m sgdet end to end on the GPU, as a subprocess, the way a user runs it: a detector checkpoint
    (reference-format file name, `vgdet/vg-N.tar`; a random detector made confident like tests/test_gpu_sgdet.py does, so that it
    detects something) -> 3 training batches with the frozen detector stage two batches ahead of the step, the loop left early by
    -max_iters (stages in flight are discarded) -> the validation epoch (SGDet evaluation with the stage ahead) -> Recall@K"""
    if not torch.cuda.is_available():
        pytest.fail('needs a HIP device')
    from lib.object_detector import ObjectDetector
    from dataloaders.synthetic import SyntheticVG
    torch.manual_seed(4)
    ds = SyntheticVG(num_images=2, seed=1)
    det = ObjectDetector(classes=ds.ind_to_classes, mode='refinerels')
    with torch.no_grad():
        det.score_fc.weight.mul_(30.0)
        det.rpn_head.conv[2].weight.mul_(4.0)
    os.makedirs(str(tmp_path / 'vgdet'))
    ckpt = str(tmp_path / 'vgdet' / 'vg-0.tar')
    torch.save({'epoch': 0, 'state_dict': det.state_dict()}, ckpt)
    del det
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, 'neural-motifs_amd'), MOTIFS_DETECT_AHEAD='2')
    cmd = [sys.executable, os.path.join(ROOT, 'neural-motifs_amd', 'models', 'train_rels.py'), '-m', 'sgdet', '-model', 'motifnet',
           '-order', 'leftright', '-nl_obj', '1', '-nl_edge', '1', '-b', '2', '-nepoch', '1', '-max_iters', '3', '-val_size', '3',
           '-synthetic', '14', '-p', '1', '-lr', '1e-3', '-hidden_dim', '128', '-pooling_dim', '4096', '-use_bias', '-clip', '5',
           '-ngpu', '1', '-ckpt', ckpt, '-save_dir', str(tmp_path / 'rel')]
    r = subprocess.run(cmd, env=env, cwd=os.path.join(ROOT, 'neural-motifs_amd'), capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    print(tail)
    assert r.returncode == 0, tail
    assert 'detector stage: 2 batch(es) ahead' in r.stdout
    assert 'overall' in r.stdout and 'R@100' in r.stdout
    # ... and models/eval_rels.py on the relation checkpoint it wrote (one image per step, the detector stage of the next two ahead)
    rel = str(tmp_path / 'rel' / 'vgrel-0.tar')
    assert os.path.exists(rel)
    cmd = [sys.executable, os.path.join(ROOT, 'neural-motifs_amd', 'models', 'eval_rels.py'), '-m', 'sgdet', '-model', 'motifnet',
           '-order', 'leftright', '-nl_obj', '1', '-nl_edge', '1', '-b', '1', '-val_size', '4', '-synthetic', '14',
           '-hidden_dim', '128', '-pooling_dim', '4096', '-use_bias', '-ngpu', '1', '-ckpt', rel]
    r = subprocess.run(cmd, env=env, cwd=os.path.join(ROOT, 'neural-motifs_amd'), capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    print(tail)
    assert r.returncode == 0, tail
    assert 'R@100' in r.stdout
