#!/usr/bin/env python3
"""
Generate tests/golden/*.npz from the REFERENCE ITSELF.

Runs only in the build container (needs /root/reference and oracle/_ref, see oracle/Makefile
target `ref`).  The reference is PyTorch-0.3 code; the modules below import and run unmodified
on CPU under today's PyTorch once three absent third-party imports are stubbed (h5py,
overrides, and the CUDA `_ext` packages that are never called here).  Nothing is copied from
the reference: we call its functions and store inputs + outputs.

    python tests/golden/make_golden.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('MOTIFS_REFERENCE', '/root/reference')


def _setup_reference_imports():
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref'))
    sys.argv = ['make_golden']
    for name in ('h5py', 'overrides'):
        sys.modules[name] = types.ModuleType(name)
    sys.modules['overrides'].overrides = lambda f: f
    import bbox as ref_bbox            # compiled from lib/fpn/box_intersections_cpu/bbox.pyx
    import draw_rectangles as ref_dr   # compiled from lib/draw_rectangles/draw_rectangles.pyx
    pkg = types.ModuleType('lib.fpn.box_intersections_cpu')
    pkg.__path__ = []
    sys.modules['lib.fpn.box_intersections_cpu'] = pkg
    sys.modules['lib.fpn.box_intersections_cpu.bbox'] = ref_bbox
    ext = types.ModuleType('lib.lstm.highway_lstm_cuda._ext')
    ext.__path__ = []
    ext.highway_lstm_layer = types.ModuleType('highway_lstm_layer')
    sys.modules['lib.lstm.highway_lstm_cuda._ext'] = ext
    sys.modules['lib.lstm.highway_lstm_cuda._ext.highway_lstm_layer'] = ext.highway_lstm_layer
    if not hasattr(torch.nn.init, 'orthogonal'):
        torch.nn.init.orthogonal = torch.nn.init.orthogonal_
    return ref_bbox, ref_dr


def rand_boxes(rs, n, lo=0.0, hi=591.0, integer=False):
    x1 = rs.uniform(lo, hi - 20, n)
    y1 = rs.uniform(lo, hi - 20, n)
    w = rs.uniform(1, 300, n)
    h = rs.uniform(1, 300, n)
    b = np.stack([x1, y1, np.minimum(x1 + w, hi), np.minimum(y1 + h, hi)], 1)
    return np.round(b) if integer else b


def main():
    ref_bbox, ref_dr = _setup_reference_imports()
    rs = np.random.RandomState(20240925)
    out = {}

    # ---------------------------------------------------------------- N4 union-box rasteriser
    pairs = np.concatenate([rand_boxes(rs, 24), rand_boxes(rs, 24)], 1).astype(np.float32)
    pairs[0, 4:] = pairs[0, :4]                       # identical boxes
    pairs[1] = [10, 10, 50, 50, 20, 20, 30, 30]       # containment
    pairs[2] = [0, 0, 591, 591, 0, 0, 591, 591]       # full image
    pairs[3] = [5, 7, 5.5, 90, 5, 20, 5.25, 60]       # very thin union (w == 0 raises ZeroDivisionError in the reference)
    pairs[4] = [100.5, 100.25, 180.75, 140.5, 150.125, 90.0, 300.0, 200.0]
    for P in (27, 7):
        masks = ref_dr.draw_union_boxes(pairs, P)
        out['draw_P%d' % P] = dict(pairs=pairs, masks=masks)

    # ---------------------------------------------------------------- N5 float64 IoU
    a, q = rand_boxes(rs, 17, integer=True), rand_boxes(rs, 9, integer=True)
    a[3] = q[2]
    out['bbox64'] = dict(a=a, q=q, overlaps=ref_bbox.bbox_overlaps(a, q),
                         intersections=ref_bbox.bbox_intersections(a, q))

    # ---------------------------------------------------------------- anchors + constants
    config = importlib.import_module('config')
    ga = importlib.import_module('lib.fpn.generate_anchors')
    anchors = ga.generate_anchors(base_size=config.ANCHOR_SIZE, feat_stride=16,
                                  anchor_scales=config.ANCHOR_SCALES, anchor_ratios=config.ANCHOR_RATIOS)
    out['anchors'] = dict(shape=np.array(anchors.shape), base=anchors[0, 0], corner=anchors[36, 36],
                          mid=anchors[5, 17], total=np.array([anchors.sum(), np.abs(anchors).sum()]),
                          ratios=np.array(config.ANCHOR_RATIOS), scales=np.array(config.ANCHOR_SCALES),
                          consts=np.array([config.IM_SCALE, config.BOX_SCALE, config.RELS_PER_IMG,
                                           config.REL_FG_FRACTION, config.BATCHNORM_MOMENTUM]))

    # ---------------------------------------------------------------- box codec (torch fp32)
    bu = importlib.import_module('lib.fpn.box_utils')
    boxes = torch.from_numpy(rand_boxes(rs, 40).astype(np.float32))
    deltas = torch.from_numpy((rs.randn(40, 4) * 0.3).astype(np.float32))
    boxes_b = torch.from_numpy(rand_boxes(rs, 13).astype(np.float32))
    per_cls = torch.from_numpy(np.stack([rand_boxes(rs, 5) for _ in range(6)], 0).astype(np.float32))
    out['box_utils'] = dict(
        boxes=boxes.numpy(), deltas=deltas.numpy(), boxes_b=boxes_b.numpy(), per_cls=per_cls.numpy(),
        center_size=bu.center_size(boxes).numpy(), point_form=bu.point_form(bu.center_size(boxes)).numpy(),
        bbox_preds=bu.bbox_preds(boxes, deltas).numpy(),
        bbox_overlaps=bu.bbox_overlaps(boxes, boxes_b).numpy(),
        bbox_intersections=bu.bbox_intersections(boxes, boxes_b).numpy(),
        nms_overlaps=bu.nms_overlaps(per_cls).numpy())

    # ---------------------------------------------------------------- index helpers
    pm = importlib.import_module('lib.pytorch_misc')
    tp = {}
    for k, lengths in enumerate([[5], [4, 3, 1, 1], [6, 6, 6], [20, 17, 17, 9, 2, 1], [3, 2, 2, 2, 1]]):
        inds, lens = pm.transpose_packed_sequence_inds(list(lengths))
        tp['len%d' % k] = np.array(lengths)
        tp['inds%d' % k] = np.asarray(inds)
        tp['lens%d' % k] = np.array(lens)
    im_inds = torch.LongTensor([0, 0, 0, 1, 1, 3, 3, 3, 3, 4])
    tp['ebi_in'] = im_inds.numpy()
    tp['ebi_out'] = np.array(list(pm.enumerate_by_image(im_inds)))
    x = torch.from_numpy(rs.randn(3, 4, 5, 6).astype(np.float32))
    index = torch.LongTensor([[0, 1, 2], [2, 3, 4], [1, 0, 0]])
    tp['gnd_x'], tp['gnd_index'] = x.numpy(), index.numpy()
    tp['gnd_out'] = pm.gather_nd(x, index).numpy()
    vec = torch.LongTensor([3, 0, 7, 7, 1])
    tp['onehot_in'] = vec.numpy()
    tp['onehot_out'] = pm.to_onehot(vec, 9).numpy()
    tp['diag'] = pm.diagonal_inds(torch.zeros(7, 7)).numpy()
    x1 = rs.randint(0, 3, (11, 3))
    x2 = rs.randint(0, 3, (8, 3))
    tp['i2d_x1'], tp['i2d_x2'], tp['i2d_out'] = x1, x2, pm.intersect_2d(x1, x2)
    sc = rs.rand(5, 7)
    tp['asd_in'], tp['asd_out'] = sc, pm.argsort_desc(sc)
    out['misc'] = tp

    # ---------------------------------------------------------------- filter_dets (surgery.py)
    surgery = importlib.import_module('lib.surgery')
    nb, nr = 7, 30
    fb = torch.from_numpy(rand_boxes(rs, nb).astype(np.float32))
    f_scores = torch.from_numpy(rs.rand(nb).astype(np.float32))
    f_cls = torch.from_numpy(rs.randint(1, 151, nb).astype(np.int64))
    f_rel = torch.from_numpy(np.array([(i, j) for i in range(nb) for j in range(nb) if i != j][:nr], dtype=np.int64))
    f_pred = torch.softmax(torch.from_numpy(rs.randn(nr, 51).astype(np.float32)), 1)
    r = surgery.filter_dets(fb, f_scores, f_cls, f_rel, f_pred)
    out['filter_dets'] = dict(boxes=fb.numpy(), obj_scores=f_scores.numpy(), obj_classes=f_cls.numpy(),
                              rel_inds=f_rel.numpy(), pred_scores=f_pred.numpy(),
                              o_boxes=r[0], o_objs=r[1], o_scores=r[2], o_rels=r[3], o_pred=r[4])

    # ---------------------------------------------------------------- Recall@K evaluator
    sg = importlib.import_module('lib.evaluation.sg_eval')
    ev = {}
    for case in range(4):
        n_gt = 6 + case
        gt_boxes = rand_boxes(rs, n_gt, hi=1023, integer=True)
        gt_classes = rs.randint(1, 151, n_gt)
        allp = np.array([(i, j) for i in range(n_gt) for j in range(n_gt) if i != j])
        sel = rs.choice(len(allp), 8, replace=False)
        gt_rels = np.column_stack((allp[sel], rs.randint(1, 51, 8)))
        jitter = rs.uniform(-12, 12, gt_boxes.shape)
        pred_boxes = gt_boxes + jitter
        pred_classes = gt_classes.copy()
        flip = rs.rand(n_gt) < 0.25
        pred_classes[flip] = rs.randint(1, 151, flip.sum())
        obj_scores = rs.rand(n_gt)
        rel_scores = rs.rand(len(allp), 51)
        rel_scores /= rel_scores.sum(1, keepdims=True)
        for gi, (s, o, p) in enumerate(gt_rels[:5]):      # make some GT relations likely hits
            row = np.where((allp[:, 0] == s) & (allp[:, 1] == o))[0][0]
            rel_scores[row, p] += 0.5
        # the reference's eval tuple is sorted by max-pred * subj * obj score (surgery.py:44-49)
        order = np.argsort(-(rel_scores[:, 1:].max(1) * obj_scores[allp[:, 0]] * obj_scores[allp[:, 1]]),
                           kind='stable')
        pred_rel_inds, rel_scores = allp[order], rel_scores[order]
        gt_entry = dict(gt_classes=gt_classes, gt_relations=gt_rels, gt_boxes=gt_boxes)
        pred_entry = dict(pred_boxes=pred_boxes, pred_classes=pred_classes, pred_rel_inds=pred_rel_inds,
                          obj_scores=obj_scores, rel_scores=rel_scores)
        for k, v in list(gt_entry.items()) + list(pred_entry.items()):
            ev['c%d_%s' % (case, k)] = v
        for mode in ('predcls', 'sgcls', 'sgdet'):
            evaluator = sg.BasicSceneGraphEvaluator.all_modes()
            p2g, _, _ = evaluator[mode].evaluate_scene_graph_entry(gt_entry, pred_entry)
            rd = evaluator[mode].result_dict[mode + '_recall']
            ev['c%d_%s_recall' % (case, mode)] = np.array([rd[20][0], rd[50][0], rd[100][0]])
            ev['c%d_%s_nmatch' % (case, mode)] = np.array([len(m) for m in p2g])
    out['sg_eval'] = ev

    # ---------------------------------------------------------------- DecoderRNN (Python LSTM decoder)
    dec_mod = importlib.import_module('lib.lstm.decoder_rnn')
    ah = importlib.import_module('lib.lstm.highway_lstm_cuda.alternating_highway_lstm')
    # torch>=0.4: every Tensor is a Variable, which sends block_orthogonal's
    # `isinstance(tensor, Variable)` unwrapping branch into infinite recursion; neutralise it.
    ah.Variable = type('NeverAVariable', (), {})
    # torch-0.3's PackedSequence was a 2-field namedtuple (data, batch_sizes) with python-int sizes,
    # which is what DecoderRNN.forward destructures (decoder_rnn.py:159-161).
    import collections
    dec_mod.PackedSequence = collections.namedtuple('PackedSequence', ['data', 'batch_sizes'])
    global _packed
    _packed = lambda data, batch_lengths: dec_mod.PackedSequence(data, list(batch_lengths))
    n_cls, H, D = 11, 16, 24
    classes = ['c%d' % i for i in range(n_cls)]
    gen = torch.Generator().manual_seed(7)
    dec_mod.obj_edge_vectors = lambda names, wv_dim=100, **kw: torch.randn(len(names), wv_dim, generator=gen)
    recorded = {}
    orig_mask = dec_mod.get_dropout_mask

    def recording_mask(p, t):
        m = orig_mask(p, t)
        recorded['mask'] = m.detach().clone()
        return m
    dec_mod.get_dropout_mask = recording_mask
    dec = {}
    for tag, p_drop in (('p0', 0.0), ('p2', 0.2)):
        torch.manual_seed(11)
        net = dec_mod.DecoderRNN(classes, embed_dim=100, inputs_dim=D, hidden_dim=H,
                                 recurrent_dropout_probability=p_drop)
        with torch.no_grad():
            net.out.weight.normal_(0, 0.5)
            net.input_linearity.bias.normal_(0, 0.1)
        for k, v in net.state_dict().items():
            dec['%s_param_%s' % (tag, k)] = v.numpy().copy()
        batch_lengths = [3, 3, 2, 1, 1]
        seq = torch.randn(sum(batch_lengths), D)
        labels = torch.LongTensor([1, 4, 0, 2, 0, 9, 3, 10, 5, 0])
        net.train()
        recorded.clear()
        dists, commits = net(_packed(seq, batch_lengths), labels=labels)
        dec[tag + '_train_seq'], dec[tag + '_train_lengths'] = seq.numpy(), np.array(batch_lengths)
        dec[tag + '_train_labels'] = labels.numpy()
        dec[tag + '_train_dists'], dec[tag + '_train_commits'] = dists.detach().numpy(), commits.numpy()
        if 'mask' in recorded:
            dec[tag + '_train_mask'] = recorded['mask'].numpy()
        net.eval()
        T = 6
        seq1 = torch.randn(T, D)
        dists, commits = net(_packed(seq1, [1] * T))
        dec[tag + '_eval_seq'] = seq1.numpy()
        dec[tag + '_eval_dists'], dec[tag + '_eval_commits'] = dists.detach().numpy(), commits.numpy()
        bfn = torch.from_numpy(np.stack([rand_boxes(rs, n_cls, hi=200) for _ in range(T)], 0).astype(np.float32))
        dists, commits = net(_packed(seq1, [1] * T), boxes_for_nms=bfn)
        dec[tag + '_evalnms_boxes'] = bfn.numpy()
        dec[tag + '_evalnms_dists'], dec[tag + '_evalnms_commits'] = dists.detach().numpy(), commits.numpy()
    out['decoder'] = dec

    # ---------------------------------------------------------------- LSTM parameter layout
    torch.manual_seed(3)
    lstm = ah.AlternatingHighwayLSTM(input_size=12, hidden_size=8, num_layers=3)
    out['ahlstm_layout'] = dict(weight=lstm.weight.detach().numpy(), bias=lstm.bias.detach().numpy(),
                                dims=np.array([12, 8, 3]))

    # ---------------------------------------------------------------- detector-training host samplers + box loss
    # anchor_target_layer (lib/fpn/anchor_targets.py:16-105) with the global numpy RNG seeded per case;
    # _sel_inds (proposal_assignments_det.py:94-118) is the sampling core of proposal_assignments_det (the rest of
    # that function needs a CUDA device in the reference: `.cuda(rpn_rois.get_device())`);
    # bbox_loss (lib/fpn/box_utils.py:8-25).
    at = importlib.import_module('lib.fpn.anchor_targets')
    det = {}
    for case, (n_gt, seed) in enumerate([(3, 11), (20, 12), (1, 13)]):
        gtb = rand_boxes(rs, n_gt, integer=True).astype(np.float32)
        gtb[:, 2:] = np.maximum(gtb[:, 2:], gtb[:, :2] + 8)
        np.random.seed(seed)
        anchors, anchor_inds, targets, labels = at.anchor_target_layer(gtb, (592, 592))
        det['at%d_gt' % case], det['at%d_seed' % case] = gtb, np.array(seed)
        det['at%d_anchors' % case], det['at%d_inds' % case] = anchors, anchor_inds
        det['at%d_targets' % case], det['at%d_labels' % case] = targets, labels
    fake = types.ModuleType('lib.pytorch_misc_stub')
    pad = importlib.import_module('lib.fpn.proposal_assignments.proposal_assignments_det')
    for case, (n, seed) in enumerate([(300, 21), (40, 22), (1000, 23)]):
        mo = rs.uniform(0, 1, n) ** 2
        mo[rs.uniform(0, 1, n) < 0.1] = 0.0
        np.random.seed(seed)
        keep, num_fg = pad._sel_inds(mo.copy(), 0.5, 64, 256)
        det['sel%d_overlaps' % case], det['sel%d_seed' % case] = mo, np.array(seed)
        det['sel%d_keep' % case], det['sel%d_numfg' % case] = keep, np.array(num_fg)
    bu = importlib.import_module('lib.fpn.box_utils')
    prior = torch.from_numpy(rand_boxes(rs, 33).astype(np.float32))
    gtbx = torch.from_numpy(rand_boxes(rs, 33).astype(np.float32))
    deltas = torch.randn(33, 4) * 0.5
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        det['bl_loss'] = np.array(float(bu.bbox_loss(prior, deltas, gtbx)))
    det['bl_prior'], det['bl_deltas'], det['bl_gt'] = prior.numpy(), deltas.numpy(), gtbx.numpy()
    out['det_train'] = det

    # ---------------------------------------------------------------- Visual Genome on-disk formats
    # the reference's load_graphs / load_info / VG.__getitem__ geometry on a synthetic VG-SGG-shaped roidb.  h5py,
    # torchvision, pycocotools are absent and dataloaders/blob.py does not parse under Python 3.10 (`async` keyword):
    # all four are stubbed; h5py.File hands back the in-memory arrays (the reference only does roi_h5[key][...]).
    import json
    import tempfile
    from PIL import Image
    n_img = 40
    split = np.array([2 if i % 4 == 3 else 0 for i in range(n_img)], dtype=np.int32)
    first_box, last_box, first_rel, last_rel = [], [], [], []
    boxes_xywh, labels, rel_pairs, rel_preds = [], [], [], []
    for i in range(n_img):
        nb = 0 if i in (5, 17) else rs.randint(2, 9)
        if nb == 0:
            first_box.append(-1); last_box.append(-1); first_rel.append(-1); last_rel.append(-1)
            continue
        b0 = len(boxes_xywh)
        for _ in range(nb):
            w_, h_ = rs.randint(20, 400), rs.randint(20, 400)
            xc, yc = rs.randint(w_ // 2 + 1, 1024 - w_ // 2 - 1), rs.randint(h_ // 2 + 1, 1024 - h_ // 2 - 1)
            boxes_xywh.append((xc, yc, w_, h_)); labels.append(rs.randint(1, 151))
        first_box.append(b0); last_box.append(len(boxes_xywh) - 1)
        nr = 0 if i in (2, 9, 22) else rs.randint(1, 12)
        if nr == 0:
            first_rel.append(-1); last_rel.append(-1)
            continue
        r0 = len(rel_pairs)
        for _ in range(nr):
            a_, b_ = rs.choice(nb, 2, replace=False)
            rel_pairs.append((b0 + a_, b0 + b_)); rel_preds.append(rs.randint(1, 51))
        if nr >= 3:                                         # duplicates of one pair with different predicates
            rel_pairs.append(rel_pairs[r0]); rel_preds.append(rs.randint(1, 51))
        first_rel.append(r0); last_rel.append(len(rel_pairs) - 1)
    roidb = {'split': split, 'img_to_first_box': np.array(first_box, np.int32), 'img_to_last_box': np.array(last_box, np.int32),
             'img_to_first_rel': np.array(first_rel, np.int32), 'img_to_last_rel': np.array(last_rel, np.int32),
             'labels': np.array(labels, np.int32)[:, None], 'boxes_1024': np.array(boxes_xywh, np.int32),
             'relationships': np.array(rel_pairs, np.int32), 'predicates': np.array(rel_preds, np.int32)[:, None]}

    class _FakeH5(dict):
        pass
    h5stub = types.ModuleType('h5py')
    h5stub.File = lambda path, mode='r': _FakeH5({k: v.copy() for k, v in roidb.items()})
    sys.modules['h5py'] = h5stub
    tv = types.ModuleType('torchvision'); tvt = types.ModuleType('torchvision.transforms')
    for nm in ('Resize', 'Compose', 'ToTensor', 'Normalize'):
        setattr(tvt, nm, type(nm, (), {'__init__': lambda self, *a, **k: None}))
    tv.transforms = tvt
    sys.modules['torchvision'], sys.modules['torchvision.transforms'] = tv, tvt
    pc = types.ModuleType('pycocotools'); pcc = types.ModuleType('pycocotools.coco'); pcc.COCO = object
    sys.modules['pycocotools'], sys.modules['pycocotools.coco'] = pc, pcc
    blob_stub = types.ModuleType('dataloaders.blob'); blob_stub.Blob = object
    sys.modules['dataloaders.blob'] = blob_stub
    refvg = importlib.import_module('dataloaders.visual_genome')
    vg = {('in_' + k): v for k, v in roidb.items()}
    cases = [('train', -1, 6, True, True), ('train', -1, 6, True, False), ('val', -1, 6, True, False),
             ('test', -1, 0, True, False), ('train', 20, 4, False, False), ('test', 5, 0, False, False)]
    for ci, (mode, num_im, num_val, fer, fno) in enumerate(cases):
        mask, bxs, cls_, rels = refvg.load_graphs('unused.h5', mode, num_im, num_val_im=num_val, filter_empty_rels=fer,
                                                  filter_non_overlap=fno)
        vg['lg%d_args' % ci] = np.array([{'train': 0, 'val': 1, 'test': 2}[mode], num_im, num_val, int(fer), int(fno)])
        vg['lg%d_mask' % ci] = mask
        vg['lg%d_counts' % ci] = np.array([b.shape[0] for b in bxs] + [-1] + [r.shape[0] for r in rels])
        vg['lg%d_boxes' % ci] = np.concatenate(bxs, 0) if bxs else np.zeros((0, 4))
        vg['lg%d_classes' % ci] = np.concatenate(cls_, 0) if cls_ else np.zeros((0,))
        vg['lg%d_rels' % ci] = np.concatenate(rels, 0) if rels else np.zeros((0, 3))
    with tempfile.TemporaryDirectory() as td:
        info = {'label_to_idx': {'cls%03d' % i: i for i in range(1, 151)},
                'predicate_to_idx': {'pred%02d' % i: i for i in range(1, 51)}}
        jp = os.path.join(td, 'dicts.json')
        json.dump(info, open(jp, 'w'))
        itc, itp = refvg.load_info(jp)
        vg['info_classes'], vg['info_predicates'] = np.array(itc), np.array(itp)
        # __getitem__ geometry: flips, box clipping, im_size, duplicate-relation sampling (numpy draw order)
        sizes = [(640, 480), (375, 500), (512, 512), (800, 333)]
        fns = []
        for i, (w_, h_) in enumerate(sizes):
            fn = os.path.join(td, 'im%d.jpg' % i)
            Image.fromarray(rs.randint(0, 255, (h_, w_, 3)).astype(np.uint8)).save(fn)
            fns.append(fn)
        mask, bxs, cls_, rels = refvg.load_graphs('unused.h5', 'train', -1, num_val_im=0, filter_empty_rels=True)
        for mode in ('train', 'val'):
            ds_ = refvg.VG.__new__(refvg.VG)
            ds_.mode, ds_.filenames = mode, fns
            ds_.gt_boxes, ds_.gt_classes, ds_.relationships = bxs[:4], cls_[:4], rels[:4]
            ds_.filter_duplicate_rels = (mode == 'train')
            ds_.rpn_rois = None
            ds_.transform_pipeline = lambda im: torch.zeros(3, 592, 592)
            for rep in range(3):
                for idx in range(4):
                    np.random.seed(1000 * rep + idx)
                    e = ds_[idx]
                    key = 'gi_%s_%d_%d_' % (mode, rep, idx)
                    vg[key + 'boxes'], vg[key + 'rels'] = e['gt_boxes'], np.asarray(e['gt_relations'])
                    vg[key + 'size'] = np.array(e['img_size'], dtype=np.float64)
                    vg[key + 'flipped'] = np.array(int(bool(e['flipped'])))
        vg['gi_sizes'] = np.array(sizes)
        vg['gi_first4_counts'] = np.array([b.shape[0] for b in bxs[:4]] + [-1] + [r.shape[0] for r in rels[:4]])
        vg['gi_first4_boxes'] = np.concatenate(bxs[:4], 0)
        vg['gi_first4_rels'] = np.concatenate(rels[:4], 0)
    out['vg_formats'] = vg

    # ---------------------------------------------------------------- GloVe text format (lib/word_vectors.py:49-113)
    wvm = importlib.import_module('lib.word_vectors')
    with tempfile.TemporaryDirectory() as td:
        vocab = ['person', 'dog', 'fire', 'hydrant', 'table', 'of', 'on', 'tennis', 'racket', 'caf\u00e9']
        rows = rs.randn(len(vocab), 6)
        with open(os.path.join(td, 'glove.tiny.6d.txt'), 'wb') as f:
            for w_, r_ in zip(vocab, rows):
                f.write(w_.encode('utf-8') + b' ' + b' '.join(('%.5f' % v).encode() for v in r_) + b'\n')
        names = ['__background__', 'person', 'fire hydrant', 'tennis racket', 'on', 'zebra crossing', 'dog']
        torch.manual_seed(0)
        vec = wvm.obj_edge_vectors(names, wv_type='glove.tiny', wv_dir=td, wv_dim=6)
        wv_dict, wv_arr, wv_size = wvm.load_word_vectors(td, 'glove.tiny', 6)
        out['glove'] = dict(txt=np.frombuffer(open(os.path.join(td, 'glove.tiny.6d.txt'), 'rb').read(), dtype=np.uint8),
                            names=np.array(names), vectors=vec.numpy(), arr=wv_arr.numpy(),
                            tokens=np.array([t for t, _ in sorted(wv_dict.items(), key=lambda kv: kv[1])]),
                            known=np.array([1 if (n in wv_dict or sorted(n.split(' '), key=len, reverse=True)[0] in wv_dict)
                                            else 0 for n in names]))

    for name, d in out.items():
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in d.items()})
        print('wrote %-28s %7.1f KB' % (os.path.basename(path), os.path.getsize(path) / 1024.0))


_packed = None


if __name__ == '__main__':
    main()
