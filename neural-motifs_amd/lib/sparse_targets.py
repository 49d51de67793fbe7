"""
FrequencyBias: log P(predicate | subject class, object class) as an embedding table [151*151, 51]
(reference lib/sparse_targets.py:11-37).  The reference builds the counts by scanning the VG training set at
construction time (`get_counts(must_overlap=True)`, :20); here they are injectable (`fg_matrix` [C,C,P], `bg_matrix`
[C,C] -- the drivers models/train_rels.py / eval_rels.py pass lib.get_dataset_counts.get_counts(train), which is the
reference's scan).  With no counts given the constructor uses a seeded synthetic count tensor (benchmarks /
parity tests on random weights), whatever is on this machine's disk: model construction is deterministic and fast on
every host.  `FrequencyBias.from_dataset()` is the reference's constructor behaviour (scan the VG training split).  The
arithmetic on the counts is the reference's (:20-24).
"""
import os

import numpy as np
import torch
import torch.nn as nn


def synthetic_counts(num_objs=151, num_rels=51, seed=1234):
    rs = np.random.RandomState(seed)
    fg = rs.gamma(0.3, 20.0, size=(num_objs, num_objs, num_rels)).astype(np.int64)
    bg = rs.gamma(0.5, 40.0, size=(num_objs, num_objs)).astype(np.int64)
    return fg, bg


class FrequencyBias(nn.Module):
    def __init__(self, eps=1e-3, fg_matrix=None, bg_matrix=None, num_objs=151, num_rels=51):
        super(FrequencyBias, self).__init__()
        if fg_matrix is None or bg_matrix is None:
            fg_matrix, bg_matrix = synthetic_counts(num_objs, num_rels)
        fg_matrix = np.array(fg_matrix, dtype=np.int64)
        bg_matrix = np.array(bg_matrix, dtype=np.int64) + 1
        fg_matrix[:, :, 0] = bg_matrix
        pred_dist = np.log(fg_matrix / fg_matrix.sum(2)[:, :, None] + eps)
        self.num_objs = pred_dist.shape[0]
        pred_dist = torch.FloatTensor(pred_dist).view(-1, pred_dist.shape[2])
        self.obj_baseline = nn.Embedding(pred_dist.size(0), pred_dist.size(1))
        self.obj_baseline.weight.data = pred_dist

    @classmethod
    def from_dataset(cls, eps=1e-3):
        """the reference's `FrequencyBias()` (lib/sparse_targets.py:17-20): counts from a scan of the VG training split"""
        from dataloaders.visual_genome import VG, VG_SGG_FN
        if not os.path.exists(VG_SGG_FN):
            raise FileNotFoundError('%s: the Visual Genome files are not on this machine' % VG_SGG_FN)
        from lib.get_dataset_counts import get_counts
        fg, bg = get_counts(VG(mode='train', filter_duplicate_rels=False, num_val_im=5000), must_overlap=True)
        return cls(eps=eps, fg_matrix=fg, bg_matrix=bg)

    def index_with_labels(self, labels):
        """labels [n,2] (subject class, object class) -> [n,51]"""
        return self.obj_baseline(labels[:, 0] * self.num_objs + labels[:, 1])

    def forward(self, obj_cands0, obj_cands1):
        joint = obj_cands0[:, :, None] * obj_cands1[:, None]
        return joint.view(joint.size(0), -1) @ self.obj_baseline.weight
