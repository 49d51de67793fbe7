"""
The relation driver's two losses as one autograd node (round 6).

Reference models/train_rels.py:140-141:
    losses['class_loss'] = F.cross_entropy(result.rm_obj_dists, result.rm_obj_labels)
    losses['rel_loss'] = F.cross_entropy(result.rel_dists, result.rel_labels[:, -1])
The framework evaluates each as log_softmax + nll_loss and their two backward kernels plus fills: ~25 launches of a few
microseconds on the main stream between the relation tail and the first product of the backward pass.  `relation_losses`
returns both means as ONE [2] tensor from csrc/exact_ops.hip ce_pair_* (two launches forward, one backward); the script
sums (or weights) that tensor.  MOTIFS_FUSED_LOSS=0, CPU tensors and anything but fp32 logits / int64 labels take the
framework's own functions.
"""
import os

import torch
import torch.nn.functional as F

from lib import _hip

FUSED = os.environ.get('MOTIFS_FUSED_LOSS', '1') != '0'


class _CrossEntropyPairFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits_a, labels_a, logits_b, labels_b):
        la, lb = logits_a.contiguous(), logits_b.contiguous()
        losses, lse = _hip.ce_pair_fwd(la, labels_a, lb, labels_b)
        ctx.save_for_backward(la, labels_a, lb, labels_b, lse)
        return losses

    @staticmethod
    def backward(ctx, g):
        la, labels_a, lb, labels_b, lse = ctx.saved_tensors
        ga, gb = _hip.ce_pair_bwd(la, labels_a, lb, labels_b, lse, g.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[2])
        return ga, None, gb, None


def relation_losses(result):
    """[class_loss, rel_loss] of a training `Result` (mean cross-entropy over the object rows / the sampled relation rows)"""
    a, la = result.rm_obj_dists, result.rm_obj_labels
    b, lb = result.rel_dists, result.rel_labels[:, -1]
    if (FUSED and a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and la.dtype == torch.int64 and lb.dtype == torch.int64
            and a.shape[0] > 0 and b.shape[0] > 0 and la.dim() == 1):
        return _CrossEntropyPairFn.apply(a, la, b, lb)
    return torch.stack((F.cross_entropy(a, la), F.cross_entropy(b, lb)))
