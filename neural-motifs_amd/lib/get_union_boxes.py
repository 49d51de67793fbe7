"""
Union-box visual features (reference lib/get_union_boxes.py:15-93): RoIAlign of the union rectangle of every
(subject, object) pair + a small conv tower over two 27x27 soft masks of the boxes drawn in union coordinates.

MI355X data flow: the masks are rasterised on the device straight into NHWC (no rois->CPU->Cython->GPU round trip,
reference :47-50), the tower runs in NHWC on the MFMA GEMM / implicit-GEMM conv kernels, and the result is returned
as a logical [N,C,7,7] tensor (channels_last memory) so that `Flattener` + fc6 see the reference's (c,y,x) order.
State-dict keys are the reference's: conv.{0,2,4,6}.*
"""
import torch
from torch import nn
from torch.nn import functional as F

from config import BATCHNORM_MOMENTUM
from lib.draw_rectangles.draw_rectangles import draw_union_boxes
from lib.fpn.roi_align.functions.roi_align import RoIAlignFunction
from lib.hip_ops import Conv2dNHWC, ReLU


class _ChannelsLastBN(nn.BatchNorm2d):
    """BatchNorm2d applied to an NHWC tensor (statistics per last-dim channel)."""

    def forward(self, x_nhwc):
        y = super(_ChannelsLastBN, self).forward(x_nhwc.permute(0, 3, 1, 2))
        return y.permute(0, 2, 3, 1)


class _MaxPoolNHWC(nn.MaxPool2d):
    def forward(self, x_nhwc):
        y = super(_MaxPoolNHWC, self).forward(x_nhwc.permute(0, 3, 1, 2))
        return y.permute(0, 2, 3, 1)


class UnionBoxesAndFeats(nn.Module):
    def __init__(self, pooling_size=7, stride=16, dim=256, concat=False, use_feats=True):
        super(UnionBoxesAndFeats, self).__init__()
        self.pooling_size = pooling_size
        self.stride = stride
        self.dim = dim
        self.use_feats = use_feats
        self.concat = concat
        # same child indices as the reference Sequential: Conv(0) ReLU(1) BN(2) MaxPool(3) Conv(4) ReLU(5) BN(6)
        self.conv = nn.Sequential(
            Conv2dNHWC(2, dim // 2, kernel_size=7, stride=2, padding=3),
            ReLU(),
            _ChannelsLastBN(dim // 2, momentum=BATCHNORM_MOMENTUM),
            _MaxPoolNHWC(kernel_size=3, stride=2, padding=1),
            Conv2dNHWC(dim // 2, dim, kernel_size=3, stride=1, padding=1),
            ReLU(),
            _ChannelsLastBN(dim, momentum=BATCHNORM_MOMENTUM),
        )

    def forward(self, fmap, rois, union_inds):
        union_pools = union_boxes(fmap, rois, union_inds, pooling_size=self.pooling_size, stride=self.stride)
        if not self.use_feats:
            return union_pools.detach()
        pair_rois = torch.cat((rois[:, 1:][union_inds[:, 0]], rois[:, 1:][union_inds[:, 1]]), 1).detach()
        rects = draw_union_boxes(pair_rois, self.pooling_size * 4 - 1, offset=-0.5, channels_last=True)   # [N,27,27,2]
        tower = self.conv(rects).permute(0, 3, 1, 2)                      # logical NCHW
        if self.concat:
            return torch.cat((union_pools, tower), 1)
        return union_pools + tower


def union_boxes(fmap, rois, union_inds, pooling_size=14, stride=16):
    """RoIAlign over the union rectangle of each pair (reference :72-93)"""
    assert union_inds.size(1) == 2
    im_inds = rois[:, 0][union_inds[:, 0]]
    union_rois = torch.cat((
        im_inds[:, None],
        torch.min(rois[:, 1:3][union_inds[:, 0]], rois[:, 1:3][union_inds[:, 1]]),
        torch.max(rois[:, 3:5][union_inds[:, 0]], rois[:, 3:5][union_inds[:, 1]]),
    ), 1)
    return RoIAlignFunction(pooling_size, pooling_size, spatial_scale=1 / stride)(fmap, union_rois)
