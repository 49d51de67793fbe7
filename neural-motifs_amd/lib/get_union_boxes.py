"""
Union-box visual features (reference lib/get_union_boxes.py:15-93): RoIAlign of the union rectangle of every
(subject, object) pair + a small conv tower over two 27x27 soft masks of the boxes drawn in union coordinates.

MI355X data flow: the masks are rasterised on the device straight into NHWC (no rois->CPU->Cython->GPU round trip,
reference :47-50), the tower runs in NHWC on the MFMA GEMM / implicit-GEMM conv kernels, and the result is returned
as a logical [N,C,7,7] tensor (channels_last memory) so that `Flattener` + fc6 see the reference's (c,y,x) order.
State-dict keys are the reference's: conv.{0,2,4,6}.*
"""
import os

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from config import BATCHNORM_MOMENTUM
from lib.draw_rectangles.draw_rectangles import draw_union_boxes
from lib.fpn.roi_align.functions.roi_align import RoIAlignFunction
from lib import _hip
from lib.pytorch_misc import h2d, has_host, host_np
from lib import hip_ops
from lib.hip_ops import Conv2dNHWC, ReLU, EPI_NONE, EPI_RELU


class _ChannelsLastBN(nn.BatchNorm2d):
    """BatchNorm2d applied to an NHWC tensor (statistics per last-dim channel)."""

    def forward(self, x_nhwc):
        y = super(_ChannelsLastBN, self).forward(x_nhwc.permute(0, 3, 1, 2))
        return y.permute(0, 2, 3, 1)


class _MaxPoolNHWC(nn.MaxPool2d):
    def forward(self, x_nhwc):
        y = super(_MaxPoolNHWC, self).forward(x_nhwc.permute(0, 3, 1, 2))
        return y.permute(0, 2, 3, 1)


# the tower's first convolution: 'direct' = csrc/tower.hip, no column matrix -- since round 6 on the matrix cores (t1::tower_conv1_mfma_*:
# MFMA fragments built straight from the zero-padded masks, f16x3 forward, bf16x6 weight gradient; MH_TOWER_CONV1=valu = the round-4
# kernels, mask values through the scalar cache and exact fp32 FMAs; shapes the kernels do not cover fall through to the other path),
# 'gemm' = im2col + the small-product engine.  MOTIFS_TOWER_CONV1=gemm for A/B runs.  gpurun r06_c10, same box, alternating:
# 415.0 / 418.5 img/s (14.46 / 14.34 ms per cfg2 step) on the matrix cores against 408.1 / 407.2 (14.70 / 14.74) on the VALU kernels.
TOWER_CONV1 = os.environ.get('MOTIFS_TOWER_CONV1', 'direct')
# union rectangles / pair boxes computed on the host when boxes and pair list carry their host mirrors (round 6); =0: on the device (A/B)
HOST_GEOMETRY = os.environ.get('MOTIFS_HOST_GEOMETRY', '1') != '0'


# test hook (tests/parity_util.py): a dict here receives the tower's ReLU masks and pool arg-max table of the next forward,
# so that the parity tests can hand the oracle the product's own kink decisions (oracle/model.py: TAPS)
TAPS = None


class _TowerFn(torch.autograd.Function):
    """Conv7x7/2+ReLU -> BN -> MaxPool3x3/2 -> Conv3x3+ReLU -> BN -> (+ union features, NCHW out) as ONE autograd
    node over the fused kernels of csrc/tower.hip (ReLU lives in the conv epilogues; BN statistics / apply / pool /
    residual add and their backward are single HBM passes)."""

    @staticmethod
    def forward(ctx, rects, union_pools, w0, b0, g1, be1, w4, b4, g2, be2, bn1, bn2, training):
        N = rects.shape[0]
        C0, C1 = w0.shape[0], w4.shape[0]
        K0 = w0.shape[1] * w0.shape[2] * w0.shape[3]
        rects = rects.contiguous()
        ctx.direct = TOWER_CONV1 == 'direct' and _hip.tower_conv1_supported(rects, w0)
        if ctx.direct:
            cols0 = _hip.tower_conv1_pad(rects)                  # zero-padded masks (13 MB at 1536 pairs), kept for the weight gradient
            y0 = _hip.tower_conv1_fwd(cols0, w0.permute(2, 3, 1, 0).reshape(K0, C0).contiguous(), b0)
        else:
            ld0 = (K0 + 3) // 4 * 4
            cols0, Ho, Wo = _hip.im2col_nhwc(rects, w0.shape[2], w0.shape[3], 2, 3, ldo=ld0)
            wmat = w0.new_zeros(C0, ld0)
            wmat[:, :K0] = w0.permute(0, 2, 3, 1).reshape(C0, K0)
            y0 = _hip.gemm(cols0, wmat, False, True, bias=b0, epilogue=EPI_RELU).view(N, Ho, Wo, C0)

        def stats(x2d, bn):
            if training:
                return _hip.bn_stats(x2d, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
            return bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)
        mean1, invstd1 = stats(y0.view(-1, C0), bn1)
        if training:
            if bn1.num_batches_tracked.is_cuda:
                torch._foreach_add_([bn1.num_batches_tracked, bn2.num_batches_tracked], 1)       # one launch for both counters
            else:
                bn1.num_batches_tracked += 1
                bn2.num_batches_tracked += 1
        z, arg = _hip.bn_pool_fwd(y0, mean1, invstd1, g1, be1)
        ctx.maps = hip_ops._small_maps_on_planes(z, w4.shape[1], C1) and w4.shape[1] >= 128
        if ctx.maps:                  # the 3x3 conv over the N 7x7 maps on the ring engine (lib/hip_ops.py: conv3x3_small_maps)
            y1 = hip_ops.conv3x3_small_maps(z, w4, b4, EPI_RELU)
        else:
            y1 = _hip.conv3x3_nhwc(z, _hip.conv3x3_pack_weight(w4.contiguous(), False), b4, EPI_RELU)
        mean2, invstd2 = stats(y1.view(-1, C1), bn2)
        out = _hip.bn_residual_nchw(y1, mean2, invstd2, g2, be2, union_pools.contiguous())
        if TAPS is not None:
            TAPS.update({'union_boxes.conv.0': (y0 > 0).permute(0, 3, 1, 2).cpu(), 'union_boxes.pool': arg.cpu(),
                         'union_boxes.conv.4': (y1 > 0).permute(0, 3, 1, 2).cpu()})
        ctx.save_for_backward(cols0, y0, arg, z, y1, mean1, invstd1, mean2, invstd2, w0, w4, g1, g2)
        return out

    @staticmethod
    def backward(ctx, dout):
        cols0, y0, arg, z, y1, mean1, invstd1, mean2, invstd2, w0, w4, g1, g2 = ctx.saved_tensors
        C0, C1 = w0.shape[0], w4.shape[0]
        K0 = w0.shape[1] * w0.shape[2] * w0.shape[3]
        g_nhwc = _hip.nchw_to_nhwc_small(dout.contiguous())
        dx2, dg2, db2 = _hip.bn_bwd(y1, g_nhwc, None, mean2, invstd2, g2, True)          # through BN2 and conv.4's ReLU
        if ctx.maps and w4.shape[1] >= 128:
            dz = hip_ops.conv3x3_small_maps(dx2, w4, None, EPI_NONE, flip_transpose=True)
        else:
            dz = _hip.conv3x3_nhwc(dx2, _hip.conv3x3_pack_weight(w4.contiguous(), True), None, EPI_NONE)
        dw4 = _hip.conv3x3_wgrad(z, dx2)                 # implicit GEMM over the pixels: no 0.7 GB patch matrix
        if dw4 is None:                                  # f32-MFMA build
            cols4, _, _ = _hip.im2col_nhwc(z, 3, 3, 1, 1)
            dw4 = _hip.gemm(dx2.view(-1, C1), cols4, True, False)
        dw4 = dw4.view(C1, 3, 3, w4.shape[1]).permute(0, 3, 1, 2).contiguous()
        db4 = dx2.view(-1, C1).sum(0)
        dx1, dg1, db1 = _hip.bn_bwd(y0, dz, arg, mean1, invstd1, g1, True)               # pool + BN1 + conv.0's ReLU
        if ctx.direct:
            dwk, db0 = _hip.tower_conv1_wgrad(cols0, dx1)                    # [K0, C0] in (ky, kx, ci) order + the bias gradient
            dw0 = dwk.view(w0.shape[2], w0.shape[3], w0.shape[1], C0).permute(3, 2, 0, 1).contiguous()
            db0 = db0.clone()
        else:
            dwm = _hip.gemm(dx1.view(-1, C0), cols0, True, False)
            dw0 = dwm[:, :K0].reshape(C0, w0.shape[2], w0.shape[3], w0.shape[1]).permute(0, 3, 1, 2).contiguous()
            db0 = dx1.view(-1, C0).sum(0)
        d_up = dout if ctx.needs_input_grad[1] else None
        return None, d_up, dw0, db0, dg1, db1, dw4, db4, dg2, db2, None, None, None


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
        if HOST_GEOMETRY and rois.is_cuda and has_host(rois) and has_host(union_inds):
            r, u = host_np(rois), host_np(union_inds)                  # GT-box modes: the pairs' boxes gathered on the host, one upload
            pair_rois = h2d(np.ascontiguousarray(np.concatenate((r[u[:, 0], 1:], r[u[:, 1], 1:]), 1), dtype=np.float32), rois.device)
        else:
            pair_rois = torch.cat((rois[:, 1:][union_inds[:, 0]], rois[:, 1:][union_inds[:, 1]]), 1).detach()
        rects = draw_union_boxes(pair_rois, self.pooling_size * 4 - 1, offset=-0.5, channels_last=True)   # [N,27,27,2]
        if not self.concat and rects.is_cuda:
            c = self.conv
            return _TowerFn.apply(rects, union_pools, c[0].weight, c[0].bias, c[2].weight, c[2].bias, c[4].weight,
                                  c[4].bias, c[6].weight, c[6].bias, c[2], c[6], self.training)
        tower = self.conv(rects).permute(0, 3, 1, 2)                      # generic path (logical NCHW)
        if self.concat:
            return torch.cat((union_pools, tower), 1)
        return union_pools + tower


def union_boxes(fmap, rois, union_inds, pooling_size=14, stride=16):
    """RoIAlign over the union rectangle of each pair (reference :72-93)"""
    assert union_inds.size(1) == 2
    if HOST_GEOMETRY and rois.is_cuda and has_host(rois) and has_host(union_inds):
        # boxes and pair list came from the host (GT-box modes; the sampled relations of a training step): the union rectangles are
        # min / max of fp32 values -- exact on either side -- computed in numpy and uploaded once (was ~10 launches on the main stream
        # in front of the RoIAlign, profiles/r06_step_launches_c13.txt)
        r, u = host_np(rois), host_np(union_inds)
        a, b = r[u[:, 0]], r[u[:, 1]]
        union_np = np.ascontiguousarray(np.concatenate((a[:, :1], np.minimum(a[:, 1:3], b[:, 1:3]), np.maximum(a[:, 3:5], b[:, 3:5])), 1),
                                        dtype=np.float32)
        return RoIAlignFunction(pooling_size, pooling_size, spatial_scale=1 / stride)(fmap, h2d(union_np, rois.device))
    im_inds = rois[:, 0][union_inds[:, 0]]
    union_rois = torch.cat((
        im_inds[:, None],
        torch.min(rois[:, 1:3][union_inds[:, 0]], rois[:, 1:3][union_inds[:, 1]]),
        torch.max(rois[:, 3:5][union_inds[:, 0]], rois[:, 3:5][union_inds[:, 1]]),
    ), 1)
    return RoIAlignFunction(pooling_size, pooling_size, spatial_scale=1 / stride)(fmap, union_rois)
