"""ResNet-101 trunk (conv1 .. layer3) for `ObjectDetector(use_resnet=True)` -- the reference's `load_resnet()`
(lib/object_detector.py:615-620: torchvision `resnet101` minus layer4/avgpool/fc; feature_map :119-127) and the
bottleneck of lib/resnet.py:8-46 (1x1 -> BN -> ReLU -> 3x3 (stride here) -> BN -> ReLU -> 1x1 -> BN, + identity or
1x1/stride projection, ReLU).

Module and parameter names are torchvision's (`conv1.weight`, `bn1.*`, `layer2.0.downsample.0.weight`, ...), so the
reference's state dicts load.  Everything runs NHWC on the HIP kernels of libmotifs_hip.so:
  1x1 convs            -> MFMA GEMM on the [pixels, C] view (weights used in place, [Cout, Cin] K-contiguous)
  3x3 stride-1 convs   -> the implicit-GEMM conv kernel (packed weights cached)
  7x7/2 stem, 3x3/2    -> im2col + MFMA GEMM
  BatchNorm            -> mh_bn_stats (batch statistics + running-stat update in train mode, exactly like the
                          reference, which calls detector.train() with frozen weights: train_rels.py:101) fused with
                          the residual add and the ReLU in mh_bn_apply_nhwc; the stem's BN is fused with its 3x3/2 max-pool
models/train_rels.py freezes the detector: that path is forward-only (no_grad, fused BN + pool, cached weight images).  With
trainable parameters (detector pre-training, SURVEY.md §8f: models/train_detector.py -resnet) the same blocks run through the
autograd Functions of lib/hip_ops.py -- no kernel of their own: a strided 1x1 conv is the product on the subsampled rows, a
strided 3x3 conv is the stride-1 conv subsampled (identical arithmetic at the kept pixels; two layers of the trunk), the 7x7/2
stem is im2col + the product (its input is the image: no input gradient), BN + ReLU + max-pool of the stem is mh_bn_pool_fwd
with mh_bn_bwd's pooled form behind it (_StemPoolFn)."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from lib import _hip
from lib.hip_ops import _c, _Conv3x3Fn, linear, EPI_NONE, EPI_RELU

BN_EPS = 1e-5
BN_MOMENTUM = 0.1            # torchvision's BatchNorm2d default (the detector uses torchvision.models.resnet)
L4_BN_MOMENTUM = 0.01        # config.BATCHNORM_MOMENTUM: the relation model's resnet_l4 blocks (reference lib/resnet.py:14-19)


class _Conv(nn.Module):
    """bias-free conv holder with torchvision's parameter name (`weight` [Cout,Cin,k,k]) and a HIP forward on NHWC"""

    def __init__(self, cin, cout, k, stride=1, pad=0):
        super(_Conv, self).__init__()
        self.k, self.stride, self.pad = k, stride, pad
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        n = k * k * cout
        self.weight.data.normal_(0, (2.0 / n) ** 0.5)                       # lib/resnet.py:66-69
        self._cache = (None, None)

    def _derived(self):
        """weight matrix / packed weights for the kernel in use, rebuilt only when the parameter changes"""
        key = _hip.version_of(self.weight)
        with _hip.cache_lock:               # shared with the detect-ahead worker thread / stream (_hip.built_here)
            return self._derived_locked(key)

    def _derived_locked(self, key):
        if self._cache[0] != key:
            w = self.weight.detach()
            cout, cin, k = w.shape[0], w.shape[1], self.k
            if k == 1:
                d = _c(w.view(cout, cin))
                # frozen 1x1 convs (the detector trunk): the weight's plane image is made ONCE and the product runs on the ring /
                # plane engine whatever its size -- what keeps a 4 GFLOP product off that engine is the cost of TWO operand
                # images per call, and the weight's is free here (round 5: the trunk's 81 products of 1-17 GFLOP ran on the
                # small-product engine at 37 TFLOP/s, 9.5 ms of the cfg4 step; profiles/r05_cfg4_kernel_stats.csv)
                self._image = None
                if (w.is_cuda and not self.weight.requires_grad and cin % 16 == 0 and cin >= 64
                        and os.environ.get('MOTIFS_RESNET_1X1', 'planes') != 'small'):          # (=small: the A/B arm)
                    self._image = _hip.make_planes(d, True)
            elif k == 3 and self.stride == 1 and cin % 16 == 0:
                d = _hip.conv3x3_pack_weight(_c(w))
            else:                                                            # im2col order: (ky*kw + kx)*C + c
                K = k * k * cin
                ld = (K + 3) // 4 * 4
                d = w.new_zeros(cout, ld)
                d[:, :K] = w.permute(0, 2, 3, 1).reshape(cout, K)
            self._built = _hip.built_here(w.device)
            self._cache = (key, d)
        _hip.use_built(getattr(self, '_built', None))
        return self._cache[1]

    def forward(self, x):                       # x NHWC
        B, H, W, C = x.shape
        cout = self.weight.shape[0]
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            # autograd path: the relation model's trainable layer4 copies (stride-1 1x1 / 3x3) and, for detector pre-training,
            # the trunk's strided layers
            s = self.stride
            if self.k == 1:
                if s != 1:
                    x = x[:, ::s, ::s, :].contiguous()
                    B, H, W, C = x.shape
                return linear(x.reshape(-1, C), self.weight.view(cout, C)).view(B, H, W, cout)
            if self.k == 3 and self.pad == 1 and C % 16 == 0:
                y = _Conv3x3Fn.apply(x, self.weight, None, EPI_NONE)
                # stride s, pad 1: output (i, j) reads the window centred on input (s i, s j) = the stride-1 output there
                return y if s == 1 else y[:, ::s, ::s, :].contiguous()
            if x.requires_grad:
                raise NotImplementedError('trainable ResNet conv %dx%d/%d with an input gradient (only the stem, whose input is '
                                          'the image, takes the im2col path)' % (self.k, self.k, s))
            K = self.k * self.k * C
            ld = (K + 3) // 4 * 4
            cols, Ho, Wo = _hip.im2col_nhwc(_c(x), self.k, self.k, s, self.pad, ldo=ld)       # pad columns are zero
            wm = F.pad(self.weight.permute(0, 2, 3, 1).reshape(cout, K), (0, ld - K))         # im2col order (ky*kw + kx)*C + c
            return linear(cols, wm).view(B, Ho, Wo, cout)
        d = self._derived()
        if self.k == 1:
            if self.stride != 1:
                x = x[:, ::self.stride, ::self.stride, :].contiguous()
                B, H, W, C = x.shape
            x2 = x.view(-1, C)
            if getattr(self, '_image', None) is not None and x2.shape[0] >= 1024:
                return _hip.gemm_planes(_hip.make_planes(x2, True), self._image).view(B, H, W, cout)
            return _hip.gemm(x2, d, False, True).view(B, H, W, cout)
        if self.k == 3 and self.stride == 1 and C % 16 == 0:
            return _hip.conv3x3_nhwc(_c(x), d, None, 0)
        cols, Ho, Wo = _hip.im2col_nhwc(_c(x), self.k, self.k, self.stride, self.pad, ldo=d.shape[1])
        return _hip.gemm(cols, d, False, True).view(B, Ho, Wo, cout)


class _BNFn(torch.autograd.Function):
    """train-mode BatchNorm + residual + ReLU on NHWC through the HIP kernels, with its backward: the ReLU mask on
    the saved output (mh_act_bwd), then mh_bn_bwd (dense form: gradient through the batch statistics, dgamma, dbeta);
    the residual branch receives the masked gradient unchanged."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, relu, mean, invstd):
        x = _c(x)
        y = _hip.bn_apply_nhwc(x, mean, invstd, gamma.detach(), beta.detach(), None if residual is None else _c(residual), relu)
        ctx.relu, ctx.has_res = bool(relu), residual is not None
        ctx.save_for_backward(x, gamma.detach(), mean, invstd, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, mean, invstd, y = ctx.saved_tensors
        g = _c(g)
        if ctx.relu:
            g = _hip.act_bwd(g, y, EPI_RELU)
        dx, dgamma, dbeta = _hip.bn_bwd(x, g, None, mean, invstd, gamma, False)
        return dx, dgamma, dbeta, (g if ctx.has_res else None), None, None, None


class _StemPoolFn(torch.autograd.Function):
    """relu(maxpool3x3/2(BN(x))) of the stem in train mode (== maxpool(relu(BN(x))), lib/object_detector.py:121-124) on
    mh_bn_pool_fwd; backward: the ReLU mask on the saved output, then mh_bn_bwd in its pooled form (gradient gathered through
    the saved arg-max, through the batch statistics, dgamma, dbeta) -- the kernels of the union-box mask tower"""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, invstd):
        x = _c(x)
        y, argmax = _hip.bn_pool_fwd(x, mean, invstd, gamma.detach(), beta.detach())
        y = torch.relu_(y)
        ctx.save_for_backward(x, gamma.detach(), mean, invstd, argmax, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, mean, invstd, argmax, y = ctx.saved_tensors
        g = _hip.act_bwd(_c(g), y, EPI_RELU)
        dx, dgamma, dbeta = _hip.bn_bwd(x, g, argmax, mean, invstd, gamma, False)
        return dx, dgamma, dbeta, None, None


class _BN(nn.Module):
    """BatchNorm2d parameters/buffers under torchvision's names; statistics + apply on the HIP kernels"""

    def __init__(self, c, momentum=BN_MOMENTUM):
        super(_BN, self).__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.momentum = momentum

    def stats(self, x):
        if self.training:
            return _hip.bn_stats(x.view(-1, x.shape[-1]), BN_EPS, self.momentum, self.running_mean, self.running_var)
        return self.running_mean, torch.rsqrt(self.running_var + BN_EPS)

    def forward(self, x, residual=None, relu=False):
        mean, invstd = self.stats(x)
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            if not self.training:
                raise NotImplementedError('BatchNorm backward is built for train mode (batch statistics) only')
            return _BNFn.apply(x, self.weight, self.bias, residual, relu, mean, invstd)
        return _hip.bn_apply_nhwc(x, mean, invstd, self.weight.detach(), self.bias.detach(), residual, relu)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, momentum=BN_MOMENTUM):
        super(Bottleneck, self).__init__()
        self.conv1 = _Conv(inplanes, planes, 1)
        self.bn1 = _BN(planes, momentum)
        self.conv2 = _Conv(planes, planes, 3, stride=stride, pad=1)
        self.bn2 = _BN(planes, momentum)
        self.conv3 = _Conv(planes, planes * 4, 1)
        self.bn3 = _BN(planes * 4, momentum)
        self.downsample = downsample
        self.stride = stride
        self.relu_end = True            # lib/resnet.py:126-131: the last block of the relation model's layer4 ends without ReLU

    def forward(self, x):                       # NHWC
        out = self.bn1(self.conv1(x), relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        out = self.conv3(out)
        residual = x if self.downsample is None else self.downsample[1](self.downsample[0](x))
        return self.bn3(out, residual=residual, relu=self.relu_end)


class ResNet101Trunk(nn.Module):
    """resnet101 with layer4 / avgpool / fc deleted (load_resnet, lib/object_detector.py:615-620)"""

    LAYERS = (3, 4, 23)

    def __init__(self):
        super(ResNet101Trunk, self).__init__()
        self.inplanes = 64
        self.conv1 = _Conv(3, 64, 7, stride=2, pad=3)
        self.bn1 = _BN(64)
        self.layer1 = self._make_layer(64, self.LAYERS[0])
        self.layer2 = self._make_layer(128, self.LAYERS[1], stride=2)
        self.layer3 = self._make_layer(256, self.LAYERS[2], stride=2)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(_Conv(self.inplanes, planes * 4, 1, stride=stride), _BN(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)

    def stem(self, x):
        """conv1 -> bn1 -> relu -> maxpool of a [B,3,S,S] image with autograd (trainable trunk) -> NHWC [B,S/4,S/4,64]"""
        if not self.training:
            raise NotImplementedError('BatchNorm backward is built for train mode (batch statistics) only')
        y = self.conv1(_hip.nchw_to_nhwc(_c(x)))
        if y.shape[1] % 2 or y.shape[2] % 2:
            raise ValueError('the fused BN + 3x3/2 max-pool needs an even stem output (image side % 4 == 0)')
        mean, invstd = self.bn1.stats(y.detach())
        return _StemPoolFn.apply(y, self.bn1.weight, self.bn1.bias, mean, invstd)

    def forward(self, x):
        """[B,3,S,S] NCHW image -> c4 [B,1024,S/16,S/16] (channels_last memory), lib/object_detector.py:119-127"""
        if any(p.requires_grad for p in self.parameters()) and torch.is_grad_enabled():
            # detector pre-training (models/train_detector.py -resnet): the same blocks through their autograd Functions
            y = self.stem(x)
            for layer in (self.layer1, self.layer2, self.layer3):
                for block in layer:
                    y = block(y)
            return y.permute(0, 3, 1, 2)
        with torch.no_grad():
            y = self.conv1(_hip.nchw_to_nhwc(_c(x)))
            if y.shape[1] % 2 or y.shape[2] % 2:
                raise ValueError('the fused BN + 3x3/2 max-pool needs an even stem output (image side % 4 == 0)')
            mean, invstd = self.bn1.stats(y)
            # relu(maxpool(BN(y))) == maxpool(relu(BN(y))): one fused kernel, then the ReLU on the 4x smaller tensor
            y, _ = _hip.bn_pool_fwd(y, mean, invstd, self.bn1.weight.detach(), self.bn1.bias.detach())
            y = torch.relu_(y)
            for layer in (self.layer1, self.layer2, self.layer3):
                for block in layer:
                    y = block(y)
        return y.permute(0, 3, 1, 2)


class Layer4Stack(nn.Sequential):
    """`resnet_l4(relu_end)` of the reference (lib/resnet.py:126-133): torchvision's layer4 (3 bottlenecks, 1024 ->
    2048) with the stride taken out of the first block, as the relation model's RoI feature extractor
    (lib/rel_model.py:360-365).  Child names are torchvision's (`0.conv1.weight`, `0.downsample.0.weight`, ...), so
    `roi_fmap.0.*` keys of a reference checkpoint load.  Input: RoI features [n, 1024, 7, 7] (logical NCHW); the blocks run
    NHWC on the HIP kernels, with autograd when the parameters train."""

    def __init__(self, relu_end=True):
        # the reference's OWN lib/resnet.py builds these blocks with momentum=BATCHNORM_MOMENTUM = 0.01 (lib/resnet.py:14-19,
        # config.py:57); only the detector trunk comes from torchvision (0.1)
        m = L4_BN_MOMENTUM
        down = nn.Sequential(_Conv(1024, 2048, 1), _BN(2048, m))
        blocks = [Bottleneck(1024, 512, 1, down, momentum=m), Bottleneck(2048, 512, momentum=m), Bottleneck(2048, 512, momentum=m)]
        blocks[-1].relu_end = relu_end
        super(Layer4Stack, self).__init__(*blocks)

    def forward(self, x):
        y = x.permute(0, 2, 3, 1).contiguous()                  # NCHW-shaped RoI features -> NHWC
        for block in self:
            y = block(y)
        return y                                                # [n, 7, 7, 2048]


class AvgPoolNHWC(nn.Module):
    """nn.AvgPool2d(7) + Flattener of the reference on the NHWC output of Layer4Stack: [n, 7, 7, C] -> [n, C]"""

    def forward(self, x):
        return x.mean((1, 2))
