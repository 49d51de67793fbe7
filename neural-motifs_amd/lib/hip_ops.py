"""
autograd glue over the HIP kernels (lib/_hip.py): the nn.Module / Function layer that replaces the cuDNN / cuBLAS
backed nn.Linear and nn.Conv2d calls of the reference on the hot path.

Every op here runs hand-written gfx950 kernels through the C ABI; nothing falls back to rocBLAS/MIOpen or to the
CPU.  Parameter names and shapes are the reference's (nn.Linear: weight [out,in], bias [out]; conv: weight
[Cout,Cin,kh,kw]) so checkpoints load unchanged.
"""
import math
import os
import weakref

import torch
import torch.nn as nn

from lib import _hip
from lib.rng import alpha_dropout as rng_alpha_dropout, dropout as rng_dropout

EPI_NONE, EPI_RELU, EPI_RELU6 = 0, 1, 2

# test hook (tests/parity_util.py): a dict here receives the activation masks / pool arg-max tables of the trainable trunk
# and RPN head during the next forward, in the oracle's naming, so that the parity tests can hand the oracle the product's
# own kink decisions (oracle/model.py: TAPS).  None in production.
TAPS = None


def _tap_act(name, y_nhwc, epilogue):
    if TAPS is not None:
        m = (y_nhwc > 0) if epilogue == EPI_RELU else ((y_nhwc > 0) & (y_nhwc < 6))
        TAPS[name] = m.permute(0, 3, 1, 2).cpu()


def _tap_pool2x2(name, x_nhwc):
    """arg-max table [B,H/2,W/2,C] (dy * 2 + dx, first maximum in scan order = where mh_maxpool2x2_bwd_nhwc routes)"""
    if TAPS is not None:
        B, H, W, C = x_nhwc.shape
        w = x_nhwc[:, :H // 2 * 2, :W // 2 * 2].reshape(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 5, 2, 4)
        TAPS[name] = w.reshape(B, H // 2, W // 2, C, 4).argmax(-1).to(torch.uint8).cpu()


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _rows2d(t):
    """2-D view with unit inner stride (row stride may be larger); copies only if it has to"""
    if t.dim() != 2:
        raise ValueError('expected a 2-D tensor')
    return t if t.stride(1) == 1 else t.contiguous()


# =========================================================================================== Linear
# plane images of weights (lib/_hip.py: PlaneImage), cached per parameter VALUE: key = _hip.version_of(weight), which the
# fused optimizer bumps through note_raw_update.  'w' = rows are output features (forward), 'wt' = rows are input features
# (input gradient).  Frozen weights (the detector's fc6/fc7: 411 MB) are split once for the life of the process.
_weight_images = {}


def _weight_image(weight, transposed, want_both=False):
    """image of the weight [out, in]: rows = output features (forward) or, transposed, rows = input features (input
    gradient).  want_both (the forward of a layer whose input needs a gradient): both images from ONE pass over the weight
    -- the backward of this step will ask for the other one."""
    key = id(weight)
    ver = _hip.version_of(weight)
    with _hip.cache_lock:                  # entries are shared with the detect-ahead worker thread and its stream (_hip.built_here)
        hit = _weight_images.get(key)
        if hit is None or hit[0] != ver:
            if hit is None:
                weakref.finalize(weight, _weight_images.pop, key, None)      # the images die with their parameter
            hit = [ver, None, None, None, None]                              # version, image, transposed image, their build marks
            _weight_images[key] = hit
        idx = 2 if transposed else 1
        if hit[idx] is None:
            w2 = _rows2d(weight.detach())
            if want_both and hit[1] is None and hit[2] is None:
                hit[1], hit[2] = _hip.make_planes_both(w2)
                hit[3] = hit[4] = _hip.built_here(weight.device)
            else:
                hit[idx] = _hip.make_planes(w2, k_contiguous=not transposed)
                hit[idx + 2] = _hip.built_here(weight.device)
        _hip.use_built(hit[idx + 2])
        return hit[idx]


def drop_weight_images():
    _weight_images.clear()


_SKINNY_ROWS = 128          # see _hip.gemm_inloop
_LINEAR_ENGINE = os.environ.get('MOTIFS_LINEAR', 'auto')      # 'inloop': every Linear on the in-loop-split kernel (A/B switch)

# multi-GPU: while a backward pass runs under lib.dist.OverlappedGradReducer this is its `grad_view`: a weight-gradient GEMM
# writes its result straight into the parameter's slot of the gradient bucket (no copy into the bucket afterwards)
GRAD_SINK = None
# ... and the reducer itself: for a weight it reduces in row ranges (fc6: 411 MB in ranges of <= 64 MB) the weight gradient is
# produced range by range, each range reported as soon as its GEMM is enqueued, so that its all-reduce runs under the next one
GRAD_REDUCER = None


def _wgrad_planes(gy, x_cols, weight):
    """gw [N, K] = gy^T [N, M] . x [M, K] on plane images; `x_cols` = image of x with operand rows = x columns.  Written into
    the parameter's gradient bucket when a reducer is armed (GRAD_SINK), in the reducer's row ranges when it has any."""
    sink = GRAD_SINK(weight) if (GRAD_SINK is not None and weight.is_contiguous()) else None
    segs = GRAD_REDUCER.segments(weight) if (sink is not None and GRAD_REDUCER is not None) else None
    if segs is None or len(segs) < 2:
        return _hip.gemm_planes(_hip.make_planes(gy, False), x_cols, out=sink)
    red = GRAD_REDUCER
    for i, (r0, r1) in enumerate(segs):
        _hip.gemm_planes(_hip.make_planes(gy[:, r0:r1], False), x_cols, out=sink[r0:r1])
        red.segment_done(weight, i)
    return sink


class _LinearFn(torch.autograd.Function):
    """y = act(x @ W^T + b): fp32-accurate product on the f16 matrix cores (csrc/pl_gemm.hip), ReLU fused into the
    epilogue.  Every operand is split into its plane images ONCE: the weight images are cached per parameter value, the
    input's second image (for the weight gradient) is made together with the first and kept for backward instead of the
    fp32 input, the output gradient gets both of its images from one pass."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        x2, w2 = _rows2d(x), _rows2d(weight)
        epi = EPI_RELU if relu else EPI_NONE
        ctx.x_cols = None
        x_keep = None
        need_wgrad, need_dgrad = ctx.needs_input_grad[1], ctx.needs_input_grad[0]      # (grad mode is off inside forward)
        M, K, N = x2.shape[0], x2.shape[1], w2.shape[0]
        # plane images pay for big, deep products (and for cached weights); small / thin ones and the skinny product against a
        # big weight that changes every step read their fp32 operands once in the in-loop-split kernel (_hip.gemm_inloop)
        ctx.images = (M > 0 and 2.0 * M * N * K >= 20e9 and K >= 512 and not (M <= _SKINNY_ROWS and weight.requires_grad)
                      and _LINEAR_ENGINE != 'inloop')
        if M == 0:
            y = x2.new_zeros(0, N)
        elif not ctx.images:
            y = _hip.gemm_inloop(x2, w2, False, True, bias=bias, epilogue=epi)
            x_keep = x2 if need_wgrad else None
        else:
            if need_wgrad:
                x_rows, ctx.x_cols = _hip.make_planes_both(x2)
            else:
                x_rows = _hip.make_planes(x2, True)
            y = _hip.gemm_planes(x_rows, _weight_image(weight, False, want_both=need_dgrad), bias=bias, epilogue=epi)
        ctx.relu = relu
        ctx.has_bias = bias is not None
        ctx.weight = weight
        ctx.x_shape = tuple(x2.shape)
        # fp32 tensors go through save_for_backward (autograd's in-place-modification check); the plane image kept for the weight
        # gradient is this function's own buffer
        ctx.save_for_backward(y if relu else None, x_keep)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, x_keep = ctx.saved_tensors
        weight = ctx.weight
        gy = _rows2d(gy)
        if ctx.relu:
            # gradient where the OUTPUT is > 0 (nn.ReLU's rule): one kernel instead of compare + cast + multiply
            if (gy.is_cuda and gy.is_contiguous() and y.is_contiguous() and gy.numel() % 4 == 0 and gy.numel() > 0
                    and (gy.data_ptr() | y.data_ptr()) % 16 == 0):
                gy = _hip.act_bwd(gy, y, EPI_RELU)
            else:
                gy = gy * (y > 0).to(gy.dtype)
        gx = gw = gb = None
        if gy.shape[0] == 0:
            return (gy.new_zeros(ctx.x_shape) if ctx.needs_input_grad[0] else None,
                    torch.zeros_like(weight) if ctx.needs_input_grad[1] else None,
                    torch.zeros_like(weight[:, 0]) if (ctx.has_bias and ctx.needs_input_grad[2]) else None, None)
        if not ctx.images:
            w2 = _rows2d(weight.detach())
            if ctx.needs_input_grad[0]:
                gx = _hip.gemm_inloop(gy, w2, False, False)                                # [M,N] . [N,K]
            if ctx.needs_input_grad[1]:
                M, K, N = ctx.x_shape[0], ctx.x_shape[1], gy.shape[1]
                if 2.0 * M * N * K >= 20e9 and M >= 96:        # the skinny layer's weight gradient is a big product again
                    gw = _wgrad_planes(gy, _hip.make_planes(x_keep, False), weight)
                else:
                    sink = GRAD_SINK(weight) if (GRAD_SINK is not None and weight.is_contiguous()) else None
                    gw = _hip.gemm_inloop(gy, x_keep, True, False, out=sink)               # [M,N]^T . [M,K]
        else:
            gy_rows = gy_cols = None
            if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
                gy_rows, gy_cols = _hip.make_planes_both(gy)
            if ctx.needs_input_grad[0]:
                gy_rows = gy_rows if gy_rows is not None else _hip.make_planes(gy, True)
                gx = _hip.gemm_planes(gy_rows, _weight_image(weight, True))                 # [M,N] . (W^T image [K,N])^T
            if ctx.needs_input_grad[1]:
                if GRAD_REDUCER is not None and GRAD_REDUCER.segments(weight) is not None:
                    gw = _wgrad_planes(gy, ctx.x_cols, weight)                              # row ranges, reduced as they complete
                else:
                    gy_cols = gy_cols if gy_cols is not None else _hip.make_planes(gy, False)
                    sink = GRAD_SINK(weight) if (GRAD_SINK is not None and weight.is_contiguous()) else None
                    gw = _hip.gemm_planes(gy_cols, ctx.x_cols, out=sink)                   # gy^T [N,M] . (x^T [K,M])^T
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(0)
        ctx.x_cols = None
        return gx, gw, gb, None


def linear(x, weight, bias=None, relu=False):
    return _LinearFn.apply(x, weight, bias, relu)


class Linear(nn.Module):
    """Drop-in for nn.Linear (same parameter names / init) running on the HIP GEMM."""

    def __init__(self, in_features, out_features, bias=True):
        super(Linear, self).__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1.0 / math.sqrt(self.in_features)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, relu=False):
        lead = x.shape[:-1]
        y = linear(x.reshape(-1, x.shape[-1]), self.weight, self.bias, relu)
        return y.view(*lead, self.out_features)

    def extra_repr(self):
        return 'in_features=%d, out_features=%d, bias=%s' % (self.in_features, self.out_features, self.bias is not None)


class ReLU(nn.Module):
    """Marker module (keeps the reference's Sequential indices); fused into the preceding Linear/conv when the
    parent container runs the stack, plain clamp otherwise."""

    def forward(self, x):
        # torch.relu, not clamp_min: its backward passes the gradient where the OUTPUT is > 0 (nn.ReLU / threshold_backward, what the
        # reference runs); clamp_min passes it where the input is >= 0 -- a pre-activation that is exactly 0.0 then trains
        # differently (found by the kink accounting of tests/test_gpu_sgdet.py: one such unit in 16384)
        return torch.relu(x)


class Dropout(nn.Module):
    """nn.Dropout whose mask source can be switched to the seeded host stream (lib/rng.py)."""

    def __init__(self, p=0.5):
        super(Dropout, self).__init__()
        self.p = p

    def forward(self, x):
        return rng_dropout(x, self.p, self.training)


class AlphaDropout(nn.AlphaDropout):
    """nn.AlphaDropout whose mask source can be switched to the seeded host stream (lib/rng.py): the SELU RoI head of the ResNet
    detector branch in train mode is then comparable with the oracle draw for draw."""

    def forward(self, x):
        return rng_alpha_dropout(x, self.p, self.training)


class FCStack(nn.Sequential):
    """A VGG-classifier-style Sequential of Linear / ReLU / Dropout children (indices as in torchvision's
    vgg16.classifier after the reference's deletions, lib/object_detector.py:623-633).  Running the container
    fuses each Linear with a following ReLU into one GEMM epilogue."""

    def forward(self, x):
        mods = list(self.children())
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, Linear) and i + 1 < len(mods) and isinstance(mods[i + 1], ReLU):
                x = m(x, relu=True)
                i += 2
            else:
                x = m(x)
                i += 1
        return x


class Flattener(nn.Module):
    def forward(self, x):
        return x.reshape(x.size(0), -1)


# =========================================================================================== conv stack
class Conv3x3(nn.Module):
    """3x3 / stride 1 / pad 1 convolution parameters in the API layout [Cout,Cin,3,3]; the packed [9][Cin][Cout]
    copy the implicit-GEMM kernel reads is cached and refreshed when the weight tensor changes."""

    def __init__(self, cin, cout):
        super(Conv3x3, self).__init__()
        self.in_channels, self.out_channels = cin, cout
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.kaiming_normal_(self.weight, mode='fan_out', nonlinearity='relu')
        self._packed = None
        self._packed_key = None
        self._plane = None
        self._plane_key = None
        self._packed_built = self._plane_built = None      # _hip.built_here marks of the two cached copies

    def plane_weight(self):
        """packed plane image of the weights for the plane-engine conv (csrc/pl_conv.hip), cached per parameter value"""
        key = (_hip.version_of(self.weight), self.weight.device)
        with _hip.cache_lock:
            if self._plane_key != key:
                self._plane = _hip.plconv_pack_weight(_c(self.weight.detach()), False)
                self._plane_built = _hip.built_here(self.weight.device)
                self._plane_key = key
            _hip.use_built(self._plane_built)
            return self._plane

    def packed_weight(self, flip_transpose=False):
        key = (_hip.version_of(self.weight), flip_transpose, self.weight.device)
        with _hip.cache_lock:
            if self._packed_key != key:
                self._packed = _hip.conv3x3_pack_weight(_c(self.weight.detach()), flip_transpose)
                self._packed_built = _hip.built_here(self.weight.device)
                self._packed_key = key
            _hip.use_built(self._packed_built)
            return self._packed

    def forward_nhwc(self, x_nhwc, epilogue=EPI_RELU):
        if self.in_channels % 16 != 0:
            raise ValueError('Conv3x3.forward_nhwc needs Cin % 16 == 0 (use the stem kernel for the image layer)')
        return _hip.conv3x3_nhwc(x_nhwc, self.packed_weight(), self.bias.detach(), epilogue)

    def forward(self, x):
        """API-compatible NCHW in / NCHW-shaped out (channels_last memory)."""
        y = self.forward_nhwc(_hip.nchw_to_nhwc(_c(x)) if not _is_nhwc(x) else x.permute(0, 2, 3, 1), EPI_NONE)
        return y.permute(0, 3, 1, 2)


def _is_nhwc(x):
    """logical NCHW tensor whose memory is NHWC-contiguous"""
    return x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous()


class MaxPool2x2(nn.Module):
    def forward(self, x):
        return _hip.maxpool2x2_nhwc(_c(x.permute(0, 2, 3, 1))).permute(0, 3, 1, 2)


class VGG16Features(nn.Sequential):
    """torchvision vgg16().features minus the last max-pool (lib/object_detector.py:623-626): same child indices
    (convs at 0,2,5,7,10,12,14,17,19,21,24,26,28 -> state-dict keys features.N.{weight,bias}).

    forward: NCHW image in; every layer runs in NHWC on the MFMA implicit-GEMM kernel with fused bias+ReLU; the
    returned feature map is a logical [B,512,H/16,W/16] tensor with channels_last strides (physically NHWC), which
    RoIAlign consumes coalesced over channels.  With frozen parameters (models/train_rels.py:50-52) the forward-only
    fast path runs; with trainable parameters (models/train_detector.py) the same layers run as autograd Functions."""

    CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512)

    def __init__(self):
        layers, cin = [], 3
        for v in self.CFG:
            if v == 'M':
                layers.append(MaxPool2x2())
            else:
                layers += [Conv3x3(cin, v), ReLU()]
                cin = v
        super(VGG16Features, self).__init__(*layers)

    def forward(self, x):
        if any(p.requires_grad for p in self.parameters()) and torch.is_grad_enabled():
            return self._forward_trainable(x)          # detector pre-training (models/train_detector.py)
        # Engine of the frozen trunk: plane images (csrc/pl_conv.hip) for batches, the round-2 in-loop kernels for a single
        # image -- at b = 1 most layers have fewer tiles than one round of resident blocks and the converters' launches are
        # not paid back (cfg1 evaluation: 2.06 ms of conv per image on planes, 1.5 ms on the in-loop kernels).
        # MOTIFS_TRUNK=planes|v2 forces one engine.
        engine = os.environ.get('MOTIFS_TRUNK', 'auto')
        if engine == 'planes' or (engine == 'auto' and x.shape[0] >= 2):
            # csrc/pl_conv.hip addresses an activation image with 32-bit byte offsets (image bytes = B*H*W*C*4 < 0x7ff00000,
            # B <= 256): the widest layer (64 channels at full resolution) bounds the images of one pass -- 23 at 592x592.
            # Larger batches run in chunks (the trunk is per-image arithmetic; per-image scales make chunking exact).
            cap = max(1, min(256, (0x7ff00000 - 1) // (x.shape[2] * x.shape[3] * 64 * 4)))
            if x.shape[0] <= cap:
                return self._forward_planes(x)
            return torch.cat([self._forward_planes(x[i:i + cap]) for i in range(0, x.shape[0], cap)], 0)
        with torch.no_grad():
            mods = list(self.children())
            first = mods[0]
            y = _hip.conv_first_nchw(_c(x), _c(first.weight), first.bias, EPI_RELU)   # NHWC out
            i = 2
            while i < len(mods):
                m = mods[i]
                if isinstance(m, Conv3x3):
                    y = m.forward_nhwc(y, EPI_RELU)
                    i += 2                      # its ReLU is fused
                elif isinstance(m, MaxPool2x2):
                    y = _hip.maxpool2x2_nhwc(y)
                    i += 1
                else:
                    raise RuntimeError('unexpected module in VGG16Features')
        return y.permute(0, 3, 1, 2)

    def _forward_planes(self, x):
        """the frozen trunk on the plane engine (csrc/pl_conv.hip): every conv reads its input as a pre-split f16 plane
        image (no split arithmetic in the K loop) and reports the per-image maxima of what it writes; ONE converter pass
        per layer turns the fp32 output into the next layer's image, through the 2x2 max-pool where the reference has
        one (no separate pool launches, no per-pixel exponent passes).  MOTIFS_TRUNK=v2 selects the round-2 kernels."""
        with torch.no_grad():
            mods = list(self.children())
            B = x.shape[0]
            nconv = sum(isinstance(m, Conv3x3) for m in mods)
            mb = torch.zeros(nconv, B, dtype=torch.int32, device=x.device)     # per-layer, per-image |y| maxima (fp32 bits)
            first = mods[0]
            direct = os.environ.get('MOTIFS_TRUNK_DIRECT', '1') != '0' and B <= 32
            # a layer that is NOT followed by a pool hands its output to the next one as a plane image straight from its
            # epilogue (no fp32 tensor, no converter).  A layer in front of a pool does the same THROUGH the pool since round 6
            # (_hip.plconv3x3_pool_to_image: tile rows in pool order, the window maximum taken in the epilogue; even map sizes --
            # MOTIFS_TRUNK_POOL=converter restores the fp32 output + pooling converter); the last layer writes fp32 NHWC
            fused_pool = direct and os.environ.get('MOTIFS_TRUNK_POOL', 'epilogue') != 'converter'
            def pool_follows(idx):
                return idx + 2 < len(mods) and isinstance(mods[idx + 2], MaxPool2x2)
            is_last = lambda idx: idx + 2 >= len(mods)
            y = img = None
            if direct and not pool_follows(0):
                img = _hip.stem_to_image(_c(x), _c(first.weight), first.bias, EPI_RELU, mb[0])
            else:
                y = _hip.conv_first_nchw_max(_c(x), _c(first.weight), first.bias, EPI_RELU, mb[0])
            i, layer, pool = 2, 0, False
            while i < len(mods):
                m = mods[i]
                if isinstance(m, Conv3x3):
                    if img is None:
                        img = _hip.act_planes(y, mb[layer], pool=pool)
                    pool = False
                    if direct and not pool_follows(i) and not is_last(i):
                        img, y = _hip.plconv3x3_to_image(img, mb[layer], m.plane_weight(), m.out_channels, m.bias.detach(), EPI_RELU,
                                                         mb[layer + 1]), None
                    elif (fused_pool and pool_follows(i) and i + 3 < len(mods) and img.H % 2 == 0 and img.W % 2 == 0
                          and m.out_channels % 16 == 0):
                        img, y = _hip.plconv3x3_pool_to_image(img, mb[layer], m.plane_weight(), m.out_channels, m.bias.detach(),
                                                              EPI_RELU, mb[layer + 1]), None
                        i += 1                  # the MaxPool2x2 module behind the ReLU has been applied
                    else:
                        y = _hip.plconv3x3(img, m.plane_weight(), m.out_channels, m.bias.detach(), EPI_RELU, mb[layer + 1])
                        img = None
                    layer += 1
                    i += 2                      # its ReLU is fused
                elif isinstance(m, MaxPool2x2):
                    pool = True                 # folded into the next layer's converter
                    i += 1
                else:
                    raise RuntimeError('unexpected module in VGG16Features')
            if pool:
                y = _hip.maxpool2x2_nhwc(y)
        return y.permute(0, 3, 1, 2)

    def _forward_trainable(self, x):
        """same layers through autograd Functions (conv backward = dgrad on the conv kernel + im2col/GEMM wgrad,
        pool backward, activation masks): the trunk as the detector pre-training step needs it"""
        mods = list(self.children())
        y = _ConvFirstFn.apply(x, mods[0].weight, mods[0].bias)
        _tap_act('detector.features.0', y, EPI_RELU)
        i, last_conv = 2, 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, Conv3x3):
                y = _Conv3x3Fn.apply(y, m.weight, m.bias, EPI_RELU)
                _tap_act('detector.features.%d' % i, y, EPI_RELU)
                last_conv = i
                i += 2
            elif isinstance(m, MaxPool2x2):
                _tap_pool2x2('detector.features.pool%d' % last_conv, y)
                y = _MaxPool2x2Fn.apply(y)
                i += 1
            else:
                raise RuntimeError('unexpected module in VGG16Features')
        return y.permute(0, 3, 1, 2)


# =========================================================================================== generic conv (im2col)
class _ConvIm2colFn(torch.autograd.Function):
    """NHWC convolution as im2col + MFMA GEMM (used for the 2-channel 7x7/2 mask conv, whose K = 98 is too
    small for the implicit-GEMM kernel).  No input gradient (its input is a constant mask)."""

    @staticmethod
    def forward(ctx, x_nhwc, weight, bias, kh, kw, stride, pad):
        B, H, W, C = x_nhwc.shape
        K = kh * kw * C
        ldo = (K + 3) // 4 * 4
        cols, Ho, Wo = _hip.im2col_nhwc(_c(x_nhwc), kh, kw, stride, pad, ldo=ldo)
        wmat = weight.new_zeros(weight.shape[0], ldo)
        wmat[:, :K] = weight.permute(0, 2, 3, 1).reshape(weight.shape[0], K)
        y = _hip.gemm(cols, wmat, False, True, bias=bias)
        ctx.save_for_backward(cols)
        ctx.meta = (weight.shape, K)
        return y.view(B, Ho, Wo, weight.shape[0])

    @staticmethod
    def backward(ctx, gy):
        (cols,) = ctx.saved_tensors
        wshape, K = ctx.meta
        g2 = _c(gy).view(-1, wshape[0])
        gw = gb = None
        if ctx.needs_input_grad[1]:
            gwm = _hip.gemm(g2, cols, True, False)                       # [Cout, ldo]
            gw = gwm[:, :K].reshape(wshape[0], wshape[2], wshape[3], wshape[1]).permute(0, 3, 1, 2).contiguous()
        if ctx.needs_input_grad[2]:
            gb = g2.sum(0)
        return None, gw, gb, None, None, None, None


# 3x3 convolutions over MANY SMALL maps (the 1536 7x7 RoI maps of the mask tower and of the ResNet layer4 stacks) on the ring
# engine of the frozen trunk (round 5): per-map maxima (one small pass), the activation image, weights packed per call (they
# train), mh_plconv3x3 with fp32 output.  The in-loop-split conv kernel (round 2) ran these at 220-285 TFLOP/s, the ring kernel
# runs the same shapes at ~350.  Forward and input gradient (flip-transposed weights); the weight gradient stays on
# conv3x3_wgrad.  MOTIFS_CONV3X3_MAPS=inloop: the round-2 kernels (A/B).
_MAPS_ENGINE = os.environ.get('MOTIFS_CONV3X3_MAPS', 'planes')


def _small_maps_on_planes(x_nhwc, cin, cout):
    B, H, W, _ = x_nhwc.shape
    return (_MAPS_ENGINE == 'planes' and x_nhwc.is_cuda and B >= 64 and H * W <= 1024 and B * H * W >= 8192
            and _hip.plconv_many_images_ok(B, H, W, cin, cout))


def conv3x3_small_maps(x_nhwc, weight, bias, epilogue, flip_transpose=False):
    """epi(conv3x3(x, weight) + bias) (or, flip_transpose, the input-gradient convolution of that layer applied to x = dy) on the
    plane / ring engine; x [B,H,W,C] fp32 NHWC, weight [Cout,Cin,3,3]"""
    cout = weight.shape[1] if flip_transpose else weight.shape[0]
    img = _hip.act_planes(x_nhwc, _hip.image_maxbits(x_nhwc))
    return _hip.plconv3x3(img, _hip.plconv_pack_weight(_c(weight.detach()), flip_transpose), cout, bias, epilogue, None)


class _Conv3x3Fn(torch.autograd.Function):
    """Trainable NHWC 3x3 conv with the fused bias + activation epilogue: forward = implicit GEMM; backward =
    activation mask (mh_act_bwd on the saved output), dgrad = the same conv kernel on flip-transposed weights,
    wgrad = im2col(x)^T x dY on the GEMM."""

    @staticmethod
    def forward(ctx, x_nhwc, weight, bias, epilogue=EPI_NONE):
        x_nhwc = _c(x_nhwc)
        ctx.maps = _small_maps_on_planes(x_nhwc, weight.shape[1], weight.shape[0]) and weight.shape[1] >= 128
        if ctx.maps:
            y = conv3x3_small_maps(x_nhwc, weight, bias, epilogue)
        else:
            wt = _hip.conv3x3_pack_weight(_c(weight), False)
            y = _hip.conv3x3_nhwc(x_nhwc, wt, bias, epilogue)
        ctx.epilogue = epilogue
        ctx.save_for_backward(x_nhwc, weight, y if epilogue != EPI_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x_nhwc, weight, y = ctx.saved_tensors
        gy = _c(gy)
        if ctx.epilogue != EPI_NONE:
            gy = _hip.act_bwd(gy, y, ctx.epilogue)
        Cout, Cin = weight.shape[0], weight.shape[1]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if ctx.maps:
                gx = conv3x3_small_maps(gy, weight, None, EPI_NONE, flip_transpose=True)
            else:
                wt_t = _hip.conv3x3_pack_weight(_c(weight), True)          # the dgrad conv's weights (roles swapped)
                gx = _hip.conv3x3_nhwc(gy, wt_t, None, EPI_NONE)
        if ctx.needs_input_grad[1]:
            gwm = _hip.conv3x3_wgrad(x_nhwc, gy)                       # implicit GEMM over the pixels, no patch matrix
            if gwm is None:                                            # f32-MFMA build
                cols, _, _ = _hip.im2col_nhwc(x_nhwc, 3, 3, 1, 1)      # [M, 9*Cin]
                gwm = _hip.gemm(gy.view(-1, Cout), cols, True, False)  # [Cout, 9*Cin]  (tap-major, then cin)
            gw = gwm.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).contiguous()
        if ctx.needs_input_grad[2]:
            gb = gy.view(-1, Cout).sum(0)
        return gx, gw, gb, None


class _ConvFirstFn(torch.autograd.Function):
    """conv1_1 (3 -> Cout, NCHW image in, NHWC out, fused bias + ReLU) with a weight gradient: the image needs no
    gradient, dW = im2col(image)^T x dY (K = 27, padded to 28)."""

    @staticmethod
    def forward(ctx, img_nchw, weight, bias):
        img_nchw = _c(img_nchw)
        y = _hip.conv_first_nchw(img_nchw, _c(weight), bias, EPI_RELU)
        ctx.save_for_backward(img_nchw, y)
        ctx.wshape = weight.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        img, y = ctx.saved_tensors
        Cout, Cin = ctx.wshape[0], ctx.wshape[1]
        g = _hip.act_bwd(_c(gy), y, EPI_RELU)
        gw = gb = None
        if ctx.needs_input_grad[1]:
            K = 9 * Cin
            cols, _, _ = _hip.im2col_nhwc(_hip.nchw_to_nhwc(img), 3, 3, 1, 1, ldo=(K + 3) // 4 * 4)
            gwm = _hip.gemm(g.view(-1, Cout), cols, True, False)[:, :K]
            gw = gwm.reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2).contiguous()
        if ctx.needs_input_grad[2]:
            gb = g.view(-1, Cout).sum(0)
        return None, gw, gb


class _MaxPool2x2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_nhwc):
        x_nhwc = _c(x_nhwc)
        ctx.save_for_backward(x_nhwc)
        return _hip.maxpool2x2_nhwc(x_nhwc)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return _hip.maxpool2x2_bwd_nhwc(x, _c(gy))


class Conv2dNHWC(nn.Module):
    """nn.Conv2d-compatible parameters ([Cout,Cin,kh,kw], bias); forward takes / returns NHWC tensors."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0):
        super(Conv2dNHWC, self).__init__()
        self.cin, self.cout, self.k, self.stride, self.padding = cin, cout, kernel_size, stride, padding
        self.weight = nn.Parameter(torch.empty(cout, cin, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(cin * kernel_size * kernel_size)
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x_nhwc):
        if self.k == 3 and self.stride == 1 and self.padding == 1 and self.cin % 16 == 0 and self.cout % 4 == 0:
            return _Conv3x3Fn.apply(x_nhwc, self.weight, self.bias, EPI_NONE)
        return _ConvIm2colFn.apply(x_nhwc, self.weight, self.bias, self.k, self.k, self.stride, self.padding)


# =========================================================================================== RoIAlign
class _RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, rois, ph, pw, spatial_scale):
        nhwc = _is_nhwc(features)
        feat = features.permute(0, 2, 3, 1) if nhwc else _c(features)
        rois = _c(rois.detach().float())
        if rois.dim() != 2 or rois.size(1) != 5:
            raise ValueError('rois must be [n,5] (im, x1, y1, x2, y2)')     # roi_align_cuda.c:19-22 returned 0
        B, C, H, W = features.shape
        ctx.meta = (B, C, H, W, spatial_scale, nhwc)
        ctx.save_for_backward(rois)
        return _hip.roi_align_fwd(feat, rois, ph, pw, spatial_scale, nhwc)

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        B, C, H, W, scale, nhwc = ctx.meta
        gf = _hip.roi_align_bwd(_c(g), rois, B, C, H, W, scale, nhwc)
        return (gf.permute(0, 3, 1, 2) if nhwc else gf), None, None, None, None


def roi_align(features, rois, ph, pw, spatial_scale):
    return _RoIAlignFn.apply(features, rois, int(ph), int(pw), float(spatial_scale))
