"""Comparison helpers of the GPU parity tests (product on the GPU vs oracle/ on the CPU).

Gradients are compared at a TRUE relative tolerance: the bound is `rtol` times the largest magnitude of the reference
tensor itself (round 2 clamped that scale at 1.0, which turned the bound into an absolute 1e-4 that an all-zero gradient
would have passed for every tensor whose gradients are O(1e-5)).

What makes that possible is kink accounting.  A ReLU input within rounding of 0 (or two max-pool candidates within
rounding of each other) can fall on either side in two correct fp32 evaluations: the forward value does not care, the
backward mask does, and ONE such unit changes a whole row of the weight gradient below it by O(1) of that row.  So the
tests run the product first, capture ITS ReLU masks / pool arg-max table (`ProductMasks`), and make the oracle evaluate
with exactly those decisions (oracle/model.py: TAPS['force']).  The oracle reports, per forced site, how many units
differ from its own decision and how far the farthest of them is from the kink; `assert_genuine_kinks` bounds both.
With the decisions identical, every gradient must agree to rtol of its own scale -- no row is excused.
"""
import contextlib

import numpy as np


def rel_close(got, ref, rtol=1e-4, what='', own_scale=False):
    """max |got - ref| <= rtol * scale; scale = max(1, max|ref|) for logits / losses (the north star's absolute 1e-4 on
    O(1..10) values), the tensor's own max|ref| with own_scale (statistics, gradients)"""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    mag = float(np.abs(ref).max()) if ref.size else 0.0
    scale = mag if own_scale else max(1.0, mag)
    err = float(np.abs(got - ref).max()) if ref.size else 0.0
    print('%-30s max|ref|=%.4e  max abs err=%.3e  (%.2e of %s)' % (what, mag, err, err / max(scale, 1e-300),
                                                                    'own max' if own_scale else 'scale'))
    assert err <= rtol * scale, '%s: max abs err %.3e > %.1e * %.4e' % (what, err, rtol, scale)


def grad_close(got, ref, what='', rtol=1e-4, max_flipped_rows=0):
    """|got - ref| <= rtol * max|ref| (the tensor's OWN largest magnitude, no clamp), element-wise.
    max_flipped_rows > 0 (only for paths whose kink decisions are NOT forced, e.g. the detector pre-training step):
    that many rows (first index) may exceed the bound -- they are printed with their error; the rest must hold it."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, what
    mag = float(np.abs(ref).max()) if ref.size else 0.0
    if mag == 0.0:
        assert float(np.abs(got).max() if got.size else 0.0) == 0.0, '%s: reference gradient is identically 0, got is not' % what
        print('%-34s identically zero on both sides' % what)
        return
    err = np.abs(got - ref).reshape(got.shape[0], -1) if got.ndim else np.abs(got - ref).reshape(1, 1)
    bad_rows = np.nonzero((err > rtol * mag).any(1))[0]
    ok_err = float(np.delete(err, bad_rows, axis=0).max()) if len(bad_rows) < err.shape[0] else 0.0
    print('%-34s max|ref|=%.3e  max abs err=%.3e  = %.2e of own max%s' % (
        what, mag, ok_err, ok_err / mag,
        '' if not len(bad_rows) else '  + %d rows beyond the bound %s (max err %.3e = %.2e of own max)' % (
            len(bad_rows), bad_rows.tolist()[:8], err.max(), err.max() / mag)))
    assert len(bad_rows) <= max_flipped_rows, '%s: %d rows differ by more than %.1e of the tensor\'s own max %.3e (max err %.3e)' % (
        what, len(bad_rows), rtol, mag, err.max())
    assert err.max() <= 5e-2 * mag, what


def unforced_report(tag, named_grads, forced, unforced):
    """the gradient comparison AGAIN with the oracle's OWN ReLU / max-pool decisions, printed beside the
    asserted, forced one (never asserted: a unit within rounding of a kink that the two sides decide differently moves a whole
    row of a gradient by its full contribution -- that is the kink, not an arithmetic error; how rare and how close to the kink
    those units are IS asserted, by assert_genuine_kinks).  named_grads: [(name, product grad ndarray)]; forced / unforced:
    {name: oracle grad ndarray}.  Returns (worst forced, worst unforced) as fractions of each tensor's own maximum."""
    worst_f, worst_u, rows = 0.0, 0.0, []
    for name, got in named_grads:
        got = np.asarray(got, dtype=np.float64)
        out = []
        for ref in (forced[name], unforced[name]):
            ref = np.asarray(ref, dtype=np.float64)
            mag = float(np.abs(ref).max()) if ref.size else 0.0
            out.append(float(np.abs(got - ref).max()) / mag if mag > 0.0 else 0.0)
        worst_f, worst_u = max(worst_f, out[0]), max(worst_u, out[1])
        rows.append((name, out[0], out[1]))
    print('%s: error of every trainable gradient as a fraction of its own maximum -- oracle with the product\'s kink decisions '
          '(asserted above) | oracle with its OWN decisions (printed only)' % tag)
    for name, f, u in rows:
        print('  %-44s forced %.2e | un-forced %.2e%s' % (name[-44:], f, u, '   <- differs' if u > 2.0 * max(f, 1e-7) else ''))
    print('%s: worst of %d gradients: forced %.2e, un-forced %.2e of own max' % (tag, len(rows), worst_f, worst_u))
    return worst_f, worst_u


class ProductMasks(object):
    """Context manager: captures the product's ReLU masks (forward hooks on the Linear / ReLU modules; the fused tower
    reports through lib.get_union_boxes.TAPS) during the forward passes run inside it.  .force = {site name: CPU tensor}
    in the oracle's naming (state-dict prefix of the layer in front of the ReLU)."""

    def __init__(self, model, extra_sites=None, nhwc_sites=None):
        self.model = model
        self.extra_sites = dict(extra_sites or {})        # {oracle site name: module whose OUTPUT is the activation}
        self.nhwc_sites = dict(nhwc_sites or {})          # the same for modules whose output is [n, h, w, c] (oracle: [n, c, h, w])
        self.force = {}
        self._handles = []

    def _hook(self, name, nhwc=False):
        def fn(_mod, _inp, out):
            m = out.detach() > 0
            self.force[name] = (m.permute(0, 3, 1, 2) if nhwc else m).cpu()
        return fn

    def __enter__(self):
        import lib.get_union_boxes as GUB
        m = self.model
        sites = []
        if hasattr(m, 'roi_fmap') and hasattr(m, 'union_boxes'):          # RelModel: Sequential(UnionBoxesAndFeats, FCStack)
            try:
                sites.append(('roi_fmap.1.0', m.roi_fmap[1][0]))
            except (TypeError, IndexError):
                pass
        if hasattr(m, 'roi_fmap_obj') and not getattr(m, 'use_resnet', False):      # VGG fc6 / fc7 (the ResNet stacks: nhwc_sites)
            sites += [('roi_fmap_obj.0', m.roi_fmap_obj[0]), ('roi_fmap_obj.3', m.roi_fmap_obj[3])]
        if hasattr(m, 'context') and hasattr(m.context, 'pos_embed'):
            sites.append(('context.pos_embed.1', m.context.pos_embed[2]))
        sites += list(self.extra_sites.items())
        for name, mod in sites:
            self._handles.append(mod.register_forward_hook(self._hook(name)))
        for name, mod in self.nhwc_sites.items():
            self._handles.append(mod.register_forward_hook(self._hook(name, nhwc=True)))
        import lib.hip_ops as HO
        self._gub, self._ho = GUB, HO
        self._tower, self._trunk = {}, {}
        GUB.TAPS = self._tower
        HO.TAPS = self._trunk                             # trainable trunk / RPN head (detector pre-training)
        return self

    def __exit__(self, *exc):
        for h in self._handles:
            h.remove()
        self._gub.TAPS = None
        self._ho.TAPS = None
        self.force.update(self._tower)
        self.force.update(self._trunk)
        return False


@contextlib.contextmanager
def oracle_forced(force):
    """run the oracle with the product's kink decisions; yields the TAPS dict (flips are in ['flips'] afterwards)"""
    from oracle import model as OM
    taps = {'force': dict(force)}
    OM.TAPS = taps
    try:
        yield taps
    finally:
        OM.TAPS = None


def assert_genuine_kinks(taps, max_frac=2e-5, max_far=2e-6):
    """every unit where the product decided differently from the oracle must sit within rounding of the kink:
    |pre-activation| <= max_far * max|pre-activation| of its tensor (pool: the two candidates differ by that little),
    and such units must be rare (max_frac of the site's units, at least 4 allowed)"""
    flips = taps.get('flips', {})
    assert set(flips) >= set(taps['force']), 'forced sites the oracle never visited: %s' % (set(taps['force']) - set(flips))
    for name, (n, far, numel) in sorted(flips.items()):
        where = ''
        if n and 'mask' in taps and name in taps['mask'] and name in taps['force']:
            own, forced = taps['mask'][name], taps['force'][name].reshape(taps['mask'][name].shape)
            idx = (own != forced).nonzero()[:4].tolist()
            where = '  at %s (product says %s)' % (idx, [bool(forced[tuple(i)]) for i in idx])
        print('kink site %-24s %8d units, %3d decided differently by product and oracle (farthest: %.2e of max)%s' % (
            name, numel, n, far, where))
        assert n <= max(4, max_frac * numel), '%s: %d of %d units differ' % (name, n, numel)
        assert far <= max_far, '%s: a differing unit lies %.2e of the tensor max away from the kink' % (name, far)


def resnet_trunk_gradients(device, seed=5, size=64, batch=2):
    """Detector pre-training with the ResNet-101 trunk (models/train_detector.py -resnet): loss = <compress(trunk(x)), G> for a
    fixed random G, through the PRODUCT (lib.resnet.ResNet101Trunk + lib.object_detector.ResNetCompress, train mode, all
    parameters trainable) and through the oracle's differentiable restatement in float32 and float64.
    Returns (names, product grads, oracle fp32 grads, oracle fp64 grads, product output, fp32 output, fp64 output) as numpy."""
    import torch
    from lib.object_detector import ResNetCompress
    from lib.resnet import ResNet101Trunk
    from oracle import model as OM
    torch.manual_seed(seed)
    trunk, comp = ResNet101Trunk(), ResNetCompress()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in list(trunk.named_parameters()) + list(comp.named_parameters()):
            if n.endswith('bn3.weight') or 'downsample.1.weight' in n:
                p.copy_(0.25 + 0.5 * torch.rand(p.shape, generator=g))      # keep 33 residual blocks from blowing the scale up
            elif p.dim() == 1 and n.endswith('weight'):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif p.dim() == 1:
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
    sd = {'features.' + k: v.detach().clone() for k, v in trunk.state_dict().items()}
    sd.update({'compress.' + k: v.detach().clone() for k, v in comp.state_dict().items()})
    x = torch.randn(batch, 3, size, size, generator=g)
    G = torch.randn(batch, 256, size // 16, size // 16, generator=g)
    trunk.to(device).train()
    comp.to(device).train()
    out = comp(trunk(x.to(device)))
    (out * G.to(device)).sum().backward()
    names = ['features.' + n for n, _ in trunk.named_parameters()] + ['compress.' + n for n, _ in comp.named_parameters()]
    got = [p.grad.detach().cpu().numpy() for p in list(trunk.parameters()) + list(comp.parameters())]
    assert all(p.grad is not None for p in list(trunk.parameters()) + list(comp.parameters()))

    def oracle(dt):
        s = {k: (v.detach().to(dt).clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        for k in names:
            s[k].requires_grad_(True)
        o = OM.resnet_compress(s, OM.resnet_features(s, x.to(dt), True, prefix='features.'), True, prefix='compress.')
        (o * G.to(dt)).sum().backward()
        return [s[k].grad.numpy().astype(np.float64) for k in names], o.detach().numpy().astype(np.float64)
    g32, o32 = oracle(torch.float32)
    g64, o64 = oracle(torch.float64)
    return names, got, g32, g64, out.detach().cpu().numpy(), o32, o64


def resnet_piece_gradients(device, seed=9):
    """the pieces of the trainable ResNet trunk one by one, each on its own random input (a chain of 33 random residual blocks
    amplifies fp32 noise to percents, which says nothing about a kernel): the stem (7x7/2 conv as im2col + product, BN + ReLU +
    3x3/2 max-pool), a bottleneck with a stride-2 3x3 conv and a strided 1x1 projection (layer2.0), one with a projection at
    stride 1 (layer1.0), a plain one (layer3.1), the compress head.  Yields (what, name, product grad, fp32 oracle grad,
    float64 oracle grad); inputs that carry a gradient are reported as 'input'."""
    import torch
    import torch.nn.functional as F
    from lib.object_detector import ResNetCompress
    from lib.resnet import ResNet101Trunk
    from oracle import model as OM
    torch.manual_seed(seed)
    trunk, comp = ResNet101Trunk(), ResNetCompress()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in list(trunk.named_parameters()) + list(comp.named_parameters()):
            if p.dim() == 1:
                p.copy_(0.5 + torch.rand(p.shape, generator=g) if n.endswith('weight') else 0.2 * torch.randn(p.shape, generator=g))
    sd0 = {'features.' + k: v.detach().clone() for k, v in trunk.state_dict().items()}
    sd0.update({'compress.' + k: v.detach().clone() for k, v in comp.state_dict().items()})
    trunk.to(device).train()
    comp.to(device).train()

    def oracle_sd(dt, names):
        s = {k: (v.detach().to(dt).clone() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        for k in names:
            s[k].requires_grad_(True)
        return s

    def run(what, names, x_nchw, x_grad, product, oracle):
        G = None
        outs = []
        xp = x_nchw.clone().to(device).requires_grad_(x_grad)
        out = product(xp)                                    # logical NCHW
        G = torch.randn(out.shape, generator=g)
        for p in list(trunk.parameters()) + list(comp.parameters()):
            p.grad = None
        (out * G.to(device)).sum().backward()
        pmap = dict([('features.' + n, p) for n, p in trunk.named_parameters()] + [('compress.' + n, p) for n, p in comp.named_parameters()])
        got = {k: pmap[k].grad.detach().cpu().numpy().astype(np.float64) for k in names}
        if x_grad:
            got['input'] = xp.grad.detach().cpu().numpy().astype(np.float64)
        refs = []
        for dt in (torch.float32, torch.float64):
            s = oracle_sd(dt, names)
            xo = x_nchw.to(dt).clone().requires_grad_(x_grad)
            o = oracle(s, xo)
            (o * G.to(dt)).sum().backward()
            r = {k: s[k].grad.numpy().astype(np.float64) for k in names}
            if x_grad:
                r['input'] = xo.grad.numpy().astype(np.float64)
            r['output'] = o.detach().numpy().astype(np.float64)
            refs.append(r)
        got['output'] = out.detach().cpu().numpy().astype(np.float64)
        for k in got:
            outs.append((what, k, got[k], refs[0][k], refs[1][k]))
        return outs

    def nhwc_block(block):
        return lambda xp: block(xp.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2)

    def stem_oracle(s, x):
        y = F.conv2d(x, s['features.conv1.weight'], None, stride=2, padding=3)
        return F.max_pool2d(F.relu(OM._bn(s, y, 'features.bn1.', True)), 3, 2, 1)

    def block_names(p):
        return [k for k in sd0 if k.startswith(p) and not k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))]

    res = []
    res += run('stem', ['features.conv1.weight', 'features.bn1.weight', 'features.bn1.bias'], torch.randn(2, 3, 96, 96, generator=g), False,
               lambda xp: trunk.stem(xp).permute(0, 3, 1, 2), stem_oracle)
    for name, cin, side, stride in (('layer2.0', 256, 24, 2), ('layer1.0', 64, 24, 1), ('layer3.1', 1024, 12, 1)):
        block = trunk
        for part in name.split('.'):
            block = block[int(part)] if part.isdigit() else getattr(block, part)
        p = 'features.%s.' % name
        res += run(name, block_names(p), torch.randn(2, cin, side, side, generator=g), True, nhwc_block(block),
                   lambda s, x, p=p, stride=stride: OM.resnet_bottleneck(s, x, p, stride, True))
    res += run('compress', block_names('compress.'), torch.randn(2, 1024, 12, 12, generator=g), True, comp,
               lambda s, x: OM.resnet_compress(s, x, True, prefix='compress.'))
    return res


def assert_resnet_piece_gradients(device, what):
    """every output / gradient of every piece against the float64 oracle, as rel-rms: within max(2e-4, 3 x the fp32 oracle's own
    distance) -- except where ONE ReLU input sits within rounding of zero and flips in one of the evaluations: that moves the
    gradients of its channel (BN scale / shift, the conv row in front, the input) by up to percents of rms (seen: 5e-3).  So all
    tensors must be within 5e-2, and at most 15 % of them may miss the tight bound (a wrong kernel or a wrong piece of autograd
    plumbing is an O(1) distance in most tensors of its piece)"""
    n, loose = 0, []
    worst = (0.0, 0.0, '')
    for piece, name, got, r32, r64 in resnet_piece_gradients(device):
        assert got.shape == r64.shape and np.isfinite(got).all(), (piece, name)
        rms = float(np.sqrt((r64 ** 2).mean())) + 1e-30
        e, f = float(np.sqrt(((got - r64) ** 2).mean())) / rms, float(np.sqrt(((r32 - r64) ** 2).mean())) / rms
        assert e <= 5e-2, '%s %s %s: rel-rms %.3e from float64 (fp32 oracle %.3e)' % (what, piece, name, e, f)
        if e > max(2e-4, 3 * f):
            loose.append('%s %s %.1e' % (piece, name, e))
        worst = max(worst, (e, f, piece + ' ' + name))
        n += 1
    assert len(loose) <= 0.15 * n, loose
    print('%s: %d tensors, %d beside a flipped ReLU %s; worst rel-rms %.2e from float64 (fp32 oracle %.2e): %s' % (
        what, n, len(loose), loose[:4], worst[0], worst[1], worst[2]))
    return n


def assert_resnet_trunk_gradients(device, what):
    """the whole chain (conv1 .. layer3 + compress): every parameter receives a gradient of the oracle's shape, and the product is
    as far from the float64 gradient as the fp32 ORACLE is -- a 33-block random residual net amplifies fp32 summation-order noise
    and ReLU flips to percents (medians of the per-tensor rel-rms distances are compared; a plumbing error is an O(1) distance)"""
    names, got, g32, g64, out, o32, o64 = resnet_trunk_gradients(device)
    so = float(np.abs(o64).max())
    eo, fo = float(np.abs(out - o64).max()) / so, float(np.abs(o32 - o64).max()) / so
    assert eo <= max(2e-3, 3 * fo), '%s output: %.3e of scale from float64 (fp32 oracle %.3e)' % (what, eo, fo)
    E, Fo = [], []
    for n, a, b, c in zip(names, got, g32, g64):
        assert a.shape == c.shape and np.isfinite(a).all(), n
        rms = float(np.sqrt((c ** 2).mean())) + 1e-30
        E.append(float(np.sqrt(((a - c) ** 2).mean())) / rms)
        Fo.append(float(np.sqrt(((b - c) ** 2).mean())) / rms)
    assert max(E) <= max(0.2, 3 * max(Fo)), '%s: a gradient is %.2e (rel-rms) from float64; the fp32 oracle\'s worst is %.2e' % (what, max(E), max(Fo))
    assert np.median(E) <= max(1e-3, 2 * np.median(Fo)), (np.median(E), np.median(Fo))
    print('%s: %d gradients, output %.2e of scale from float64 (fp32 oracle %.2e); rel-rms of the gradients: median %.2e (fp32 oracle %.2e), '
          'worst %.2e (%.2e)' % (what, len(names), eo, fo, np.median(E), np.median(Fo), max(E), max(Fo)))
    return len(names)
