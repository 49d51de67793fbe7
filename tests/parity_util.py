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
