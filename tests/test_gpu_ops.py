"""
GPU parity tests proper: every HIP operator, called through the C ABI (lib/_hip.py -> libmotifs_hip.so),
against the CPU oracle on the same seeded inputs.

Bars: bit-exact for NMS keep lists, RoIAlign, union-box masks, fp32 IoU (integer / index / border-decision work);
1e-4 absolute for fp32 GEMM / conv / LSTM arithmetic (BASELINE.json north_star).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a HIP device (there is no CPU fallback for the hot path)')
    from lib import _hip
    _hip.lib()
    return _hip


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def rand_boxes(rs, n, hi=591.0):
    x1 = rs.uniform(0, hi - 20, n)
    y1 = rs.uniform(0, hi - 20, n)
    w = rs.uniform(2, 250, n)
    h = rs.uniform(2, 250, n)
    return np.stack([x1, y1, np.minimum(x1 + w, hi), np.minimum(y1 + h, hi)], 1).astype(np.float32)


# ----------------------------------------------------------------------------------------------- NMS
@pytest.mark.parametrize('n,thresh', [(0, 0.5), (1, 0.5), (2, 0.3), (63, 0.3), (64, 0.7), (65, 0.5), (129, 0.3),
                                      (1000, 0.3), (2500, 0.5), (6000, 0.7), (6300, 0.7), (12000, 0.7), (6000, -0.7), (6300, -0.3)])
def test_nms_bit_exact(hip, n, thresh):
    """sizes around every boundary of the sweep (csrc/exact_ops.hip nms_sweep_kernel): 64-box block-rows, the 32-block-row stages of
    the diagonal words (2048 boxes), the 96-column register chunks (6208 boxes); a negative threshold = the same test on a
    CROWDED scene (boxes within 120 px: whole block-rows without a kept box)"""
    from oracle import native
    rs = np.random.RandomState(100 + n)
    boxes = rand_boxes(rs, n) if thresh > 0 else rand_boxes(rs, n, hi=120.0)
    thresh = abs(thresh)
    if n >= 64:
        boxes[5] = boxes[3]                 # exact duplicates (IoU == 1)
        boxes[40:44] = boxes[40] + np.float32(0.5)
    keep, num = hip.nms(dev(boxes).view(-1, 4), thresh)
    k = int(num.item())
    ref = native.nms(boxes, thresh)
    assert k == len(ref)
    np.testing.assert_array_equal(keep[:k].cpu().numpy(), ref)


def test_nms_thresholds_at_the_boundary(hip):
    """IoU exactly equal to the threshold must NOT suppress (strict >), as in nms_kernel.cu:61."""
    from oracle import native
    boxes = np.array([[0, 0, 9, 9], [0, 5, 9, 14], [100, 100, 119, 119], [100, 110, 119, 129]], np.float32)
    # IoU(0,1) = 50/150 = 1/3 ; IoU(2,3) = 200/600 = 1/3
    for thr in (np.float32(1.0) / np.float32(3.0), 0.3333, 0.33334):
        keep, num = hip.nms(dev(boxes), float(thr))
        ref = native.nms(boxes, float(thr))
        np.testing.assert_array_equal(keep[:int(num.item())].cpu().numpy(), ref)


def test_nms_batched_matches_per_segment(hip):
    from oracle import native
    rs = np.random.RandomState(7)
    seg_lens = [300, 0, 64, 1, 517, 300]
    offs = np.concatenate([[0], np.cumsum(seg_lens)]).astype(np.int32)
    boxes = rand_boxes(rs, int(offs[-1]), hi=300.0)
    keep, num = hip.nms_batched(dev(boxes).view(-1, 4), dev(offs, torch.int32), max(seg_lens), 0.3)
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    for s, L in enumerate(seg_lens):
        ref = native.nms(boxes[offs[s]:offs[s + 1]], 0.3)
        assert num[s] == len(ref)
        np.testing.assert_array_equal(keep[offs[s]:offs[s] + num[s]], ref)


# ----------------------------------------------------------------------------------------------- RoIAlign
def _rois(rs, n, B):
    b = rand_boxes(rs, n)
    b[0] = [0, 0, 591, 591]            # spills over the last row/col -> zero fill
    b[1] = [575.5, 10, 576.5, 400]     # straddles the x border decision in_x > W-1
    b[2] = [16, 16, 16, 16]            # degenerate: every bin samples the same point
    im = rs.randint(0, B, n).astype(np.float32)
    return np.concatenate([im[:, None], b], 1).astype(np.float32)


@pytest.mark.parametrize('C', [7, 64, 512])
def test_roi_align_fwd_bit_exact(hip, C):
    from oracle import native
    rs = np.random.RandomState(C)
    B = 3
    feat = rs.randn(B, C, 37, 37).astype(np.float32)
    rois = _rois(rs, 37, B)
    rois[5, 0] = 9                      # image index out of range -> zeros
    ref = native.roi_align_fwd(feat, rois)
    out_nchw = hip.roi_align_fwd(dev(feat), dev(rois), 7, 7, 1.0 / 16, nhwc=False)
    np.testing.assert_array_equal(out_nchw.cpu().numpy(), ref)
    feat_nhwc = dev(feat.transpose(0, 2, 3, 1))
    out_nhwc = hip.roi_align_fwd(feat_nhwc, dev(rois), 7, 7, 1.0 / 16, nhwc=True)
    np.testing.assert_array_equal(out_nhwc.cpu().numpy(), ref)


def test_roi_align_bwd(hip):
    from oracle import native
    rs = np.random.RandomState(5)
    B, C = 2, 24
    rois = _rois(rs, 11, B)
    g = rs.randn(11, C, 7, 7).astype(np.float32)
    ref = native.roi_align_bwd(g, rois, (B, C, 37, 37))
    out = hip.roi_align_bwd(dev(g), dev(rois), B, C, 37, 37, 1.0 / 16, nhwc=False)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=1e-5)          # atomic order differs
    out2 = hip.roi_align_bwd(dev(g), dev(rois), B, C, 37, 37, 1.0 / 16, nhwc=True)
    np.testing.assert_allclose(out2.permute(0, 3, 1, 2).cpu().numpy(), ref, atol=1e-5)


def test_roi_align_bwd_gather_is_deterministic_and_equals_the_scatter(hip, monkeypatch):
    """the default backward is the gather kernel (fixed accumulation order): bit-identical run to run, equal to the
    reference-style atomicAdd scatter up to summation order, incl. heavily overlapping RoIs, RoIs outside the map, an
    out-of-range image index and 600 channels (3 channel passes per thread)"""
    from oracle import native
    rs = np.random.RandomState(8)
    B, C, n = 3, 600, 200
    rois = _rois(rs, n, B)
    rois[:40, 1:] = rois[0, 1:] + rs.uniform(-3, 3, (40, 4)).astype(np.float32)      # a pile of near-duplicates
    rois[:40, 0] = rois[0, 0]
    rois[41] = [1, -300, -300, -100, -50]                                           # entirely outside
    rois[42, 0] = 7                                                                  # image index out of range
    g = rs.randn(n, C, 7, 7).astype(np.float32)
    a = hip.roi_align_bwd(dev(g), dev(rois), B, C, 37, 37, 1.0 / 16, nhwc=True)
    for _ in range(3):
        b = hip.roi_align_bwd(dev(g), dev(rois), B, C, 37, 37, 1.0 / 16, nhwc=True)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    monkeypatch.setenv('MOTIFS_ROIALIGN_BWD', 'atomic')
    s = hip.roi_align_bwd(dev(g), dev(rois), B, C, 37, 37, 1.0 / 16, nhwc=True)
    monkeypatch.delenv('MOTIFS_ROIALIGN_BWD')
    scale = float(s.abs().max())
    assert float((a - s).abs().max()) <= 2e-6 * scale
    ref = native.roi_align_bwd(g[:, :24].copy(), rois, (B, 24, 37, 37))
    np.testing.assert_allclose(a.permute(0, 3, 1, 2)[:, :24].cpu().numpy(), ref, atol=1e-5 * max(1.0, scale))
    nchw = hip.roi_align_bwd(dev(g), dev(rois), B, C, 37, 37, 1.0 / 16, nhwc=False)
    assert torch.equal(nchw.permute(0, 2, 3, 1).contiguous().view(torch.int32), a.view(torch.int32))


# ----------------------------------------------------------------------------------------------- masks / IoU
@pytest.mark.parametrize('P', [27, 7])
def test_draw_union_boxes_matches_reference_golden(hip, golden, P):
    g = golden('draw_P%d' % P)
    out = hip.draw_union_boxes(dev(g['pairs']), P)
    np.testing.assert_array_equal(out.cpu().numpy(), g['masks'])
    out_cl = hip.draw_union_boxes(dev(g['pairs']), P, offset=-0.5, channels_last=True)
    np.testing.assert_array_equal(out_cl.permute(0, 3, 1, 2).cpu().numpy(), g['masks'] - np.float32(0.5))


def test_draw_union_boxes_random_vs_oracle(hip):
    from oracle import native
    rs = np.random.RandomState(1)
    pairs = np.concatenate([rand_boxes(rs, 700), rand_boxes(rs, 700)], 1)
    out = hip.draw_union_boxes(dev(pairs), 27)
    np.testing.assert_array_equal(out.cpu().numpy(), native.draw_union_boxes(pairs, 27))


def test_bbox_overlaps_matches_reference_golden(hip, golden):
    g = golden('box_utils')
    out = hip.bbox_overlaps(dev(g['boxes']), dev(g['boxes_b']))
    np.testing.assert_array_equal(out.cpu().numpy(), g['bbox_overlaps'])


# ----------------------------------------------------------------------------------------------- GEMM
GEMM_CASES = [
    # M, N, K, ta, tb
    (128, 128, 64, 0, 1), (120, 4096, 1024, 0, 1), (1, 151, 512, 0, 1), (257, 130, 100, 0, 1),
    (120, 3072, 4424, 0, 0), (37, 51, 4096, 0, 1), (300, 64, 48, 0, 0), (64, 200, 151, 0, 0),
    (200, 600, 120, 1, 0), (151, 512, 96, 1, 0), (98, 256, 1000, 1, 0), (70, 90, 33, 1, 1), (513, 384, 256, 0, 0),
]


@pytest.mark.parametrize('M,N,K,ta,tb', GEMM_CASES)
def test_gemm_against_fp64(hip, M, N, K, ta, tb):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g)
    b = torch.randn((N, K) if tb else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    ref = ((a.t() if ta else a).double() @ (b.t() if tb else b).double())
    out = hip.gemm(a.cuda(), b.cuda(), bool(ta), bool(tb))
    # N(0,1) operands: rms(C) = sqrt(K).  The largest error over the six-million-entry products of tools/pl_check.cpp is 5.0e-6 of
    # rms(C) on either engine (profiles/r03_pl_check.jsonl: max_rel); the bound is four times that (round 2 allowed 6.4e-5)
    tol = (2e-5 * K ** 0.5 + 1e-5) / 8
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().numpy(), atol=tol * 8)
    # asymmetric operands: a transposed C-write would show as O(1) error
    out2 = hip.gemm(a.cuda(), b.cuda(), bool(ta), bool(tb), bias=bias.cuda(), epilogue=1)
    np.testing.assert_allclose(out2.cpu().numpy(), torch.relu(ref + bias.double()).float().numpy(), atol=tol * 8)
    for sk in (2, 5):
        out3 = hip.gemm(a.cuda(), b.cuda(), bool(ta), bool(tb), bias=bias.cuda(), splitk=sk)
        np.testing.assert_allclose(out3.cpu().numpy(), (ref + bias.double()).float().numpy(), atol=tol * 8)
    acc = torch.randn(M, N, generator=g)
    out4 = acc.clone().cuda()
    hip.gemm(a.cuda(), b.cuda(), bool(ta), bool(tb), out=out4, accumulate=True)
    np.testing.assert_allclose(out4.cpu().numpy(), (ref + acc.double()).float().numpy(), atol=tol * 8)


def test_gemm_small_integers_are_exact(hip):
    """Every partial product and partial sum is representable: the result must be exact (either MFMA evaluation)."""
    g = torch.Generator().manual_seed(0)
    a = torch.randint(-8, 9, (130, 96), generator=g).float()
    b = torch.randint(-8, 9, (70, 96), generator=g).float()
    out = hip.gemm(a.cuda(), b.cuda(), False, True)
    np.testing.assert_array_equal(out.cpu().numpy(), (a.double() @ b.double().t()).float().numpy())


def test_gemm_copies_23_bit_operands_exactly(hip):
    """A selection matrix (one power of two per column) must copy operand values through the matrix cores bit for bit, from
    either operand side -- for values of up to 23 significant bits: the two-term f16 split carries 11 + 11 bits plus the
    sign of the second term (22-23 bits), not fp32's full 24 (the bf16x6 engine of round 1, which
    reconstructed all 24, is gone)."""
    g = torch.Generator().manual_seed(1)
    a = (torch.randint(-2 ** 23, 2 ** 23, (200, 150), generator=g) | 1).float() * 2.0 ** -11
    sel = torch.zeros(150, 90)
    cols = torch.randint(0, 150, (90,), generator=g)
    pw = 2.0 ** torch.randint(-6, 7, (90,), generator=g).float()
    sel[cols, torch.arange(90)] = pw
    out = hip.gemm(a.cuda(), sel.cuda(), False, False)
    np.testing.assert_array_equal(out.cpu().numpy(), (a[:, cols] * pw).numpy())
    out = hip.gemm(sel.cuda(), a.cuda(), True, True)                   # [90,150] x [200,150]^T
    np.testing.assert_array_equal(out.cpu().numpy(), (a[:, cols] * pw).t().numpy())


def test_gemm_error_is_fp32_rounding(hip):
    """Error against fp64 relative to the rms result, operands with a wide dynamic range inside a row: at the level
    of an fp32 fma chain (measured 1.7e-6 max / 2.7e-7 rms for both MFMA evaluations; rocBLAS fp32: 1.1e-5 / 1.1e-6)."""
    g = torch.Generator().manual_seed(2)
    a = torch.randn(512, 4096, generator=g)
    b = torch.randn(384, 4096, generator=g)
    a[:, ::7] *= 1e-3
    b[:, ::5] *= 1e4
    ref = a.double() @ b.double().t()
    err = (hip.gemm(a.cuda(), b.cuda(), False, True).cpu().double() - ref)
    rms = ref.pow(2).mean().sqrt().item()
    assert err.abs().max().item() / rms < 6e-6
    assert err.pow(2).mean().sqrt().item() / rms < 8e-7


def test_gemm_strided_rows(hip):
    g = torch.Generator().manual_seed(3)
    big = torch.randn(40, 700, generator=g).cuda()
    a = big[:, 100:612]                      # lda = 700, K = 512
    w = torch.randn(96, 512, generator=g).cuda()
    out = hip.gemm(a, w, False, True)
    np.testing.assert_allclose(out.cpu().numpy(), (a.cpu().double() @ w.cpu().double().t()).float().numpy(), atol=2e-4)


# ----------------------------------------------------------------------------------------------- small-product engine
# (csrc/gemm.hip, mh_gemm_small_f32: bf16x6, no row-maxima pass, split-K reduced by the last block of a tile inside the launch)
SMALL_CASES = [
    # M, N, K, ta, tb -- the small products of a cfg2 step (profiles/r04_gemm_shapes.jsonl) + ragged / unaligned ones
    (120, 4096, 4096, 0, 1), (120, 512, 8192, 0, 0), (8192, 512, 120, 1, 0), (120, 151, 512, 0, 1), (152, 100, 3072, 0, 0),
    (1536, 51, 4096, 0, 1), (51, 4096, 1536, 1, 0), (120, 4, 128, 0, 0), (120, 200, 151, 0, 1), (130, 129, 1003, 1, 1),
    (120, 3072, 4424, 0, 0), (4424, 3072, 120, 1, 0), (37, 51, 4099, 0, 1),
]


def _counters_of(hip, device):
    key = (str(device), 'gemm_counters', torch.cuda.current_stream().cuda_stream)
    return hip._ws_cache[key].view(torch.int32)


@pytest.mark.parametrize('M,N,K,ta,tb', SMALL_CASES)
def test_small_product_engine_is_one_launch_and_equals_the_two_launch_form(hip, M, N, K, ta, tb):
    g = torch.Generator().manual_seed(M * 5 + N * 11 + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g).cuda()
    b = torch.randn((N, K) if tb else (K, N), generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda()
    ref = ((a.t() if ta else a).double() @ (b.t() if tb else b).double()).cpu()
    tol = 2e-5 * K ** 0.5 + 1e-5
    out = hip.gemm_inloop(a, b, bool(ta), bool(tb), bias=bias, epilogue=1)
    np.testing.assert_allclose(out.cpu().numpy(), torch.relu(ref + bias.cpu().double()).float().numpy(), atol=tol)
    assert int(_counters_of(hip, a.device).abs().sum()) == 0, 'the arrival counters were not left zero'
    # the fused reduction adds the K slices in slice order: bit-identical to the separate reduce launch over the same split
    sk = hip.lib().mh_gemm_auto_splitk_v2(M, N, K)
    sep = hip.gemm(a, b, bool(ta), bool(tb), bias=bias, epilogue=1, splitk=sk)       # mh_gemm_f32 -> mh_gemm_f32_v2: no counters
    assert torch.equal(out, sep), 'fused and separate split-K reductions differ (split %d)' % sk
    # ... and reproducible whatever order the slices arrive in
    for _ in range(3):
        assert torch.equal(hip.gemm_inloop(a, b, bool(ta), bool(tb), bias=bias, epilogue=1), out)
    # accumulate into an existing C (the LSTM weight gradients): C += A.B
    acc = torch.randn(M, N, generator=g).cuda()
    out2 = acc.clone()
    hip.gemm_inloop(a, b, bool(ta), bool(tb), out=out2, accumulate=True)
    np.testing.assert_allclose(out2.cpu().numpy(), (ref + acc.cpu().double()).float().numpy(), atol=tol)
    assert int(_counters_of(hip, a.device).abs().sum()) == 0


def test_small_product_engine_copies_24_bit_operands_exactly(hip):
    """three bf16 terms carry all 24 significant bits of an fp32 number: a selection matrix copies operand values through the
    matrix cores bit for bit, from either side (the f16x3 engine of the big products carries 22-23)"""
    g = torch.Generator().manual_seed(5)
    a = (torch.randint(-2 ** 24 + 1, 2 ** 24, (200, 150), generator=g) | 1).float() * 2.0 ** -11
    sel = torch.zeros(150, 90)
    cols = torch.randint(0, 150, (90,), generator=g)
    pw = 2.0 ** torch.randint(-6, 7, (90,), generator=g).float()
    sel[cols, torch.arange(90)] = pw
    out = hip.gemm_inloop(a.cuda(), sel.cuda(), False, False)
    np.testing.assert_array_equal(out.cpu().numpy(), (a[:, cols] * pw).numpy())
    out = hip.gemm_inloop(sel.cuda(), a.cuda(), True, True)
    np.testing.assert_array_equal(out.cpu().numpy(), (a[:, cols] * pw).t().numpy())
    # a wide dynamic range inside a row needs no scale: rows of 1e-20 .. 1e+10 magnitudes multiply like fp32
    x = torch.randn(64, 256, generator=g) * (10.0 ** torch.randint(-20, 10, (64, 1), generator=g).float())
    w = torch.randn(96, 256, generator=g) * (10.0 ** torch.randint(-10, 10, (96, 1), generator=g).float())
    ref = x.double() @ w.double().t()
    got = hip.gemm_inloop(x.cuda(), w.cuda(), False, True).cpu().double()
    scale = (x.double().abs() @ w.double().abs().t())
    assert ((got - ref).abs() / scale).max().item() < 2e-6


def test_small_product_engine_error_is_fp32_rounding(hip):
    g = torch.Generator().manual_seed(2)
    a = torch.randn(120, 4096, generator=g)
    b = torch.randn(3072, 4096, generator=g)
    a[:, ::7] *= 1e-3
    b[:, ::5] *= 1e4
    ref = a.double() @ b.double().t()
    err = hip.gemm_inloop(a.cuda(), b.cuda(), False, True).cpu().double() - ref
    rms = ref.pow(2).mean().sqrt().item()
    assert err.abs().max().item() / rms < 6e-6
    assert err.pow(2).mean().sqrt().item() / rms < 8e-7
    pos = hip.gemm_inloop(a.abs().cuda(), b.abs().cuda(), False, True).cpu().double() - a.abs().double() @ b.abs().double().t()
    rel = pos / (a.abs().double() @ b.abs().double().t())
    assert abs(rel.mean().item()) < 1e-6 and rel.abs().max().item() < 6e-6       # no one-sided bias (round to nearest split)


def test_small_products_on_two_streams_keep_their_own_counters(hip):
    """products of the two branches of a step run concurrently on two HIP streams: each stream has its own arrival counters
    and partial-sum workspace, results are bit-identical to the sequential ones"""
    g = torch.Generator().manual_seed(9)
    a1, b1 = torch.randn(120, 4096, generator=g).cuda(), torch.randn(4096, 4096, generator=g).cuda()
    a2, b2 = torch.randn(152, 3072, generator=g).cuda(), torch.randn(3072, 100, generator=g).cuda()
    r1, r2 = hip.gemm_inloop(a1, b1, False, True), hip.gemm_inloop(a2, b2, False, False)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for _ in range(20):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            o2 = [hip.gemm_inloop(a2, b2, False, False) for _ in range(4)]
        o1 = [hip.gemm_inloop(a1, b1, False, True) for _ in range(4)]
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert all(torch.equal(o, r1) for o in o1) and all(torch.equal(o, r2) for o in o2)


# ----------------------------------------------------------------------------------------------- conv stack
@pytest.mark.parametrize('B,H,W,Cin,Cout,epi', [(2, 9, 11, 16, 32, 1), (1, 37, 37, 64, 128, 1), (2, 14, 14, 64, 64, 0),
                                                (1, 20, 23, 128, 120, 2), (3, 7, 7, 256, 512, 1)])
def test_conv3x3_nhwc(hip, B, H, W, Cin, Cout, epi):
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = {0: ref, 1: F.relu(ref), 2: F.relu6(ref)}[epi]
    wt = hip.conv3x3_pack_weight(w.cuda())
    out = hip.conv3x3_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), wt, b.cuda(), epi)
    np.testing.assert_allclose(out.permute(0, 3, 1, 2).cpu().numpy(), ref.float().numpy(), atol=1e-4)


def test_conv3x3_dgrad_via_flipped_weights(hip):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 32, 7, 7, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(48, 32, 3, 3, generator=g, dtype=torch.float64) * 0.1
    y = F.conv2d(x, w, padding=1)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    wt = hip.conv3x3_pack_weight(w.float().cuda(), flip_transpose=True)        # [9, Cout, Cin]
    gx = hip.conv3x3_nhwc(gy.float().permute(0, 2, 3, 1).contiguous().cuda(), wt, None, 0)
    np.testing.assert_allclose(gx.permute(0, 3, 1, 2).cpu().numpy(), x.grad.float().numpy(), atol=1e-4)


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(300, 7, 7, 256, 512), (1536, 7, 7, 512, 512), (100, 14, 14, 128, 128), (1203, 7, 7, 128, 256)])
def test_conv3x3_over_many_small_maps_on_the_ring_engine(hip, B, H, W, Cin, Cout):
    """the mask tower's / ResNet layer4's 3x3 conv over hundreds of 7x7 maps on the plane / ring engine (mh_image_maxbits ->
    mh_act_planes -> mh_plconv3x3 with up to 65535 images, per-map scales): forward with bias + ReLU and the input-gradient
    convolution (flip-transposed weights) against float64, maps of very different magnitude included"""
    from lib import hip_ops
    g = torch.Generator().manual_seed(B + Cin)
    x = torch.randn(B, H, W, Cin, generator=g) * (10.0 ** torch.randint(-3, 3, (B, 1, 1, 1), generator=g).float())
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    assert hip_ops._small_maps_on_planes(x.cuda(), Cin, Cout)
    y = hip_ops.conv3x3_small_maps(x.cuda(), w.cuda(), bias.cuda(), 1)
    sel = torch.arange(0, B, max(1, B // 40))                      # float64 reference on a sample of the maps (CPU time)
    ref = torch.relu(F.conv2d(x[sel].permute(0, 3, 1, 2).double(), w.double(), bias.double(), padding=1)).permute(0, 2, 3, 1)
    scale = ref.abs().amax(dim=(1, 2, 3), keepdim=True).clamp(min=1e-30)
    assert float(((y[sel.cuda()].cpu().double() - ref).abs() / scale).max()) < 2e-5          # per map: each has its own scale
    gy = torch.randn(B, H, W, Cout, generator=g) * (10.0 ** torch.randint(-4, 2, (B, 1, 1, 1), generator=g).float())
    gx = hip_ops.conv3x3_small_maps(gy.cuda(), w.cuda(), None, 0, flip_transpose=True)
    gref = F.conv_transpose2d(gy[sel].permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    gscale = gref.abs().amax(dim=(1, 2, 3), keepdim=True).clamp(min=1e-30)
    assert tuple(gx.shape) == (B, H, W, Cin)
    assert float(((gx[sel.cuda()].cpu().double() - gref).abs() / gscale).max()) < 2e-5
    # the autograd node of the trainable layers takes this path and agrees with the in-loop kernels it replaces
    if B * H * W <= 20000:
        xs = x.cuda().requires_grad_(True)
        ws = w.cuda().requires_grad_(True)
        out = hip_ops._Conv3x3Fn.apply(xs, ws, None, 0)
        out.backward(gy.cuda())
        y2 = hip.conv3x3_nhwc(x.cuda(), hip.conv3x3_pack_weight(w.cuda(), False), None, 0)
        assert float((out.detach() - y2).abs().max()) <= 2e-5 * float(y2.abs().max())
        assert ws.grad is not None and torch.isfinite(ws.grad).all() and tuple(xs.grad.shape) == tuple(x.shape)
        assert float((xs.grad - gx).abs().max()) == 0.0                # the node's input gradient IS the flip-transposed conv above


def _decode_act_image(img):
    """ActImage (csrc/pl_conv.hip: cells [C/16][B*H*W][h1[16] | h2[16]] f16, per-image scale words behind them) -> float64
    [B,H,W,C]: value = (h1 + h2) 2^-e, exact"""
    B, H, W, C = img.B, img.H, img.W, img.C
    M, G = B * H * W, C // 16
    raw = img.buf.cpu().numpy()
    cells = raw[:G * M * 64].view(np.float16).reshape(G, M, 2, 16).astype(np.float64)
    off = (G * M * 64 + 255) // 256 * 256
    bits = raw[off:off + 4 * B].view(np.uint32)
    biased = (bits >> 23) & 0xff
    e = np.where((biased == 0) | (biased == 255), 0, 14 - (biased.astype(np.int64) - 127))
    val = (cells[:, :, 0, :] + cells[:, :, 1, :]).transpose(1, 0, 2).reshape(B, H * W, C)
    return torch.from_numpy(val * (2.0 ** -e.astype(np.float64))[:, None, None]).reshape(B, H, W, C)


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(3, 20, 20, 32, 64), (2, 26, 18, 64, 128), (5, 6, 10, 16, 256), (1, 120, 136, 64, 64),
                                            (3, 36, 36, 256, 128), (6, 74, 74, 256, 512)])
def test_conv3x3_pooled_image_epilogue(hip, B, H, W, Cin, Cout):
    """mh_plconv3x3_pool_to_image (round 6: tile rows in pool order, the 2x2 window maximum taken in the epilogue, the POOLED
    output written as the next layer's activation image; K slices added up inside the launch where the planner cuts) against
    (a) float64 conv + ReLU + max-pool, (b) mh_plconv3x3 + torch max-pool (same arithmetic: fp32 rounding apart), (c) the max-pool
    of the image mh_plconv3x3_to_image writes -- cell for cell when every tile of the launch is sliced alike; true maxima reported
    for the next layer equal the un-pooled layer's.  Covers 64-channel tiles (conv1_2's ring shape), windows that wrap image rows
    inside a tile, several images per tile, the conv4_3 shape with its sliced tail tiles."""
    g = torch.Generator().manual_seed(B * H + Cout)
    x = torch.relu(torch.randn(B, H, W, Cin, generator=g)) * torch.tensor([1.0, 5.0, 0.2, 9.0, 1.0, 3.0])[:B].view(B, 1, 1, 1)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    xd, wd, bd = x.cuda(), w.cuda(), bias.cuda()
    mb_in = hip.image_maxbits(xd)
    img = hip.act_planes(xd, mb_in)
    pk = hip.plconv_pack_weight(wd, False)
    mb0, mb1, mb2 = (torch.zeros(B, dtype=torch.int32, device='cuda') for _ in range(3))
    y = hip.plconv3x3(img, pk, Cout, bd, 1, mb0)
    im1 = hip.plconv3x3_to_image(img, mb_in, pk, Cout, bd, 1, mb1)
    im2 = hip.plconv3x3_pool_to_image(img, mb_in, pk, Cout, bd, 1, mb2)
    assert (im2.B, im2.H, im2.W, im2.C) == (B, H // 2, W // 2, Cout)
    pooled = _decode_act_image(im2)
    ref = F.max_pool2d(torch.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), bias.double(), padding=1)), 2, 2).permute(0, 2, 3, 1)
    scale = ref.abs().amax(dim=(1, 2, 3), keepdim=True).clamp(min=1e-30)
    err = ((pooled - ref).abs() / scale)
    assert float(err.max()) < 4e-5 and float((err ** 2).mean().sqrt()) < 3e-6, (float(err.max()), float((err ** 2).mean().sqrt()))
    y_pool = F.max_pool2d(y.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).cpu().double()
    assert float(((pooled - y_pool).abs() / scale).max()) < 4e-5
    assert torch.equal(mb1.cpu(), mb2.cpu()) and torch.equal(mb0.cpu(), mb2.cpu())          # true per-image maxima (fp32 bits)
    pool_of_image = F.max_pool2d(_decode_act_image(im1).permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    if B * H * W * (Cout // 64) < 256 * 512 * 2:       # fewer tiles than one round of resident blocks: every tile is sliced alike
        assert torch.equal(pooled, pool_of_image)
    else:                                              # tail tiles are sliced, body tiles are not; the two orders put other pixels there
        assert float(((pooled - pool_of_image).abs() / scale).max()) < 2e-6


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(2, 9, 11, 16, 32), (3, 7, 7, 256, 512), (1, 37, 37, 64, 128),
                                            (2, 14, 14, 64, 64), (1, 20, 23, 128, 120), (5, 6, 5, 32, 8)])
def test_conv3x3_weight_gradient_implicit_gemm(hip, B, H, W, Cin, Cout):
    """mh_conv3x3_wgrad (implicit GEMM over the pixels, tap masks, split-K) vs autograd in fp64: image borders, several
    images, pixel counts that are not a multiple of 16, n-tiles that span two taps (Cin = 64), tiny outputs."""
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64)
    gy = torch.randn(B, Cout, H, W, generator=g, dtype=torch.float64)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w, None, padding=1).backward(gy)
    got = hip.conv3x3_wgrad(x.float().permute(0, 2, 3, 1).contiguous().cuda(), gy.float().permute(0, 2, 3, 1).contiguous().cuda())
    assert got is not None and got.shape == (Cout, 9 * Cin)
    got = got.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).cpu().double()
    scale = float(w.grad.abs().max())
    np.testing.assert_allclose(got.numpy(), w.grad.numpy(), atol=2e-5 * scale)

def test_maxpool_and_activation_backward(hip):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 8, 9, 7, generator=g)                         # odd H and W: trailing row / column get zero gradient
    x[0, :, :2, :2] = 1.5                                            # ties: the first maximal element takes the gradient
    xg = x.clone().requires_grad_()
    y = F.max_pool2d(xg, 2, 2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    got = hip.maxpool2x2_bwd_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), gy.permute(0, 2, 3, 1).contiguous().cuda())
    np.testing.assert_array_equal(got.permute(0, 3, 1, 2).cpu().numpy(), xg.grad.numpy())
    yv = torch.randn(3, 40, generator=g) * 4
    gv = torch.randn(3, 40, generator=g)
    for epi, act in ((1, torch.relu(yv)), (2, torch.clamp(yv, 0, 6))):
        ref = gv * ((act > 0) & ((act < 6) if epi == 2 else torch.ones_like(act, dtype=torch.bool))).float()
        np.testing.assert_array_equal(hip.act_bwd(gv.cuda(), act.cuda(), epi).cpu().numpy(), ref.numpy())


def test_conv_first_and_pool_and_layouts(hip):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 3, 38, 42, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(64, generator=g)
    out = hip.conv_first_nchw(x.cuda(), w.cuda(), b.cuda(), 1)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)).float()
    np.testing.assert_allclose(out.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy(), atol=1e-5)
    pooled = hip.maxpool2x2_nhwc(out)
    np.testing.assert_array_equal(pooled.permute(0, 3, 1, 2).cpu().numpy(),
                                  F.max_pool2d(out.permute(0, 3, 1, 2), 2, 2).cpu().numpy())
    y = torch.randn(2, 5, 6, 7, generator=g).cuda()
    np.testing.assert_array_equal(hip.nchw_to_nhwc(y).cpu().numpy(), y.permute(0, 2, 3, 1).contiguous().cpu().numpy())
    np.testing.assert_array_equal(hip.nhwc_to_nchw(hip.nchw_to_nhwc(y)).cpu().numpy(), y.cpu().numpy())


def test_im2col_conv7x7_stride2(hip):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 2, 27, 27, generator=g)
    w = torch.randn(16, 2, 7, 7, generator=g) * 0.1
    cols, Ho, Wo = hip.im2col_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), 7, 7, 2, 3, ldo=100)
    assert (Ho, Wo) == (14, 14)
    wmat = torch.zeros(16, 100)
    wmat[:, :98] = w.permute(0, 2, 3, 1).reshape(16, 98)
    out = hip.gemm(cols, wmat.cuda(), False, True).view(3, 14, 14, 16).permute(0, 3, 1, 2)
    ref = F.conv2d(x.double(), w.double(), stride=2, padding=3).float()
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=1e-4)


@pytest.mark.parametrize('kern', ['mfma', 'valu'])
@pytest.mark.parametrize('N,C0', [(1, 256), (37, 256), (200, 512)])
def test_tower_conv1_direct_forward_and_weight_gradient(hip, monkeypatch, N, C0, kern):
    """csrc/tower.hip: the mask tower's 7x7 / stride 2 convolution without a column matrix against torch's conv2d in float64 --
    `mfma` (round 6, default): fragments built straight from the padded masks, f16x3 forward / bf16x6 weight gradient on the
    matrix cores; `valu` (MH_TOWER_CONV1=valu): mask values through the scalar cache, exact fp32 FMAs.  N = 1 and 37 leave a
    partial last 32-pixel tile / a short row range per block, N = 37 and 200 a partial last block of the VALU weight-gradient
    grid (3 pairs per block), C0 = 512 is the ResNet tower (two channel groups)."""
    monkeypatch.setenv('MH_TOWER_CONV1', kern)
    g = torch.Generator().manual_seed(N)
    rects = torch.rand(N, 27, 27, 2, generator=g)
    rects[:, :5] = 0                                               # masks are exactly zero outside their box
    w = (torch.randn(C0, 2, 7, 7, generator=g) * 0.1).requires_grad_(True)
    b = (torch.randn(C0, generator=g) * 0.1).requires_grad_(True)
    dy = torch.randn(N, 14, 14, C0, generator=g)
    pre = F.conv2d(rects.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=2, padding=3)
    pre.backward(dy.permute(0, 3, 1, 2).double())
    assert hip.tower_conv1_supported(rects, w)
    xp = hip.tower_conv1_pad(rects.cuda())
    assert tuple(xp.shape) == (N, 33, 33, 2)
    np.testing.assert_array_equal(xp[:, 3:30, 3:30].cpu().numpy(), rects.numpy())
    assert float(xp[:, :3].abs().max()) == 0.0 and float(xp[:, :, 30:].abs().max()) == 0.0
    y = hip.tower_conv1_fwd(xp, w.detach().permute(2, 3, 1, 0).reshape(98, C0).contiguous().cuda(), b.detach().cuda())
    ref = F.relu(pre).detach().permute(0, 2, 3, 1).float().numpy()
    np.testing.assert_allclose(y.cpu().numpy(), ref, atol=2e-6 * max(1.0, float(np.abs(ref).max())))
    dwk, db = hip.tower_conv1_wgrad(xp, dy.cuda())
    dw0 = dwk.view(7, 7, 2, C0).permute(3, 2, 0, 1).cpu().numpy()
    np.testing.assert_allclose(dw0, w.grad.numpy(), atol=3e-6 * float(w.grad.abs().max()))
    np.testing.assert_allclose(db.cpu().numpy(), b.grad.numpy(), atol=3e-6 * float(b.grad.abs().max()))


def test_mask_tower_first_convolution_in_its_three_forms_agree(hip, monkeypatch):
    """the whole tower node (forward + every gradient) with its first convolution on the matrix-core kernels (round 6), on the
    direct VALU kernels and on im2col + GEMM.  The forms round differently, so a unit within rounding of a kink (the two ReLUs,
    the max-pool) may be decided differently: such a unit moves gradients by its whole contribution (a conv.4 flip moves EVERY
    gradient upstream of it: gpurun r06_c11, one flip of 2.4 M units = 5.5e-3 of conv.4.weight's gradient at 96 pairs) -- that is
    the kink, not arithmetic; at 96 pairs = 7.2 M kink units every draw has 1-4 such units between ANY two forms, r06_c12).  The
    kink decisions of every form are captured (lib.get_union_boxes.TAPS): at most 4 may differ between two forms, and the
    gradients are compared on a draw where none does (6 pairs = 0.45 M units: most draws)."""
    import lib.get_union_boxes as GUB
    from parity_util import grad_close
    N = 6
    forms = (('gemm', 'gemm', 'valu'), ('valu', 'direct', 'valu'), ('mfma', 'direct', 'mfma'))
    for seed in range(5, 17):
        torch.manual_seed(seed)
        tower = GUB.UnionBoxesAndFeats(pooling_size=7, stride=16, dim=512).cuda().train()
        rects = torch.rand(N, 27, 27, 2).cuda()
        pools = torch.randn(N, 512, 7, 7).cuda()
        gout = torch.randn(N, 512, 7, 7).cuda()
        res, taps = {}, {}
        for form, mode, kern in forms:
            monkeypatch.setattr(GUB, 'TOWER_CONV1', mode)
            monkeypatch.setenv('MH_TOWER_CONV1', kern)
            for bn in (tower.conv[2], tower.conv[6]):
                bn.reset_running_stats()
            tower.zero_grad(set_to_none=True)
            c = tower.conv
            taps[form] = {}
            monkeypatch.setattr(GUB, 'TAPS', taps[form])
            out = GUB._TowerFn.apply(rects, pools, c[0].weight, c[0].bias, c[2].weight, c[2].bias, c[4].weight, c[4].bias,
                                     c[6].weight, c[6].bias, c[2], c[6], True)
            monkeypatch.setattr(GUB, 'TAPS', None)
            out.backward(gout)
            res[form] = [out.detach().clone()] + [p.grad.detach().clone() for p in tower.parameters()]
        flips = {f: sum(int((taps['gemm'][k] != taps[f][k]).sum()) for k in taps['gemm']) for f in ('valu', 'mfma')}
        print('seed %d: kink decisions that differ from the GEMM form: %s' % (seed, flips))
        assert max(flips.values()) <= 4, flips
        if max(flips.values()) == 0:
            break
    else:
        raise AssertionError('no draw without a differing kink decision in 12 seeds')
    names = ['output'] + [n for n, _ in tower.named_parameters()]
    for form in ('valu', 'mfma'):
        for name, a, d in zip(names, res['gemm'], res[form]):
            grad_close(d.cpu().numpy(), a.cpu().numpy(), what='tower %s %s' % (form, name), rtol=1e-4)


@pytest.mark.parametrize('n,R,D,with_vis', [(120, 1536, 4096, True), (7, 33, 64, True), (5, 1, 8, False), (40, 300, 512, False)])
def test_pair_product_forward_and_backward(hip, n, R, D, with_vis):
    """csrc/exact_ops.hip pair_product_*: the relation tail's subj[i1] * obj[i2] (* vis) as one autograd node (lib/rel_model.py:
    _PairProductFn) against the framework's gather / multiply ops -- forward bit for bit (same multiplication order, no
    contraction), gradients against the same graph in float64; boxes without a row on a side get exact zeros."""
    from lib.pytorch_misc import set_host
    from lib.rel_model import _PairProductFn
    g = torch.Generator().manual_seed(n * 1000 + R)
    edge = torch.randn(n, 2, D, generator=g)
    vis = torch.randn(R, D, generator=g) if with_vis else None
    rel = torch.stack((torch.zeros(R, dtype=torch.int64), torch.randint(0, max(n - 2, 1), (R,), generator=g),
                       torch.randint(1, n, (R,), generator=g)), 1)             # box n-1 is never a subject, box 0 never an object
    gout = torch.randn(R, D, generator=g)
    e_d = edge.cuda().requires_grad_(True)
    v_d = vis.cuda().requires_grad_(True) if with_vis else None
    rel_d = set_host(rel.cuda(), rel.numpy())
    out = _PairProductFn.apply(e_d, v_d, rel_d)
    ref32 = edge[rel[:, 1], 0] * edge[rel[:, 2], 1]
    if with_vis:
        ref32 = ref32 * vis
    assert torch.equal(out.detach().cpu(), ref32)
    out.backward(gout.cuda())
    e64 = edge.double().requires_grad_(True)
    v64 = vis.double().requires_grad_(True) if with_vis else None
    ref = e64[rel[:, 1], 0] * e64[rel[:, 2], 1]
    if with_vis:
        ref = ref * v64
    ref.backward(gout.double())
    scale = float(e64.grad.abs().max())
    np.testing.assert_allclose(e_d.grad.cpu().numpy(), e64.grad.numpy(), atol=2e-6 * scale)
    assert float(e_d.grad[n - 1, 0].abs().max()) == 0.0 and float(e_d.grad[0, 1].abs().max()) == 0.0
    if with_vis:
        np.testing.assert_allclose(v_d.grad.cpu().numpy(), v64.grad.numpy(), atol=2e-6 * float(v64.grad.abs().max()))
    # twice the same call: deterministic
    e2 = edge.cuda().requires_grad_(True)
    _PairProductFn.apply(e2, v_d.detach() if with_vis else None, rel_d).backward(gout.cuda())
    assert torch.equal(e2.grad, e_d.grad)


@pytest.mark.parametrize('n,R', [(120, 1536), (9, 40), (3, 1), (80, 6320)])
def test_frequency_bias_add_forward_and_table_gradient(hip, n, R):
    """csrc/exact_ops.hip freq_bias_*: rel_dists + FrequencyBias[obj_preds[i1], obj_preds[i2]] as one node (lib/rel_model.py:
    _FreqBiasAddFn) against the framework's gather / embedding ops -- forward bit for bit; the table's gradient (rows of a key summed
    in ascending row order by the key's first row) against the same graph in float64, zero everywhere else, bitwise reproducible;
    few classes so that many rows share a key."""
    from lib.rel_model import _FreqBiasAddFn
    g = torch.Generator().manual_seed(n + R)
    C, P = 151, 51
    table = torch.randn(C * C, P, generator=g)
    logits = torch.randn(R, P, generator=g)
    preds = torch.randint(1, 6 if R > 8 else C, (n,), generator=g)
    rel = torch.stack((torch.zeros(R, dtype=torch.int64), torch.randint(0, n, (R,), generator=g), torch.randint(0, n, (R,), generator=g)), 1)
    gout = torch.randn(R, P, generator=g)
    t_d = table.cuda().requires_grad_(True)
    l_d = logits.cuda().requires_grad_(True)
    out = _FreqBiasAddFn.apply(l_d, t_d, preds.cuda(), rel.cuda(), C)
    keys = preds[rel[:, 1]] * C + preds[rel[:, 2]]
    assert torch.equal(out.detach().cpu(), logits + table[keys])
    out.backward(gout.cuda())
    assert torch.equal(l_d.grad.cpu(), gout)
    ref = torch.zeros(C * C, P, dtype=torch.float64).index_add_(0, keys, gout.double())
    got = t_d.grad.cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-6 * float(ref.abs().max()))
    untouched = torch.ones(C * C, dtype=torch.bool)
    untouched[keys] = False
    assert float(got[untouched].abs().max()) == 0.0
    t2 = table.cuda().requires_grad_(True)
    _FreqBiasAddFn.apply(logits.cuda(), t2, preds.cuda(), rel.cuda(), C).backward(gout.cuda())
    assert torch.equal(t2.grad, t_d.grad)


@pytest.mark.parametrize('Ra,Ca,Rb,Cb', [(120, 151, 1536, 51), (3, 5, 1, 2), (257, 1000, 64, 7)])
def test_cross_entropy_pair_forward_and_backward(hip, Ra, Ca, Rb, Cb):
    """csrc/exact_ops.hip ce_pair_* / lib/losses.py: the relation driver's two mean cross-entropy losses as one node against
    F.cross_entropy in float64 (losses, both logit gradients with unequal upstream weights), the strided label view of
    rel_labels[:, -1], bitwise reproducible; MOTIFS_FUSED_LOSS=0 and CPU tensors take the framework's functions."""
    from lib.losses import _CrossEntropyPairFn
    g = torch.Generator().manual_seed(Ra + Rb)
    a = (torch.randn(Ra, Ca, generator=g) * 3).requires_grad_(True)
    b = (torch.randn(Rb, Cb, generator=g) * 5).requires_grad_(True)
    la = torch.randint(0, Ca, (Ra,), generator=g)
    lb4 = torch.randint(0, Cb, (Rb, 4), generator=g)
    w = torch.tensor([0.7, 1.9])
    a_d, b_d = a.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
    lb_d = lb4.cuda()[:, -1]
    assert Rb == 1 or lb_d.stride(0) == 4
    ls = _CrossEntropyPairFn.apply(a_d, la.cuda(), b_d, lb_d)
    (ls * w.cuda()).sum().backward()
    ref = torch.stack((F.cross_entropy(a.double(), la), F.cross_entropy(b.double(), lb4[:, -1])))
    (ref * w.double()).sum().backward()
    # (a loss is lse - x[label]: fp32 cancellation bounds it absolutely, at the logits' scale, not relatively)
    np.testing.assert_allclose(ls.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-6, atol=2e-7 * max(1.0, float(a.abs().max()), float(b.abs().max())))
    np.testing.assert_allclose(a_d.grad.cpu().numpy(), a.grad.numpy(), atol=2e-7 * max(1.0, float(a.grad.abs().max()) * 1e3))
    np.testing.assert_allclose(b_d.grad.cpu().numpy(), b.grad.numpy(), atol=2e-7 * max(1.0, float(b.grad.abs().max()) * 1e3))
    a2, b2 = a.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
    ls2 = _CrossEntropyPairFn.apply(a2, la.cuda(), b2, lb_d)
    (ls2 * w.cuda()).sum().backward()
    assert torch.equal(ls2, ls) and torch.equal(a2.grad, a_d.grad) and torch.equal(b2.grad, b_d.grad)
    # a label outside the row: the loss of that side is NaN (no read outside the logits), the other side's is untouched
    bad = la.clone()
    bad[0] = Ca
    ls3 = _CrossEntropyPairFn.apply(a.detach().cuda(), bad.cuda(), b.detach().cuda(), lb_d)
    assert bool(torch.isnan(ls3[0])) and torch.equal(ls3[1], ls[1])


# ----------------------------------------------------------------------------------------------- LSTM
def _lstm_problem(lengths, in_size, H, nl, seed, p=0.0):
    from oracle import lstm as OL
    g = torch.Generator().manual_seed(seed)
    lengths = sorted(lengths, reverse=True)
    T, B = lengths[0], len(lengths)
    x = torch.randn(T, B, in_size, generator=g)
    for b, l in enumerate(lengths):
        x[l:, b] = 0
    _, wtot = OL.layer_offsets(in_size, H, nl)
    weight = torch.randn(wtot, generator=g) * (1.0 / in_size ** 0.5)
    bias = torch.randn(5 * H * nl, generator=g) * 0.1
    drop = torch.ones(nl, B, H)
    if p > 0:
        drop = (torch.rand(nl, B, H, generator=g) > p).float() / (1 - p)
    return x, lengths, weight, bias, drop


@pytest.mark.parametrize('lengths,in_size,H,nl,p', [([5, 3, 3, 1], 24, 16, 1, 0.0), ([6, 6, 2], 40, 32, 2, 0.3),
                                                     ([4, 2, 1], 20, 8, 3, 0.2), ([7], 12, 20, 4, 0.0),
                                                     ([3, 2], 10, 18, 2, 0.0),   # H % 4 != 0: per-step launch path
                                                     ([1, 1], 12, 16, 2, 0.0),   # T = 1: the backward-direction layer's dWh is all zero
                                                     ([20] * 3 + [17, 9, 9, 4, 2, 1, 1], 712, 512, 2, 0.1)])
def test_hwlstm_fwd_bwd(hip, lengths, in_size, H, nl, p):
    from oracle import lstm as OL
    x, lengths, weight, bias, drop = _lstm_problem(lengths, in_size, H, nl, seed=nl + H, p=p)
    T, B = x.shape[0], x.shape[1]
    out_ref, h_slots, c_slots, gates = OL.highway_lstm_forward(x, lengths, weight, bias, drop, H, nl, True,
                                                               return_state=True)
    h_data, c_data, g = hip.hwlstm_fwd(x.cuda(), lengths, weight.cuda(), bias.cuda(), drop.cuda(), H, nl, True)
    np.testing.assert_allclose(h_data[-1, 1:].cpu().numpy(), out_ref.numpy(), atol=1e-4)
    for l in range(nl):
        np.testing.assert_allclose(c_data[l, 1:].cpu().numpy(), torch.stack(c_slots[l][1:]).numpy(), atol=1e-4)
    gout = torch.randn(T, B, H, generator=torch.Generator().manual_seed(1))
    for b, l in enumerate(lengths):
        gout[l:, b] = 0
    xg_ref, wg_ref, bg_ref = OL.highway_lstm_backward(gout, x, lengths, weight, drop, H, nl, h_slots, c_slots, gates)
    xg, wg, bg = hip.hwlstm_bwd(gout.cuda(), x.cuda(), lengths, weight.cuda(), drop.cuda(), H, nl, h_data, c_data, g)
    np.testing.assert_allclose(xg.cpu().numpy(), xg_ref.numpy(), atol=2e-4)
    np.testing.assert_allclose(wg.cpu().numpy(), wg_ref.numpy(), atol=5e-4)
    np.testing.assert_allclose(bg.cpu().numpy(), bg_ref.numpy(), atol=5e-4)
    # eval mode: same outputs, no gates
    h2, _, g2 = hip.hwlstm_fwd(x.cuda(), lengths, weight.cuda(), bias.cuda(), drop.cuda(), H, nl, False)
    assert g2 is None
    np.testing.assert_array_equal(h2.cpu().numpy(), h_data.cpu().numpy())


def test_lstm_barrier_timeout_is_loud(hip):
    """A persistent LSTM launch whose grid barrier times out must not hand back plausible numbers: with the test hook
    making the barrier target unreachable, the launch (a) sets the host-visible fault word, (b) NaN-poisons its
    outputs, (c) makes check_faults() raise and (d) makes the NEXT LSTM entry point return MH_EFAULT; after
    mh_fault_clear() the same call is bit-identical to an undisturbed one."""
    L = hip.lib()
    x, lengths, weight, bias, drop = _lstm_problem([6, 6, 2], 40, 32, 2, seed=5, p=0.0)
    args = (x.cuda(), lengths, weight.cuda(), bias.cuda(), drop.cuda(), 32, 2, True)
    good_h, good_c, good_g = hip.hwlstm_fwd(*args)
    torch.cuda.synchronize()
    assert L.mh_fault_pending() == 0
    hip.check_faults()
    L.mh_debug_lstm_barrier_fault(1)
    try:
        h, c, g = hip.hwlstm_fwd(*args)                       # launches; the barrier cannot complete
        torch.cuda.synchronize()
        assert L.mh_fault_pending() == 1
        assert torch.isnan(h[-1, 1:]).any(), 'outputs of a broken launch must be poisoned'
        with pytest.raises(hip.HipKernelError, match='grid barrier'):
            hip.check_faults()
        L.mh_debug_lstm_barrier_fault(0)
        with pytest.raises(hip.HipKernelError, match='status -3'):       # MH_EFAULT, before anything is launched
            hip.hwlstm_fwd(*args)
        pre_i = torch.zeros(3, 6 * 32, device='cuda')
        with pytest.raises(hip.HipKernelError, match='status -3'):
            hip.hwcell_seq_fwd(pre_i, [1, 1, 1], torch.zeros(5 * 32, 32, device='cuda'), torch.zeros(5 * 32, device='cuda'), None)
    finally:
        L.mh_debug_lstm_barrier_fault(0)
        L.mh_fault_clear()
    h2, c2, g2 = hip.hwlstm_fwd(*args)
    torch.cuda.synchronize()
    assert L.mh_fault_pending() == 0
    np.testing.assert_array_equal(h2.cpu().numpy(), good_h.cpu().numpy())
    # the backward launch has the same protection
    gout = torch.randn(x.shape[0], x.shape[1], 32, generator=torch.Generator().manual_seed(1)).cuda()
    L.mh_debug_lstm_barrier_fault(1)
    try:
        xg, wg, bg = hip.hwlstm_bwd(gout, x.cuda(), lengths, weight.cuda(), drop.cuda(), 32, 2, good_h, good_c, good_g)
        torch.cuda.synchronize()
        assert L.mh_fault_pending() == 1 and not torch.isfinite(xg).all()
    finally:
        L.mh_debug_lstm_barrier_fault(0)
        L.mh_fault_clear()


@pytest.mark.parametrize('H,batch_sizes', [(32, [6, 6, 5, 3, 3, 1]), (512, [6] * 9 + [5, 4, 4, 2, 1]), (20, [1, 1, 1])])
def test_packed_recurrence_single_launch(hip, H, batch_sizes):
    """mh_hwcell_seq_fwd/bwd (one persistent launch, grid barrier per step) == stepping the cell kernels"""
    g = torch.Generator().manual_seed(H)
    N, B = sum(batch_sizes), batch_sizes[0]
    pre_i = (torch.randn(N, 6 * H, generator=g) * 0.5).cuda()
    w = (torch.randn(5 * H, H, generator=g) * (1.0 / H ** 0.5)).cuda()
    b = (torch.randn(5 * H, generator=g) * 0.1).cuda()
    mask = ((torch.rand(B, H, generator=g) > 0.2).float() / 0.8).cuda()
    dh = torch.randn(N, H, generator=g).cuda()
    bounds, s0 = [], 0
    for n in batch_sizes:
        bounds.append((s0, s0 + n, n)); s0 += n
    # reference: the per-step kernels
    h_ref, c_ref, g_ref = [], [], []
    h_prev = c_prev = torch.zeros(B, H, device='cuda')
    for s, e, n in bounds:
        h, c, gt = hip.hwlstm_cell_fwd(pre_i[s:e], h_prev[:n].contiguous(), c_prev[:n].contiguous(), w, b,
                                       mask[:n].contiguous(), True)
        h_ref.append(h); c_ref.append(c); g_ref.append(gt)
        h_prev, c_prev = h, c
    h_buf, c_buf, gates = hip.hwcell_seq_fwd(pre_i, batch_sizes, w, b, mask)
    np.testing.assert_array_equal(h_buf[:B].cpu().numpy(), 0)
    # (same formulas; the two kernels may contract a*b+c differently, hence not bitwise)
    np.testing.assert_allclose(h_buf[B:].cpu().numpy(), torch.cat(h_ref).cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(c_buf[B:].cpu().numpy(), torch.cat(c_ref).cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(gates.cpu().numpy(), torch.cat(g_ref).cpu().numpy(), rtol=0, atol=2e-6)
    wt = w.t().contiguous()
    d_ref = [None] * len(bounds)
    dh_rec = dc_rec = None
    for t in range(len(bounds) - 1, -1, -1):
        s, e, n = bounds[t]
        d_h = dh[s:e].clone(); d_c = torch.zeros(n, H, device='cuda')
        if dh_rec is not None:
            m = dh_rec.shape[0]; d_h[:m] += dh_rec; d_c[:m] = dc_rec
        c_prev = torch.cat(c_ref)[bounds[t - 1][0]:bounds[t - 1][0] + n].contiguous() if t > 0 else torch.zeros(n, H, device='cuda')
        dg, dc_in = hip.hwlstm_cell_bwd(d_h, d_c, c_prev, c_ref[t].contiguous(), g_ref[t].contiguous(), mask[:n].contiguous())
        d_ref[t] = dg
        if t > 0:
            dh_rec, dc_rec = hip.gemv_rows(dg[:, :5 * H], wt), dc_in
    d_pre = hip.hwcell_seq_bwd(dh, batch_sizes, c_buf, gates, mask, wt)
    np.testing.assert_allclose(d_pre.cpu().numpy(), torch.cat(d_ref).cpu().numpy(), rtol=0, atol=1e-5)

@pytest.mark.parametrize('H,batch_sizes,train', [(512, [1] * 20, False), (128, [1] * 64, False), (512, [6] * 9 + [5, 4, 4, 2, 1], True),
                                                 (64, [3, 3, 1], True)])
def test_fused_greedy_decoder_equals_the_step_loop(hip, monkeypatch, H, batch_sizes, train):
    """mh_decoder_greedy (whole greedy decode in one persistent launch: cell + class logits + arg-max + embedding
    gather) against the per-step path (2 launches + torch glue per object, MOTIFS_STEP_DECODER=1): committed labels and
    fed embedding rows EXACT, logits 1e-5; in eval the fused path also skips the second recurrence pass"""
    from torch.nn.utils.rnn import PackedSequence
    from lib.lstm.decoder_rnn import DecoderRNN
    torch.manual_seed(H + len(batch_sizes))
    classes = ['__background__'] + ['c%d' % i for i in range(1, 151)]
    dec = DecoderRNN(classes, embed_dim=100, inputs_dim=H, hidden_dim=H, recurrent_dropout_probability=0.1).cuda()
    dec.train(train)
    N = sum(batch_sizes)
    x = torch.randn(N, H, device='cuda')
    labels = None
    if train:
        labels = torch.randint(0, 151, (N,), device='cuda')
        labels[torch.rand(N, device='cuda') < 0.4] = 0                   # background rows feed back their own arg-max
    ps = PackedSequence(x, torch.tensor(batch_sizes))
    from lib import rng
    outs = {}
    for mode in ('fused', 'steps'):
        if mode == 'steps':
            monkeypatch.setenv('MOTIFS_STEP_DECODER', '1')
        else:
            monkeypatch.delenv('MOTIFS_STEP_DECODER', raising=False)
        rng.use_host_rng(4)
        ctxm = torch.enable_grad() if train else torch.no_grad()
        with ctxm:
            dists, commits = dec(ps, labels=labels)
        rng.use_host_rng(None)
        outs[mode] = (dists.detach().cpu().numpy(), commits.cpu().numpy())
    np.testing.assert_array_equal(outs['fused'][1], outs['steps'][1])
    np.testing.assert_allclose(outs['fused'][0], outs['steps'][0], atol=1e-5)
    if train:
        lab = labels.cpu().numpy()
        np.testing.assert_array_equal(outs['fused'][1][lab != 0], lab[lab != 0])      # teacher forcing where a label exists
    assert (outs['fused'][1] > 0).all()
    # the raw entry point: fed rows = committed label of the previous step of the same sequence + 1 ('start' = 0)
    enc = torch.randn(N, 6 * H, device='cuda') * 0.3
    emb = torch.randn(152, 6 * H, device='cuda') * 0.3
    h_all, logits, fed, commits = hip.decoder_greedy(enc, emb, batch_sizes, dec.state_linearity.weight.contiguous(),
                                                     dec.state_linearity.bias, None, dec.out.weight.contiguous(), dec.out.bias)
    fed, commits = fed.cpu().numpy(), commits.cpu().numpy()
    start = 0
    for t, n in enumerate(batch_sizes):
        if t == 0:
            assert (fed[:n] == 0).all()
        else:
            np.testing.assert_array_equal(fed[start:start + n], commits[start - batch_sizes[t - 1]:start - batch_sizes[t - 1] + n] + 1)
        start += n
    np.testing.assert_array_equal(commits, logits[:, 1:].argmax(1).cpu().numpy() + 1)
    np.testing.assert_allclose(logits.cpu().numpy(), (h_all @ dec.out.weight.t() + dec.out.bias).detach().cpu().numpy(), atol=1e-5)


@pytest.mark.parametrize('H,bs,train', [(64, [3, 3, 2, 1], True), (512, [6] * 5 + [4, 2], True), (128, [1] * 9, False)])
def test_plain_lstm_decoder_cell_on_the_highway_kernels(hip, H, bs, train):
    """DecoderRNN(use_highway=False): the reference's four-block parameters, evaluated by the highway-cell kernels with the
    highway gate held open (lib/lstm/decoder_rnn.py: _cell_params) -- logits, commitments and parameter gradients against the
    oracle's plain LSTM cell (persistent launch for H = 512 / 128, step kernels otherwise)"""
    from lib.lstm.decoder_rnn import DecoderRNN
    from oracle import lstm as OL
    from torch.nn.utils.rnn import PackedSequence
    torch.manual_seed(H + len(bs))
    D = 40
    classes = ['bg'] + ['c%d' % i for i in range(1, 12)]
    dec = DecoderRNN(classes, embed_dim=100, inputs_dim=D, hidden_dim=H, recurrent_dropout_probability=0.0, use_highway=False)
    with torch.no_grad():
        dec.out.weight.mul_(3.0)
    p = {k: v.detach().clone().requires_grad_(True) for k, v in dec.state_dict().items()}
    dec.cuda().train(train)
    n = sum(bs)
    x = torch.randn(n, D)
    labels = torch.randint(0, len(classes), (n,))
    labels[1] = 0
    out, commits = dec(PackedSequence(x.cuda(), torch.tensor(bs)), labels=labels.cuda() if train else None)
    ref_out, ref_commits = OL.decoder_forward(p, x, bs, H, train, labels=labels if train else None)
    np.testing.assert_array_equal(commits.cpu().numpy(), ref_commits.numpy())
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref_out.detach().numpy(), atol=1e-4)
    if train:
        g = torch.randn(out.shape)
        (out * g.cuda()).sum().backward()
        (ref_out * g).sum().backward()
        for k, v in dec.named_parameters():
            ref = p[k].grad
            np.testing.assert_allclose(v.grad.cpu().numpy(), ref.numpy(), atol=1e-4 * max(1e-3, float(ref.abs().max())), err_msg=k)


@pytest.mark.parametrize('N', [1, 7, 64, 80])
def test_decoder_nms_commitments_on_device_equal_the_host_loop(hip, N):
    """mh_decoder_nms_commit == the reference's host loop (decoder_rnn.py:230-247) on the same probabilities and class
    boxes, incl. duplicated boxes (IoU exactly 1) and tied probabilities"""
    from lib.fpn.box_utils import nms_overlaps
    g = torch.Generator().manual_seed(N)
    C = 151
    logits = torch.randn(N, C, generator=g) * 3
    if N > 2:
        logits[2] = logits[1]                                              # tied rows
    probs = torch.softmax(logits, 1).cuda()
    x1y1 = torch.rand(N, 1, 2, generator=g) * 400
    wh = torch.rand(N, 1, 2, generator=g) * 150 + 10
    boxes = torch.cat((x1y1, x1y1 + wh), 2).repeat(1, C, 1) + torch.randn(N, C, 4, generator=g) * 3
    if N > 2:
        boxes[2] = boxes[1]
    boxes = boxes.cuda().contiguous()
    got = hip.decoder_nms_commit(probs.contiguous(), boxes, 0.3).cpu().numpy()
    is_overlap = (nms_overlaps(boxes).cpu().numpy() >= 0.3)
    sampled = probs.cpu().numpy().copy()
    sampled[:, 0] = 0
    ref = np.zeros(N, dtype=np.int64)
    for _ in range(N):
        b, c = np.unravel_index(sampled.argmax(), sampled.shape)
        ref[int(b)] = int(c)
        sampled[is_overlap[b, :, c], c] = 0.0
        sampled[b] = -1.0
    np.testing.assert_array_equal(got, ref)



def test_decoder_cell_and_gemv(hip):
    from oracle import lstm as OL
    g = torch.Generator().manual_seed(11)
    n, H, D = 6, 32, 44
    p = {'input_linearity.weight': torch.randn(6 * H, D, generator=g) * 0.2,
         'input_linearity.bias': torch.randn(6 * H, generator=g) * 0.1,
         'state_linearity.weight': torch.randn(5 * H, H, generator=g) * 0.2,
         'state_linearity.bias': torch.randn(5 * H, generator=g) * 0.1}
    x = torch.randn(n, D, generator=g)
    h0, c0 = torch.randn(n, H, generator=g), torch.randn(n, H, generator=g)
    mask = (torch.rand(n, H, generator=g) > 0.2).float() / 0.8
    h_ref, c_ref = OL.decoder_lstm_equations(p, x, h0, c0, H, mask, True)
    pre_i = hip.gemm(x.cuda(), p['input_linearity.weight'].cuda(), False, True, bias=p['input_linearity.bias'].cuda())
    h1, c1, gates = hip.hwlstm_cell_fwd(pre_i, h0.cuda(), c0.cuda(), p['state_linearity.weight'].cuda(),
                                        p['state_linearity.bias'].cuda(), mask.cuda(), True)
    np.testing.assert_allclose(h1.cpu().numpy(), h_ref.numpy(), atol=1e-5)
    np.testing.assert_allclose(c1.cpu().numpy(), c_ref.numpy(), atol=1e-5)
    # cell backward vs autograd of the oracle cell
    x64 = {k: v.double() for k, v in p.items()}
    pi = pre_i.cpu().double().requires_grad_()
    h0d, c0d = h0.double().requires_grad_(), c0.double().requires_grad_()
    ps = F.linear(h0d, x64['state_linearity.weight'], x64['state_linearity.bias'])
    ig = torch.sigmoid(pi[:, :H] + ps[:, :H]); fg = torch.sigmoid(pi[:, H:2 * H] + ps[:, H:2 * H])
    mi = torch.tanh(pi[:, 2 * H:3 * H] + ps[:, 2 * H:3 * H]); og = torch.sigmoid(pi[:, 3 * H:4 * H] + ps[:, 3 * H:4 * H])
    mem = ig * mi + fg * c0d
    hg = torch.sigmoid(pi[:, 4 * H:5 * H] + ps[:, 4 * H:5 * H])
    hout = (hg * (og * torch.tanh(mem)) + (1 - hg) * pi[:, 5 * H:]) * mask.double()
    dh, dc = torch.randn(n, H, generator=g), torch.randn(n, H, generator=g)
    (hout * dh.double()).sum().add((mem * dc.double()).sum()).backward()
    d_gates, d_c_in = hip.hwlstm_cell_bwd(dh.cuda(), dc.cuda(), c0.cuda(), c1, gates, mask.cuda())
    np.testing.assert_allclose(d_gates.cpu().numpy(), pi.grad.float().numpy(), atol=1e-5)
    np.testing.assert_allclose(d_c_in.cpu().numpy(), c0d.grad.float().numpy(), atol=1e-5)
    # small-batch GEMV: recurrent dgrad  d_h_prev = d_gates[:, :5H] @ W_state  (W_state^T rows are K=5H contiguous)
    wt = p['state_linearity.weight'].t().contiguous().cuda()               # [H, 5H]
    dhp = hip.gemv_rows(d_gates[:, :5 * H], wt)
    np.testing.assert_allclose(dhp.cpu().numpy(), h0d.grad.float().numpy(), atol=1e-5)
    for nn, R, K in [(1, 151, 512), (9, 37, 100), (17, 8, 2560)]:
        v = torch.randn(nn, K, generator=g)
        w = torch.randn(R, K, generator=g)
        bb = torch.randn(R, generator=g)
        out = hip.gemv_rows(v.cuda(), w.cuda(), bb.cuda())
        np.testing.assert_allclose(out.cpu().numpy(), (v.double() @ w.double().t() + bb.double()).float().numpy(),
                                   atol=2e-4)


# ----------------------------------------------------------------------------------------------- optimiser tail
def test_fused_clip_sgd_matches_torch(hip):
    from lib.optim import FusedClipSGD
    from lib.pytorch_misc import clip_grad_norm
    g = torch.Generator().manual_seed(5)
    shapes = [(300, 257), (70001,), (5,), (128, 64, 3, 3), (65536,), (65537,)]
    ref_p = [torch.randn(s, generator=g).cuda().requires_grad_() for s in shapes]
    new_p = [p.detach().clone().requires_grad_() for p in ref_p]
    groups = lambda ps: [{'params': ps[:2], 'lr': 0.01}, {'params': ps[2:]}]
    ref_opt = torch.optim.SGD(groups(ref_p), lr=0.1, momentum=0.9, weight_decay=1e-4)
    new_opt = FusedClipSGD(groups(new_p), lr=0.1, momentum=0.9, weight_decay=1e-4)
    for step in range(4):
        grads = [torch.randn(s, generator=g).cuda() * (10.0 if step % 2 == 0 else 0.01) for s in shapes]
        for p, q, gr in zip(ref_p, new_p, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        tn = clip_grad_norm([('p%d' % i, p) for i, p in enumerate(ref_p)], max_norm=5.0, clip=True)
        ref_opt.step()
        new_opt.step(max_norm=5.0)
        assert abs(new_opt.last_total_norm() - tn) <= 1e-4 * tn
        for p, q in zip(ref_p, new_p):
            np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)
    if True:   # lr change through param_groups (what ReduceLROnPlateau does) is honoured
        for opt in (ref_opt, new_opt):
            for grp in opt.param_groups:
                grp['lr'] *= 0.1
        grads = [torch.randn(s, generator=g).cuda() for s in shapes]
        for p, q, gr in zip(ref_p, new_p, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        ref_opt.step()
        new_opt.step(max_norm=0.0)
        for p, q in zip(ref_p, new_p):
            np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)

def test_packed_conv_weights_follow_the_fused_optimizer(hip):
    """FusedClipSGD updates parameters through raw pointers (torch's _version does not move): the cached packed copy
    of a 3x3 conv weight must still be refreshed, or a no_grad forward after a step runs on stale weights (validation
    epochs of models/train_detector.py)."""
    from lib.hip_ops import Conv3x3
    from lib.optim import FusedClipSGD
    from lib.resnet import _Conv
    torch.manual_seed(3)
    conv = Conv3x3(16, 32).cuda()
    rconv = _Conv(16, 32, 3, stride=1, pad=1).cuda()
    x = torch.randn(2, 9, 11, 16, device='cuda')
    with torch.no_grad():
        y0 = conv.forward_nhwc(x, 0)
        r0 = rconv(x)
    opt = FusedClipSGD(list(conv.parameters()) + list(rconv.parameters()), lr=0.5, momentum=0.9, weight_decay=0.0)
    for p in list(conv.parameters()) + list(rconv.parameters()):
        p.grad = torch.randn_like(p)
    opt.step(max_norm=0.0)
    with torch.no_grad():
        y1 = conv.forward_nhwc(x, 0)
        r1 = rconv(x)
    for got, w, b, before in ((y1, conv.weight, conv.bias, y0), (r1, rconv.weight, None, r0)):
        ref = F.conv2d(x.cpu().permute(0, 3, 1, 2), w.detach().cpu(), None if b is None else b.detach().cpu(), padding=1)
        np.testing.assert_allclose(got.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy(), atol=1e-4)
        assert (got - before).abs().max() > 1e-2            # the step really moved the output



def test_device_recall_matches_the_reference_evaluator(hip, golden):
    """lib/evaluation/sg_eval_device.py + mh_triplet_match vs the host evaluator (lib/evaluation/sg_eval.py, itself
    pinned to the reference's BasicSceneGraphEvaluator by tests/golden/sg_eval.npz): Recall@20/50/100 and matches per
    prediction in sgdet mode (boxes and labels both matter).  Model boxes are float32, so the stored float64 prediction
    boxes are rounded to float32 for BOTH sides."""
    from lib.evaluation.sg_eval import evaluate_from_dict
    from lib.evaluation.sg_eval_device import recall_at_k
    g = golden('sg_eval')
    cases = sorted({k.split('_')[0] for k in g if k.startswith('c')})
    assert len(cases) >= 3
    nonzero = 0
    for c in cases:
        a = lambda name: np.asarray(g[c + '_' + name])
        pred_boxes = a('pred_boxes').astype(np.float32)
        result = {'sgdet_recall': {20: [], 50: [], 100: []}}
        pred_to_gt, _, _ = evaluate_from_dict(
            {'gt_relations': a('gt_relations'), 'gt_boxes': a('gt_boxes'), 'gt_classes': a('gt_classes')},
            {'pred_rel_inds': a('pred_rel_inds'), 'rel_scores': a('rel_scores'), 'pred_boxes': pred_boxes.astype(np.float64),
             'pred_classes': a('pred_classes'), 'obj_scores': a('obj_scores')}, 'sgdet', result)
        t = lambda x: torch.from_numpy(np.asarray(x)).cuda()
        rec, nmatch = recall_at_k(t(a('gt_relations')), t(a('gt_boxes')), t(a('gt_classes')), t(a('pred_rel_inds')),
                                  t(a('rel_scores')).float(), t(pred_boxes), t(a('pred_classes')))
        for k in (20, 50, 100):
            assert rec[k] == result['sgdet_recall'][k][0], (c, k)
        np.testing.assert_array_equal(nmatch.cpu().numpy(), np.array([len(m) for m in pred_to_gt]))
        nonzero += int(nmatch.sum())
        # multiple predictions per pair (top 100 of all pair x predicate scores) and phrase detection (union boxes)
        for mode, multi in (('sgdet', True), ('phrdet', False), ('phrdet', True)):
            res2 = {mode + '_recall': {20: [], 50: [], 100: []}}
            p2g, _, _ = evaluate_from_dict(
                {'gt_relations': a('gt_relations'), 'gt_boxes': a('gt_boxes'), 'gt_classes': a('gt_classes')},
                {'pred_rel_inds': a('pred_rel_inds'), 'rel_scores': a('rel_scores').astype(np.float32),
                 'pred_boxes': pred_boxes.astype(np.float64), 'pred_classes': a('pred_classes'),
                 'obj_scores': a('obj_scores').astype(np.float32)}, mode, res2, multiple_preds=multi)
            rec2, nm2 = recall_at_k(t(a('gt_relations')), t(a('gt_boxes')), t(a('gt_classes')), t(a('pred_rel_inds')),
                                    t(a('rel_scores')).float(), t(pred_boxes), t(a('pred_classes')), multiple_preds=multi,
                                    obj_scores=t(a('obj_scores')).float(), phrdet=(mode == 'phrdet'))
            for k in (20, 50, 100):
                assert rec2[k] == res2[mode + '_recall'][k][0], (c, mode, multi, k)
            np.testing.assert_array_equal(nm2.cpu().numpy(), np.array([len(m) for m in p2g]))
    assert nonzero > 0


# ----------------------------------------------------------------------------------------------- reference-kernel goldens
# tests/golden/cuda_ref.npz = outputs of the reference's OWN .cu files compiled for the CPU (oracle/build_ref_cuda.py):
# the HIP kernels are compared with the reference's arithmetic directly, not only with our restatement of it
def test_nms_equals_the_reference_kernel(hip, golden):
    g = golden('cuda_ref')
    for i in g['nms_cases']:
        b, thr = g['nms%d_boxes' % i], float(g['nms%d_thresh' % i])
        keep, num = hip.nms(dev(b).view(-1, 4), thr)
        np.testing.assert_array_equal(keep[:int(num.item())].cpu().numpy(), g['nms%d_keep' % i])
    for j in range(3):
        keep, num = hip.nms(dev(g['nmsb%d_boxes' % j]), float(g['nmsb%d_thresh' % j]))
        np.testing.assert_array_equal(keep[:int(num.item())].cpu().numpy(), g['nmsb%d_keep' % j])


def test_roi_align_equals_the_reference_kernel(hip, golden):
    g = golden('cuda_ref')
    feat, rois = g['roi_feat'], g['roi_rois']
    B, C = feat.shape[:2]
    out = hip.roi_align_fwd(dev(feat), dev(rois), 7, 7, 1.0 / 16, nhwc=False)
    np.testing.assert_array_equal(out.cpu().numpy().view(np.int32), g['roi_out'].view(np.int32))
    out2 = hip.roi_align_fwd(dev(feat.transpose(0, 2, 3, 1)), dev(rois), 7, 7, 1.0 / 16, nhwc=True)
    np.testing.assert_array_equal(out2.cpu().numpy().view(np.int32), g['roi_out'].view(np.int32))
    # backward: the product's gather sums the same terms in a different (fixed) order than the reference's serialised atomics
    gf = hip.roi_align_bwd(dev(g['roi_grad']), dev(rois), B, C, 37, 37, 1.0 / 16, nhwc=False)
    scale = float(np.abs(g['roi_gfeat']).max())
    np.testing.assert_allclose(gf.cpu().numpy(), g['roi_gfeat'], atol=2e-6 * scale)


def test_hwlstm_equals_the_reference_kernels(hip, golden):
    g = golden('cuda_ref')
    H, nl, _ = [int(v) for v in g['lstm_dims']]
    lengths = [int(v) for v in g['lstm_lengths']]
    x, w, bias, drop = dev(g['lstm_x']), dev(g['lstm_w']), dev(g['lstm_bias']), dev(g['lstm_drop'])
    h, c, gates = hip.hwlstm_fwd(x, lengths, w, bias, drop, H, nl, True)
    np.testing.assert_allclose(h.cpu().numpy(), g['lstm_h'], atol=2e-5)
    np.testing.assert_allclose(c.cpu().numpy(), g['lstm_c'], atol=2e-5)
    xg, wg, bg = hip.hwlstm_bwd(dev(g['lstm_gout']), x, lengths, w, drop, H, nl, h, c, gates)
    for got, name in ((xg, 'lstm_gx'), (wg, 'lstm_gw'), (bg, 'lstm_gb')):
        np.testing.assert_allclose(got.cpu().numpy(), g[name], atol=2e-5 * max(1.0, float(np.abs(g[name]).max())))
    h2, _, _ = hip.hwlstm_fwd(x, lengths, w, bias, torch.ones_like(drop), H, nl, False)
    np.testing.assert_allclose(h2.cpu().numpy(), g['lstm_h_eval'], atol=2e-5)


def test_fused_sgd_skips_a_step_with_a_non_finite_gradient_norm(hip):
    """csrc/optim.hip: the clip + SGD kernel must not write weights when the gradient norm is NaN / inf (gradients
    poisoned by a timed-out persistent launch) -- the host runs steps ahead and cannot stop it.  The skip is counted in a
    host-pinned word, FusedClipSGD.step raises on the NEXT call, weights and momentum are untouched, and after the counter
    is cleared training continues from the untouched state (a skipped FIRST step leaves zeroed momentum buffers)."""
    from lib.optim import FusedClipSGD
    L = hip.lib()
    L.mh_opt_skipped_clear()
    torch.manual_seed(0)
    p = [torch.nn.Parameter(torch.randn(300, 70, device='cuda')), torch.nn.Parameter(torch.randn(1000, device='cuda'))]
    opt = FusedClipSGD(p, lr=0.1, momentum=0.9, weight_decay=1e-4)
    before = [q.detach().clone() for q in p]
    for q in p:
        q.grad = torch.randn_like(q)
    p[0].grad[5, 5] = float('nan')
    opt.step(max_norm=5.0)                      # first step, poisoned: skipped on the device
    torch.cuda.synchronize()
    assert L.mh_opt_skipped_steps() == 1
    for q, b in zip(p, before):
        assert torch.equal(q.detach(), b)
    with pytest.raises(hip.HipKernelError, match='skipped on the device'):
        opt.step(max_norm=5.0)
    L.mh_opt_skipped_clear()
    for q in p:
        q.grad = torch.randn_like(q)
    ref = [b - 0.1 * (g * min(1.0, 5.0 / (float(torch.sqrt(sum((x.grad ** 2).sum() for x in p))) + 1e-6)) + 1e-4 * b)
           for b, g in zip(before, [q.grad for q in p])]
    opt.step(max_norm=5.0)                      # momentum buffers were zeroed by the skipped first step: buf = 0.9 * 0 + d
    torch.cuda.synchronize()
    assert L.mh_opt_skipped_steps() == 0
    for q, r in zip(p, ref):
        np.testing.assert_allclose(q.detach().cpu().numpy(), r.cpu().numpy(), atol=1e-6)


# ----------------------------------------------------------------------------------------------- plane engine (round 3)
def test_linear_on_plane_images_forward_backward_and_cache(hip):
    """hip_ops.Linear above the image threshold (2 M N K >= 20 GFLOP): forward, input gradient and weight gradient run
    on cached / freshly made plane images (csrc/pl_gemm.hip) and agree with a float64 product; the weight images follow the
    parameter's value (a raw-pointer update invalidates them)"""
    from lib import hip_ops
    torch.manual_seed(0)
    M, K, N = 1024, 5120, 2048
    lin = hip_ops.Linear(K, N).cuda()
    x = torch.randn(M, K, device='cuda', requires_grad=True)
    y = lin(x, relu=True)
    g = torch.randn_like(y)
    y.backward(g)
    x64, w64, b64 = x.detach().double(), lin.weight.detach().double(), lin.bias.detach().double()
    pre = x64 @ w64.t() + b64
    ref = pre.clamp_min(0)
    gm = g.double() * (pre > 0)
    scale = float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) <= 2e-6 * scale
    flips = ((y.detach() > 0) != (pre > 0))
    assert int(flips.sum()) <= 4
    gm = g.double() * (y.detach() > 0)          # the product's own kink decisions
    gx_ref, gw_ref, gb_ref = gm @ w64, gm.t() @ x64, gm.sum(0)
    for got, want in ((x.grad, gx_ref), (lin.weight.grad, gw_ref), (lin.bias.grad, gb_ref)):
        assert float((got.double() - want).abs().max()) <= 3e-6 * float(want.abs().max())
    key = id(lin.weight)
    assert key in hip_ops._weight_images and hip_ops._weight_images[key][1] is not None and hip_ops._weight_images[key][2] is not None
    old = hip_ops._weight_images[key][1]
    with torch.no_grad():
        lin.weight.data_ptr()
        lin.weight.add_(1.0)                     # torch version counter moves
    y2 = lin(x.detach(), relu=False)
    assert hip_ops._weight_images[key][1] is not old
    ref2 = x64 @ (w64 + 1.0).t() + b64
    assert float((y2.double() - ref2).abs().max()) <= 2e-6 * float(ref2.abs().max())


@pytest.mark.parametrize('direct', ['1', '0'])
def test_vgg_trunk_on_the_plane_engine_matches_the_in_loop_engine(hip, direct, monkeypatch):
    """VGG16Features frozen forward: plane engine (image-output epilogues / converter passes) vs the round-2 kernels vs a
    float64 torch reference, on 3 images of different brightness (per-image scales)"""
    from lib import hip_ops
    torch.manual_seed(1)
    f = hip_ops.VGG16Features().cuda()
    for p in f.parameters():
        p.requires_grad = False
    x = torch.randn(3, 3, 96, 80, device='cuda') * torch.tensor([1.0, 7.0, 0.1], device='cuda').view(3, 1, 1, 1)
    monkeypatch.setenv('MOTIFS_TRUNK', 'planes')
    monkeypatch.setenv('MOTIFS_TRUNK_DIRECT', direct)
    y_pl = f(x).float()
    monkeypatch.setenv('MOTIFS_TRUNK', 'v2')
    y_v2 = f(x).float()
    ref = x.double()
    mods = list(f.children())
    for m in mods:
        if isinstance(m, hip_ops.Conv3x3):
            ref = torch.nn.functional.conv2d(ref, m.weight.double(), m.bias.double(), padding=1).clamp_min(0)
        elif isinstance(m, hip_ops.MaxPool2x2):
            ref = torch.nn.functional.max_pool2d(ref, 2, 2)
    assert y_pl.shape == ref.shape == y_v2.shape
    for b in range(3):
        s = float(ref[b].abs().max())
        e_pl, e_v2 = float((y_pl[b].double() - ref[b]).abs().max()), float((y_v2[b].double() - ref[b]).abs().max())
        print('trunk image %d: max|ref| %.3e  plane engine err %.2e  in-loop engine err %.2e' % (b, s, e_pl, e_v2))
        assert e_pl <= 1e-5 * s and e_v2 <= 1e-5 * s


# ----------------------------------------------------------------------------------------------- co-residency
def test_roi_align_is_exact_next_to_the_in_loop_split_conv_on_another_stream(hip):
    """Regression guard for the round-3 two-stream corruption: RoIAlign launched on a side stream while the in-loop-split conv
    (MFMA + packed-VALU waves) runs on the main stream must return exactly what it returns alone.  With packed FP32 VALU
    instructions in the RoIAlign kernel 44 of 45 such launches were wrong (lanes 48-63, low halves): csrc/build.py compiles
    everything outside the tile engines without them (profiles/r03_packed_f32_coresidency.txt)."""
    torch.manual_seed(0)
    fmap = torch.randn(1, 37, 37, 512, device='cuda').relu_()
    n = 20
    xy = torch.rand(n, 2, device='cuda') * 300
    wh = torch.rand(n, 2, device='cuda') * 250 + 20
    rois = torch.cat((torch.zeros(n, 1, device='cuda'), xy, (xy + wh).clamp(max=591)), 1).contiguous()
    z = torch.randn(380, 7, 7, 256, device='cuda')
    wt = hip.conv3x3_pack_weight(torch.randn(512, 256, 3, 3, device='cuda') * 0.01, False)
    bias = torch.zeros(512, device='cuda')
    big, w6 = torch.randn(380, 25088, device='cuda'), torch.randn(4096, 25088, device='cuda') * 0.01
    ref = hip.roi_align_fwd(fmap, rois, 7, 7, 1 / 16, True)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    wrong = 0
    for aggressor in (lambda: hip.conv3x3_nhwc(z, wt, bias, 1), lambda: hip.gemm_inloop(big, w6, False, True)):
        for _ in range(10):
            torch.cuda.synchronize()
            side.wait_stream(torch.cuda.current_stream())
            keep = [aggressor() for _ in range(2)]
            with torch.cuda.stream(side):
                outs = [hip.roi_align_fwd(fmap, rois, 7, 7, 1 / 16, True) for _ in range(3)]
            keep += [aggressor() for _ in range(2)]
            torch.cuda.synchronize()
            wrong += sum(not torch.equal(o, ref) for o in outs)
    assert wrong == 0, '%d of 60 concurrent RoIAlign launches differ from the stand-alone result' % wrong
