"""Oracle C restatement vs the reference's own outputs (tests/golden) + brute-force self checks."""
import numpy as np
import pytest

from oracle import native


@pytest.mark.parametrize('P', [27, 7])
def test_draw_union_boxes_matches_reference_pyx(golden, P):
    g = golden('draw_P%d' % P)
    out = native.draw_union_boxes(g['pairs'], P)
    # bit-exact: same fp32 expression order as draw_rectangles.pyx:41-66
    assert out.dtype == np.float32
    np.testing.assert_array_equal(out, g['masks'])


def test_draw_union_boxes_properties():
    rs = np.random.RandomState(0)
    b = rs.uniform(0, 500, (50, 4)).astype(np.float32)
    b[:, 2:] = b[:, :2] + rs.uniform(2, 90, (50, 2)).astype(np.float32)
    pairs = np.concatenate([b, b[::-1]], 1)
    m = native.draw_union_boxes(pairs, 27)
    assert m.min() >= 0 and m.max() <= 1
    # swapping the two boxes swaps the channels
    m2 = native.draw_union_boxes(np.concatenate([pairs[:, 4:], pairs[:, :4]], 1), 27)
    np.testing.assert_array_equal(m[:, 0], m2[:, 1])
    # a box equal to the union fills its channel with ones
    u = np.concatenate([np.minimum(pairs[:, :2], pairs[:, 4:6]), np.maximum(pairs[:, 2:4], pairs[:, 6:8])], 1)
    m3 = native.draw_union_boxes(np.concatenate([u, pairs[:, 4:]], 1), 27)
    np.testing.assert_allclose(m3[:, 0], 1.0, atol=2e-5)


def test_bbox64_matches_reference_pyx(golden):
    g = golden('bbox64')
    np.testing.assert_array_equal(native.bbox_overlaps(g['a'], g['q']), g['overlaps'])
    np.testing.assert_array_equal(native.bbox_intersections(g['a'], g['q']), g['intersections'])


def _rand_sorted_boxes(rs, n):
    x1 = rs.uniform(0, 500, n)
    y1 = rs.uniform(0, 500, n)
    w = rs.uniform(4, 200, n)
    h = rs.uniform(4, 200, n)
    return np.stack([x1, y1, x1 + w, y1 + h], 1).astype(np.float32)


def _py_nms(boxes, thresh):
    """Independent float32 python restatement (numpy scalars keep every op in fp32)."""
    f = np.float32
    n = len(boxes)
    removed = np.zeros(n, bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        a = boxes[i]
        for j in range(i + 1, n):
            b = boxes[j]
            w = max(f(min(a[2], b[2]) - max(a[0], b[0])) + f(1), f(0))
            h = max(f(min(a[3], b[3]) - max(a[1], b[1])) + f(1), f(0))
            inter = f(w * h)
            sa = f(f(a[2] - a[0] + f(1)) * f(a[3] - a[1] + f(1)))
            sb = f(f(b[2] - b[0] + f(1)) * f(b[3] - b[1] + f(1)))
            if f(inter / f(f(sa + sb) - inter)) > f(thresh):
                removed[j] = True
    return np.array(keep, dtype=np.int32)


@pytest.mark.parametrize('n,thresh', [(0, 0.5), (1, 0.5), (63, 0.3), (64, 0.7), (65, 0.5), (300, 0.3), (1000, 0.7)])
def test_nms_greedy_equals_bitmask_and_bruteforce(n, thresh):
    rs = np.random.RandomState(n + 1)
    boxes = _rand_sorted_boxes(rs, n)
    k1 = native.nms(boxes, thresh)
    k2 = native.nms(boxes, thresh, bitmask=True)
    np.testing.assert_array_equal(k1, k2)
    if n <= 300:
        np.testing.assert_array_equal(k1, _py_nms(boxes, thresh))
    # idempotence: NMS of the kept set keeps everything
    if n:
        np.testing.assert_array_equal(native.nms(boxes[k1], thresh), np.arange(len(k1)))


def test_roi_align_constant_map_and_outside():
    feat = np.full((2, 3, 37, 37), 2.5, np.float32)
    rois = np.array([[0, 16, 16, 200, 300], [1, 0, 0, 591, 591], [1, 300, 300, 700, 700], [5, 0, 0, 10, 10]],
                    np.float32)
    out = native.roi_align_fwd(feat, rois)
    np.testing.assert_allclose(out[0], 2.5, rtol=1e-6)
    # 591/576 > 1 -> the last sampling row/col fall outside the map and are zero-filled
    assert np.all(out[1][:, :6, :6] == 2.5) and np.all(out[1][:, 6, :] == 0) and np.all(out[1][:, :, 6] == 0)
    assert np.all(out[2][:, -1, :] == 0)
    assert np.all(out[3] == 0)          # image index out of range: untouched (zero) output


def test_roi_align_bilinear_exact_on_linear_ramp():
    H = W = 37
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing='ij')
    feat = (3 * yy + 0.5 * xx)[None, None]
    rois = np.array([[0, 32, 48, 400, 320]], np.float32)
    out = native.roi_align_fwd(feat, rois)[0, 0]
    x1, y1, x2, y2 = [v / 576.0 for v in rois[0, 1:]]
    for y in range(7):
        for x in range(7):
            in_y = y1 * 36 + y * (y2 - y1) * 36 / 6
            in_x = x1 * 36 + x * (x2 - x1) * 36 / 6
            np.testing.assert_allclose(out[y, x], 3 * in_y + 0.5 * in_x, rtol=1e-5)


def test_roi_align_backward_is_adjoint_of_forward():
    rs = np.random.RandomState(3)
    feat = rs.randn(2, 4, 37, 37).astype(np.float32)
    rois = np.array([[0, 10, 20, 300, 400], [1, 100, 50, 580, 560], [0, 5, 5, 60, 90]], np.float32)
    g = rs.randn(3, 4, 7, 7).astype(np.float32)
    out = native.roi_align_fwd(feat, rois)
    gf = native.roi_align_bwd(g, rois, feat.shape)
    # <fwd(feat), g> == <feat, bwd(g)>  (the op is linear in feat)
    np.testing.assert_allclose((out.astype(np.float64) * g).sum(), (feat.astype(np.float64) * gf).sum(), rtol=1e-4)
