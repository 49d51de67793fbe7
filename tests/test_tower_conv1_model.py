"""Index model of the direct first convolution of the mask tower (neural-motifs_amd/csrc/tower.hip tower_conv1_*): the padded NHWC
mask copy, the window address of an output pair, the k = (ky*7 + kx)*2 + ci weight order and the host-side layout conversions
of lib/get_union_boxes.py, replayed in numpy against torch's conv2d (/root/reference lib/get_union_boxes.py:31:
nn.Conv2d(2, dim // 2, kernel_size=7, stride=2, padding=3)).  The kernels themselves are compared with the same reference on
the GPU (tests/test_gpu_ops.py::test_tower_conv1_direct_*); this file needs none."""
import numpy as np
import torch
import torch.nn.functional as F

K, CI, STRIDE, PAD = 7, 2, 2, 3
ROW, PAIR = K * CI, K * CI + STRIDE * CI            # 14 floats of one kernel row, 18 for two neighbouring outputs


def pad_nhwc(rects):                                 # tower_pad_kernel
    N, S = rects.shape[0], rects.shape[1]
    xp = np.zeros((N, S + 2 * PAD, S + 2 * PAD, CI), np.float32)
    xp[:, PAD:PAD + S, PAD:PAD + S] = rects
    return xp


def fwd_model(xp, w_kc, bias):                       # tower_conv1_fwd_kernel, all channels at once
    N, Sp = xp.shape[0], xp.shape[1]
    Ho = (Sp - K) // STRIDE + 1
    flat = xp.reshape(N, -1)
    y = np.zeros((N, Ho, Ho, w_kc.shape[1]), np.float32)
    for oy in range(Ho):
        for ox in range(0, Ho, 2):
            win = ((oy * STRIDE) * Sp + ox * STRIDE) * CI
            a0 = np.tile(bias, (N, 1)).astype(np.float64)
            a1 = a0.copy()
            for ky in range(K):
                r = flat[:, win + ky * Sp * CI: win + ky * Sp * CI + PAIR]                  # [N, 18]
                a0 += r[:, :ROW] @ w_kc[ky * ROW:(ky + 1) * ROW]
                a1 += r[:, STRIDE * CI:STRIDE * CI + ROW] @ w_kc[ky * ROW:(ky + 1) * ROW]
            y[:, oy, ox], y[:, oy, ox + 1] = np.maximum(a0, 0), np.maximum(a1, 0)
    return y


def wgrad_model(xp, dy):                             # tower_conv1_wgrad_kernel + reduce
    N, Sp = xp.shape[0], xp.shape[1]
    Ho, C0 = dy.shape[1], dy.shape[3]
    flat = xp.reshape(N, -1)
    acc = np.zeros((K * ROW + 1, C0), np.float64)
    for oy in range(Ho):
        for j in range(0, Ho, 2):
            win = (oy * STRIDE * Sp + j * STRIDE) * CI
            v0, v1 = dy[:, oy, j].astype(np.float64), dy[:, oy, j + 1].astype(np.float64)   # [N, C0]
            acc[K * ROW] += (v0 + v1).sum(0)
            for ky in range(K):
                x = flat[:, win + ky * Sp * CI: win + ky * Sp * CI + PAIR].astype(np.float64)
                acc[ky * ROW:(ky + 1) * ROW] += x[:, :ROW].T @ v0 + x[:, STRIDE * CI:STRIDE * CI + ROW].T @ v1
    return acc[:K * ROW], acc[K * ROW]


def test_direct_conv1_index_model_equals_conv2d():
    rs = np.random.RandomState(3)
    N, S, C0 = 5, 27, 8
    rects = rs.rand(N, S, S, CI).astype(np.float32)
    w = torch.tensor(rs.randn(C0, CI, K, K).astype(np.float32) * 0.1, requires_grad=True)
    b = torch.tensor(rs.randn(C0).astype(np.float32) * 0.1, requires_grad=True)
    x = torch.tensor(rects).permute(0, 3, 1, 2)
    y_ref = F.relu(F.conv2d(x, w, b, stride=STRIDE, padding=PAD))                           # [N, C0, 14, 14]
    # host-side conversions of lib/get_union_boxes.py
    w_kc = w.detach().permute(2, 3, 1, 0).reshape(K * K * CI, C0).numpy()
    xp = pad_nhwc(rects)
    y = fwd_model(xp, w_kc, b.detach().numpy())
    assert y.shape == (N, 14, 14, C0)
    np.testing.assert_allclose(y, y_ref.detach().permute(0, 2, 3, 1).numpy(), rtol=1e-5, atol=1e-6)
    # weight / bias gradient for a gradient arriving at the PRE-ReLU output
    dy = rs.randn(N, 14, 14, C0).astype(np.float32)
    pre = F.conv2d(x, w, b, stride=STRIDE, padding=PAD)
    pre.backward(torch.tensor(dy).permute(0, 3, 1, 2))
    dwk, db = wgrad_model(xp, dy)
    dw0 = torch.tensor(dwk).view(K, K, CI, C0).permute(3, 2, 0, 1)                          # as in _TowerFn.backward
    np.testing.assert_allclose(dw0.numpy(), w.grad.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(db, b.grad.numpy(), rtol=1e-5, atol=1e-5)
