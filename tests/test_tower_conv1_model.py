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


# ---- round 6: the same layer on the matrix cores (csrc/tower.hip t1::tower_conv1_mfma_*), lane by lane ---------------------------------
# v_mfma_f32_32x32x16: operand 1 = rows i, operand 2 = columns j; lane (j = lane & 31, g = lane >> 5) of an operand holds k = 8 g .. 8 g + 7
# of row / column j; the result D[i][j] lives in lane (j, g) register r with i = 8 (r >> 2) + 4 g + (r & 3).
def mfma_32x32x16(op1, op2):
    """op1 / op2 [64 lanes][8] -> D as [64 lanes][16 registers] (fp64 arithmetic: the index model, not the rounding)"""
    A = np.zeros((32, 16)); B = np.zeros((32, 16))
    for lane in range(64):
        j, g = lane & 31, lane >> 5
        A[j, 8 * g:8 * g + 8] = op1[lane]
        B[j, 8 * g:8 * g + 8] = op2[lane]
    D = A @ B.T                                       # [i][j]
    out = np.zeros((64, 16))
    for lane in range(64):
        j, g = lane & 31, lane >> 5
        for r in range(16):
            out[lane, r] = D[8 * (r >> 2) + 4 * g + (r & 3), j]
    return out


def test_matrix_core_forward_fragments_are_the_convolution():
    """forward: the padded k' = 16 ky + (2 kx + ci) order, a lane's window-row reads (floats 8 g .. 8 g + 7 of kernel row ky: 14 real, the
    last two of the second half forced to zero), weights first / pixels second (transposed accumulators), registers 4 q .. 4 q + 3 =
    channels 8 q + 4 g + (0..3) of pixel j"""
    rs = np.random.RandomState(0)
    N, S, C0 = 2, 27, 32                                # one 32-channel block of a wave, two masks
    rects = rs.rand(N, S, S, CI).astype(np.float32)
    w = rs.randn(C0, CI, K, K).astype(np.float32) * 0.1
    bias = rs.randn(C0).astype(np.float32) * 0.1
    xp = pad_nhwc(rects)
    Sp, Ho = S + 2 * PAD, (S + 2 * PAD - K) // STRIDE + 1
    flat = xp.reshape(-1)
    w_kc = np.transpose(w, (2, 3, 1, 0)).reshape(K * ROW, C0)            # lib/get_union_boxes.py: k = (ky*7 + kx)*2 + ci
    ref = F.relu(F.conv2d(torch.from_numpy(rects).permute(0, 3, 1, 2).double(), torch.from_numpy(w).double(), torch.from_numpy(bias).double(),
                          stride=STRIDE, padding=PAD)).permute(0, 2, 3, 1).numpy()
    M = N * Ho * Ho
    y = np.zeros((M, C0))
    for tile in range((M + 31) // 32):
        acc = np.zeros((64, 16))
        for ky in range(K):
            wfrag, afrag = np.zeros((64, 8)), np.zeros((64, 8))
            for lane in range(64):
                j, g = lane & 31, lane >> 5
                for i in range(8):
                    t = 8 * g + i
                    wfrag[lane, i] = w_kc[ky * ROW + t, j] if t < ROW else 0.0                 # channel j of the block
                p = min(tile * 32 + j, M - 1)
                n, rem = divmod(p, Ho * Ho)
                oy, ox = divmod(rem, Ho)
                row = ((n * Sp + oy * STRIDE + ky) * Sp + ox * STRIDE) * CI + 8 * g
                vals = list(flat[row:row + 6])                                                 # 16-byte + 8-byte loads: always inside the row
                vals += list(flat[row + 6:row + 8]) if g == 0 else [0.0, 0.0]                  # floats 14, 15 of the padded kernel row
                afrag[lane] = vals
            acc += mfma_32x32x16(wfrag, afrag)
        for lane in range(64):
            j, g = lane & 31, lane >> 5
            p = tile * 32 + j
            if p < M:
                for q in range(4):
                    for r in range(4):
                        c = 8 * q + 4 * g + r
                        y[p, c] = max(acc[lane, 4 * q + r] + bias[c], 0.0)
    np.testing.assert_allclose(y.reshape(N, Ho, Ho, C0), ref, atol=1e-6)


def test_matrix_core_weight_gradient_fragments_are_the_gradient():
    """weight gradient: one OUTPUT ROW (14 pixels, padded to 16) per k-tile; tap row 32 mb + j = kernel row 2 mb + (j >> 4), column j & 15;
    a lane's 8 mask values sit 4 floats apart; register r of tap block mb = tap row 32 mb + 8 (r >> 2) + 4 g + (r & 3) of channel j; pad
    rows and pad pixels contribute nothing; the bias gradient is the plain sum of the gradients"""
    rs = np.random.RandomState(1)
    N, S, C0 = 3, 27, 32
    rects = rs.rand(N, S, S, CI).astype(np.float32)
    Sp, Ho = S + 2 * PAD, (S + 2 * PAD - K) // STRIDE + 1
    dy = rs.randn(N, Ho, Ho, C0).astype(np.float32)
    xp = pad_nhwc(rects)
    flat = xp.reshape(-1)
    x64 = torch.from_numpy(rects).permute(0, 3, 1, 2).double()
    w = torch.zeros(C0, CI, K, K, dtype=torch.float64, requires_grad=True)
    F.conv2d(x64, w, stride=STRIDE, padding=PAD).backward(torch.from_numpy(dy).permute(0, 3, 1, 2).double())
    ref = np.transpose(w.grad.numpy(), (2, 3, 1, 0)).reshape(K * ROW, C0)               # [k][c], k = (ky*7 + kx)*2 + ci
    acc = np.zeros((4, 64, 16))
    bsum = np.zeros(64)
    for row in range(N * Ho):
        n, oy = divmod(row, Ho)
        dfrag = np.zeros((64, 8))
        for lane in range(64):
            j, g = lane & 31, lane >> 5
            for i in range(8):
                if 8 * g + i < Ho:
                    dfrag[lane, i] = dy[n, oy, 8 * g + i, j]
            bsum[lane] += dfrag[lane].sum()
        for mb in range(4):
            afrag = np.zeros((64, 8))
            for lane in range(64):
                j, g = lane & 31, lane >> 5
                ky, t = 2 * mb + (j >> 4), j & 15
                if ky < K and t < ROW:
                    base = ((n * Sp + oy * STRIDE) * Sp) * CI + (8 * g) * STRIDE * CI + ky * Sp * CI + t
                    for i in range(8):
                        if 8 * g + i < Ho:
                            afrag[lane, i] = flat[base + i * STRIDE * CI]
            acc[mb] += mfma_32x32x16(afrag, dfrag)
    got = np.zeros((K * ROW, C0))
    for mb in range(4):
        for lane in range(64):
            j, g = lane & 31, lane >> 5
            for r in range(16):
                kp = 32 * mb + 8 * (r >> 2) + 4 * g + (r & 3)
                ky, t = kp >> 4, kp & 15
                if ky < K and t < ROW:
                    got[ky * ROW + t, j] = acc[mb, lane, r]
    np.testing.assert_allclose(got, ref, atol=1e-9 * max(1.0, float(np.abs(ref).max())) + 1e-6)
    db = np.array([bsum[j] + bsum[j + 32] for j in range(32)])
    np.testing.assert_allclose(db, dy.reshape(-1, C0).sum(0), atol=1e-4)
