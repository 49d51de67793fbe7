"""
TEST-ONLY stand-in for the HIP kernels, so that the product's *host logic* (module wiring, index arithmetic,
autograd plumbing, samplers, drivers) can be exercised in the GPU-less build container.

`install()` monkey-patches the functions of `lib._hip` with CPU implementations backed by the oracle / plain torch.
Nothing in the product imports this module; the product itself has no CPU path and raises without a HIP device
(tests/test_cabi.py::test_hot_path_fails_loudly_without_gpu).  Numerical parity is NOT established here -- that is
the job of the `-m gpu` tests, which run the real kernels.
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import lstm as OL
from oracle import native


_GATES = {}


def _np(t):
    return t.detach().cpu().numpy()


def install():
    from lib import _hip
    from dataloaders import blob as blob_mod

    def gemm(a, b, trans_a=False, trans_b=False, bias=None, epilogue=0, out=None, accumulate=False, splitk=0):
        r = (a.t() if trans_a else a) @ (b.t() if trans_b else b)
        if bias is not None:
            r = r + bias
        r = {0: r, 1: F.relu(r), 2: F.relu6(r)}[epilogue]
        if out is not None:
            out.copy_(out + r if accumulate else r)
            return out
        return r

    def make_planes(x, k_contiguous=True):
        m = x.detach() if k_contiguous else x.detach().t()
        return _hip.PlaneImage(m.contiguous().clone(), m.shape[0], m.shape[1])      # .buf = the dense [rows, K] operand

    def make_planes_both(x):
        return make_planes(x, True), make_planes(x, False)

    def gemm_planes(a, b, bias=None, epilogue=0, out=None, accumulate=False, splitk=0):
        return gemm(a.buf, b.buf, False, True, bias=bias, epilogue=epilogue, out=out, accumulate=accumulate)

    def gemm_inloop(a, b, trans_a=False, trans_b=False, bias=None, epilogue=0, out=None, accumulate=False):
        return gemm(a, b, trans_a, trans_b, bias=bias, epilogue=epilogue, out=out, accumulate=accumulate)

    def nms(boxes_sorted, thresh):
        keep = native.nms(_np(boxes_sorted), thresh)
        k = torch.zeros(max(boxes_sorted.shape[0], 1), dtype=torch.int32)
        k[:len(keep)] = torch.from_numpy(keep)
        return k, torch.tensor([len(keep)], dtype=torch.int32)

    def nms_batched(boxes_sorted, seg_offsets, max_seg, thresh):
        offs = seg_offsets.tolist()
        keep = torch.zeros(max(boxes_sorted.shape[0], 1), dtype=torch.int32)
        num = torch.zeros(max(len(offs) - 1, 1), dtype=torch.int32)
        for s in range(len(offs) - 1):
            k = native.nms(_np(boxes_sorted[offs[s]:offs[s + 1]]), thresh)
            keep[offs[s]:offs[s] + len(k)] = torch.from_numpy(k)
            num[s] = len(k)
        return keep, num

    def roi_align_fwd(feat, rois, ph, pw, spatial_scale, nhwc):
        f = feat.permute(0, 3, 1, 2) if nhwc else feat
        return torch.from_numpy(native.roi_align_fwd(_np(f), _np(rois), ph, pw, spatial_scale))

    def roi_align_bwd(grad_out, rois, B, C, H, W, spatial_scale, nhwc):
        g = torch.from_numpy(native.roi_align_bwd(_np(grad_out), _np(rois), (B, C, H, W), spatial_scale))
        return g.permute(0, 2, 3, 1).contiguous() if nhwc else g

    def draw_union_boxes(box_pairs, P, offset=0.0, channels_last=False):
        m = torch.from_numpy(native.draw_union_boxes(_np(box_pairs), P)) + np.float32(offset)
        return m.permute(0, 2, 3, 1).contiguous() if channels_last else m

    def bbox_overlaps(a, b):
        from oracle import boxes as OB
        return OB.bbox_overlaps(a, b)

    def conv3x3_pack_weight(w, flip_transpose=False):
        # packed layout [9][Cout'][Cin'] of the conv that consumes it (flip_transpose: the dgrad conv, roles swapped)
        if flip_transpose:
            return w.flip(2, 3).permute(2, 3, 1, 0).reshape(9, w.shape[1], w.shape[0]).contiguous()
        return w.permute(2, 3, 0, 1).reshape(9, w.shape[0], w.shape[1]).contiguous()

    def conv3x3_nhwc(x, wt, bias, epilogue):
        cout, cin = wt.shape[1], wt.shape[2]
        w = wt.view(3, 3, cout, cin).permute(2, 3, 0, 1)
        y = F.conv2d(x.permute(0, 3, 1, 2), w, bias, padding=1)
        y = {0: y, 1: F.relu(y), 2: F.relu6(y)}[epilogue]
        return y.permute(0, 2, 3, 1).contiguous()

    def conv_first_nchw_max(x, w, bias, epilogue, maxbits):
        y = conv_first_nchw(x, w, bias, epilogue)
        maxbits.copy_(y.abs().amax(dim=(1, 2, 3)).view(torch.int32))
        return y

    def act_planes(x_nhwc, maxbits, pool=False):
        y = F.max_pool2d(x_nhwc.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).contiguous() if pool else x_nhwc
        return _hip.ActImage(y, y.shape[0], y.shape[1], y.shape[2], y.shape[3])      # .buf = the dense NHWC tensor

    def plconv_pack_weight(w, flip_transpose=False):
        return conv3x3_pack_weight(w, flip_transpose)

    def plconv3x3(img, packed, cout, bias, epilogue, out_maxbits=None):
        y = conv3x3_nhwc(img.buf, packed, bias, epilogue)
        if out_maxbits is not None:
            out_maxbits.copy_(y.abs().amax(dim=(1, 2, 3)).view(torch.int32))
        return y

    def plconv3x3_to_image(img, in_true_maxbits, packed, cout, bias, epilogue, out_maxbits):
        y = plconv3x3(img, packed, cout, bias, epilogue, out_maxbits)
        return _hip.ActImage(y, y.shape[0], y.shape[1], y.shape[2], y.shape[3])

    def plconv3x3_pool_to_image(img, in_true_maxbits, packed, cout, bias, epilogue, out_maxbits):
        y = maxpool2x2_nhwc(plconv3x3(img, packed, cout, bias, epilogue, out_maxbits))
        return _hip.ActImage(y, y.shape[0], y.shape[1], y.shape[2], y.shape[3])

    def stem_to_image(x, w, bias, epilogue, out_maxbits):
        y = conv_first_nchw_max(x, w, bias, epilogue, out_maxbits)
        return _hip.ActImage(y, y.shape[0], y.shape[1], y.shape[2], y.shape[3])

    def conv_first_nchw(x, w, bias, epilogue):
        y = F.conv2d(x, w, bias, padding=1)
        y = {0: y, 1: F.relu(y), 2: F.relu6(y)}[epilogue]
        return y.permute(0, 2, 3, 1).contiguous()

    def maxpool2x2_nhwc(x):
        return F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).contiguous()

    def conv3x3_wgrad(x, gy):
        return None                       # the plumbing tests exercise the im2col + gemm alternative

    def maxpool2x2_bwd_nhwc(x, gy):
        with torch.enable_grad():                        # called from inside a backward pass
            xx = x.permute(0, 3, 1, 2).detach().clone().requires_grad_()
            F.max_pool2d(xx, 2, 2).backward(gy.permute(0, 3, 1, 2))
        return xx.grad.permute(0, 2, 3, 1).contiguous()

    def act_bwd(g, y, epilogue):
        m = (y > 0) if epilogue == 1 else ((y > 0) & (y < 6))
        return g * m.to(g.dtype)

    # ---- BatchNorm kernels (ResNet branch) ----
    def bn_stats(x2d, eps, momentum, running_mean=None, running_var=None):
        M = x2d.shape[0]
        mean = x2d.double().mean(0)
        var = x2d.double().var(0, unbiased=False)
        if running_mean is not None:
            running_mean.mul_(1 - momentum).add_(momentum * mean.float())
            running_var.mul_(1 - momentum).add_(momentum * (var * M / max(M - 1, 1)).float())
        return mean.float(), (1.0 / torch.sqrt(var + eps)).float()

    def bn_apply_nhwc(x, mean, invstd, gamma, beta, residual=None, relu=False):
        y = (x - mean) * invstd * gamma + beta
        if residual is not None:
            y = y + residual
        return torch.relu(y) if relu else y

    def bn_pool_fwd(x, mean, invstd, gamma, beta):
        # arg-max as the kernel stores it: window position ky * 3 + kx of the 3x3 / 2 / pad 1 window, first maximum wins
        N, H, W, C = x.shape
        y = ((x - mean) * invstd * gamma + beta).permute(0, 3, 1, 2)
        z, flat = F.max_pool2d(y, 3, 2, 1, return_indices=True)               # flat = iy * W + ix in the input plane
        oy = torch.arange(H // 2).view(1, 1, -1, 1)
        ox = torch.arange(W // 2).view(1, 1, 1, -1)
        arg = (torch.div(flat, W, rounding_mode='floor') - (2 * oy - 1)) * 3 + (flat % W - (2 * ox - 1))
        return z.permute(0, 2, 3, 1).contiguous(), arg.permute(0, 2, 3, 1).contiguous().to(torch.uint8)

    def bn_bwd(x, g, argmax, mean, invstd, gamma, relu_mask):
        assert not relu_mask, 'shim: BatchNorm backward without the producer-ReLU mask only'
        if argmax is not None:
            # pooled form: g belongs to the 3x3 / 2 max-pool's output; route it to the arg-max positions first
            N, H, W, C = x.shape
            a = argmax.long()
            oy = torch.arange(H // 2).view(1, -1, 1, 1)
            ox = torch.arange(W // 2).view(1, 1, -1, 1)
            iy, ix = 2 * oy - 1 + torch.div(a, 3, rounding_mode='floor'), 2 * ox - 1 + a % 3
            n = torch.arange(N).view(-1, 1, 1, 1).expand_as(a)
            c = torch.arange(C).view(1, 1, 1, -1).expand_as(a)
            dense = torch.zeros(N * H * W * C, dtype=torch.float64)
            dense.index_add_(0, (((n * H + iy) * W + ix) * C + c).reshape(-1), g.reshape(-1).double())
            g = dense.view(N, H, W, C)
        C = x.shape[-1]
        x2, g2 = x.reshape(-1, C).double(), g.reshape(-1, C).double()
        M = x2.shape[0]
        xhat = (x2 - mean.double()) * invstd.double()
        dbeta, dgamma = g2.sum(0), (g2 * xhat).sum(0)
        dx = gamma.double() * invstd.double() * (g2 - dbeta / M - xhat * dgamma / M)
        return dx.float().view(x.shape), dgamma.float(), dbeta.float()

    def im2col_nhwc(x, kh, kw, stride, pad, ldo=None):
        B, H, W, C = x.shape
        Ho = (H + 2 * pad - kh) // stride + 1
        Wo = (W + 2 * pad - kw) // stride + 1
        cols = F.unfold(x.permute(0, 3, 1, 2), (kh, kw), padding=pad, stride=stride)      # [B, C*kh*kw, L]
        cols = cols.view(B, C, kh * kw, Ho * Wo).permute(0, 3, 2, 1).reshape(B * Ho * Wo, kh * kw * C)
        K = kh * kw * C
        ldo = K if ldo is None else ldo
        out = x.new_zeros(B * Ho * Wo, ldo)
        out[:, :K] = cols
        return out, Ho, Wo

    def nchw_to_nhwc(x):
        return x.permute(0, 2, 3, 1).contiguous()

    def nhwc_to_nchw(x):
        return x.permute(0, 3, 1, 2).contiguous()

    def hwlstm_fwd(x, lengths, weight, bias, dropout, H, L_, training):
        out, h_slots, c_slots, gates = OL.highway_lstm_forward(x, lengths, weight, bias, dropout, H, L_, True,
                                                               return_state=True)
        T, B = x.shape[0], x.shape[1]
        h_data = torch.stack([torch.stack(h_slots[l]) for l in range(L_)]).detach()
        c_data = torch.stack([torch.stack(c_slots[l]) for l in range(L_)]).detach()
        if not training:
            return h_data, c_data, None
        token = torch.zeros(1)                      # a tensor autograd can save; the python-side gates ride along
        _GATES[token.data_ptr()] = (token, gates)
        return h_data, c_data, token

    def hwlstm_bwd(out_grad, x, lengths, weight, dropout, H, L_, h_data, c_data, gates, need_weight_grad=True):
        h_slots = [[h_data[l, t] for t in range(h_data.shape[1])] for l in range(L_)]
        c_slots = [[c_data[l, t] for t in range(c_data.shape[1])] for l in range(L_)]
        gates = _GATES.pop(gates.data_ptr())[1]
        xg, wg, bg = OL.highway_lstm_backward(out_grad, x, lengths, weight, dropout, H, L_, h_slots, c_slots, gates)
        return xg, (wg if need_weight_grad else None), (bg if need_weight_grad else None)

    def hwcell_seq_supported(H, B):
        return False                      # the plumbing tests exercise the per-step path

    def hwlstm_cell_fwd(pre_i, h_prev, c_prev, wh_t, bias_h, dropout, want_gates):
        H = h_prev.shape[1]
        ps = h_prev @ wh_t.t() + (bias_h if bias_h is not None else 0)
        g = pre_i[:, :5 * H] + ps
        ig, fg = torch.sigmoid(g[:, :H]), torch.sigmoid(g[:, H:2 * H])
        ag, og, rg = torch.tanh(g[:, 2 * H:3 * H]), torch.sigmoid(g[:, 3 * H:4 * H]), torch.sigmoid(g[:, 4 * H:5 * H])
        lin = pre_i[:, 5 * H:]
        c = fg * c_prev + ig * ag
        h = rg * (og * torch.tanh(c)) + (1 - rg) * lin
        if dropout is not None:
            h = h * dropout
        return h, c, (torch.cat((ig, fg, ag, og, rg, lin), 1) if want_gates else None)

    def hwlstm_cell_bwd(d_h, d_c_out, c_prev, c_out, gates, dropout):
        H = d_h.shape[1]
        ig, fg, ag, og, rg, lin = [gates[:, k * H:(k + 1) * H] for k in range(6)]
        dh = d_h * dropout if dropout is not None else d_h
        tc = torch.tanh(c_out)
        d_o = dh * rg
        d_c = d_o * og * (1 - tc * tc) + (d_c_out if d_c_out is not None else 0)
        dg = torch.cat((d_c * ag * ig * (1 - ig), d_c * c_prev * fg * (1 - fg), d_c * ig * (1 - ag * ag),
                        d_o * tc * og * (1 - og), dh * (og * tc - lin) * rg * (1 - rg), dh * (1 - rg)), 1)
        return dg, fg * d_c

    def gemv_rows(v, wt, bias=None):
        r = v @ wt.t()
        return r + bias if bias is not None else r

    def check_faults():
        return None

    # ---- FusedClipSGD's two multi-tensor kernels over the SAME chunk table, on host memory (raw pointers via ctypes):
    # lets the world_size-2 gloo test drive the real optimizer object (pointer table, re-pointed .grad buffers)
    import ctypes
    _rec = np.dtype([('p', '<u8'), ('g', '<u8'), ('buf', '<u8'), ('n', '<i4'), ('lr', '<f4')])

    def _val(a):
        return a.value if hasattr(a, 'value') else a

    def _arr(addr, n, dt=np.float32):
        return np.ctypeslib.as_array((ctypes.c_float * int(n)).from_address(int(addr))) if dt is np.float32 else None

    def _table(tptr, n):
        raw = (ctypes.c_uint8 * (int(n) * _rec.itemsize)).from_address(int(_val(tptr)))
        return np.frombuffer(raw, dtype=_rec, count=int(n))

    class ShimLib(object):
        @staticmethod
        def mh_opt_chunk_elems():
            return 1 << 16

        @staticmethod
        def mh_fault_pending():
            return 0

        skipped = [0]

        @staticmethod
        def mh_opt_skipped_steps():
            return ShimLib.skipped[0]

        @staticmethod
        def mh_opt_skipped_clear():
            ShimLib.skipped[0] = 0
            return 0

        @staticmethod
        def mh_opt_build_chunks(params, nparams, chunks, nchunks, stream):
            prm = _table(params, _val(nparams))
            out = _table(chunks, _val(nchunks))
            chunk, c = 1 << 16, 0
            for r in prm:
                for off in range(0, int(r['n']), chunk):
                    out[c] = (r['p'] + 4 * off, r['g'] + 4 * off, r['buf'] + 4 * off, min(chunk, int(r['n']) - off), r['lr'])
                    c += 1
            assert c == _val(nchunks)
            return 0

        @staticmethod
        def mh_multi_sumsq(tptr, n, partial, sumsq, stream):
            tot = 0.0
            for r in _table(tptr, _val(n)):
                g = _arr(r['g'], r['n'])
                tot += float(np.dot(g.astype(np.float64), g.astype(np.float64)))
            _arr(_val(sumsq), 1)[0] = tot
            return 0

        @staticmethod
        def mh_multi_sgd_step(tptr, n, sumsq, max_norm, momentum, wd, first, stream):
            max_norm, momentum, wd, first = float(_val(max_norm)), float(_val(momentum)), float(_val(wd)), int(_val(first))
            scale = 1.0
            if _val(sumsq) and max_norm > 0:
                ss = float(_arr(_val(sumsq), 1)[0])
                if not np.isfinite(ss):                  # the kernel's device-side guard: skip, count, zero a first-step buffer
                    ShimLib.skipped[0] += 1
                    if first:
                        for r in _table(tptr, _val(n)):
                            _arr(r['buf'], r['n'])[:] = 0
                    return 0
                scale = min(1.0, max_norm / (float(np.sqrt(ss)) + 1e-6))
            for r in _table(tptr, _val(n)):
                p, g, buf = _arr(r['p'], r['n']), _arr(r['g'], r['n']), _arr(r['buf'], r['n'])
                d = g * np.float32(scale) + np.float32(wd) * p
                buf[:] = d if first else np.float32(momentum) * buf + d
                p -= np.float32(r['lr']) * buf
            return 0

    def ptr(t):
        return ctypes.c_void_p(0) if t is None else ctypes.c_void_p(t.data_ptr())

    def stream():
        return None

    for name, fn in list(locals().items()):
        if callable(fn) and hasattr(_hip, name) and name not in ('install',):
            setattr(_hip, name, fn)
    _hip.f32 = _hip.i32 = ptr
    _hip.lib = lambda: ShimLib
    def _stay(self, x, mirror=False):
        if mirror:
            from lib.pytorch_misc import set_host
            set_host(x, x.numpy())
        return x
    blob_mod.Blob._to_device = _stay
    return _hip
