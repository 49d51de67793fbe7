"""RCCL self-test on ONE GPU (the only multi-GPU-path check possible on the 1-GPU test box; the 2-rank logic is covered
by the gloo tests, the 8-GPU curve is the driver's): backend "nccl" (= RCCL on ROCm) at world_size 1 with the
OverlappedGradReducer forced on -- hooks launch bucket all-reduces during backward from two HIP streams, .grad is
re-pointed at the reduced buffers, the fused optimizer consumes them."""
import socket

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_rccl_world1_overlapped_reducer_and_fused_optimizer():
    if not torch.cuda.is_available():
        pytest.fail('needs a HIP device')
    import torch.distributed as dist
    from lib import dist as D
    from lib.optim import FusedClipSGD
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0, world_size=1)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(64, 96), torch.nn.Tanh(), torch.nn.Linear(96, 10)).cuda()
        ref = torch.nn.Sequential(torch.nn.Linear(64, 96), torch.nn.Tanh(), torch.nn.Linear(96, 10)).cuda()
        ref.load_state_dict(net.state_dict())
        x = torch.randn(32, 64, device='cuda')
        y = torch.randint(0, 10, (32,), device='cuda')
        red = D.OverlappedGradReducer(list(net.parameters()), bucket_bytes=4096, force=True)
        assert red.enabled and len(red.buckets) >= 2
        opt = FusedClipSGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        opt_ref = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        side = torch.cuda.Stream()
        for step in range(3):
            w = D.global_row_weights([32], x.device)
            assert float(w[0]) == 1.0
            opt.zero_grad(set_to_none=True)
            red.prepare()
            h = net[1](net[0](x))
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                       # part of the graph runs (and back-propagates) on a 2nd stream
                h.record_stream(side)
                out = net[2](h)
            torch.cuda.current_stream().wait_stream(side)
            (F.cross_entropy(out, y) * w[0]).backward()
            red.finish()
            assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
            opt.step(max_norm=5.0)
            opt_ref.zero_grad()
            F.cross_entropy(ref(x), y).backward()
            torch.nn.utils.clip_grad_norm_(ref.parameters(), 5.0)
            opt_ref.step()
        torch.cuda.synchronize()
        for a, b in zip(net.parameters(), ref.parameters()):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), atol=2e-6)
    finally:
        dist.destroy_process_group()


def test_rccl_reducer_beside_the_two_compute_streams_of_the_real_backward():
    """A THIRD stream next to the two compute streams: the overlapped reducer (forced on at world_size 1: RCCL's all-reduce
    over one rank is the identity, but its kernels run on RCCL's own stream for the whole backward pass) while the real
    RelModel backward runs its union-box branch and its context branch on two HIP streams.  Two hazards only this design has
    (VERDICT r03): RCCL's reduction kernels co-resident with our MFMA kernels (the packed-FP32 fault of round 3 was a
    co-residency fault) and the persistent LSTM launches, whose grid barrier needs every workgroup resident, next to RCCL's
    channel blocks.  Logits and loss must be BITWISE those of the same step without the reducer, gradients equal to rounding (most of them bitwise)."""
    if not torch.cuda.is_available():
        pytest.fail('needs a HIP device')
    import torch.distributed as dist
    from dataloaders.synthetic import SyntheticVG, make_blob
    from lib import _hip
    from lib import dist as D
    from lib.rel_model import RelModel
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0, world_size=1)
    try:
        torch.manual_seed(5)
        ds = SyntheticVG(num_images=4, seed=23, n_boxes=12, n_rels=14)
        model = RelModel(classes=ds.ind_to_classes, rel_classes=ds.ind_to_predicates, mode='sgcls', num_gpus=1,
                         hidden_dim=512, pooling_dim=4096, nl_obj=2, nl_edge=2, order='leftright', rec_dropout=0.0,
                         use_bias=True, pass_in_obj_feats_to_decoder=False, pass_in_obj_feats_to_edge=False,
                         use_tanh=False, limit_vision=False)
        for _, p in model.detector.named_parameters():
            p.requires_grad = False
        model.cuda().train()
        for m in model.modules():
            if m.__class__.__name__ in ('Dropout', 'AlphaDropout'):
                m.eval()
        model.overlap_streams = True
        blob = make_blob(ds, [0, 1, 2, 3], is_train=True)
        params = [p for p in model.parameters() if p.requires_grad]
        red = D.OverlappedGradReducer(params, bucket_bytes=32 << 20, force=True)
        assert red.enabled and len(red.buckets) >= 8                     # ~1.1 GB of gradients in 32 MB buckets

        def step(with_reducer):
            model.zero_grad(set_to_none=True)
            model.sampler_rs = np.random.RandomState(9)
            res = model[blob]
            loss = F.cross_entropy(res.rm_obj_dists, res.rm_obj_labels) + F.cross_entropy(res.rel_dists, res.rel_labels[:, -1])
            if with_reducer:
                red.prepare()
            loss.backward()
            if with_reducer:
                red.finish()
            torch.cuda.synchronize()
            _hip.check_faults()                                          # a timed-out grid barrier of the persistent LSTM raises here
            return (res.rm_obj_dists.detach().clone(), res.rel_dists.detach().clone(), float(loss),
                    {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})

        plain = step(False)
        for trial in range(3):
            got = step(True)
            assert torch.equal(plain[0], got[0]) and torch.equal(plain[1], got[1]) and plain[2] == got[2]
            assert set(plain[3]) == set(got[3])
            exact = 0
            for name, g in plain[3].items():
                # rounding only: several gradients are accumulated with atomics (embedding / index_add / the persistent LSTM's
                # weight-gradient partial sums) and move in their last bits from run to run with or without the reducer; a
                # co-residency fault (round 3: wrong low halves in 16 of 64 lanes) is orders of magnitude above this bound
                err = float((g - got[3][name]).abs().max())
                assert err <= 1e-6 * float(g.abs().max()) + 1e-30, '%s: gradient differs with the reducer running: %.3e (max %.3e)' % (
                    name, err, float(g.abs().max()))
                exact += int(err == 0.0)
            assert exact >= len(plain[3]) // 2, 'only %d of %d gradients are bitwise equal' % (exact, len(plain[3]))
        assert red.stats['in_place_bytes'] > 0                           # fc6 / fc7 weight gradients were born inside the buckets
        # the N > 1-only producer path ran HERE: the relation head's fc6 gradient (411 MB) was written range by
        # range into its bucket (lib/hip_ops.py: _wgrad_planes) and every range reduced as its own collective, in unit order
        fc6 = model.roi_fmap[1][0].weight
        rows = red.segments(fc6)
        assert rows is not None and len(rows) >= 6 and all((r1 - r0) * fc6.shape[1] * 4 <= red.split_bytes for r0, r1 in rows)
        assert red.launch_log == list(range(len(red.units))) and len(red.units) > len(red.buckets)
        assert red.stats['in_place_bytes'] >= 3 * 2 * fc6.numel() * 4       # both trainable fc6 copies, three trials
        red.remove()
    finally:
        dist.destroy_process_group()
