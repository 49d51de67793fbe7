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
