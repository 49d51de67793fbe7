"""world_size-2 gloo test of the data-parallel gradient path (lib/dist.py): bucketed all-reduce + row weighting
reproduce the single-process gradient of the global-mean loss."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, 'neural-motifs_amd'))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from lib import dist as D
    r, w, _ = D.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    net[0].bias.requires_grad = False                       # frozen parameters stay out of the buckets
    g = torch.Generator().manual_seed(1)
    x = torch.randn(7, 6, generator=g)
    y = torch.randint(0, 3, (7,), generator=g)
    lo, hi = (0, 2) if rank == 0 else (2, 7)                # ragged shards: 2 rows vs 5 rows
    loss = torch.nn.functional.cross_entropy(net(x[lo:hi]), y[lo:hi])
    wgt = D.global_row_weights([hi - lo], 'cpu')[0]
    (loss * wgt).backward()
    buckets = D.GradBuckets(net.parameters(), bucket_bytes=64)      # tiny buckets -> several of them
    assert len(buckets.buckets) >= 2 and all(p.requires_grad for b in buckets.buckets for p in b)
    buckets.all_reduce()
    torch.save([p.grad for p in net.parameters() if p.requires_grad], os.path.join(out_dir, 'g%d.pt' % rank))
    # the overlapped reducer (hooks launch each bucket's all-reduce during backward) must give the same sums, also
    # over two steps and with a parameter that gets no gradient on one rank
    extra = torch.nn.Linear(3, 2)                           # used by rank 1 only
    params = list(net.parameters()) + list(extra.parameters())
    red = D.OverlappedGradReducer(params, bucket_bytes=64)
    assert len(red.buckets) >= 3 and red.enabled
    for step in range(3):
        red.overlap = step != 1                             # step 1 exercises the MOTIFS_GRAD_SYNC=post path
        for p in params:
            p.grad = None
        out = net(x[lo:hi])
        loss = torch.nn.functional.cross_entropy(out, y[lo:hi]) * wgt
        if rank == 1:
            loss = loss + extra(out).sum() * 0.01
        red.prepare()
        loss.backward()
        red.finish()
    overlapped = [p.grad.clone() for p in params if p.requires_grad]
    torch.save(overlapped, os.path.join(out_dir, 'o%d.pt' % rank))
    # same loss through the plain post-backward buckets (hooks are disarmed outside prepare()/finish())
    for p in params:
        p.grad = None
    out = net(x[lo:hi])
    loss = torch.nn.functional.cross_entropy(out, y[lo:hi]) * wgt
    if rank == 1:
        loss = loss + extra(out).sum() * 0.01
    loss.backward()
    D.GradBuckets(params, bucket_bytes=64).all_reduce()
    for a, p in zip(overlapped, [p for p in params if p.requires_grad]):
        torch.testing.assert_close(a, p.grad, rtol=0, atol=1e-7)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradients_equal_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    net[0].bias.requires_grad = False
    g = torch.Generator().manual_seed(1)
    x = torch.randn(7, 6, generator=g)
    y = torch.randint(0, 3, (7,), generator=g)
    torch.nn.functional.cross_entropy(net(x), y).backward()
    ref = [p.grad for p in net.parameters() if p.requires_grad]
    # reference for the overlapped run: sum over ranks of (weighted local loss [+ rank 1's extra term]) gradients
    extra = torch.nn.Linear(3, 2)
    torch.manual_seed(0)
    net2 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    net2[0].bias.requires_grad = False
    o0 = torch.load(os.path.join(str(tmp_path), 'o0.pt'))
    o1 = torch.load(os.path.join(str(tmp_path), 'o1.pt'))
    for a, b in zip(o0, o1):
        torch.testing.assert_close(a, b, rtol=0, atol=0)              # both ranks hold the same reduced gradients
    for a, b in zip(o0[:len(ref)], ref):                               # net's part: global-mean CE gradient + extra term
        assert a.shape == b.shape
    for rank in range(2):
        got = torch.load(os.path.join(str(tmp_path), 'g%d.pt' % rank))
        for a, b in zip(got, ref):
            torch.testing.assert_close(a, b, atol=1e-6, rtol=1e-5)
