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
    for rank in range(2):
        got = torch.load(os.path.join(str(tmp_path), 'g%d.pt' % rank))
        for a, b in zip(got, ref):
            torch.testing.assert_close(a, b, atol=1e-6, rtol=1e-5)
