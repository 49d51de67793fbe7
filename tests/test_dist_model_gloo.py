"""world_size-2 gloo run of the REAL data-parallel training step (what bench.py / models/train_rels.py do per rank):
RelModel (actual parameter set: 3 fc6/fc7 copies, flat LSTM weights, embeddings, BN) on the CPU shim, ragged shards,
OverlappedGradReducer armed (hooks launch bucket all-reduces during backward, .grad re-pointed at the reduced flat
buffers), row-weighted losses, FusedClipSGD driven through its chunk table of raw pointers.

Checked: after two steps both ranks hold IDENTICAL parameters, and they equal a single-process emulation that runs the
two shards through one model, sums the row-weighted losses' gradients and takes the same optimizer steps (the
reference's semantics: CE averaged over the gathered rows of all replicas, per-replica BatchNorm statistics,
models/train_rels.py:140-150, lib/rel_model.py:549-560)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(hidden_dim=32, pooling_dim=4096, nl_obj=1, nl_edge=1, order='leftright', rec_dropout=0.0, use_bias=True,
          pass_in_obj_feats_to_decoder=False, pass_in_obj_feats_to_edge=False, use_tanh=False, limit_vision=False)
SHARDS = ([0, 1], [2])            # ragged: 2 images on rank 0, 1 image on rank 1
STEPS = 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup():
    for p in (ROOT, os.path.join(ROOT, 'neural-motifs_amd'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import cpu_shim
    cpu_shim.install()
    from dataloaders.synthetic import SyntheticVG
    from lib.rel_model import RelModel
    torch.manual_seed(0)
    ds = SyntheticVG(num_images=3, seed=4, n_boxes=[4, 6, 3], n_rels=4, im_size=96)
    model = RelModel(classes=ds.ind_to_classes, rel_classes=ds.ind_to_predicates, mode='sgcls', num_gpus=1, **KW)
    for _, p in model.detector.named_parameters():
        p.requires_grad = False
    for m in model.modules():                     # the VGG classifier's Dropout(0.5): off, so the ranks need no shared mask stream
        if m.__class__.__name__ == 'Dropout':
            m.p = 0.0
    model.train()
    return ds, model


def _optimizer(model, world):
    from lib.optim import FusedClipSGD
    lr = 1e-3 * world * 2
    fc = [p for n, p in model.named_parameters() if n.startswith('roi_fmap') and p.requires_grad]
    rest = [p for n, p in model.named_parameters() if not n.startswith('roi_fmap') and p.requires_grad]
    return FusedClipSGD([{'params': fc, 'lr': lr / 10.0}, {'params': rest}], lr=lr, momentum=0.9, weight_decay=1e-4)


def _losses(model, ds, idx, step):
    from dataloaders.synthetic import make_blob
    model.sampler_rs = np.random.RandomState(100 * step + idx[0])
    res = model[make_blob(ds, idx, is_train=True)]
    return (F.cross_entropy(res.rm_obj_dists, res.rm_obj_labels), F.cross_entropy(res.rel_dists, res.rel_labels[:, -1]),
            res.rm_obj_labels.shape[0], res.rel_labels.shape[0])


def _pids(world):
    return []          # (the ranks' pids are not known to each other: the identity strings are only checked for distinctness)


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    ds, model = _setup()
    import torch.distributed as dist
    from lib import dist as D
    D.init_from_env(backend='gloo')
    opt = _optimizer(model, world)
    # split_bytes 16 MB: the three fc6 weights (411 MB each) are reduced in 26 row ranges each, produced range by range by the
    # weight-gradient GEMM (lib/hip_ops.py: _wgrad_planes); fc7 (67 MB) in 4
    red = D.OverlappedGradReducer([p for p in model.parameters() if p.requires_grad], bucket_bytes=8 << 20, split_bytes=16 << 20)
    assert red.enabled and len(red.buckets) >= 3
    big = [p for p in red.params if p.numel() * 4 > red.split_bytes]
    assert len(big) >= 4 and all(red.segments(p) is not None and len(red.segments(p)) >= 4 for p in big)
    assert len(red.units) > len(red.buckets) + 40 and max(red.bucket_mb) <= 16.0 + 1e-6
    ptrs = None
    roww = D.RowWeights('cpu')
    model.rows_hook = roww.start              # the row-count all-reduce is launched inside the forward pass, asynchronously
    for step in range(STEPS):
        l_obj, l_rel, n_obj, n_rel = _losses(model, ds, SHARDS[rank], step)
        w = roww.get()
        assert torch.equal(w, D.global_row_weights([n_obj, n_rel], 'cpu'))
        opt.zero_grad(set_to_none=True)
        red.prepare()
        (l_obj * w[0] + l_rel * w[1]).backward()
        red.finish()
        assert red.launch_log == list(range(len(red.units))), 'collectives were not issued in unit order'
        now = [p.grad.data_ptr() for p in red.params]
        assert ptrs is None or ptrs == now, 'reduced gradients must keep their addresses (fused optimizer pointer table)'
        ptrs = now
        opt.step(max_norm=5.0)
    # the big weight gradients (fc6 / fc7 / post_lstm ...) were written straight into their buckets by the GEMMs
    assert red.stats['in_place_bytes'] > 0.8 * (red.stats['in_place_bytes'] + red.stats['copied_bytes']), red.stats
    # what bench.py adds to rank 0's line at N > 1 (lib/dist.py: scaling_diagnostics; every rank takes part)
    diag = D.scaling_diagnostics(red, 'cpu', 12.5 + rank)
    assert diag['ranks_seen'] == ['cpu:pid%d' % pid for pid in _pids(world)] or len(diag['ranks_seen']) == world
    assert diag['distinct_devices'] == world and diag['collective_order_identical'] is True
    assert diag['ms_per_step_per_rank'] == [12.5, 13.5] and len(diag['allreduce_exposed_ms']) == world
    assert all(v >= 0.0 for v in diag['allreduce_exposed_ms']) and diag['bucket_mb'] == red.bucket_mb
    assert diag['grad_bytes_in_place_frac'] > 0.8
    torch.save({n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad},
               os.path.join(out_dir, 'params%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_real_model_training_steps(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    p0 = torch.load(os.path.join(str(tmp_path), 'params0.pt'))
    p1 = torch.load(os.path.join(str(tmp_path), 'params1.pt'))
    assert len(p0) >= 30
    for n in p0:
        assert torch.equal(p0[n], p1[n]), n                       # replicas stay bit-identical
    # single-process emulation of the same two-replica step (same intra-op thread count as the workers: the summation
    # order inside torch's CPU kernels depends on it)
    threads = torch.get_num_threads()
    torch.set_num_threads(2)
    try:
        _compare_with_emulation(p0)
    finally:
        torch.set_num_threads(threads)


def _compare_with_emulation(p0):
    ds, model = _setup()
    opt = _optimizer(model, 2)
    init = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    for step in range(STEPS):
        parts = [_losses(model, ds, idx, step) for idx in SHARDS]
        n_obj, n_rel = sum(p[2] for p in parts), sum(p[3] for p in parts)
        loss = sum(p[0] * (p[2] / n_obj) + p[1] * (p[3] / n_rel) for p in parts)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step(max_norm=5.0)
    moved = 0
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        scale = max(1e-6, float((p.detach() - init[n]).abs().max()))
        np.testing.assert_allclose(p0[n].numpy(), p.detach().numpy(), rtol=0,
                                   atol=2e-4 * scale + 4e-7 * max(1.0, float(p.detach().abs().max())), err_msg=n)   # + fp32 ulps of p
        moved += int(scale > 1e-6)
    assert moved >= 25                                             # the steps really changed (almost) every parameter
