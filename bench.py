#!/usr/bin/env python3
"""
bench.py -- headline metric of BASELINE.json: images/sec of a MotifNet-SGCls training step (forward + backward +
grad-clip + SGD step) on synthetic 592x592 VG-shaped batches, `configs[1]`:
    SGCls MotifNet (order=leftright, 2-layer highway LSTMs, hidden 512) VGG16, batch 6 per GPU, 20 GT boxes/img.

    python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU over RCCL; every rank trains on its own images (weak scaling), the only collective is the gradient
all-reduce; rank 0 prints ONE JSON line.  Started either by torchrun (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`: RANK / WORLD_SIZE in the environment) or plainly as `python bench.py --gpus N`, in which case
bench.py starts the N ranks itself (`launch_ranks`: re-executes itself under torch.distributed.run on 127.0.0.1) -- as easy
to start as the reference's single-process DataParallel (lib/rel_model.py:549-560).  Fewer than N visible devices, or a
process group of another size, is an error, never a silent smaller run.

Step-time distribution: every timed step is bracketed by HIP events on the main stream and by host timestamps;
`ms_per_step` stays the wall-clock mean the driver expects, `step_ms` carries p50 / p90 / max / the per-step list.
`h2d_inclusive` re-times a few steps with a batch uploaded from page-locked host memory in EVERY step (what
models/train_rels.py:137 + dataloaders/blob.py:155-180 do per step): the next batch's copies run on a copy stream while the
step computes (Blob.prefetch, the shipped training loop), `h2d_inclusive.inline` = the copies on the compute stream in front
of the step (the reference's placement); `value` itself is quoted with inputs resident in HBM.

Extra objects in the line:
  roofline     -- the dominant kernel class, the 3x3 convolutions as implicit GEMMs on the matrix cores (12 trunk layers on
                  the plane engine, csrc/pl_conv.hip, + the union tower's forward / input-gradient conv): algorithmic fp32
                  FLOPs / HIP-event time of the calls during the timed steps, against the matrix-core peak of the f16x3
                  evaluation (2500/3 = 833 TFLOP/s fp32-equivalent; `frac_of_f32_mfma_peak` is given too).  `traffic` =
                  fabric-side bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over a replay
                  of the trunk's launches (profiles/r03_conv_traffic_summary.json; a PMC pass over the whole step does not
                  finish); `traffic_algorithmic` = input + weights + output bytes of those launches.
  roofline_gemm-- the same for every matrix product of the step: mh_gemm_planes on ready plane images (fc6/fc7 of the RoI
                  heads fwd / dgrad / wgrad), the generic mh_gemm_f32 calls (operand preparation inside the call) and the
                  one skinny in-loop-split product.
  hbm_kernels  -- the HBM-bound kernels of the step, each as algorithmic bytes / HIP-event time against the 8 TB/s peak:
                  RoIAlign forward, the activation converter (fp32 -> plane image, with the fused 2x2 pool), the operand
                  preparation of the GEMMs, the fused clip + SGD step; and the latency-bound ones in microseconds per
                  call: the persistent highway-LSTM layer launches.
  cpu_baseline -- the CPU oracle (oracle/model.py, "port") timed on this host: the same cfg2 step at b = 6 (1 warm-up +
                  3 timed iterations) and the cfg1 PredCls evaluation (1 image per step), rank 0 at N=1 only.
  --config cfgN-- secondary rows for the other BASELINE.json configurations (see `secondary`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'neural-motifs_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_FP32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md chip table (f32-input MFMA)
PEAK_BF16_MFMA_TFLOPS = 2500.0         # same table, dense bf16 MFMA
GEMM_TRAFFIC_SUMMARY = 'profiles/r06_gemm_traffic_summary.json'   # tools/r05/gemm_traffic_summary.py (the step's big products on plane images, XCD-banded order)
TRAFFIC_SUMMARY = 'profiles/r06_conv_traffic_summary.json'   # tools/r04/traffic_summary.py (ring trunk with the step's image / pooled-image epilogues, round 6)
PEAK_HBM_TBS = 8.0                     # same table, HBM3E
# every product is an fp32 product (operands, accumulation and results fp32, 1e-4 parity against the fp32 oracle); the matrix cores
# evaluate it from 16-bit terms: big products as f16x3 (two f16 terms of row-scaled operands, 3 MFMAs), small ones as bf16x6
# (three bf16 terms, 6 MFMAs) -- DESIGN.md section 3.1
DTYPE = 'f32 (f16x3 / bf16x6 on the 16-bit matrix cores, fp32 accumulate)'
MODEL_KW = dict(hidden_dim=512, pooling_dim=4096, nl_obj=2, nl_edge=2, order='leftright', rec_dropout=0.1,
                use_bias=True, pass_in_obj_feats_to_decoder=False, pass_in_obj_feats_to_edge=False, use_tanh=False,
                limit_vision=False)
BATCH, N_BOXES, N_RELS = 6, 20, 30


class KernelMeter(object):
    """HIP-event timing of every call of one binding function (events recorded on the stream the kernel is launched
    on) plus its algorithmic work.  `work_of(args, kwargs, result)` -> flops, or (flops, bytes)."""

    sampling = True          # class-wide: False on the steps that are not sampled (METER_EVERY)

    def __init__(self, hip, name, work_of, obj=None, tag_of=None):
        self.hip, self.name, self.work_of, self.tag_of = hip if obj is None else obj, name, work_of, tag_of
        self.orig = getattr(self.hip, name)
        self.records, self.enabled = [], False
        setattr(self.hip, name, self)

    def __call__(self, *args, **kwargs):
        if not (self.enabled and KernelMeter.sampling):
            return self.orig(*args, **kwargs)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        y = self.orig(*args, **kwargs)
        e.record()
        w = self.work_of(args, kwargs, y)
        self.records.append((s, e) + (tuple(w) if isinstance(w, tuple) else (w, 0.0)) + (self.tag_of(args, kwargs) if self.tag_of else None,))
        return y

    def summary(self, tag=None):
        """totals of the recorded calls; tag: only the calls `tag_of` labelled so (the plane conv serves the trunk AND, since round
        5, the 3x3 convs over many small RoI maps: two rows of the report)"""
        torch.cuda.synchronize()
        records = self.records if tag is None else [r for r in self.records if r[4] == tag]
        ms = sum(r[0].elapsed_time(r[1]) for r in records)
        flops = sum(r[2] for r in records)
        nbytes = sum(r[3] for r in records)
        n = max(len(records), 1)
        return dict(launches=len(records), avg_ms=ms / n, total_ms=ms, flops_per_launch=flops / n, flops=flops,
                    bytes=nbytes, tflops=(flops / (ms * 1e-3) / 1e12) if ms > 0 else 0.0,
                    gbps=(nbytes / (ms * 1e-3) / 1e9) if ms > 0 else 0.0)


def merge(*sums):
    """one summary over several meters (same kernel class)"""
    out = dict(launches=0, total_ms=0.0, flops=0.0, bytes=0.0)
    for m in sums:
        for k in out:
            out[k] += m[k]
    n = max(out['launches'], 1)
    out['avg_ms'], out['flops_per_launch'] = out['total_ms'] / n, out['flops'] / n
    out['tflops'] = out['flops'] / (out['total_ms'] * 1e-3) / 1e12 if out['total_ms'] > 0 else 0.0
    out['gbps'] = out['bytes'] / (out['total_ms'] * 1e-3) / 1e9 if out['total_ms'] > 0 else 0.0
    return out


def _conv_flops(args, kwargs, y):
    x, wt = args[0], args[1]
    B, H, W, Cin = x.shape
    return 2.0 * B * H * W * Cin * wt.shape[1] * 9


def _plconv_flops(args, kwargs, y):
    img, cout = args[0], args[2]
    return 2.0 * img.B * img.H * img.W * img.C * cout * 9


def _gemm_flops(args, kwargs, y):
    a = args[0]
    return 2.0 * y.shape[0] * y.shape[1] * (a.shape[0] if (args[2] if len(args) > 2 else kwargs.get('trans_a', False)) else a.shape[1])


def _gemm_planes_flops(args, kwargs, y):
    return 2.0 * args[0].rows * args[1].rows * args[0].K


def _roi_bytes(args, kwargs, y):
    feat = args[0]                              # algorithmic: the output written once + the feature map read once
    return 0.0, 4.0 * (y.numel() + feat.numel())


def _act_planes_bytes(args, kwargs, y):
    x = args[0]                                 # fp32 in (read once) + plane image out (4 B per output element)
    return 0.0, 4.0 * x.numel() + 4.0 * y.B * y.H * y.W * y.C


def _make_planes_bytes(args, kwargs, y):
    x = args[0]                                 # read once + one 4 B/element image per orientation written
    return 0.0, 4.0 * x.numel() * (1 + (2 if isinstance(y, tuple) else 1))


def install_meters(_hip):
    from lib.optim import FusedClipSGD
    m = dict(
        plconv=KernelMeter(_hip, 'plconv3x3', _plconv_flops, tag_of=lambda a, k: 'maps' if a[0].B >= 64 else 'trunk'),
        conv=KernelMeter(_hip, 'conv3x3_nhwc', _conv_flops),
        plconv_img=KernelMeter(_hip, 'plconv3x3_to_image', lambda a, k, y: 2.0 * a[0].B * a[0].H * a[0].W * a[0].C * a[3] * 9),
        # round 6: the layers in front of a pool write the POOLED image from their epilogue (same algorithmic flops: the full-resolution conv)
        plconv_pool=KernelMeter(_hip, 'plconv3x3_pool_to_image', lambda a, k, y: 2.0 * a[0].B * a[0].H * a[0].W * a[0].C * a[3] * 9),
        stem=KernelMeter(_hip, 'stem_to_image', lambda a, k, y: (0.0, 4.0 * a[0].numel() + float(y.buf.numel()))),
        gemm_planes=KernelMeter(_hip, 'gemm_planes', _gemm_planes_flops), gemm=KernelMeter(_hip, 'gemm', _gemm_flops),
        gemm_inloop=KernelMeter(_hip, 'gemm_inloop', _gemm_flops),
        roi=KernelMeter(_hip, 'roi_align_fwd', _roi_bytes), act=KernelMeter(_hip, 'act_planes', _act_planes_bytes),
        planes=KernelMeter(_hip, 'make_planes', _make_planes_bytes), planes_both=KernelMeter(_hip, 'make_planes_both', _make_planes_bytes),
        lstm_fwd=KernelMeter(_hip, 'hwlstm_fwd', lambda a, k, y: 0.0), lstm_bwd=KernelMeter(_hip, 'hwlstm_bwd', lambda a, k, y: 0.0),
    )
    return m


def set_meters(meters, on):
    for v in meters.values():
        v.enabled = on


def sample_step(i, every):
    """meters record on every `every`-th timed step (i = 0, every, 2 every, ...).  An event pair around a call is not free on
    the GPU: each timing event is a barrier packet, so consecutive kernels no longer overlap their ramp-down / ramp-up --
    ~100 metered calls per step cost 0.8 ms of an 18.8 ms step (gpurun r04_c1: 19.56 ms p50 fully metered vs 18.75 unmetered).
    Sampling keeps the rooflines measured INSIDE the timed region at a quarter of that price; `meter_every` is in the line."""
    KernelMeter.sampling = (i % max(every, 1) == 0)
    return KernelMeter.sampling


def hbm_rows(meters, steps, opt_ms=None, opt_bytes=None):
    """the HBM-bound / latency-bound kernels of the step as roofline rows"""
    rows = {}
    def row(name, summ, what):
        if summ['launches']:
            rows[name] = {'bound': 'hbm', 'what': what, 'achieved': summ['gbps'], 'peak': PEAK_HBM_TBS * 1e3, 'unit': 'GB/s',
                          'frac': summ['gbps'] / (PEAK_HBM_TBS * 1e3), 'launches_per_step': summ['launches'] / steps,
                          'ms_per_step': summ['total_ms'] / steps, 'bytes_per_step': summ['bytes'] / steps}
    row('roi_align_fwd', meters['roi'].summary(), 'RoIAlign 7x7 forward (objects + union boxes): output bytes + feature map once')
    row('stem_to_image', meters['stem'].summary(), 'conv1_1 (3 -> 64 channels, on the matrix cores since round 6: im2col rows built in LDS, K = 27 -> 32) writing its output as a plane image: NCHW input + image bytes')
    row('act_planes', meters['act'].summary(), 'fp32 NHWC -> plane image (2x2 pool fused where the trunk has one): bytes in + bytes out')
    row('make_planes', merge(meters['planes'].summary(), meters['planes_both'].summary()),
        'GEMM operand preparation (row maxima + split, both orientations from one read where both are needed): read once + images written')
    if opt_ms is not None:
        rows['fused_clip_sgd'] = {'bound': 'hbm', 'what': 'global-norm clip + SGD(momentum, wd) over all trainable parameters: 20 B/param',
                                  'achieved': opt_bytes / (opt_ms * 1e-3) / 1e9, 'peak': PEAK_HBM_TBS * 1e3, 'unit': 'GB/s',
                                  'frac': opt_bytes / (opt_ms * 1e-3) / 1e9 / (PEAK_HBM_TBS * 1e3), 'ms_per_step': opt_ms}
    for k, what in (('lstm_fwd', 'persistent highway-LSTM forward (all layers of one context LSTM; state exchanged as tagged 8-byte granules, one launch per layer)'),
                    ('lstm_bwd', 'persistent highway-LSTM backward')):
        sm = meters[k].summary()
        if sm['launches']:
            rows[k] = {'bound': 'latency', 'what': what, 'us_per_call': 1e3 * sm['avg_ms'], 'calls_per_step': sm['launches'] / steps,
                       'ms_per_step': sm['total_ms'] / steps}
    return rows


def calibration():
    """a fixed product on ready plane images (4096^3, 10 launches, HIP events): the matrix-pipe rate THIS box sustains.  The
    sustained clock under MFMA load differs between boxes of the pool (power management: MI355X_MICROARCH.md), so step
    times of different gpurun calls are compared through this number"""
    from lib import _hip
    a = torch.randn(4096, 4096, device='cuda')
    b = torch.randn(4096, 4096, device='cuda')
    ia, ib = _hip.make_planes(a, True), _hip.make_planes(b, True)
    out = torch.empty(4096, 4096, device='cuda')
    for _ in range(3):
        _hip.gemm_planes(ia, ib, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _hip.gemm_planes(ia, ib, out=out)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    return {'plane_gemm_4096_tflops': 2.0 * 4096 ** 3 / ms * 1e-9, 'ms': ms,
            'what': 'pl::gemm_kernel on ready images, 4096^3, mean of 10 back-to-back launches after the timed region'}


def cpu_baseline(ds, model_sd, iters=3, eval_images=3, budget_s=150.0):
    """The CPU oracle (oracle/model.py, kind "port") on this host's cores, as SURVEY.md section 8d prescribes: the SAME
    cfg2 step (b = 6: forward + backward of the trainable part) with 1 warm-up + up to `iters` timed iterations, and the
    cfg1 PredCls evaluation (one image per step, `eval_images` images after one warm-up).  Bounded: timed iterations stop
    once `budget_s` seconds of CPU baseline have been spent (at least one is always timed; the count is reported).
    Threads: torch's default for this host (one per physical core; oversubscribing the hyperthreads is slower).
    Baseline only -- not a target."""
    from oracle import model as OM
    from dataloaders.synthetic import make_blob
    from lib.fpn.proposal_assignments.proposal_assignments_gtbox import proposal_assignments_gtbox
    t_start = time.time()
    blob = make_blob(ds, range(BATCH), is_train=True)
    a = blob[0]
    rois = torch.cat((a[4][:, :1].float(), a[3]), 1)
    _, _, rel_labels = proposal_assignments_gtbox(rois, a[3], a[4], a[5], 0, rs=np.random.RandomState(0))
    trainable = {k for k in model_sd if not k.startswith('detector.') and model_sd[k].is_floating_point()
                 and 'running_' not in k and 'num_batches' not in k}
    cfg = dict(MODEL_KW, mode='sgcls')

    def train_step():
        params = {k: v.clone().requires_grad_(k in trainable) for k, v in model_sd.items()}
        t0 = time.time()
        out = OM.relmodel_forward(params, cfg, a[0], a[1], 0, a[3], a[4], True, OM.HostRNG(0), rel_labels=rel_labels)
        loss = F.cross_entropy(out['rm_obj_dists'], out['rm_obj_labels']) + \
            F.cross_entropy(out['rel_dists'], out['rel_labels'][:, -1])
        loss.backward()
        return time.time() - t0

    # cfg1 first (cheap): PredCls eval, one image per step
    ecfg = dict(MODEL_KW, mode='predcls')
    etimes = []
    with torch.no_grad():
        for i in range(eval_images + 1):
            eb = make_blob(ds, [i], is_train=False)[0]
            t0 = time.time()
            OM.relmodel_forward({k: v for k, v in model_sd.items()}, ecfg, eb[0], eb[1], 0, eb[3], eb[4], False, OM.HostRNG(0))
            etimes.append(time.time() - t0)
    edt = sum(etimes[1:]) / max(len(etimes) - 1, 1)
    warm = train_step()
    times = [train_step()]
    while len(times) < iters and time.time() - t_start + times[-1] < budget_s:
        times.append(train_step())
    dt = sum(times) / len(times)
    return dict(value=BATCH / dt, unit='img/s', cores=os.cpu_count(), threads=torch.get_num_threads(), kind='port',
                sample='cfg2 SGCls step at b=%d (fwd+bwd, %d relation rows): 1 warm-up (%.1f s) + %d timed iteration(s) '
                       '(%s s); %d torch threads on %d logical cores' % (
                           BATCH, rel_labels.shape[0], warm, len(times), ', '.join('%.1f' % t for t in times),
                           torch.get_num_threads(), os.cpu_count() or 0),
                cfg1_predcls_eval={'value': 1.0 / edt, 'unit': 'img/s',
                                   'sample': '1 warm-up + %d images, one per step (%s s)' % (
                                       eval_images, ', '.join('%.2f' % t for t in etimes[1:]))})


def rank_command(args, argv):
    """(command line, environment) of the N ranks `python bench.py --gpus N` starts under torch.distributed.run: one process per
    GPU, standalone c10d rendezvous on 127.0.0.1."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL's cross-process buffer sharing needs it on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1))))
    # --standalone: the c10d rendezvous binds a free port ITSELF (no bind-close-rebind window in which another process could
    # take a port picked here); --local-addr: the container's hostname may not resolve
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), os.path.abspath(__file__)] + list(argv)
    return cmd, env


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without torchrun's environment: start N ranks of this file under torch.distributed.run
    (one process per GPU, standalone c10d rendezvous on 127.0.0.1) and pass rank 0's JSON line through.  Returns the exit code."""
    import subprocess
    if not args.launch_selftest:
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < args.gpus:
            raise SystemExit('bench.py --gpus %d: only %d HIP device(s) visible -- refusing to run a smaller job under that name'
                             % (args.gpus, n_dev))
    cmd, env = rank_command(args, argv)
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------------- --dry
# `python bench.py --gpus 8 --dry` -- everything of the first N-GPU run that does NOT need N GPUs, on any box:
#   * the N rank command lines and their environment, exactly as launch_ranks() would start them (rank_command);
#   * the REAL trainable-parameter list of the benched model (built once by the parent, shapes only) handed to N gloo ranks on the
#     CPU: every rank runs lib.dist.OverlappedGradReducer's own planning code on it (buckets, <= 64 MB collective units, the row
#     ranges of fc6) and the plans are compared ACROSS the ranks through the process group;
#   * the step's collectives at their real sizes, in the reducer's launch order, asynchronously like the reducer issues them,
#     on a reused <= 64 MB scratch buffer (a different order or size on any rank deadlocks or fails HERE, not on the 8-GPU node),
#     with the sums checked; lib.dist.scaling_diagnostics (its all_gather_object) over the same group;
#   * the learning-rate rule (models/train_rels.py:192: 1e-3 x world x batch) and the per-rank host-thread budget: N ranks x
#     (main thread + the detect-ahead worker of the SGDet configurations) against the node's cores, OMP threads per rank.
# One JSON line from rank 0; exit code != 0 when a check fails.  No scaling number comes out of this -- SCALE_rNN.json does that.
_INIT_FNS = ('uniform_', 'normal_', 'trunc_normal_', 'constant_', 'ones_', 'zeros_', 'eye_', 'dirac_', 'xavier_uniform_',
             'xavier_normal_', 'kaiming_uniform_', 'kaiming_normal_', 'orthogonal_', 'sparse_')


def trainable_parameter_list(config):
    """[(name, shape, requires_grad)] of the benched model: the constructor runs with torch.nn.init's functions stubbed (the
    block-orthogonal LSTM initialisation and three 411 MB fc6 draws are most of its 50 s; shapes do not depend on values)"""
    import torch.nn.init as I
    from dataloaders.synthetic import SyntheticVG
    from lib.rel_model import RelModel
    saved = {n: getattr(I, n) for n in _INIT_FNS if hasattr(I, n)}
    for n in saved:
        setattr(I, n, lambda t, *a, **k: t)
    try:
        ds = SyntheticVG(num_images=2, seed=1, n_boxes=N_BOXES, n_rels=N_RELS)
        kw = dict(MODEL_KW, nl_edge=4) if config == 'recipe' else MODEL_KW
        mode = 'sgdet' if config in ('cfg3', 'cfg5') else 'sgcls'
        model = RelModel(classes=ds.ind_to_classes, rel_classes=ds.ind_to_predicates, mode=mode, num_gpus=1, **kw)
    finally:
        for n, f in saved.items():
            setattr(I, n, f)
    for _, p in model.detector.named_parameters():           # models/train_rels.py:50-52
        p.requires_grad = False
    return [(n, list(p.shape), bool(p.requires_grad)) for n, p in model.named_parameters()]


def dry_parent(args, argv):
    import subprocess
    import tempfile
    real_cmd, env = rank_command(args, [a for a in argv if a != '--dry'])
    plist = trainable_parameter_list(args.config)
    with tempfile.NamedTemporaryFile('w', suffix='.json', delete=False) as f:
        json.dump({'params': plist, 'real_cmd': real_cmd,
                   'env': {k: env[k] for k in ('HSA_ENABLE_IPC_MODE_LEGACY', 'OMP_NUM_THREADS')}}, f)
        path = f.name
    try:
        cmd, env = rank_command(args, argv)
        env['MOTIFS_DRY_PLAN'] = path
        env['HIP_VISIBLE_DEVICES'] = ''                       # the rehearsal is a CPU job wherever it runs
        return subprocess.call(cmd, env=env)
    finally:
        os.unlink(path)


def dry_rank(args):
    import hashlib
    import torch.distributed as dist
    from lib import dist as D
    rank, world, local_rank = D.init_from_env(backend='gloo')
    if world != args.gpus or (world > 1 and dist.get_world_size() != args.gpus):
        raise SystemExit('--dry: asked for %d ranks, the process group has %d' % (args.gpus, world))
    plan = json.load(open(os.environ['MOTIFS_DRY_PLAN']))
    names, params = [], []
    for name, shape, rg in plan['params']:
        if rg:
            names.append(name)
            params.append(torch.nn.Parameter(torch.empty(shape, device='meta')))
    reducer = D.OverlappedGradReducer(params, force=True)      # the product's planning code on the real shapes
    name_of = {id(p): n for n, p in zip(names, params)}
    layout = [[name_of[id(p)] for p in b] for b in reducer.buckets]
    digest = hashlib.sha256(json.dumps([layout, reducer.units]).encode()).hexdigest()
    seen = [None] * world
    if world > 1:
        dist.all_gather_object(seen, digest)
    else:
        seen = [digest]
    plan_identical = all(d == seen[0] for d in seen)
    # the collectives of one backward at their real sizes, in launch order, in flight together like the reducer's
    biggest = max(hi - lo for _, lo, hi in reducer.units)
    scratch = [torch.empty(biggest) for _ in range(2)]
    t0 = time.time()
    ok = True
    work = []
    for u, (_, lo, hi) in enumerate(reducer.units):
        buf = scratch[u % 2][:hi - lo]
        if len(work) >= 2:                                      # two scratch buffers: wait for the unit that used this one
            w, b, n = work.pop(0)
            w.wait()
            ok = ok and bool((b[:8] == n).all()) and bool((b[-8:] == n).all())
        buf.fill_(float(rank + 1))
        expect = float(world * (world + 1) // 2)
        if world > 1:
            work.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True), buf, expect))
        else:
            expect = 1.0
        reducer.launch_log.append(u)
    for w, b, n in work:
        w.wait()
        ok = ok and bool((b[:8] == n).all()) and bool((b[-8:] == n).all())
    t_coll = time.time() - t0
    diag = D.scaling_diagnostics(reducer, 'cpu', 0.0)
    cores = os.cpu_count() or 1
    sgdet = args.config in ('cfg3', 'cfg5')
    per_rank = 1 + (1 if sgdet else 0)                          # main thread + RelModel.detect_ahead's worker (SGDet only)
    omp = int(os.environ.get('OMP_NUM_THREADS', '1'))
    threads = {'cores': cores, 'ranks': world, 'python_threads_per_rank': per_rank, 'omp_threads_per_rank': omp,
               'ok': world * per_rank <= cores and world * omp <= max(cores, world)}
    grad_bytes = sum(p.numel() for p in params) * 4
    checks = {'plan_identical_across_ranks': plan_identical, 'collectives_ok': ok,
              'collective_order_identical': bool(diag['collective_order_identical']),
              'distinct_ranks_seen': diag['distinct_devices'] == world, 'host_threads_ok': threads['ok'],
              'units_cover_gradients': sum(hi - lo for _, lo, hi in reducer.units) * 4 == grad_bytes,
              'unit_bytes_le_split': all((hi - lo) * 4 <= reducer.split_bytes or len(reducer.buckets[bi]) > 1
                                         for bi, lo, hi in reducer.units)}
    if rank == 0:
        print(json.dumps({'dry': True, 'n_gpus': world, 'config': args.config, 'backend': 'gloo (CPU rehearsal; RCCL on the GPU node)',
                          'rank_command': plan['real_cmd'], 'rank_env': plan['env'],
                          'trainable_parameters': len(params), 'gradient_mb': round(grad_bytes / 2.0 ** 20, 1),
                          'collective_units': len(reducer.units), 'bucket_mb': reducer.bucket_mb, 'plan_sha256': digest[:16],
                          'rehearsal_s': round(t_coll, 2), 'lr': 1e-3 * world * BATCH, 'lr_rule': 'models/train_rels.py:192',
                          'host_threads': threads, 'checks': checks, 'ok': all(checks.values()),
                          'note': 'no scaling number: nothing here ran on a GPU'}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not all(checks.values()):
        raise SystemExit('--dry: failed checks: %s' % [k for k, v in checks.items() if not v])


def launch_selftest(args):
    """the launcher path without the model (CPU + gloo when there is no HIP device): every rank joins the group, one
    all-reduce, rank 0 prints a line.  tests/test_bench_launcher.py runs `python bench.py --gpus 2 --launch-selftest`."""
    import torch.distributed as dist
    from lib import dist as D
    rank, world, local_rank = D.init_from_env()
    if world != args.gpus or (world > 1 and dist.get_world_size() != args.gpus):
        raise SystemExit('launch self-test: asked for %d ranks, the process group has %d' % (args.gpus, world))
    t = torch.tensor([float(rank + 1)])
    if torch.cuda.is_available():
        t = t.cuda(local_rank)
    if world > 1:
        dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({'launch_selftest': True, 'n_gpus': world, 'sum_of_ranks_plus_1': float(t.item()),
                          'backend': dist.get_backend() if world > 1 else None}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def pct(xs, q):
    xs = sorted(xs)
    if not xs:
        return None
    k = (len(xs) - 1) * q
    lo, hi = int(k), min(int(k) + 1, len(xs) - 1)
    return xs[lo] + (xs[hi] - xs[lo]) * (k - lo)


def step_stats(events, host_t, t_end):
    """per-step distribution of the timed region: `gpu` = HIP-event time between the starts of consecutive steps on the main
    stream (the last step ends at the final event), `host` = the host's enqueue time per step (run-ahead: it can be
    shorter than the GPU's)"""
    gpu = [events[i].elapsed_time(events[i + 1]) for i in range(len(events) - 1)]
    host = [1e3 * (b - a) for a, b in zip(host_t, host_t[1:] + [t_end])]
    r = lambda v: None if v is None else round(v, 3)
    return {'gpu_p50': r(pct(gpu, 0.5)), 'gpu_p90': r(pct(gpu, 0.9)), 'gpu_max': r(max(gpu)), 'gpu_min': r(min(gpu)),
            'host_p50': r(pct(host, 0.5)), 'host_max': r(max(host)),
            'gpu_per_step': [round(x, 2) for x in gpu] if len(gpu) <= 64 else None,
            'what': 'ms; gpu = HIP events at the start of every step on the main stream, host = enqueue time per step'}


def secondary(args, rank, world, dev):
    """Secondary rows (BASELINE.json configs other than the headline cfg2), each ONE JSON line with its own workload name:
      cfg1  PredCls evaluation, one 592x592 image with 20 GT boxes per step (380 candidate pairs)      -- eval img/s
      cfg3  SGDet training step, b = 6: RPN -> NMS -> RoI head -> per-class NMS -> <=64 detections/img -> GT matching ->
            rel_assignments (<=64 rows/img) -> context + relation head, fwd + bwd + clip + SGD             -- train img/s
      cfg4  SGCls training step of the ResNet-101 MotifNet, b = 6: the reference's `-resnet` RelModel cannot run (lib/rel_model.py:360-365
            vs :448); the row times the model with the documented repair (resnet_obj_fmap='layer4').  MOTIFS_CFG4=trunk: the
            conv1..layer3 trunk forward alone                                                          -- train img/s
      cfg5  SGDet evaluation stress, one image per step (the reference's eval decoder asserts batch 1), max_per_img = 80 ->
            all 80*79 = 6320 ordered pairs (require_overlap off) through the union-box relation head                                                                        -- eval img/s
    The random-weight detector is made confident (score_fc x30, RPN objectness x4, as tests/test_gpu_sgdet.py does) so that
    the per-class NMS keeps max_per_img detections per image: the workload is then the configuration's worst case."""
    from dataloaders.synthetic import SyntheticVG, make_blob
    from lib import _hip
    from lib.optim import FusedClipSGD
    from lib.rel_model import RelModel
    cfg = args.config
    torch.manual_seed(1234)
    cfg_id = int(cfg[3])
    seed = 1234 + 100 * cfg_id + rank
    np.random.seed(seed)
    meters = install_meters(_hip)
    extra = {}
    if cfg == 'cfg4' and os.environ.get('MOTIFS_CFG4', 'model') == 'trunk':
        from lib.object_detector import ObjectDetector
        det = ObjectDetector(classes=['bg'] + ['c%d' % i for i in range(150)], mode='gtbox', use_resnet=True).to(dev).eval()
        x = torch.randn(BATCH, 3, 592, 592, device=dev)
        per_step, unit_name = BATCH, 'images/sec ResNet-101 trunk (conv1..layer3) forward'
        workload = 'ResNet-101 conv1..layer3 forward (eval-mode BN), batch 6, 592x592 -> [6,1024,37,37]; 97.6 GFLOP/img'

        def step(i):
            with torch.no_grad():
                return det.feature_map(x)
    elif cfg == 'cfg4':
        # the config's MODEL: SGCls train step of RelModel(use_resnet=True) with the documented repair (the reference never builds
        # roi_fmap_obj for this configuration, lib/rel_model.py:360-365 vs :448): frozen ResNet-101 trunk (conv1..layer3, train-mode
        # BN like the reference), layer4 copies as the object / relation RoI heads (1.46 GFLOP per union RoI), pooling_dim 2048
        n_img = BATCH * 4
        ds = SyntheticVG(num_images=n_img, seed=seed, n_boxes=N_BOXES, n_rels=N_RELS)
        kw = dict(MODEL_KW, pooling_dim=2048)
        model = RelModel(classes=ds.ind_to_classes, rel_classes=ds.ind_to_predicates, mode='sgcls', num_gpus=1, use_resnet=True,
                         resnet_obj_fmap='layer4', **kw)
        for _, p in model.detector.named_parameters():
            p.requires_grad = False
        model.to(dev).train()
        blobs = [make_blob(ds, range(i * BATCH, (i + 1) * BATCH), is_train=True) for i in range(n_img // BATCH)]
        for bl in blobs:
            bl.scatter()
        lr = 1e-3 * world * BATCH
        fc = [p for n, p in model.named_parameters() if n.startswith('roi_fmap') and p.requires_grad]
        rest = [p for n, p in model.named_parameters() if not n.startswith('roi_fmap') and p.requires_grad]
        opt = FusedClipSGD([{'params': fc, 'lr': lr / 10.0}, {'params': rest}], lr=lr, momentum=0.9, weight_decay=1e-4)
        per_step, unit_name = BATCH, 'images/sec MotifNet-SGCls (ResNet-101) fwd+bwd'
        workload = ('SGCls MotifNet ResNet-101 train step (fwd+bwd+clip+SGD): frozen conv1..layer3 trunk, trainable layer4 copies as object / '
                    'relation RoI heads, batch 6/GPU, 20 GT boxes/img, <=256 rel rows/img, 592x592; use_resnet repaired with '
                    "resnet_obj_fmap='layer4'")

        def step(i):
            res = model[blobs[i % len(blobs)]]
            loss = F.cross_entropy(res.rm_obj_dists, res.rm_obj_labels) + F.cross_entropy(res.rel_dists, res.rel_labels[:, -1])
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step(max_norm=5.0)
            extra['rows'] = int(res.rel_labels.shape[0])
            return loss
    else:
        mode = {'cfg1': 'predcls', 'cfg3': 'sgdet', 'cfg5': 'sgdet'}[cfg]
        b = {'cfg1': 1, 'cfg3': BATCH, 'cfg5': 1}[cfg]      # evaluation decodes one image per step (reference decoder_rnn.py:215)
        n_img = b * 4
        ds = SyntheticVG(num_images=n_img, seed=seed, n_boxes=N_BOXES, n_rels=N_RELS)
        kw = dict(MODEL_KW)
        if mode == 'sgdet':
            kw['order'] = 'leftright'
        model = RelModel(classes=ds.ind_to_classes, rel_classes=ds.ind_to_predicates, mode=mode, num_gpus=1,
                         max_per_img=80 if cfg == 'cfg5' else 64, **kw)
        for _, p in model.detector.named_parameters():
            p.requires_grad = False
        if mode == 'sgdet':
            with torch.no_grad():
                model.detector.score_fc.weight.mul_(30.0)
                model.detector.rpn_head.conv[2].weight.mul_(4.0)
        if cfg == 'cfg5':
            model.require_overlap = False                     # the stress case: ALL N(N-1) ordered pairs, not only overlapping ones
        model.to(dev)
        train = cfg == 'cfg3'
        if train:
            # A random-weight detector matches (almost) no synthetic GT box, and rel_assignments would then sample ~1 row per
            # image.  Make the workload what training sees: run the detector once in eval mode and use ITS detections as the
            # images' ground-truth boxes (random classes, 30 random relations): GT matching then labels the detections of
            # the timed steps (same proposals, detector dropout off) and <= 64 rows / image are sampled.
            model.eval()
            model.eval_on_device = True
            rs = np.random.RandomState(seed)
            with torch.no_grad():
                for i in range(n_img):
                    out = model[make_blob(ds, [i], is_train=False)]
                    boxes = model.last_eval_result.rm_box_priors.float().cpu().numpy()     # the boxes GT matching compares (:319-326)
                    nb = boxes.shape[0]
                    ds.gt_boxes[i] = (boxes * (1024.0 / 592.0)).astype(np.float32)          # stored at BOX_SCALE like VG
                    ds.gt_classes[i] = rs.randint(1, 151, nb).astype(np.int64)
                    pairs = np.array([(a_, b_) for a_ in range(nb) for b_ in range(nb) if a_ != b_])
                    sel = pairs[rs.choice(len(pairs), size=min(N_RELS, len(pairs)), replace=False)]
                    ds.relationships[i] = np.column_stack((sel, rs.randint(1, 51, sel.shape[0]))).astype(np.int64)
            model.eval_on_device = False
        model.train(train)
        if train:
            # the frozen detector's RoI-head dropout would move the detections away from the boxes just taken as ground
            # truth (fewer matches -> fewer relation rows): keep it off, as a trained detector's detections match their GT
            from lib.hip_ops import Dropout
            for m in model.detector.modules():
                if isinstance(m, Dropout):
                    m.eval()
        blobs = [make_blob(ds, range(i * b, (i + 1) * b), is_train=train) for i in range(n_img // b)]
        for bl in blobs:
            bl.scatter()
        per_step = b
        if train:
            lr = 1e-3 * world * b
            fc = [p for n, p in model.named_parameters() if n.startswith('roi_fmap') and p.requires_grad]
            rest = [p for n, p in model.named_parameters() if not n.startswith('roi_fmap') and p.requires_grad]
            opt = FusedClipSGD([{'params': fc, 'lr': lr / 10.0}, {'params': rest}], lr=lr, momentum=0.9, weight_decay=1e-4)
            unit_name = 'images/sec MotifNet-SGDet fwd+bwd'
            workload = ('SGDet MotifNet VGG16 train step (RPN + NMS + RoI head + per-class NMS -> <=64 detections/img, GT matching, '
                        'rel_assignments <=64 rows/img, context LSTMs, relation head; fwd+bwd+clip+SGD), batch %d/GPU, 592x592' % b)

            # the frozen detector stage runs one batch ahead on its own thread / stream (RelModel.detect_ahead: the host's waits
            # for proposal counts, kept detections and the sampled relations no longer leave the device idle); every timed
            # step still contains one detector stage and one relation stage -- barrier() below drains the stage in flight, so
            # the stage of timed step 0 is made in the warm-up and the one the last step starts is made inside the region.
            # MOTIFS_DETECT_AHEAD=0: the in-line order (A/B); =N: N batches ahead
            ahead = min(int(os.environ.get('MOTIFS_DETECT_AHEAD', '2')), len(blobs) - 1)
            extra['detector_stage'] = '%d batch(es) ahead (worker thread + own HIP stream)' % ahead if ahead else 'in line'

            def step(i):
                res = model[blobs[i % len(blobs)]]
                for j in range(1, ahead + 1):                     # (a stage already in flight is not started twice)
                    model.detect_ahead_blob(blobs[(i + j) % len(blobs)])
                loss = F.cross_entropy(res.rm_obj_dists, res.rm_obj_labels) + F.cross_entropy(res.rel_dists, res.rel_labels[:, -1])
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step(max_norm=5.0)
                extra['dets'], extra['rows'] = int(res.rm_obj_labels.shape[0]), int(res.rel_labels.shape[0])
                return loss
        else:
            model.eval_on_device = True                       # Recall@K inputs stay on the device (no [Nrel,51] D2H per image)
            unit_name = 'images/sec MotifNet-%s eval' % ('PredCls' if cfg == 'cfg1' else 'SGDet')
            workload = ('PredCls evaluation forward, 1 image (20 GT boxes -> 380 pairs) per step, VGG16, 592x592' if cfg == 'cfg1' else
                        'SGDet evaluation forward, 1 image per step, max_per_img 80 -> ALL ordered pairs (80*79 = 6320) through the '
                        'union-box relation head, VGG16, 592x592')

            # evaluation: the detector stage of the next images is started before this image's relation stage is issued
            # (RelModel.detect_ahead; MOTIFS_DETECT_AHEAD=0 = in line, =N: N images ahead)
            # GT-box modes stay in line: their detector stage is the trunk alone, no waits (PredCls, r05_c15/16: 275 img/s in
            # line, 292 one image ahead, 171 two ahead)
            ahead = min(int(os.environ.get('MOTIFS_DETECT_AHEAD', '2')), len(blobs) - 1) if mode == 'sgdet' else 0
            extra['detector_stage'] = '%d image(s) ahead (worker thread + own HIP stream)' % ahead if ahead else 'in line'

            def step(i):
                with torch.no_grad():
                    for j in range(1, ahead + 1):
                        model.detect_ahead_blob(blobs[(i + j) % len(blobs)])
                    out = model[blobs[i % len(blobs)]]
                extra['dets'], extra['rows'] = int(out[0].shape[0]), int(out[3].shape[0])
                return out

    def barrier():
        if hasattr(model, 'ahead_drain'):
            model.ahead_drain()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    set_meters(meters, True)
    t0 = time.time()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    dt = time.time() - t0
    set_meters(meters, False)
    if args.host_profile and rank == 0:
        # where the host spends a step of this configuration (cProfile over 5 more steps): the D2H round trips show up as time
        # inside .item() / .cpu() / nonzero() -- the host waiting for the queue to drain
        import cProfile
        import io
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for i in range(5):
            step(args.warmup + args.steps + i)
        torch.cuda.synchronize()
        pr.disable()
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats('tottime').print_stats(40)
        sys.stderr.write(buf.getvalue())
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank == 0:
        _hip.check_faults()
        split = _hip.lib().mh_mfma_split()
        peak = PEAK_BF16_MFMA_TFLOPS / split if split else PEAK_FP32_MFMA_TFLOPS
        c = merge(meters['plconv'].summary(), meters['plconv_img'].summary(), meters['plconv_pool'].summary(), meters['conv'].summary())
        g = merge(meters['gemm_planes'].summary(), meters['gemm'].summary(), meters['gemm_inloop'].summary())
        dom, dom_name = (g, 'gemm_kernel (relation-head / RoI-head / 1x1-conv GEMMs)') if g['total_ms'] >= c['total_ms'] else \
            (c, 'conv3x3_nhwc_kernel (implicit GEMM)')
        line = {'metric': unit_name, 'value': world * per_step * args.steps / dt, 'unit': 'img/s', 'n_gpus': world,
                'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
                'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE, 'data': 'synthetic',
                'config': dict({'workload': workload, 'baseline_config': cfg, 'global_batch': world * per_step,
                                'parallelism': 'dp%d' % world}, **extra),
                'roofline': {'bound': 'mfma', 'kernel': dom_name, 'achieved': dom['tflops'], 'peak': peak, 'unit': 'TFLOP/s',
                             'frac': dom['tflops'] / peak, 'traffic': None, 'ms_per_step': dom['total_ms'] / args.steps,
                             'launches': dom['launches']},
                'kernels': {'conv3x3': {'tflops': c['tflops'], 'ms_per_step': c['total_ms'] / args.steps, 'launches': c['launches']},
                            'gemm': {'tflops': g['tflops'], 'ms_per_step': g['total_ms'] / args.steps, 'launches': g['launches']}},
                'hbm_kernels': hbm_rows(meters, args.steps)}
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-iters', type=int, default=3, help='timed iterations of the CPU baseline (after 1 warm-up)')
    ap.add_argument('--host-profile', action='store_true',
                    help='after the timed region: host enqueue time of 3 unsynchronised steps + a cProfile of 3 more (stderr)')
    ap.add_argument('--config', default='cfg2', choices=['cfg1', 'cfg2', 'cfg3', 'cfg4', 'cfg5', 'recipe'],
                    help='BASELINE.json configs[i-1]; cfg2 (default) is the headline metric, the others are secondary rows; '
                         'recipe = cfg2 with the edge-context LSTM of the shipped training script (-nl_edge 4, '
                         'scripts/train_models_sgcls.sh:19-21; SURVEY.md 8d)')
    ap.add_argument('--launch-selftest', action='store_true', help='only exercise the rank launcher / process group (no model)')
    ap.add_argument('--meter-every', type=int, default=10, help='kernel meters (HIP event pairs) record on every n-th timed step '
                    '(a metered step is 0.4-0.5 ms longer: gpurun r05_c5; every 10th keeps the headline within 0.3 %% of the unmetered step)')
    ap.add_argument('--h2d-steps', type=int, default=8, help='steps of the second, H2D-inclusive timing (0 = skip)')
    ap.add_argument('--gemm-shapes', default='', help='after the timed region: one more cfg2 step with every matrix-product call '
                    'logged (binding, M, N, K, transposes, HIP-event time, stream) as JSON lines into this file')
    ap.add_argument('--dry', action='store_true', help='rehearse the N-rank launch on the CPU (gloo): rank command lines, the real '
                    'parameter list through the gradient reducer\'s planning code on every rank, the collectives at their real sizes and '
                    'order, host-thread budget; one JSON line, no GPU needed, no scaling number produced')
    args = ap.parse_args()

    if args.dry:
        if 'WORLD_SIZE' not in os.environ:
            raise SystemExit(dry_parent(args, sys.argv[1:]))
        return dry_rank(args)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))       # not started by torchrun: start the ranks ourselves
    if args.launch_selftest:
        return launch_selftest(args)
    from lib import dist as D
    rank, world, local_rank = D.init_from_env()
    if world != args.gpus or (world > 1 and torch.distributed.get_world_size() != args.gpus):
        raise SystemExit('--gpus %d but the process group has %d rank(s): refusing to report a job of another size' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (no CPU fallback for the hot path)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1 and os.environ.get('OMP_NUM_THREADS') == '1':
        # torch.distributed.run gives every rank ONE host thread unless told otherwise; building the model (the block-orthogonal
        # LSTM initialisation is 12 QR factorisations of 4424^2 / 4808^2 matrices) then takes 45 s per rank instead of 12
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // world)))

    from dataloaders.synthetic import SyntheticVG, make_blob
    from lib import _hip
    from lib.losses import relation_losses
    from lib.optim import FusedClipSGD
    from lib.rel_model import RelModel

    if args.config not in ('cfg2', 'recipe'):
        return secondary(args, rank, world, dev)
    model_kw = dict(MODEL_KW, nl_edge=4) if args.config == 'recipe' else MODEL_KW
    torch.manual_seed(1234)
    np.random.seed(1234 + 200 + rank)
    n_img = BATCH * 4
    ds = SyntheticVG(num_images=n_img, seed=1234 + 200 + rank, n_boxes=N_BOXES, n_rels=N_RELS)
    model = RelModel(classes=ds.ind_to_classes, rel_classes=ds.ind_to_predicates, mode='sgcls', num_gpus=1, **model_kw)
    for _, p in model.detector.named_parameters():           # models/train_rels.py:50-52
        p.requires_grad = False
    sd_cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == 'cfg2':
        sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(dev).train()
    lr = 1e-3 * world * BATCH                                 # train_rels.py:192
    fc = [p for n, p in model.named_parameters() if n.startswith('roi_fmap') and p.requires_grad]
    rest = [p for n, p in model.named_parameters() if not n.startswith('roi_fmap') and p.requires_grad]
    opt = FusedClipSGD([{'params': fc, 'lr': lr / 10.0}, {'params': rest}], lr=lr, momentum=0.9, weight_decay=1e-4)
    reducer = D.OverlappedGradReducer([p for p in model.parameters() if p.requires_grad])   # inert at world 1
    import copy
    host_blobs = [make_blob(ds, range(i * BATCH, (i + 1) * BATCH), is_train=True).pin_memory() for i in range(n_img // BATCH)]
    blobs = [copy.copy(b) for b in host_blobs]                # shallow: the device copies below replace the tensor attributes
    for b in blobs:
        b.scatter()                                           # inputs resident in HBM before the timed region
    meters = install_meters(_hip)
    opt_events = []
    roww = D.RowWeights(dev)
    if world > 1:
        model.rows_hook = roww.start          # the row-count all-reduce starts inside the forward pass, asynchronously

    ahead = {}

    def step(i, upload=False):
        blob = blobs[i % len(blobs)]
        if upload == 'inline':               # the reference's per-step scatter (train_rels.py:137, blob.py:155-180): 25 MB of
            blob = copy.copy(host_blobs[i % len(host_blobs)])      # page-locked batch -> HBM, asynchronous, ordered on the stream
            blob.scatter()
        elif upload:                         # the shipped loop (models/train_rels.py): the NEXT batch's copies start on the copy stream
            blob = ahead.pop(i, None) or copy.copy(host_blobs[i % len(host_blobs)])      # while this step runs (Blob.prefetch)
            blob.scatter()
            ahead[i + 1] = copy.copy(host_blobs[(i + 1) % len(host_blobs)]).prefetch()
        res = model[blob]
        ls = relation_losses(res)            # [class loss, relation loss]: models/train_rels.py:140-141 as one node (lib/losses.py)
        loss = (ls * roww.get()).sum() if world > 1 else ls.sum()
        opt.zero_grad(set_to_none=True)
        reducer.prepare()
        loss.backward()                  # N > 1: each 32 MB gradient bucket is all-reduced (RCCL) as soon as it is complete
        reducer.finish()
        timed = meters['roi'].enabled and KernelMeter.sampling
        if opt.overlap_next_forward:
            opt.meter_events = opt_events if timed else None      # events on the optimizer's own stream (lib/optim.py)
        elif timed:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        opt.step(max_norm=5.0)           # global-norm clip (5.0) + SGD(momentum, wd) in three multi-tensor launches
        if timed and not opt.overlap_next_forward:
            ev[1].record()
            opt_events.append(ev)
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    from lib.pytorch_misc import quiet_gc
    quiet_gc()                 # what models/train_rels.py does before its first epoch: no full garbage collection inside a step.
                               # BEFORE the warm-up (round 5): the collection takes ~0.2 s with the GPU idle, and the first dozen steps
                               # after such a pause run ~6 % slower than the steady state (17.0 against 16.1 ms, gpurun r05_c4: the
                               # part's power management ramping up again) -- the warm-up steps are there to absorb exactly that
    for i in range(args.warmup):
        step(i)
    barrier()
    set_meters(meters, True)
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    host_t = []
    metered_steps = 0
    t0 = time.time()
    for i in range(args.steps):
        step_ev[i].record()
        host_t.append(time.perf_counter())
        metered_steps += bool(sample_step(i, args.meter_every))
        loss = step(args.warmup + i)
    step_ev[args.steps].record()
    t_enq = time.perf_counter()
    barrier()
    dt = time.time() - t0
    dt_local = dt
    KernelMeter.sampling = True
    set_meters(meters, False)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    stats = step_stats(step_ev, host_t, t_enq)

    # second timing, meters off: the same steps with the batch uploaded from page-locked host memory inside every step
    h2d = None
    if args.h2d_steps > 0:
        for i in range(2):
            step(i, upload=True)
        barrier()
        t1 = time.time()
        for i in range(args.h2d_steps):
            step(2 + i, upload=True)
        barrier()
        dth = torch.tensor([time.time() - t1], dtype=torch.float64, device=dev)
        if world > 1:
            torch.distributed.all_reduce(dth, op=torch.distributed.ReduceOp.MAX)
        dth = float(dth.item())
        nbytes = sum(getattr(host_blobs[0], n).numel() * getattr(host_blobs[0], n).element_size()
                     for n in ('imgs', 'gt_boxes', 'gt_classes', 'gt_rels'))
        # ... and the reference's placement of the copy: on the compute stream, in front of the step's first kernel
        ahead.clear()
        for i in range(2):
            step(i, upload='inline')
        barrier()
        t1b = time.time()
        for i in range(args.h2d_steps):
            step(2 + i, upload='inline')
        barrier()
        dti = torch.tensor([time.time() - t1b], dtype=torch.float64, device=dev)
        if world > 1:
            torch.distributed.all_reduce(dti, op=torch.distributed.ReduceOp.MAX)
        dti = float(dti.item())
        h2d = {'value': world * BATCH * args.h2d_steps / dth, 'unit': 'img/s', 'ms_per_step': 1e3 * dth / args.h2d_steps,
               'steps': args.h2d_steps, 'bytes_per_step': nbytes,
               'what': 'same step with a batch uploaded in every step (pinned host -> HBM), unmetered: the NEXT batch\'s copies run on a copy '
                       'stream while the step computes (Blob.prefetch, what models/train_rels.py does)',
               'inline': {'value': world * BATCH * args.h2d_steps / dti, 'ms_per_step': 1e3 * dti / args.h2d_steps,
                          'what': 'the same with the copies on the compute stream in front of the step (the reference\'s placement, round 5\'s figure)'}}

    # third timing, meters off, inputs resident: what the headline would be without the sampled HIP-event pairs
    unmetered = None
    if args.h2d_steps > 0:
        barrier()
        t2 = time.time()
        for i in range(args.h2d_steps):
            step(i)
        barrier()
        dtu = torch.tensor([time.time() - t2], dtype=torch.float64, device=dev)
        if world > 1:
            torch.distributed.all_reduce(dtu, op=torch.distributed.ReduceOp.MAX)
        dtu = float(dtu.item())
        unmetered = {'value': world * BATCH * args.h2d_steps / dtu, 'unit': 'img/s', 'ms_per_step': 1e3 * dtu / args.h2d_steps,
                     'steps': args.h2d_steps, 'what': 'the same step, inputs resident, kernel meters off'}

    # where the UNPROFILED step spends its time on the main stream: five events per step (start, detector stage done, forward done,
    # backward done, optimizer done) over a few extra steps -- under rocprofv3 the host becomes the bottleneck and the queues'
    # overlap in its traces is not the production one
    segments = None
    if args.h2d_steps > 0 and world == 1:
        marks = []
        hook = model.detector.register_forward_hook(lambda m, i, o: marks[-1].__setitem__(1, _ev()))

        def _ev():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        model.stream_marks = []
        for i in range(args.h2d_steps):
            marks.append([_ev(), None, None, None, None])
            blob = blobs[i % len(blobs)]
            res = model[blob]
            loss_i = relation_losses(res).sum()
            marks[-1][2] = _ev()
            opt.zero_grad(set_to_none=True)
            reducer.prepare()
            loss_i.backward()
            reducer.finish()
            marks[-1][3] = _ev()
            opt.step(max_norm=5.0)
            opt.synchronize()
            marks[-1][4] = _ev()
        hook.remove()
        torch.cuda.synchronize()
        seg = [[a.elapsed_time(b) for a, b in zip(m[:-1], m[1:])] for m in marks[1:]]
        n = max(len(seg), 1)
        waits = [a.elapsed_time(b) for a, b in model.stream_marks[1:]]
        model.stream_marks = None
        segments = {'detector_stage_ms': sum(x[0] for x in seg) / n, 'rest_of_forward_ms': sum(x[1] for x in seg) / n,
                    'backward_ms': sum(x[2] for x in seg) / n, 'optimizer_ms': sum(x[3] for x in seg) / n, 'steps': len(seg),
                    'main_waits_for_context_ms': (sum(waits) / len(waits)) if waits else None,
                    'what': 'HIP events on the main stream, unprofiled, meters off; the detector stage = frozen trunk + RoI head of the GT boxes; '
                            'main_waits_for_context_ms = part of rest_of_forward in which the main stream (union-box branch done) waits for '
                            'the context branch on the other stream'}
    if world > 1:
        torch.distributed.barrier()

    if args.gemm_shapes and rank == 0:
        log = []

        def logged(name, shape_of):
            orig = getattr(_hip, name)

            def call(*a, **k):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                y = orig(*a, **k)
                e.record()
                log.append((name, shape_of(a, k, y), s, e, int(torch.cuda.current_stream().cuda_stream)))
                return y
            setattr(_hip, name, call)
            return orig

        def dense_shape(a, k, y):
            ta = bool(a[2] if len(a) > 2 else k.get('trans_a', False))
            tb = bool(a[3] if len(a) > 3 else k.get('trans_b', False))
            return dict(M=int(y.shape[0]), N=int(y.shape[1]), K=int(a[0].shape[0] if ta else a[0].shape[1]), trans_a=ta, trans_b=tb)
        saved = {n: logged(n, dense_shape) for n in ('gemm', 'gemm_inloop')}
        saved['gemm_planes'] = logged('gemm_planes', lambda a, k, y: dict(M=a[0].rows, N=a[1].rows, K=a[0].K, trans_a=False, trans_b=True))
        step(0)
        barrier()
        for n, f in saved.items():
            setattr(_hip, n, f)
        streams = sorted({r[4] for r in log})
        with open(args.gemm_shapes, 'w') as f:
            for name, shp, s, e, st in log:
                f.write(json.dumps(dict(shp, binding=name, us=round(1e3 * s.elapsed_time(e), 1), stream=streams.index(st),
                                        gflop=round(2e-9 * shp['M'] * shp['N'] * shp['K'], 2))) + '\n')

    if args.host_profile and rank == 0:
        # where the HOST spends a step: enqueue time without synchronisation (the GPU queue is empty at the start, so this
        # is Python + launch overhead, not waiting), then the same under cProfile
        import cProfile
        import io
        import pstats
        hs = []
        for i in range(3):
            t1 = time.perf_counter()
            step(i)
            hs.append(1e3 * (time.perf_counter() - t1))
        barrier()
        # stall probes: device-allocator activity per step (segments = hipMalloc / hipFree calls of the caching allocator),
        # the same steps with Python's cyclic GC off, and the Python stack of the main thread every 20 ms
        import faulthandler
        import gc
        def seg():
            st = torch.cuda.memory_stats()
            return (st.get('segment.all.allocated', 0), st.get('segment.all.freed', 0), st.get('num_alloc_retries', 0),
                    st.get('reserved_bytes.all.current', 0) >> 20)
        for label, nogc in (('gc on', False), ('gc off', True)):
            if nogc:
                gc.disable()
            rows = []
            for i in range(6):
                s0 = seg()
                t1 = time.perf_counter()
                step(i)
                rows.append(('%.1f' % (1e3 * (time.perf_counter() - t1)), tuple(b_ - a_ for a_, b_ in zip(s0, seg()))))
            barrier()
            gc.enable()
            sys.stderr.write('probe [%s] host ms per step, (segments allocated, freed, retries, reserved MiB delta): %s; gc counts %s\n'
                             % (label, rows, gc.get_count()))
        faulthandler.dump_traceback_later(0.02, repeat=True, file=sys.stderr)
        for i in range(6):
            step(i)
        faulthandler.cancel_dump_traceback_later()
        barrier()
        pr = cProfile.Profile()
        pr.enable()
        for i in range(3):
            step(i)
        pr.disable()
        barrier()
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats('tottime').print_stats(50)
        sys.stderr.write('host enqueue ms per step, unsynchronised: %s (timed region: %.2f ms/step)\n%s\n'
                         % (['%.2f' % h for h in hs], 1e3 * dt / args.steps, buf.getvalue()))

    msteps = max(metered_steps, 1)              # the steps the kernel meters recorded (sample_step)
    # what a first N > 1 run needs to be readable (lib/dist.py: scaling_diagnostics): gathered through the process group
    diag = D.scaling_diagnostics(reducer, dev, 1e3 * dt_local / args.steps)
    if rank == 0:
        # the plane conv's calls: the trunk's (<= 6 images) and the 3x3 convs over many small RoI maps (hip_ops.conv3x3_small_maps)
        plc, c2 = merge(meters['plconv'].summary('trunk'), meters['plconv_img'].summary(), meters['plconv_pool'].summary()), meters['conv'].summary()
        cmaps = meters['plconv'].summary('maps')
        conv = merge(plc, cmaps, c2)
        gpl, gg, gi = meters['gemm_planes'].summary(), meters['gemm'].summary(), meters['gemm_inloop'].summary()
        gm = merge(gpl, gg, gi)
        _hip.check_faults()
        # `achieved` counts ALGORITHMIC fp32 flops (2*M*N*K of the convolution).  The peak is the matrix-core peak for
        # the way this build evaluates an fp32 product: six bf16 MFMAs per product (bf16x6, fp32-accurate) or the
        # f32-input MFMA.
        split = _hip.lib().mh_mfma_split()
        if split:
            peak = PEAK_BF16_MFMA_TFLOPS / split
            how = 'fp32 products as %d bf16 MFMAs (exact 3-way split, fp32 accumulate): peak = %.0f/%d' % (
                split, PEAK_BF16_MFMA_TFLOPS, split)
            if _hip.lib().mh_split_f16():
                how = 'fp32 products as 3 f16 MFMAs (f16x3: two-term f16 split of row-scaled operands, fp32 accumulate): peak = %.0f/3' % PEAK_BF16_MFMA_TFLOPS
        else:
            peak, how = PEAK_FP32_MFMA_TFLOPS, 'v_mfma_f32_32x32x2_f32'
        traffic = gemm_traffic = None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), TRAFFIC_SUMMARY)
        if os.path.exists(tpath) and _hip.lib().mh_split_f16():   # collected on the f16x3 build, on exactly these 14 launches
            with open(tpath) as f:
                traffic = json.load(f)
            if traffic.get('launches') * msteps != plc['launches']:
                traffic = None                               # another launch mix: the offline figure does not apply
        gpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), GEMM_TRAFFIC_SUMMARY)
        if os.path.exists(gpath) and _hip.lib().mh_split_f16():
            with open(gpath) as f:
                gemm_traffic = json.load(f)
        line = {
            'metric': 'images/sec MotifNet-SGCls fwd+bwd' + (' (shipped recipe: nl_edge 4)' if args.config == 'recipe' else ''),
            'value': world * BATCH * args.steps / dt, 'unit': 'img/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE, 'data': 'synthetic',
            'ms_per_step_p50': stats['gpu_p50'], 'ms_per_step_p90': stats['gpu_p90'], 'ms_per_step_max': stats['gpu_max'],
            'step_ms': stats, 'h2d_inclusive': h2d, 'unmetered': unmetered, 'main_stream_segments': segments, 'meter_every': args.meter_every, 'metered_steps': metered_steps,
            'config': {'workload': 'SGCls MotifNet VGG16 train step (fwd+bwd+clip+SGD), order=leftright, nl_obj=2, '
                                   'nl_edge=%d, hidden 512, batch 6/GPU, 20 GT boxes/img, <=256 rel rows/img, 592x592' % model_kw['nl_edge'],
                       'global_batch': world * BATCH, 'parallelism': 'dp%d' % world, 'final_loss': float(loss.item())},
            'roofline': {'bound': 'mfma', 'kernel': 'pl::conv3x3_ring_kernel (implicit GEMM on pre-split plane images, LDS-DMA ring K loop: the 12 VGG trunk layers '
                                                    'conv1_2 .. conv5_3 -- image / pooled-image / fp32 epilogues, K slices added up in the launch -- '
                                                    '+ the union tower\'s conv over 1536 7x7 maps, fwd / dgrad); ' + how,
                         'achieved': conv['tflops'], 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': conv['tflops'] / peak, 'traffic': traffic['bytes_per_launch'] if traffic else None,
                         'traffic_unit': 'bytes per launch (L2 fabric side: HBM + Infinity-Cache)',
                         'traffic_algorithmic': traffic['algorithmic_bytes_per_launch'] if traffic else None,
                         'traffic_source': TRAFFIC_SUMMARY + ' (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over one '
                                           'launch per trunk layer of this step with the shipped library and the step\'s epilogues (image / pooled image / fp32, '
                                           'conv5 in K slices added up in the launch), tools/r04/traffic.sh, gpurun r06_c13; replayed offline: a '
                                           'PMC pass over the whole step does not finish); covers the 12 plane-trunk launches',
                         'frac_of_f32_mfma_peak': conv['tflops'] / PEAK_FP32_MFMA_TFLOPS,
                         'frac_of_bf16x6_peak': conv['tflops'] / (PEAK_BF16_MFMA_TFLOPS / 6.0),
                         'launches': conv['launches'], 'avg_launch_ms': conv['avg_ms'],
                         'flops_per_launch': conv['flops_per_launch'],
                         'trunk_only': {'tflops': plc['tflops'], 'frac': plc['tflops'] / peak, 'ms_per_step': plc['total_ms'] / msteps,
                                        'launches': plc['launches']},
                         'small_maps': {'tflops': cmaps['tflops'], 'frac': cmaps['tflops'] / peak, 'ms_per_step': cmaps['total_ms'] / msteps,
                                        'launches': cmaps['launches'], 'what': 'the plane conv over many small RoI maps (mask tower fwd / dgrad)'}},
        }
        line['roofline_gemm'] = {
            'bound': 'mfma', 'kernel': 'every matrix product of the step: pl::gemm_ring_kernel / pl::gemm_kernel on ready plane images (mh_gemm_planes: fc6/fc7 '
                                       'of the RoI heads fwd, dgrad, wgrad) + the small-product engine (mh_gemm_small_f32: bf16x6, one launch per product, '
                                       'split-K reduced inside the launch); priced against the f16x3 peak',
            'achieved': gm['tflops'], 'peak': peak, 'unit': 'TFLOP/s', 'frac': gm['tflops'] / peak,
            'launches': gm['launches'], 'ms_per_step': gm['total_ms'] / msteps,
            'flops_per_step': gm['flops_per_launch'] * gm['launches'] / msteps,
            'products_on_images': {'tflops': gpl['tflops'], 'frac': gpl['tflops'] / peak, 'ms_per_step': gpl['total_ms'] / msteps,
                                   'launches': gpl['launches']},
            'note': 'HIP-event time of the calls; some run concurrently with the other HIP stream (context branch)',
            # fabric-side bytes of the big products (the 16 launches of products_on_images are these five shapes, forward and backward)
            'traffic': gemm_traffic['bytes_per_launch'] if gemm_traffic else None,
            'traffic_unit': 'bytes per launch (L2 fabric side: HBM + Infinity-Cache), mean over the five big product shapes of the step',
            'traffic_algorithmic': gemm_traffic['algorithmic_bytes_per_launch'] if gemm_traffic else None,
            'traffic_source': GEMM_TRAFFIC_SUMMARY + ' (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over one launch per shape -- fc6 forward / '
                              'input gradient / weight gradient at 1536 rows, the 120-row object fc6, fc7 -- with the shipped library, tools/traffic_run.sh gemm, '
                              'gpurun r06_c1 (XCD-banded tile order; round 5: 3.09x); replayed offline: a PMC pass over the whole step does not finish)'}
        # which kernel class takes more of the step by HIP-event time (the verdict of round 2 noted that GEMM-class work exceeds
        # the conv's: both rooflines are reported, this names the larger one)
        line['dominant_by_time'] = {'class': 'gemm' if gm['total_ms'] > conv['total_ms'] else 'conv3x3',
                                    'conv3x3_ms_per_step': conv['total_ms'] / msteps, 'gemm_ms_per_step': gm['total_ms'] / msteps}
        # `roofline` headlines the class that takes more of the step; the other class's object stays beside it and both fractions
        # are repeated in the headline object
        line['roofline_conv'] = line['roofline']
        if line['dominant_by_time']['class'] == 'gemm':
            line['roofline'] = dict(line['roofline_gemm'], avg_launch_ms=gm['avg_ms'], flops_per_launch=gm['flops_per_launch'])
        line['roofline'] = dict(line['roofline'], dominant_class=line['dominant_by_time']['class'],
                                frac_conv3x3=conv['tflops'] / peak, frac_gemm=gm['tflops'] / peak, frac_trunk_only=plc['tflops'] / peak)
        opt_ms = sum(a.elapsed_time(b) for a, b in opt_events) / max(len(opt_events), 1) if opt_events else None
        n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
        line['hbm_kernels'] = hbm_rows(meters, msteps, opt_ms, 20.0 * n_train)
        line['roofline_conv']['ms_per_step'] = conv['total_ms'] / msteps
        line['scaling_diagnostics'] = diag
        line['optimizer'] = {'deferred_to_own_stream': bool(opt.overlap_next_forward),
                             'what': 'norm + update enqueued on the optimizer stream, beside the next step\'s frozen trunk' if opt.overlap_next_forward
                                     else 'norm + update on the compute stream'}
        line['calibration'] = calibration()
        if sd_cpu is not None:
            try:
                line['cpu_baseline'] = cpu_baseline(ds, sd_cpu, iters=args.cpu_iters)
            except Exception as ex:                          # the baseline must never take the bench line down
                line['cpu_baseline'] = {'value': None, 'unit': 'img/s', 'cores': os.cpu_count(),
                                        'kind': 'port', 'sample': 'failed: %r' % (ex,)}
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
