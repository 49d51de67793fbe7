"""
Data parallelism the MI355X way: one process per GPU (torchrun), parameters resident on every rank, gradients
summed with bucketed RCCL all-reduce over xGMI (backend "nccl" is RCCL on ROCm).  Replaces the reference's
single-process `replicate` + `parallel_apply` + `Gather` scheme, which re-broadcast all 419 M parameters every step
(lib/rel_model.py:549-560).

Images are independent units, so there is no data-path collective: each rank builds its own Blob and the only
exchange is the gradient reduction.  Semantics match the reference's single-process loss (cross-entropy averaged
over ALL rows of the global batch): each rank scales its loss by rows_rank / rows_global before backward
(`global_row_weights`), gradients are then SUMMED.  Frozen detector parameters never enter a bucket.

Buckets are ~32 MB of fp32 (xGMI is point-to-point, 7 links x ~153 GB/s per GPU: a few large messages, not many
small ones).  `OverlappedGradReducer` launches each bucket's all-reduce from a gradient hook while backward is still
running; `GradBuckets` is the plain post-backward variant.  Both work unchanged on CPU tensors with the gloo backend
(used by the world_size-2 tests).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world_size, local_rank).  A single process without the env stays un-initialised."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def global_row_weights(row_counts, device):
    """row_counts: python list of this rank's row counts per loss term.  Returns, per term, rows_rank/rows_global --
    the factor that turns a rank-local mean loss into its share of the global mean (one tiny all-reduce)."""
    t = torch.tensor([float(c) for c in row_counts], dtype=torch.float64)
    if torch.device(device).type == 'cuda':
        from lib.pytorch_misc import h2d
        t = h2d(t, device)                 # page-locked + asynchronous: a pageable upload would drain the GPU queue every step
    tot = t.clone()
    if world_size() > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return (t / tot.clamp(min=1.0)).float()


class RowWeights(object):
    """`global_row_weights` without a blocking collective in front of backward: the two row counts are known as soon as
    the relation sampler has run (early in the forward pass: RelModel.rows_hook), so the tiny all-reduce is launched THERE,
    asynchronously, and has long finished when the loss is formed.  `get()` returns rows_rank / rows_global per term."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._local = self._tot = self._work = None

    def start(self, *row_counts):
        t = torch.tensor([float(c) for c in row_counts], dtype=torch.float64)
        if self.device.type == 'cuda':
            from lib.pytorch_misc import h2d
            t = h2d(t, self.device)
        self._local, self._tot, self._work = t, t.clone(), None
        if world_size() > 1:
            self._work = dist.all_reduce(self._tot, op=dist.ReduceOp.SUM, async_op=True)

    def get(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        return (self._local / self._tot.clamp(min=1.0)).float()


class GradBuckets(object):
    """Flatten-and-all-reduce of the gradients of `params` in fixed buckets."""

    def __init__(self, params, bucket_bytes=32 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, cur_bytes = [], [], 0
        for p in self.params:                      # parameters in registration order ~ reverse of backward order
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)
        self._work = []

    def start(self):
        """pack each bucket and launch its asynchronous all-reduce (SUM)"""
        self._work = []
        if world_size() == 1:
            return
        for i, bucket in enumerate(self.buckets):
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
            flat = torch.cat([g.reshape(-1) for g in grads])
            self._flat[i] = flat
            self._work.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """wait for the reductions and scatter the sums back into .grad"""
        if world_size() == 1:
            return
        for i, (bucket, work) in enumerate(zip(self.buckets, self._work)):
            work.wait()
            off = 0
            for p in bucket:
                n = p.numel()
                g = self._flat[i][off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n
            self._flat[i] = None
        self._work = []

    def all_reduce(self):
        self.start()
        self.finish()


class OverlappedGradReducer(object):
    """Bucketed gradient all-reduce OVERLAPPED with the backward pass.

    Buckets are formed in reverse registration order (the order in which backward produces gradients).  A
    post-accumulate hook on every parameter copies its fresh gradient into the bucket's persistent flat buffer; the
    moment a bucket and all buckets before it are complete its asynchronous all-reduce (SUM) is launched (same order
    on every rank), so the RCCL traffic of the early
    buckets (relation head, 0.9 GB of fc6/fc7 gradients) runs under the rest of the backward pass.  `finish()`
    launches whatever is left (parameters that received no gradient contribute zeros), waits, and re-points each
    `.grad` at its slice of the reduced flat buffer -- no unpack copy, and stable gradient addresses for the fused
    optimizer's pointer table.

    Collective UNITS (round 5).  A parameter larger than `split_bytes` (fc6: 411 MB; the flat 68 MB weight vector of the object
    context LSTM: "rows" of one element) sits in a bucket of its own whose flat buffer is reduced in row ranges of <= split_bytes,
    one all-reduce per range.  A producer that writes the gradient range
    by range (lib/hip_ops.py: the weight-gradient GEMM issued per row range, straight into `grad_view`) reports each range
    with `segment_done`, and that range's all-reduce starts while the GEMM of the next one runs -- the reduction of the
    step's largest gradient no longer waits for its last row.  A gradient that arrives whole (hook) releases all of its
    ranges at once.  Units are launched strictly in (bucket, range) order: the same sequence of collectives on every rank
    (`launch_log`, compared across ranks by tests/test_dist_model_gloo.py).

    Stream rule (two HIP streams run the backward of the two RelModel branches): every copy records an event; the
    launch makes the current stream wait for all events of the unit before handing the buffer to RCCL.
    With world_size 1 the object is inert.  CPU tensors + gloo work the same way (tests)."""

    def __init__(self, params, bucket_bytes=32 << 20, force=False, split_bytes=64 << 20):
        """force=True arms the hooks and the collectives even at world_size 1 (RCCL self-test on a single GPU: the
        all-reduce over one rank is the identity but runs through the same streams / events / buffers)"""
        self.params = [p for p in params if p.requires_grad]
        self.enabled = world_size() > 1 or (force and dist.is_initialized())
        # MOTIFS_GRAD_SYNC=post: no hooks, every bucket is reduced in finish() (after backward) -- an escape hatch
        self.overlap = os.environ.get('MOTIFS_GRAD_SYNC', 'overlap') != 'post'
        self.split_bytes = int(split_bytes)
        self.buckets, cur, cur_bytes = [], [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or nbytes > self.split_bytes):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
            if nbytes > self.split_bytes:                  # a bucket of its own, reduced in row ranges
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur)
        self._where = {}                                   # id(param) -> (bucket index, offset)
        self.units = []                                    # (bucket index, first element, end element), in collective order
        self._units_of = []                                # bucket index -> indices into self.units
        self._rows_of = {}                                 # id(param) of a split parameter -> [(row0, row1), ...]
        for bi, bucket in enumerate(self.buckets):
            off = 0
            for p in bucket:
                self._where[id(p)] = (bi, off)
                off += p.numel()
            first = len(self.units)
            p0 = bucket[0]
            if len(bucket) == 1 and p0.dim() >= 1 and p0.numel() * p0.element_size() > self.split_bytes and p0.shape[0] > 1:
                row = p0.numel() // p0.shape[0]
                step = max(1, self.split_bytes // (row * p0.element_size()))
                rows = [(r, min(r + step, p0.shape[0])) for r in range(0, p0.shape[0], step)]
                self._rows_of[id(p0)] = rows
                self.units += [(bi, r0 * row, r1 * row) for r0, r1 in rows]
            else:
                self.units.append((bi, 0, off))
            self._units_of.append(list(range(first, len(self.units))))
        self._flat = [None] * len(self.buckets)
        self._pending = [0] * len(self.buckets)
        self._seen = [set() for _ in self.buckets]
        self._ready = [False] * len(self.units)
        self._events = [[] for _ in self.units]
        self._work = [None] * len(self.units)
        self._armed = False
        self._next = 0
        self._handed = set()
        self.launch_log = []                                # unit indices in launch order of the last backward
        self.stats = {'copied_bytes': 0, 'in_place_bytes': 0}      # gradient bytes copied into / born inside the buckets
        self._exposed = []                                  # per finish(): (start, end) events or host times
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params] if self.enabled else []

    @property
    def bucket_mb(self):
        """size of every collective unit in MB, in launch order"""
        es = self.params[0].element_size() if self.params else 4
        return [round((hi - lo) * es / 2.0 ** 20, 2) for _, lo, hi in self.units]

    def grad_view(self, p):
        """a FRESH view of p's slot in its bucket's flat buffer, or None (not a bucketed parameter / reducer inert).  A
        producer that writes its gradient through this view (lib/hip_ops.py: the weight-gradient GEMM's `out=`) and returns
        it from backward lets autograd adopt the view as `.grad` -- the gradient is born inside the bucket and the hook has
        nothing to copy (0.9 of the 1.1 GB of gradients per step are fc6 / fc7 weight gradients)."""
        if not (self.enabled and self.overlap and self._armed):
            return None
        w = self._where.get(id(p))
        if p.grad is not None:                      # a kept .grad (zero_grad(set_to_none=False), gradient accumulation) may BE
            return None                             # this slot: writing the new gradient into it and then `grad += gw` would double it
        if w is None or id(p) in self._handed:      # a slot is handed out ONCE per backward: a parameter used twice gets its
            return None                             # second contribution as an ordinary tensor, which autograd adds on top
        self._handed.add(id(p))
        bi, off = w
        return self._buffer(bi)[off:off + p.numel()].view_as(p)

    def segments(self, p):
        """row ranges [(r0, r1), ...] in which a producer holding `grad_view(p)` should write the gradient (None: whole)"""
        return self._rows_of.get(id(p))

    def segment_done(self, p, i):
        """rows segments(p)[i] of the gradient view are written (enqueued on the current stream): that range may be reduced"""
        if not (self._armed and self.overlap):
            return
        bi, _ = self._where[id(p)]
        u = self._units_of[bi][i]
        if self._ready[u]:
            return
        if self._buffer(bi).is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self._events[u].append(ev)
        self._ready[u] = True
        self._advance()

    def _buffer(self, bi):
        if self._flat[bi] is None:
            ref = self.buckets[bi][0]
            self._flat[bi] = torch.zeros(sum(p.numel() for p in self.buckets[bi]), dtype=ref.dtype, device=ref.device)
        return self._flat[bi]

    def prepare(self):
        """call before every backward()"""
        if not self.enabled:
            return
        for bi, bucket in enumerate(self.buckets):
            self._pending[bi] = len(bucket)
            self._seen[bi] = set()
        for u in range(len(self.units)):
            self._ready[u] = False
            self._events[u] = []
            self._work[u] = None
        self._next = 0
        self._armed = True
        self._handed = set()
        self.launch_log = []
        from lib import hip_ops
        hip_ops.GRAD_SINK = self.grad_view          # producers may write weight gradients straight into the buckets
        hip_ops.GRAD_REDUCER = self                 # ... range by range for the parameters of segments()

    def _hook(self, p):
        if not self._armed or not self.overlap:
            return
        bi, off = self._where[id(p)]
        flat = self._buffer(bi)
        if p.grad.data_ptr() != flat.data_ptr() + off * flat.element_size():     # not already written in place (grad_view)
            flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
            self.stats['copied_bytes'] += p.numel() * p.element_size()
        else:
            self.stats['in_place_bytes'] += p.numel() * p.element_size()
        ev = None
        if flat.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
        self._seen[bi].add(id(p))
        self._pending[bi] -= 1
        for u in self._units_of[bi]:
            if ev is not None and not self._ready[u]:
                self._events[u].append(ev)
            if self._pending[bi] == 0:
                self._ready[u] = True
        self._advance()

    def _advance(self):
        # collectives must be issued in the same order on every rank: strictly by unit index (a unit that is complete
        # early waits for its predecessors)
        while self._next < len(self.units) and self._ready[self._next]:
            self._launch(self._next)
            self._next += 1

    def _launch(self, u):
        bi, lo, hi = self.units[u]
        flat = self._buffer(bi)
        if flat.is_cuda:
            cur = torch.cuda.current_stream()
            for ev in self._events[u]:
                cur.wait_event(ev)
        self.launch_log.append(u)
        self._work[u] = dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True)

    def finish(self):
        """call after backward(): reduce the stragglers, wait, re-point .grad at the reduced buffers"""
        if not self.enabled:
            return
        import time
        self._armed = False
        from lib import hip_ops
        hip_ops.GRAD_SINK = None
        hip_ops.GRAD_REDUCER = None
        cuda = any(f is not None and f.is_cuda for f in self._flat) or (self.params and self.params[0].is_cuda)
        if cuda:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record()
        else:
            t0 = time.perf_counter()
        for bi, bucket in enumerate(self.buckets):
            if all(self._ready[u] for u in self._units_of[bi]):
                continue
            flat = self._buffer(bi)
            for p in bucket:
                if id(p) not in self._seen[bi]:
                    _, off = self._where[id(p)]
                    if p.grad is not None and not self.overlap:
                        flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
                    elif not any(self._ready[u] for u in self._units_of[bi]):
                        flat[off:off + p.numel()].zero_()
            for u in self._units_of[bi]:
                self._ready[u] = True
        self._advance()
        assert self._next == len(self.units)
        for u in range(len(self.units)):
            self._work[u].wait()
        for bi, bucket in enumerate(self.buckets):
            flat = self._flat[bi]
            for p in bucket:
                _, off = self._where[id(p)]
                p.grad = flat[off:off + p.numel()].view_as(p)
        if cuda:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record()
        else:
            t1 = time.perf_counter()
        if len(self._exposed) < 4096:
            self._exposed.append((t0, t1))

    def exposed_ms(self, reset=True):
        """mean time per step between the end of backward (entry of finish()) and the completion of the last collective on
        the compute stream: the part of the gradient reduction that backward did not hide.  0.0 when inert."""
        if not self._exposed:
            return 0.0
        vals = []
        for a, b in self._exposed:
            if isinstance(a, float):
                vals.append(1e3 * (b - a))
            else:
                b.synchronize()
                vals.append(a.elapsed_time(b))
        if reset:
            self._exposed = []
        return sum(vals) / len(vals)

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def device_identity(device):
    """a string that names the physical device behind `device` (uuid when the runtime reports one, else PCI location + name)"""
    device = torch.device(device)
    if device.type != 'cuda':
        return 'cpu:pid%d' % os.getpid()
    pr = torch.cuda.get_device_properties(device)
    uuid = getattr(pr, 'uuid', None)
    if uuid is not None and str(uuid).strip('0-') != '':
        return 'uuid:%s' % uuid
    loc = '%s:%s:%s' % (getattr(pr, 'pci_domain_id', '?'), getattr(pr, 'pci_bus_id', '?'), getattr(pr, 'pci_device_id', '?'))
    return 'pci:%s:%s:local%d' % (loc, pr.name, device.index if device.index is not None else torch.cuda.current_device())


def scaling_diagnostics(reducer, device, ms_per_step_local):
    """What a first multi-GPU run needs in order to be readable without a second one (bench.py puts it into rank 0's line):
      ranks_seen             one identity string per rank, gathered THROUGH the process group (RCCL at N > 1): N distinct
                             devices, or the launch put two ranks on one GPU
      ms_per_step_per_rank   every rank's own wall-clock step time (the line's ms_per_step is their maximum)
      allreduce_exposed_ms   per rank: time per step between the end of backward and the completion of the last gradient
                             collective on the compute stream (what the overlap did not hide); 0 at N = 1
      bucket_mb              the collective units of a step in launch order (row ranges of <= 64 MB for fc6)
      collective_order_identical   every rank launched the same sequence of units in its last backward
    All ranks must call it (collectives inside)."""
    ident = device_identity(device)
    exposed = float(reducer.exposed_ms()) if reducer is not None else 0.0
    mine = {'id': ident, 'ms': float(ms_per_step_local), 'exposed': exposed,
            'order': list(reducer.launch_log) if reducer is not None else []}
    if world_size() > 1:
        allv = [None] * world_size()
        dist.all_gather_object(allv, mine)
    else:
        allv = [mine]
    return {'ranks_seen': [v['id'] for v in allv], 'distinct_devices': len({v['id'] for v in allv}),
            'ms_per_step_per_rank': [round(v['ms'], 3) for v in allv],
            'allreduce_exposed_ms': [round(v['exposed'], 3) for v in allv],
            'bucket_mb': reducer.bucket_mb if reducer is not None else [],
            'collective_order_identical': all(v['order'] == allv[0]['order'] for v in allv),
            'grad_bytes_in_place_frac': (reducer.stats['in_place_bytes'] /
                                         max(1, reducer.stats['in_place_bytes'] + reducer.stats['copied_bytes']))
                                        if (reducer is not None and reducer.enabled) else None}
