"""
Small tensor / numpy helpers with the reference's names (lib/pytorch_misc.py) -- host-side glue on the path:
index arithmetic for packed sequences, per-image enumeration, gradient clipping, checkpoint restore.
"""
import os
import threading

import numpy as np
import torch
from torch import nn


def optimistic_restore(network, state_dict):
    """copy every same-named, same-shaped tensor; report the rest (reference :14-33). True iff nothing mismatched."""
    own = network.state_dict()
    ok = True
    for name, param in state_dict.items():
        if name not in own:
            print("Unexpected key {} in state_dict with size {}".format(name, tuple(param.size())))
            ok = False
        elif param.size() == own[name].size():
            own[name].copy_(param)
        else:
            print("Network has {} with size {}, ckpt has {}".format(name, tuple(own[name].size()), tuple(param.size())))
            ok = False
    missing = set(own.keys()) - set(state_dict.keys())
    if missing:
        print("We couldn't find {}".format(','.join(sorted(missing))))
        ok = False
    return ok


# ---- host mirrors ---------------------------------------------------------------------------------------------------
# The index work the host does in a step (relation sampling, the packing order of the LSTMs) needs the VALUES of a few
# small integer tensors -- the image index / class / relation list of the GT boxes -- that came from the host in the first
# place (dataloaders/blob.py).  Reading them back with .cpu() makes the host wait until the GPU has drained its queue (the
# whole trunk), after which every small launch of the step is issued with the GPU idle: 4-5 ms of a 27 ms SGCls step
# (profiles/r02_trace_gaps.txt).  A tensor can therefore carry a numpy copy of itself; derived tensors get theirs from
# the places that derive them.  Mirrors are only attached to tensors nobody writes to afterwards.
D2H_READS = [0]


def set_host(t, arr):
    """attach `arr` (numpy, same values as `t`) to the tensor object; returns t"""
    t._host_np = np.asarray(arr)
    return t


def has_host(t):
    return getattr(t, '_host_np', None) is not None


def host_np(t):
    """numpy values of `t`: the mirror when there is one (no device synchronisation), a device->host copy otherwise"""
    if not torch.is_tensor(t):
        return np.asarray(t)
    a = getattr(t, '_host_np', None)
    if a is not None:
        return a
    D2H_READS[0] += 1                  # counted: tests assert that a GT-box training step never gets here
    return t.detach().cpu().numpy()


class _PinnedRing(object):
    """Page-locked staging memory for the small host->device uploads of a step, allocated ONCE (two halves of `nbytes`/2).
    Uploads take consecutive 64-byte-aligned slices of the active half; when it is full the other half becomes active
    after the copies that last used it have completed (events recorded on every stream that read it -- thousands of
    uploads ago, so this never waits in practice).  No allocation in steady state: torch's caching pinned allocator cannot
    hand a block back while its copy is still queued, and with the host running ahead it kept calling hipHostMalloc --
    tens of milliseconds each, with the device idle (gpurun r02_c8: every other step took 76 ms instead of 25)."""

    def __init__(self, nbytes=8 << 20):
        self.half = nbytes // 2
        self.buf = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        self.active, self.pos = 0, 0
        self.streams = [set(), set()]          # streams that copied out of each half since it became active
        self.events = [[], []]
        self.lock = threading.Lock()

    def stage(self, t, device):
        with self.lock:                     # callers: the main thread and autograd engine threads (index uploads in backward)
            return self._stage(t, device)

    def _stage(self, t, device):
        n = t.numel() * t.element_size()
        if n == 0:
            return torch.empty(t.shape, dtype=t.dtype, device=device)
        nb = (n + 63) // 64 * 64
        if nb > self.half:
            return t.pin_memory().to(device, non_blocking=True)
        if self.pos + nb > self.half:
            h = self.active
            self.events[h] = []
            for st in self.streams[h]:
                ev = torch.cuda.Event()
                ev.record(st)
                self.events[h].append(ev)
            self.streams[h] = set()
            self.active, self.pos = 1 - h, 0
            for ev in self.events[self.active]:
                ev.synchronize()
            self.events[self.active] = []
        off = self.active * self.half + self.pos
        self.pos += nb
        slot = self.buf[off:off + n].view(t.dtype).view(t.shape)
        slot.copy_(t)
        self.streams[self.active].add(torch.cuda.current_stream(device))
        return slot.to(device, non_blocking=True)


_rings = {}
_rings_lock = threading.Lock()


def h2d(x, device):
    """numpy array / CPU tensor -> `device` WITHOUT stalling the host.  A copy from pageable memory makes the host wait
    until the stream has drained (the runtime stages it synchronously); nine such copies per training step -- sampler
    output, packing permutations, the optimizer's chunk table -- each waited for the whole queue (12 of 21 ms of host time
    per SGCls step, gpurun r02_c7).  Staging through page-locked memory (_PinnedRing) + non_blocking keeps the host
    running ahead; the copy itself stays ordered on the current stream."""
    t = torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x
    device = torch.device(device)
    if device.type != 'cuda':
        return t.to(device)
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    how = os.environ.get('MOTIFS_H2D', 'ring')       # A/B switch: 'pageable' = plain .to(), 'alloc' = torch's pinned allocator
    if how == 'pageable':
        return t.to(device)
    if how == 'alloc':
        return t.pin_memory().to(device, non_blocking=True)
    ring = _rings.get(device.index)
    if ring is None:
        with _rings_lock:
            ring = _rings.get(device.index)
            if ring is None:
                ring = _rings[device.index] = _PinnedRing()
    return ring.stage(t.contiguous(), device)


def with_next(iterable):
    """(item, following item or None) pairs of an iterable: what a loop needs to start work on the next batch
    (RelModel.detect_ahead_blob) while the current one is in flight"""
    it = iter(iterable)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, nxt
        cur = nxt
    yield cur, None


def with_ahead(iterable, distance=1):
    """(item, [items to start now]) pairs: a loop that keeps the detector stage of the next `distance` batches in flight
    (RelModel.detect_ahead_blob) starts items 1..distance with the first item and item i+distance with item i afterwards"""
    import collections
    it = iter(iterable)
    buf = collections.deque()
    for x in it:
        buf.append(x)
        if len(buf) == distance + 1:
            break
    first, end = True, object()
    while buf:
        cur = buf.popleft()
        if first:
            start, first = list(buf), False
        else:
            nxt = next(it, end)
            start = [] if nxt is end else [nxt]
            buf.extend(start)
        yield cur, start


def quiet_gc():
    """Keep Python's cyclic collector out of the training step.  A full (generation-2) collection walks every tracked
    object of the process -- modules, parameters, the dataset's index arrays, ~10^6 objects here -- for ~100 ms, during
    which the host enqueues nothing and the GPU drains its queue: one such pause inside a 20-step window is +5 ms / step
    (round 3's unexplained 77-100 ms single-gap stalls; profiles/r04_variance.jsonl: one 49 ms step in the one region with a
    gen-2 collection, none with the collector off, spread 0.01 %).  `gc.freeze()` moves everything alive NOW (the model,
    optimizer state, datasets -- built once, alive for the whole run) into the permanent generation that collections skip;
    the garbage of the steps themselves (autograd graphs are freed by reference counting) stays collectable and cheap.
    Call after the model, optimizer and data pipeline are built, and again at every epoch boundary (after the validation pass:
    models/train_rels.py, train_detector.py) -- objects frozen once are never reclaimed otherwise, and what validation allocates
    would bring the generation-2 pauses back."""
    import gc
    gc.unfreeze()        # a repeated call (epoch boundary): what was frozen and has died since becomes collectable again ...
    gc.collect()         # ... is collected here, outside any step ...
    gc.freeze()          # ... and what is alive now (incl. what an evaluation pass left behind) joins the permanent generation


def restore_rel_checkpoint(rel_model, ckpt, ckpt_name):
    """What models/train_rels.py:76-96 of the reference does with `-ckpt`: a relation-model checkpoint ('.../vgrel-N.tar')
    restores everything and resumes at its epoch; any other file is a DETECTOR checkpoint ('vg-faster-rcnn.tar',
    'vgdet/vg-N.tar'): it fills `rel_model.detector` and seeds the relation model's two copies of the VGG fc6 / fc7
    (`roi_fmap[1]`, `roi_fmap_obj`) from the detector's `roi_fmap`.  Returns the epoch to resume after (-1 = start)."""
    sd = ckpt['state_dict']
    if ckpt_name.split('-')[-2].split('/')[-1] == 'vgrel':
        print("Loading EVERYTHING")
        return ckpt['epoch'] if optimistic_restore(rel_model, sd) else -1
    optimistic_restore(rel_model.detector, sd)
    for dst in (rel_model.roi_fmap[1], rel_model.roi_fmap_obj):
        for idx in (0, 3):
            dst[idx].weight.data.copy_(sd['roi_fmap.%d.weight' % idx])
            dst[idx].bias.data.copy_(sd['roi_fmap.%d.bias' % idx])
    return -1


class Flattener(nn.Module):
    def forward(self, x):
        return x.reshape(x.size(0), -1)


def arange(base_tensor, n=None):
    n = base_tensor.size(0) if n is None else n
    return torch.arange(n, dtype=torch.long, device=base_tensor.device)


def to_onehot(vec, num_classes, fill=1000):
    """[n,num_classes] float tensor with +fill at vec[i] and -fill elsewhere (reference :110-125)"""
    out = torch.full((vec.size(0), num_classes), -float(fill), dtype=torch.float32, device=vec.device)
    out[torch.arange(vec.size(0), device=vec.device), vec.long()] = float(fill)
    return out


def gather_nd(x, index):
    """x [d0..d{n-1}, dim], index [num, n] -> rows x[index[i,0],...,index[i,n-1]]  (reference :255-275)"""
    nd = x.dim() - 1
    assert nd > 0 and index.dim() == 2 and index.size(1) == nd
    flat = index[:, nd - 1].clone()
    mult = x.size(nd - 1)
    for col in range(nd - 2, -1, -1):
        flat += index[:, col] * mult
        mult *= x.size(col)
    return x.reshape(-1, x.size(-1))[flat]


def enumerate_by_image(im_inds):
    """yield (image id, start, end) for every run of equal image indices (reference :278-287)"""
    arr = host_np(im_inds)
    if arr.shape[0] == 0:
        return
    start, cur = 0, int(arr[0])
    for i in range(1, arr.shape[0]):
        if arr[i] != cur:
            yield cur, start, i
            start, cur = i, int(arr[i])
    yield cur, start, arr.shape[0]


def diagonal_inds(tensor):
    assert tensor.dim() >= 2 and tensor.size(0) == tensor.size(1)
    n = tensor.size(0)
    return (n + 1) * torch.arange(n, dtype=torch.long, device=tensor.device)


def nonintersecting_2d_inds(x):
    rs = 1 - np.diag(np.ones(x, dtype=np.int32))
    return np.column_stack(np.where(rs))


def intersect_2d(x1, x2):
    """[m1,m2] bool: row i of x1 equals row j of x2"""
    if x1.shape[1] != x2.shape[1]:
        raise ValueError("Input arrays must have same #columns")
    return (x1[..., None] == x2.T[None, ...]).all(1)


def argsort_desc(scores):
    """indices (one row per element) that sort `scores` descending"""
    return np.column_stack(np.unravel_index(np.argsort(-scores.ravel()), scores.shape))


def unravel_index(index, dims):
    out, rem = [], index.clone()
    for d in dims[::-1]:
        out.append(rem % d)
        rem = rem // d
    return torch.stack(out[::-1], 1)


def de_chunkize(tensor, chunks):
    s = 0
    for c in chunks:
        yield tensor[s:s + c]
        s += c


def random_choose(tensor, num, rs=np.random):
    """`num` rows without replacement (numpy RNG like the reference, :347-363); `rs` makes it seedable"""
    if min(tensor.size(0), num) == tensor.size(0):
        return tensor
    idx = rs.choice(tensor.size(0), size=num, replace=False)
    return tensor[h2d(idx, tensor.device)].contiguous()


def transpose_packed_sequence_inds(lengths):
    """indices that turn a batch-major concatenation (sequences sorted by decreasing length) into time-major
    packed order, plus the batch size per timestep (reference :365-384)"""
    lengths = list(lengths)
    starts = np.cumsum([0] + lengths[:-1])
    inds, sizes = [], []
    alive = len(lengths)
    for t in range(lengths[0]):
        while alive > 1 and lengths[alive - 1] <= t:
            alive -= 1
        inds.append(starts[:alive] + t)
        sizes.append(alive)
    return np.concatenate(inds, 0), sizes


def clip_grad_norm(named_parameters, max_norm, clip=False, verbose=False):
    """global L2 norm over all gradients; scales them in place when `clip` (reference :416-455).
    One device reduction for the whole list instead of one host sync per parameter."""
    named_parameters = [(n, p) for n, p in named_parameters if p.grad is not None]
    max_norm = float(max_norm)
    if not named_parameters:
        return 0.0
    norms = torch.stack([p.grad.detach().norm(2) for _, p in named_parameters])
    total_norm = float(norms.pow(2).sum().sqrt().item())
    clip_coef = max_norm / (total_norm + 1e-6)
    if clip_coef < 1 and clip:
        for _, p in named_parameters:
            p.grad.detach().mul_(clip_coef)
    if verbose:
        print('---Total norm {:.3f} clip coef {:.3f}-----------------'.format(total_norm, clip_coef))
        for (name, p), nv in sorted(zip(named_parameters, norms.tolist()), key=lambda x: -x[1]):
            print("{:<50s}: {:.3f}, ({})".format(name, nv, tuple(p.size())))
        print('-------------------------------', flush=True)
    return total_norm


def print_para(model):
    rows, total = [], 0
    for name, p in model.named_parameters():
        total += p.numel()
        if 'bias' not in name.split('.')[-1]:
            rows.append((name, list(p.size()), p.numel(), p.requires_grad))
    rows.sort(key=lambda r: -r[2])
    lines = ["{:<50s}: {:<16s}({:8d}) ({})".format(n, '[{}]'.format(','.join(map(str, s))), k, 'grad' if g else '    ')
             for n, s, k, g in rows]
    return '\n {:.1f}M total parameters \n ----- \n \n{}'.format(total / 1e6, '\n'.join(lines))
