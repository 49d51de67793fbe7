"""
Fused tail of the training step on the GPU: global gradient-norm clipping + SGD(momentum, weight decay)
(reference: lib/pytorch_misc.py:416-455 `clip_grad_norm`, models/train_rels.py:57-72 `get_optim`, :143-150).

`FusedClipSGD` looks like a torch optimizer (param_groups with per-group 'lr', `zero_grad`, `step`, `state_dict`), so
`ReduceLROnPlateau` keeps working, but one `step(max_norm)` is two multi-tensor launches (norm) + one (update) over a
chunk table that lives on the device; the norm never travels to the host unless `last_total_norm()` is asked for.
Gradients must be dense fp32 and present for every parameter of the table (parameters whose .grad is None in a step
are skipped by rebuilding the table -- this only happens if the set of used parameters changes).
"""
import ctypes
import os

import numpy as np
import torch

from lib import _hip


class FusedClipSGD(torch.optim.Optimizer):
    def __init__(self, params, lr, momentum=0.9, weight_decay=0.0, overlap_next_forward=None):
        """overlap_next_forward: enqueue the step (norm + update: 20 B per parameter, HBM-bound, ~1.2 ms for MotifNet's 279 M
        trainable parameters) on the optimizer's OWN stream, behind the gradients, instead of the compute stream.  The
        next forward pass starts with the frozen detector trunk (matrix-core-bound, reads no trainable parameter):
        the two run side by side, and the compute stream waits for the update only where RelModel.forward leaves the
        detector stage (lib/_hip.py: wait_param_update).  Only for models whose first stage is frozen (the relation
        drivers: models/train_rels.py:50-52); zero_grad() / synchronize() / state_dict() wait as well.  Default: the
        environment variable MOTIFS_OPT_DEFER (unset = off)."""
        super(FusedClipSGD, self).__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        if overlap_next_forward is None:
            overlap_next_forward = os.environ.get('MOTIFS_OPT_DEFER', '0') == '1'
        self.overlap_next_forward = bool(overlap_next_forward)
        self._opt_stream = None
        self.meter_events = None          # a list here receives (start, end) timing events of every deferred step (bench.py)
        mom = {g['momentum'] for g in self.param_groups}
        wd = {g['weight_decay'] for g in self.param_groups}
        if len(mom) != 1 or len(wd) != 1:
            raise ValueError('FusedClipSGD needs one momentum / weight_decay for all groups (per-group lr is fine)')
        # The step is asynchronous end to end (no device->host read, no blocking copy), so nothing stops the host from
        # queueing launches far ahead of the GPU (8 ms of host time for a 20 ms step).  That is what keeps the GPU fed; the
        # optimizer only bounds it: every max_ahead // 2 steps it records an event and waits for the one recorded
        # max_ahead steps earlier, so the host is never more than max_ahead steps ahead (pinned staging, logging and
        # Ctrl-C stay responsive).  Measured on one MI355X (gpurun r02_c13, img/s of the SGCls step): unbounded 292.7,
        # bound 3 / 2 / 1 with an event EVERY step 283.5 / 281.6 / 268.8 -- the per-step event costs more than the wait,
        # hence the sparse events.  MOTIFS_MAX_AHEAD: -1 = unbounded, 0 = synchronise every step.
        # Round 5 re-measured the bound on the lighter step (gpurun r05_c6, one box, first 12 / last 8 of 20 timed steps after 5
        # warm-up steps): bound 8: 16.5-16.7 / 16.0 ms -- while the host is still racing ahead about every second step takes 17+ ms;
        # bound 4: 16.17 / 16.01; bound 2: 16.16 / 16.00 (365.8-368.2 -> 372.5 / 372.7 img/s).
        # Round 6 (gpurun r06_c23, c28-c31): the slow steps are a cluster of four or five +1.2 ms steps around step 9-13 of a PROCESS --
        # while the host is still gaining on the GPU -- whatever is metered, with or without matrix-core load in front of the
        # warm-up, with no device allocation in sight; bounds 2 / 3 / 4 / 8 all show it, bound 1 does not (three of three runs:
        # 437.7 / 438.2 / 433.3 img/s against 429.5 / 432.8 / 433.5 at bound 4 over 20 timed steps after 5 warm-up steps), and the
        # steady state is the same or better (60 steps: 438.7 / 438.8 against 437.7 / 436.5; recipe 409-414 against 390-402;
        # cfg3 / cfg4 equal).  The host needs 7-8 ms to enqueue a 13.7 ms step, so one step of lead keeps the GPU fed.  Default 1.
        self.max_ahead = int(os.environ.get('MOTIFS_MAX_AHEAD', '1'))
        self._done_events = []
        self._table = None
        self._table_key = None
        self._steps = 0
        self._sumsq = None
        self._partial = None

    # -- chunk table -----------------------------------------------------------------------------
    def _build_table(self):
        """(re)build the device chunk table when any pointer / lr changed (cheap check per step; numpy build)"""
        chunk = _hip.lib().mh_opt_chunk_elems()
        key, dev = [], None
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or not p.is_contiguous():
                    raise _hip.HipKernelError('FusedClipSGD needs contiguous fp32 parameters and gradients')
                st = self.state[p]
                if 'momentum_buffer' not in st:
                    st['momentum_buffer'] = torch.zeros_like(p)
                    st['fresh'] = True
                dev = p.device
                key.append((p.data_ptr(), p.grad.data_ptr(), st['momentum_buffer'].data_ptr(), p.numel(),
                            float(group['lr'])))
        key = tuple(key)
        if key != self._table_key:
            # per-parameter records (32 B each) -> device chunk table, expanded by a kernel that receives the records in
            # its arguments: no host->device copy (autograd hands out new .grad tensors, i.e. a new key, every step)
            rec = np.dtype([('p', '<u8'), ('g', '<u8'), ('buf', '<u8'), ('n', '<i4'), ('lr', '<f4')])
            prm = np.empty(len(key), dtype=rec)
            for i, (pp, gp, bp, n, lr) in enumerate(key):
                prm[i] = (pp, gp, bp, n, lr)
            nchunks = int(sum((n + chunk - 1) // chunk for _, _, _, n, _ in key))
            if dev is not None:
                if self._table is None or self._nchunks != nchunks or self._table.device != dev:
                    self._table = torch.empty(max(nchunks, 1) * rec.itemsize, dtype=torch.uint8, device=dev)
                    self._partial = torch.empty(max(nchunks, 1), dtype=torch.float32, device=dev)
                if self._sumsq is None:
                    self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
                _hip._check(_hip.lib().mh_opt_build_chunks(prm.ctypes.data_as(ctypes.c_void_p), len(key),
                                                           ctypes.c_void_p(self._table.data_ptr()), nchunks, _hip.stream()),
                            'mh_opt_build_chunks')
            self._nchunks = nchunks
            self._table_key = key
        return self._nchunks

    def last_total_norm(self):
        """host value of the gradient norm measured by the last step (forces a sync; for logging only)"""
        return float(self._sumsq.sqrt().item()) if self._sumsq is not None else 0.0

    def zero_grad(self, set_to_none=True):
        # the gradients a deferred step is still reading must not be released (or overwritten) under it
        if torch.cuda.is_available():
            _hip.wait_param_update()
        return super(FusedClipSGD, self).zero_grad(set_to_none=set_to_none)

    def synchronize(self):
        """make the current stream wait for a deferred step (before parameters are read outside RelModel.forward:
        checkpoints, evaluation code that bypasses it)"""
        if torch.cuda.is_available():
            _hip.wait_param_update()

    def state_dict(self):
        self.synchronize()
        return super(FusedClipSGD, self).state_dict()

    @torch.no_grad()
    def step(self, max_norm=0.0, closure=None):
        if not (self.overlap_next_forward and torch.cuda.is_available()):
            return self._step(max_norm)
        dev = None
        for grp in self.param_groups:
            for p in grp['params']:
                if p.grad is not None:
                    dev = p.device
                    break
            if dev is not None:
                break
        if dev is None or dev.type != 'cuda':
            return self._step(max_norm)
        main = torch.cuda.current_stream(dev)
        if self._opt_stream is None:
            self._opt_stream = torch.cuda.Stream(device=dev)
        side = self._opt_stream
        _hip.wait_param_update()                   # (a previous deferred step nobody waited for: keep the order on `main` too)
        side.wait_stream(main)                     # every gradient of this backward pass is complete
        with torch.cuda.stream(side):
            timed = self.meter_events is not None
            if timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(side)
            out = self._step(max_norm, bound_stream=main)
            ev = torch.cuda.Event(enable_timing=timed)
            ev.record(side)
            if timed:
                self.meter_events.append((e0, ev))
        _hip.set_pending_param_update(ev)
        return out

    @torch.no_grad()
    def _step(self, max_norm=0.0, bound_stream=None):
        # Host reads, no synchronisation: they see launches that have COMPLETED, i.e. steps the host enqueued up to
        # max_ahead steps ago.  What protects the weights of the steps in between is on the device: a timed-out
        # persistent launch NaN-poisons its outputs, the gradients and their norm become NaN, and multi_sgd_kernel
        # skips the update and counts it (csrc/optim.hip).  Here the run is stopped as soon as either is visible.
        _hip.check_faults()
        _hip.check_skipped_steps()
        n = self._build_table()
        if n == 0:
            return None
        L = _hip.lib()
        g0 = self.param_groups[0]
        fresh = all(self.state[p].get('fresh', False) for grp in self.param_groups for p in grp['params']
                    if p.grad is not None)
        some_fresh = any(self.state[p].get('fresh', False) for grp in self.param_groups for p in grp['params']
                         if p.grad is not None)
        if some_fresh and not fresh:
            raise _hip.HipKernelError('parameters joined the optimizer after the first step: not supported')
        stream = _hip.stream()
        tptr = ctypes.c_void_p(self._table.data_ptr())
        sumsq_ptr = ctypes.c_void_p(0)
        if max_norm and max_norm > 0:
            _hip._check(L.mh_multi_sumsq(tptr, n, _hip.f32(self._partial), _hip.f32(self._sumsq), stream),
                        'mh_multi_sumsq')
            sumsq_ptr = _hip.f32(self._sumsq)
        _hip._check(L.mh_multi_sgd_step(tptr, n, sumsq_ptr, ctypes.c_float(float(max_norm or 0.0)),
                                        ctypes.c_float(g0['momentum']), ctypes.c_float(g0['weight_decay']),
                                        ctypes.c_int(1 if fresh else 0), stream), 'mh_multi_sgd_step')
        for grp in self.param_groups:
            for p in grp['params']:
                if p.grad is not None:
                    _hip.note_raw_update(p)          # the update went through raw pointers: torch's _version is unchanged
                    if fresh:
                        self.state[p]['fresh'] = False
        self._steps += 1
        if self.max_ahead >= 0 and self._table.is_cuda:
            period = max(1, self.max_ahead // 2)
            if self._steps % period == 0:
                if not self._done_events:
                    self._done_events = [torch.cuda.Event() for _ in range(2 if self.max_ahead > 1 else 1)]
                ev = self._done_events[(self._steps // period) % len(self._done_events)]
                ev.synchronize()                 # recorded max_ahead steps ago (a never-recorded event returns at once)
                ev.record(bound_stream) if bound_stream is not None else ev.record()
                if self.max_ahead == 0:
                    ev.synchronize()
        return None
