"""Two-stream race, sequence level: the evaluation forward is run twice in the SAME host order -- once with the 'side' stream
being the main stream itself (one stream), once with a real second stream -- and every aten op (TorchDispatchMode) and every
lib._hip binding is recorded with its outputs kept alive.  After a device sync the two sequences are compared entry by entry:
the first entries whose outputs differ are where the two-stream run went wrong."""
import inspect
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'neural-motifs_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import test_gpu_configs as T
from dataloaders.synthetic import make_blob
from lib import _hip

SKIP = {'lib', 'ptr', 'stream', 'workspace', 'check_faults', 'check_skipped_steps', 'f32', 'i32', 'version_of', 'note_raw_update'}
seq = []
on = [False]


def tensors_of(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for x in o for t in tensors_of(x)]
    if hasattr(o, 'buf') and torch.is_tensor(getattr(o, 'buf')):
        return [o.buf]
    return []


def note(name, out):
    if on[0]:
        ts = [t for t in tensors_of(out) if t.is_cuda]
        if ts:
            seq.append((name, ts, 'side' if torch.cuda.current_stream().cuda_stream else 'main'))


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        note('aten.' + func.__name__, out)
        return out


def wrap(name, fn):
    def w(*a, **k):
        out = fn(*a, **k)
        note('_hip.' + name, out)
        return out
    return w


for name, fn in list(vars(_hip).items()):
    if name.startswith('_') or name in SKIP or not inspect.isfunction(fn) or fn.__module__ != _hip.__name__:
        continue
    setattr(_hip, name, wrap(name, fn))

ds, model, sd = T.build('predcls', 1234 + 100, 4)
model.cuda().eval()
model.load_state_dict(T.calibrated(sd))
model.overlap_streams = True
real_side = torch.cuda.Stream()


def run(blob, side):
    del seq[:]
    model._side_stream = side
    on[0] = True
    with torch.no_grad(), Rec():
        model[blob]
    on[0] = False
    torch.cuda.synchronize()
    return list(seq), model.last_eval_result.rel_dists.clone()


for idx in (1, 2):
    blob = make_blob(ds, [idx], is_train=False)
    run(blob, torch.cuda.current_stream())                       # warm caches
    a, rel_a = run(blob, torch.cuda.current_stream())
    a2, rel_a2 = run(blob, torch.cuda.current_stream())
    print('img %d: one stream, same host order: run-to-run logits diff %.3e, %d entries' % (idx, float((rel_a - rel_a2).abs().max()), len(a)), flush=True)
    for trial in range(3):
        b, rel_b = run(blob, real_side)
        bad = []
        if len(a) != len(b):
            print('sequence lengths differ: %d vs %d' % (len(a), len(b)))
        for i, ((na, ta, _), (nb, tb, sb)) in enumerate(zip(a, b)):
            if na != nb:
                bad.append('%d: op differs %s vs %s' % (i, na, nb))
                break
            for j, (x, y) in enumerate(zip(ta, tb)):
                if x.shape != y.shape:
                    bad.append('%d:%s[%d] shape %s vs %s' % (i, nb, j, tuple(x.shape), tuple(y.shape)))
                elif x.dtype == torch.uint8 and x.numel() > 4096:
                    n = int((x != y).sum())
                    if n > 512:                                     # image buffers: the 256-byte tail padding is uninitialised
                        bad.append('%d:%s[%d] (%s) %d bytes differ' % (i, nb, j, sb, n))
                elif not torch.equal(x, y):
                    xf, yf = x.double(), y.double()
                    bad.append('%d:%s[%d] (%s) max diff %.3e of %.3g, %d of %d elements, shape %s' % (
                        i, nb, j, sb, float((xf - yf).abs().max()), float(xf.abs().max()), int((x != y).sum()), x.numel(), tuple(x.shape)))
        print('img %d trial %d two streams: logits diff %.3e; first differing entries: %s' % (
            idx, trial, float((rel_a - rel_b).abs().max()), bad[:8] or 'none'), flush=True)
        if bad:
            i0 = int(bad[0].split(':')[0])
            print('   context: ' + ' | '.join('%d %s (%s)' % (i, b[i][0], b[i][2]) for i in range(max(0, i0 - 6), min(len(b), i0 + 3))), flush=True)
