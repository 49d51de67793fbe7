"""Index model of nms_sweep_kernel (neural-motifs_amd/csrc/exact_ops.hip): the thread -> (row, column) mapping of the prefetched
chunks, their clamping at the segment's edge, the chunk order beyond 96 columns and the diagonal staging, replayed in numpy on
random suppression matrices and compared with the plain greedy sweep (/root/reference lib/fpn/nms/src/cuda/nms_kernel.cu:96-131).
The kernel itself is checked bit-exactly on the GPU (tests/test_gpu_ops.py::test_nms_bit_exact); this test pins the part of it
that is pure index arithmetic and runs without one."""
import numpy as np
import pytest

STAGE, P, T = 32, 4, 6          # kSweepStage, kSweepP, kSweepT


def random_mask(rs, n, density):
    """[n, cb] uint64 words as nms_mask_kernel writes them: bit j of word (i, c) set = box i suppresses box 64c + j, only for
    64c + j > i; words LEFT of the diagonal are never written (poisoned here: the sweep must not read them)."""
    cb = (n + 63) // 64
    bits = np.triu(rs.random_sample((n, cb * 64)) < density, k=1)
    bits[:, n:] = False
    words = (bits.reshape(n, cb, 64).astype(np.uint64) << np.arange(64, dtype=np.uint64)).sum(-1, dtype=np.uint64)
    poison = np.arange(cb)[None, :] < (np.arange(n) // 64)[:, None]
    return np.where(poison, np.uint64(0xdeadbeefdeadbeef), words), bits


def greedy(bits, n):
    removed = np.zeros(bits.shape[1], bool)
    keep = []
    for i in range(n):
        if not removed[i]:
            keep.append(i)
            removed |= bits[i]
    return keep


def sweep_model(mask, n):
    cb = (n + 63) // 64
    tid = np.arange(256)
    rg, cl = tid >> 4, tid & 15
    removed = np.zeros(cb, np.uint64)
    keep = []

    def fetch(r, jb):
        rows = np.minimum(r * 64 + rg[:, None] + 16 * np.arange(P)[None, :], n - 1)            # [256, P]
        cols = np.minimum(jb + cl[:, None] + 16 * np.arange(T)[None, :], cb - 1)               # [256, T]
        assert (cols[:, None, :] >= (rows // 64)[:, :, None]).all(), 'a read left of the diagonal'
        return mask[rows[:, :, None], cols[:, None, :]]                                        # [256, P, T]

    def fold(w, kept, jb):
        take = ((kept >> (rg[:, None] + 16 * np.arange(P)[None, :]).astype(np.uint64)) & np.uint64(1)).astype(bool)
        acc = np.bitwise_or.reduce(np.where(take[:, :, None], w, np.uint64(0)), axis=1)        # [256, T]
        j = jb + cl[:, None] + 16 * np.arange(T)[None, :]
        for th, t in zip(*np.nonzero((acc != 0) & (j < cb))):
            removed[j[th, t]] |= acc[th, t]

    regs = fetch(0, 1)
    diag = None
    for r in range(cb):
        if r % STAGE == 0:
            rows = np.minimum(r * 64 + np.arange(STAGE * 64), n - 1)
            diag = np.where(r * 64 + np.arange(STAGE * 64) < n, mask[rows, rows >> 6], np.uint64(0))
        d = diag[(r % STAGE) * 64:(r % STAGE) * 64 + 64]
        rows_here = min(n - r * 64, 64)
        alive = int(~removed[r]) & ((1 << rows_here) - 1)
        kept = 0
        while alive:
            i = (alive & -alive).bit_length() - 1
            kept |= 1 << i
            alive &= ~(int(d[i]) | (1 << i))
        keep += [r * 64 + i for i in range(64) if (kept >> i) & 1]
        jb = r + 1
        while True:
            if kept:
                fold(regs, np.uint64(kept), jb)
            jb += 16 * T
            if kept and jb < cb:
                regs = fetch(r, jb)
            elif r + 1 < cb:
                regs = fetch(r + 1, r + 2)
            if not (kept and jb < cb):
                break
    return keep


@pytest.mark.parametrize('n,density', [(1, 0.5), (63, 0.05), (64, 0.05), (65, 0.02), (200, 0.01), (2200, 0.002), (6300, 0.0005),
                                       (6300, 0.02)])
def test_the_sweep_index_model_equals_the_greedy_sweep(n, density):
    rs = np.random.RandomState(n)
    mask, bits = random_mask(rs, n, density)
    assert sweep_model(mask, n) == greedy(bits, n)
