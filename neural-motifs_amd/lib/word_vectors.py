"""
Initial embedding vectors for class names (reference lib/word_vectors.py:17-36).

The reference downloads GloVe-6B; there is no network here, so: if `<DATA_PATH>/glove.6B.<dim>d.pt` exists it is
used exactly like the reference does ((dict, tensor, dim) triple), otherwise every row keeps the N(0,1) draw the
reference itself falls back to for unknown tokens -- from a generator seeded by the class list, so two processes
(ranks) build identical tables.
"""
import os
import zlib

import torch

from config import DATA_PATH


def _load_glove(wv_dir, wv_type, wv_dim):
    fname = os.path.join(wv_dir, '%s.%dd.pt' % (wv_type, wv_dim))
    if os.path.isfile(fname):
        return torch.load(fname)
    return None


def obj_edge_vectors(names, wv_type='glove.6B', wv_dir=DATA_PATH, wv_dim=300):
    seed = zlib.crc32(('|'.join(names) + '#%d' % wv_dim).encode()) & 0x7FFFFFFF
    gen = torch.Generator().manual_seed(seed)
    vectors = torch.randn(len(names), wv_dim, generator=gen)
    glove = _load_glove(wv_dir, wv_type, wv_dim)
    if glove is None:
        return vectors
    wv_dict, wv_arr, _ = glove
    for i, token in enumerate(names):
        idx = wv_dict.get(token, None)
        if idx is None:      # longest word of a multi-word class name, as the reference does
            idx = wv_dict.get(sorted(token.split(' '), key=len, reverse=True)[0], None)
        if idx is not None:
            vectors[i] = wv_arr[idx]
    return vectors
