"""
Initial embedding vectors for class names (reference lib/word_vectors.py:17-36, loader :49-113).

The reference downloads GloVe-6B; there is no network here, so the loader takes what is on disk under `wv_dir`:
`<type>.<dim>d.pt` (the reference's cache: (token -> row dict, [n,dim] tensor, dim)) or `<type>.<dim>d.txt` (the GloVe
text format; parsed like the reference and cached as `.pt`).  With neither file every row keeps the N(0,1) draw the
reference itself falls back to for unknown tokens -- from a generator seeded by the class list, so two processes (ranks)
build identical tables.
"""
import array
import os
import zlib

import torch

from config import DATA_PATH


def load_word_vectors(root, wv_type, dim):
    """(wv_dict, wv_arr, wv_size) from `.pt` or `.txt` under root; None when neither exists (no download here)"""
    if isinstance(dim, int):
        dim = str(dim) + 'd'
    fname = os.path.join(root, wv_type + '.' + dim)
    if os.path.isfile(fname + '.pt'):
        return torch.load(fname + '.pt')
    if not os.path.isfile(fname + '.txt'):
        return None
    wv_tokens, wv_arr, wv_size = [], array.array('d'), None
    with open(fname + '.txt', 'rb') as f:
        for line in f:
            entries = line.strip().split(b' ')
            word, entries = entries[0], entries[1:]
            if wv_size is None:
                wv_size = len(entries)
            try:
                word = word.decode('utf-8')
            except UnicodeDecodeError:
                continue                                   # non-UTF8 token ignored, like the reference
            wv_arr.extend(float(x) for x in entries)
            wv_tokens.append(word)
    wv_dict = {word: i for i, word in enumerate(wv_tokens)}
    ret = (wv_dict, torch.tensor(wv_arr, dtype=torch.float32).view(-1, wv_size), wv_size)
    try:
        torch.save(ret, fname + '.pt')
    except OSError:
        pass
    return ret


def obj_edge_vectors(names, wv_type='glove.6B', wv_dir=DATA_PATH, wv_dim=300):
    seed = zlib.crc32(('|'.join(names) + '#%d' % wv_dim).encode()) & 0x7FFFFFFF
    gen = torch.Generator().manual_seed(seed)
    vectors = torch.randn(len(names), wv_dim, generator=gen)
    glove = load_word_vectors(wv_dir, wv_type, wv_dim)
    if glove is None:
        return vectors
    wv_dict, wv_arr, _ = glove
    for i, token in enumerate(names):
        idx = wv_dict.get(token, None)
        if idx is None:      # longest word of a multi-word class name, as the reference does
            idx = wv_dict.get(sorted(token.split(' '), key=lambda x: len(x), reverse=True)[0], None)
        if idx is not None:
            vectors[i] = wv_arr[idx]
    return vectors
