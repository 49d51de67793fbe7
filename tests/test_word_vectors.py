"""GloVe text-format loading and class-name lookup (lib/word_vectors.py) pinned to the reference's own
load_word_vectors / obj_edge_vectors on a tiny GloVe-format file (tests/golden/glove.npz)."""
import os

import numpy as np
import torch


def test_glove_txt_loading_and_lookup_match_the_reference(golden, tmp_path):
    from lib.word_vectors import load_word_vectors, obj_edge_vectors
    g = golden('glove')
    with open(os.path.join(str(tmp_path), 'glove.tiny.6d.txt'), 'wb') as f:
        f.write(g['txt'].tobytes())
    wv_dict, wv_arr, wv_size = load_word_vectors(str(tmp_path), 'glove.tiny', 6)
    assert wv_size == 6
    assert [t for t, _ in sorted(wv_dict.items(), key=lambda kv: kv[1])] == list(g['tokens'])
    np.testing.assert_array_equal(wv_arr.numpy(), g['arr'])
    assert os.path.isfile(os.path.join(str(tmp_path), 'glove.tiny.6d.pt'))       # cached like the reference
    names = list(g['names'])
    vec = obj_edge_vectors(names, wv_type='glove.tiny', wv_dir=str(tmp_path), wv_dim=6)
    known = g['known'].astype(bool)
    assert known.sum() >= 4 and (~known).sum() >= 2
    np.testing.assert_array_equal(vec.numpy()[known], g['vectors'][known])        # found tokens / longest-word fallback
    # unknown names keep a random N(0,1) row (the draw itself is implementation-specific), identical across calls
    vec2 = obj_edge_vectors(names, wv_type='glove.tiny', wv_dir=str(tmp_path), wv_dim=6)
    assert torch.equal(vec, vec2)
    assert not np.allclose(vec.numpy()[~known], 0)


def test_no_glove_file_gives_deterministic_random_vectors(tmp_path):
    from lib.word_vectors import obj_edge_vectors
    a = obj_edge_vectors(['x', 'y z'], wv_dir=str(tmp_path), wv_dim=8)
    b = obj_edge_vectors(['x', 'y z'], wv_dir=str(tmp_path), wv_dim=8)
    assert a.shape == (2, 8) and torch.equal(a, b)
