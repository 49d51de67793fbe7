"""dataloaders/h5lite.py -- the pure-Python HDF5 reader behind `load_graphs` when h5py is absent ("the real file
format is never opened by a test").  The fixtures tests/golden/vg_sgg_fixture*.h5 were written by the REAL library (h5py 3.3.0 /
libhdf5 1.10.6 of the image's conda environment, tests/golden/make_vg_h5.py): the first as the dataset's converter writes
VG-SGG.h5 (`create_dataset(name, data=...)`: contiguous), the second chunked with gzip + shuffle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'neural-motifs_amd'))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
FIXTURES = ('vg_sgg_fixture.h5', 'vg_sgg_fixture_chunked.h5')


def _expected():
    g = np.load(os.path.join(GOLDEN, 'vg_formats.npz'))
    arrays = {k[3:]: g[k] for k in g.files if k.startswith('in_')}
    arrays['boxes_512'] = (arrays['boxes_1024'] // 2).astype(np.int32)
    arrays['active_object_mask'] = arrays['labels'] > 0
    arrays['scores_f32'] = np.linspace(0.0, 1.0, arrays['labels'].shape[0], dtype=np.float32)[:, None]
    arrays['ids_i64'] = np.arange(arrays['split'].shape[0], dtype=np.int64) * 100003
    return g, arrays


@pytest.mark.parametrize('name', FIXTURES)
def test_every_dataset_of_the_real_hdf5_files_reads_back_exactly(name):
    from dataloaders import h5lite
    _, arrays = _expected()
    with h5lite.File(os.path.join(GOLDEN, name)) as f:
        assert sorted(f.keys()) == sorted(arrays) and len(f) == len(arrays) and 'labels' in f and 'nope' not in f
        for k, want in arrays.items():
            d = f[k]
            assert d.shape == want.shape and d.dtype == want.dtype and len(d) == want.shape[0], k
            got = d[:]
            assert got.dtype == want.dtype and np.array_equal(got, want), k
            np.testing.assert_array_equal(d[3:11], want[3:11])                    # h5py-style slicing
            np.testing.assert_array_equal(np.asarray(d), want)
        with pytest.raises(KeyError):
            f['missing']


@pytest.mark.parametrize('name', FIXTURES)
def test_load_graphs_on_the_hdf5_container_matches_the_reference(name, monkeypatch):
    """the reference's own load_graphs outputs (tests/golden/make_golden.py) from the .h5 file itself -- through h5lite, also
    where h5py is installed"""
    import builtins
    real_import = builtins.__import__

    def no_h5py(mod, *a, **k):
        if mod == 'h5py':
            raise ImportError('h5py hidden by the test')
        return real_import(mod, *a, **k)
    monkeypatch.setattr(builtins, '__import__', no_h5py)
    from dataloaders.visual_genome import load_graphs
    g, _ = _expected()
    path = os.path.join(GOLDEN, name)
    for ci in range(6):
        m, num_im, num_val, fer, fno = [int(v) for v in g['lg%d_args' % ci]]
        mask, boxes, classes, rels = load_graphs(path, ('train', 'val', 'test')[m], num_im, num_val_im=num_val,
                                                 filter_empty_rels=bool(fer), filter_non_overlap=bool(fno))
        np.testing.assert_array_equal(mask, g['lg%d_mask' % ci])
        counts = np.array([b.shape[0] for b in boxes] + [-1] + [r.shape[0] for r in rels])
        np.testing.assert_array_equal(counts, g['lg%d_counts' % ci])
        if boxes:
            np.testing.assert_array_equal(np.concatenate(boxes, 0), g['lg%d_boxes' % ci])
            np.testing.assert_array_equal(np.concatenate(classes, 0), g['lg%d_classes' % ci])
            np.testing.assert_array_equal(np.concatenate(rels, 0), g['lg%d_rels' % ci])


def test_what_is_not_covered_fails_loudly(tmp_path):
    from dataloaders import h5lite
    p = tmp_path / 'not.h5'
    p.write_bytes(b'definitely not hdf5' * 100)
    with pytest.raises(ValueError):
        h5lite.File(str(p))
    raw = bytearray(open(os.path.join(GOLDEN, FIXTURES[0]), 'rb').read())
    raw[8] = 2                                               # superblock version 2: a libver='latest' file
    q = tmp_path / 'v2.h5'
    q.write_bytes(bytes(raw))
    with pytest.raises(NotImplementedError):
        h5lite.File(str(q))
    with pytest.raises(ValueError):
        h5lite.File(os.path.join(GOLDEN, FIXTURES[0]), 'w')


def test_the_rest_of_the_covered_format_reads_back_exactly():
    """tests/golden/h5lite_types.h5 (same generator, real library): nested groups, a big-endian and a 64-bit float type, a 3-D and a
    scalar dataset, chunks without filters, the fletcher32 checksum, a compact dataset"""
    from dataloaders import h5lite
    exp = np.load(os.path.join(GOLDEN, 'h5lite_types_expected.npz'))
    with h5lite.File(os.path.join(GOLDEN, 'h5lite_types.h5')) as f:
        assert f.keys() == ['checksummed', 'compact', 'grp', 'plain_chunks', 'scalar']
        assert f['grp'].keys() == ['f64', 'sub'] and f['grp/sub'].keys() == ['i16_be', 'u8'] and 'grp/sub/u8' in f and 'grp/nope' not in f
        for k in exp.files:
            d = f[k.replace('__', '/')]
            got = d[()] if d.shape == () else d[:]
            assert d.shape == exp[k].shape and np.array_equal(np.asarray(got), exp[k]), k
        assert f['grp/sub/i16_be'].dtype == np.dtype('>i2') and f['grp/sub/i16_be'][:].dtype == np.dtype('int16')      # values come back in native order
        assert f['grp']['sub']['u8'].shape == (4, 3, 2)
