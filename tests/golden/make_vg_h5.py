#!/opt/conda/bin/python3.9
"""Writes tests/golden/vg_sgg_fixture.h5 (+ the chunked / deflate / shuffle variant vg_sgg_fixture_chunked.h5) WITH THE REAL
HDF5 LIBRARY: h5py 3.3.0 / libhdf5 1.10 of the image's conda environment (`/opt/conda/bin/python3.9 tests/golden/make_vg_h5.py`;
the interpreter the suite runs on has no h5py).  The arrays are the synthetic VG-SGG-shaped roidb of tests/golden/vg_formats.npz
(`in_*`: what tests/golden/make_golden.py fed the reference's own load_graphs, /root/reference dataloaders/visual_genome.py:264-330),
written the way the dataset's converter writes VG-SGG.h5 (`f.create_dataset(name, data=array)`: old-style root group, contiguous
layout, no filters), plus `boxes_512` / `active_object_mask` (a bool -> HDF5 enum) as in the real file.  The second file stores the
same arrays chunked with gzip + shuffle, the form a re-packed copy has.  dataloaders/h5lite.py (pure Python) must read both."""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    g = np.load(os.path.join(HERE, 'vg_formats.npz'))
    arrays = {k[3:]: g[k] for k in g.files if k.startswith('in_')}
    arrays['boxes_512'] = (arrays['boxes_1024'] // 2).astype(np.int32)
    arrays['active_object_mask'] = (arrays['labels'] > 0)
    arrays['scores_f32'] = np.linspace(0.0, 1.0, arrays['labels'].shape[0], dtype=np.float32)[:, None]      # a float dataset (the proposal file's kind)
    arrays['ids_i64'] = np.arange(arrays['split'].shape[0], dtype=np.int64) * 100003
    with h5py.File(os.path.join(HERE, 'vg_sgg_fixture.h5'), 'w') as f:
        for k, v in arrays.items():
            f.create_dataset(k, data=v)
    with h5py.File(os.path.join(HERE, 'vg_sgg_fixture_chunked.h5'), 'w') as f:
        for k, v in arrays.items():
            chunks = (max(1, v.shape[0] // 3),) + tuple(v.shape[1:])
            f.create_dataset(k, data=v, chunks=chunks, compression='gzip', compression_opts=4, shuffle=(v.dtype.itemsize > 1))
    # a third file for the parts of the format the two above do not touch: nested groups, big-endian and 64-bit float types, a compact
    # dataset, chunks without filters, the fletcher32 checksum, a 3-D and a scalar dataset
    rs = np.random.RandomState(7)
    extra = {'grp/sub/i16_be': rs.randint(-3000, 3000, (5, 7)).astype('>i2'), 'grp/f64': rs.randn(11, 3), 'grp/sub/u8': rs.randint(0, 255, (4, 3, 2)).astype(np.uint8),
             'plain_chunks': np.arange(1000, dtype=np.int32).reshape(100, 10), 'checksummed': rs.randn(64, 4).astype(np.float32),
             'scalar': np.float32(2.5), 'compact': np.arange(12, dtype=np.int64)}
    with h5py.File(os.path.join(HERE, 'h5lite_types.h5'), 'w') as f:
        for k in ('grp/sub/i16_be', 'grp/f64', 'grp/sub/u8', 'scalar'):
            f.create_dataset(k, data=extra[k])
        f.create_dataset('plain_chunks', data=extra['plain_chunks'], chunks=(32, 4))
        f.create_dataset('checksummed', data=extra['checksummed'], chunks=(16, 4), fletcher32=True)
        dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        dcpl.set_layout(h5py.h5d.COMPACT)
        space = h5py.h5s.create_simple((12,))
        h5py.h5d.create(f.id, b'compact', h5py.h5t.NATIVE_INT64, space, dcpl).write(h5py.h5s.ALL, h5py.h5s.ALL, extra['compact'])
    np.savez(os.path.join(HERE, 'h5lite_types_expected.npz'), **{k.replace('/', '__'): np.asarray(v) for k, v in extra.items()})
    with h5py.File(os.path.join(HERE, 'h5lite_types.h5'), 'r') as f:
        for k, v in extra.items():
            assert np.array_equal(f[k][()], v), k
    print('h5lite_types.h5', os.path.getsize(os.path.join(HERE, 'h5lite_types.h5')), 'bytes,', len(extra), 'datasets')
    for name in ('vg_sgg_fixture.h5', 'vg_sgg_fixture_chunked.h5'):
        with h5py.File(os.path.join(HERE, name), 'r') as f:
            assert sorted(f.keys()) == sorted(arrays)
            for k, v in arrays.items():
                assert np.array_equal(f[k][:], v), k
        print(name, os.path.getsize(os.path.join(HERE, name)), 'bytes,', len(arrays), 'datasets; h5py', h5py.__version__, 'hdf5', h5py.version.hdf5_version)


if __name__ == '__main__':
    main()
