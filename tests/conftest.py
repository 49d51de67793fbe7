"""pytest configuration: import paths + the `gpu` marker.

`-m "not gpu"` : oracle vs the committed golden vectors, host logic, C-ABI symbol check.
`-m gpu`       : the parity tests proper (HIP path vs oracle, through the C-ABI).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'neural-motifs_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))


@pytest.fixture
def golden():
    return load_golden


@pytest.fixture(scope='session')
def so_path():
    """libmotifs_hip.so, (re)built in-tree if a source is newer (hipcc cross-compiles without a GPU)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('mh_build', os.path.join(PKG, 'csrc', 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build()
