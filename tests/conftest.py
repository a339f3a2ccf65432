import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "golden_small.npz")
    return dict(np.load(path, allow_pickle=False))


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import pyoracle as po
    return po


@pytest.fixture(scope="session")
def oracle_backend():
    from oracle import pyoracle as po
    return "tsl" if os.path.exists(po.LIB_TSL) else "plain"


@pytest.fixture(scope="session")
def small_scene(oracle_lib, oracle_backend):
    """Oracle map (~24k pts) + 2048-keypoint sweep shared by the CPU tests."""
    from sr_livo_amd import synth
    pts, L = synth.map_candidates(777, 30_000)
    m = oracle_lib.Map(oracle_backend)
    m.add_points(pts)
    sweep = synth.make_sweep(778, 2048, L)
    return dict(map=m, sweep=sweep, L=L, candidates=pts)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.max(np.abs(a - b)) if a.size else 0.0
    s = max(np.max(np.abs(b)) if b.size else 0.0, 1e-300)
    return d / s
