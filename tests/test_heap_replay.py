"""The device kernels' restatement of libstdc++'s heap (sr_livo_amd/csrc/srl_heap.h) against the real
std::priority_queue of optimize.cpp:355-363,394-404,411-422 (oracle side), CPU only.

When candidate distances tie, which tied points survive and in which order they are read out depends on the
heap's internal arrangement; the kernels replay the literal sequence with these routines, so they must agree
with libstdc++ on every input -- ties included."""
import numpy as np
import pytest

import sr_livo_amd as srl
from oracle import pyoracle as po


def test_known_tie_survivor():
    """K = 4, five candidates tied at the cut-off -> the 3rd visited survives."""
    d = np.array([1, 2, 2, 2, 2, 0.5, 2, 1.5]) * 0.125
    ref = po.heap_topk(d, 4)
    assert list(ref) == [5, 0, 7, 3]
    assert list(srl.heap_topk(d, 4)) == list(ref)


@pytest.mark.parametrize("K", [1, 2, 3, 4, 5, 7, 8, 16, 20, 31, 32])
def test_random_tie_heavy_sequences(K):
    rng = np.random.default_rng(100 + K)
    for trial in range(400):
        n = int(rng.integers(0, 120))
        levels = int(rng.integers(1, 12))                   # few distinct values: almost everything ties
        d = rng.integers(0, levels, size=n).astype(np.float64) * 0.25
        if trial % 3 == 0:
            d = np.sort(d)
        elif trial % 3 == 1:
            d = np.sort(d)[::-1].copy()
        assert list(srl.heap_topk(d, K)) == list(po.heap_topk(d, K)), (K, trial, d)


def test_tie_free_sequences_give_sorted_order():
    rng = np.random.default_rng(5)
    for _ in range(100):
        d = rng.permutation(600).astype(np.float64)
        got = srl.heap_topk(d, 20)
        assert list(got) == list(np.argsort(d)[:20])
        assert list(got) == list(po.heap_topk(d, 20))


def test_fewer_candidates_than_k_and_empty():
    assert len(srl.heap_topk(np.zeros(0), 20)) == 0
    d = np.array([3.0, 1.0, 2.0, 1.0, 3.0])
    assert list(srl.heap_topk(d, 20)) == list(po.heap_topk(d, 20))
    assert len(srl.heap_topk(d, 20)) == 5
