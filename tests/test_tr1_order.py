"""The flat replay of std::tr1::unordered_map's bucket moves (sr_livo_amd/csrc/host/tr1_order.h, used by
srl_frame_select_keypoints to order the keypoints like gridSampling, utility.cpp:167-201) against the REAL container
(srl_grid_sampling runs subSampleFrame on std::tr1::unordered_map).  CPU only."""
import numpy as np
import pytest

import sr_livo_amd as srl


def real_order(keys):
    """one point per distinct voxel, in key order -> gridSampling's output order = the container's iteration order"""
    pts = keys.astype(np.float64) + np.where(keys >= 0, 0.5, -0.5)          # truncation toward zero maps back to the key
    return srl.grid_sampling(pts, 1.0)


@pytest.mark.parametrize("n", [0, 1, 2, 10, 11, 12, 23, 24, 47, 48, 97, 98, 199, 1000, 5000, 14000, 60000, 131072])
def test_replay_equals_the_real_container(n):
    rng = np.random.default_rng(n)
    keys = np.unique(rng.integers(-300, 300, size=(int(n * 1.3) + 8, 3)).astype(np.int16), axis=0)
    rng.shuffle(keys, axis=0)
    keys = keys[:n]
    got = srl.tr1_order(keys)
    ref = real_order(keys)
    assert len(ref) == len(keys)
    assert np.array_equal(got, ref)
    # the same order from the pairwise relation the device evaluates (host/tr1_relation.h)
    assert np.array_equal(srl.tr1_order(keys, by_relation=True), ref)


def test_adversarial_keys_one_bucket_and_negative_coordinates():
    # every key hashes to a multiple of 11 * 23 * 47 * 97 (the first bucket counts): long chains, many rehash moves
    rng = np.random.default_rng(7)
    keys = np.unique(np.column_stack([rng.integers(-3000, 3000, 4000), np.zeros(4000, int), np.zeros(4000, int)]).astype(np.int16), axis=0)
    rng.shuffle(keys, axis=0)
    assert np.array_equal(srl.tr1_order(keys), real_order(keys))
    assert np.array_equal(srl.tr1_order(keys, by_relation=True), real_order(keys))
    keys = np.array([[-1, -1, -1], [0, 0, 0], [-32768, 32767, -1], [32767, -32768, 1], [-5, 7, -9]], np.int16)
    assert np.array_equal(srl.tr1_order(keys), real_order(keys))
    assert np.array_equal(srl.tr1_order(keys, by_relation=True), real_order(keys))


@pytest.mark.parametrize("n", [9, 10, 11, 12, 13, 22, 23, 24, 25, 46, 47, 48, 49, 96, 97, 98, 99, 198, 199, 200, 201, 408, 409, 410, 411, 822, 823, 824, 825])
def test_relation_at_every_rehash_boundary(n):
    """element counts either side of the first rehashes (the insertion that triggers one is linked into the NEW table), keys that share
    buckets at several levels (multiples of small bucket counts along one axis)"""
    for seed, mult in ((0, 1), (1, 11), (2, 23 * 11), (3, 47)):
        rng = np.random.default_rng(1000 * n + seed)
        pop = np.arange(-32768 // mult + 1, 32767 // mult) * mult
        rows = n // len(pop) + 1                          # not enough multiples on one axis: more rows along y
        cand = np.array([(x, y, 0) for y in range(rows) for x in pop], np.int16)
        keys = cand[rng.permutation(len(cand))[:n]]
        keys = keys[np.sort(np.unique(keys, axis=0, return_index=True)[1])]
        ref = real_order(keys)
        assert np.array_equal(srl.tr1_order(keys), ref)
        assert np.array_equal(srl.tr1_order(keys, by_relation=True), ref)
