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


@pytest.mark.parametrize("n", [0, 1, 2, 10, 11, 12, 23, 24, 47, 48, 97, 98, 199, 1000, 5000, 14000, 60000])
def test_replay_equals_the_real_container(n):
    rng = np.random.default_rng(n)
    keys = np.unique(rng.integers(-300, 300, size=(int(n * 1.3) + 8, 3)).astype(np.int16), axis=0)
    rng.shuffle(keys, axis=0)
    keys = keys[:n]
    got = srl.tr1_order(keys)
    ref = real_order(keys)
    assert len(ref) == len(keys)
    assert np.array_equal(got, ref)


def test_adversarial_keys_one_bucket_and_negative_coordinates():
    # every key hashes to a multiple of 11 * 23 * 47 * 97 (the first bucket counts): long chains, many rehash moves
    rng = np.random.default_rng(7)
    keys = np.unique(np.column_stack([rng.integers(-3000, 3000, 4000), np.zeros(4000, int), np.zeros(4000, int)]).astype(np.int16), axis=0)
    rng.shuffle(keys, axis=0)
    assert np.array_equal(srl.tr1_order(keys), real_order(keys))
    keys = np.array([[-1, -1, -1], [0, 0, 0], [-32768, 32767, -1], [32767, -32768, 1], [-5, 7, -9]], np.int16)
    assert np.array_equal(srl.tr1_order(keys), real_order(keys))
