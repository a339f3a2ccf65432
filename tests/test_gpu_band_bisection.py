"""The pair selection's threshold search works on per-lane minima clamped into [2^-15, 2) m^2 (nine bisection steps instead of
thirteen) and falls back to the full search when the K-th smallest minimum is not below 2 m^2.  Neighbour ids must
stay the reference's (optimize.cpp:394-404) at both ends of the band and across the fallback:
  * SPARSE maps -- one to three points per voxel, so that the 20th neighbour of a keypoint lies 1.2 ... 2.5 m away (squared
    distance on either side of 2 m^2, sometimes fewer than 20 candidates at all);
  * TINY distances -- voxels uploaded with 20 points inside a 4 mm ball and the keypoint in its middle: every squared distance
    is below 2^-15 m^2 (5.5 mm), the clamp's lower end;
  * mixed pairs -- a dense and a sparse keypoint selected together (the pair shares one bisection loop)."""
import numpy as np
import pytest

import sr_livo_amd as srl
from sr_livo_amd import capi
from test_gpu_eigen_stress import _voxelise

pytestmark = pytest.mark.gpu
INT_MAX = 2**31 - 1


def _pass(oracle_lib, oracle_backend, keys, counts, xyz, raw):
    q, t, t_last = np.array([1.0, 0, 0, 0]), np.zeros(3), np.array([0.0, 0.0, 30.0])
    m = oracle_lib.Map(oracle_backend)
    m.import_(keys, counts, xyz)
    o = m.build_plane_residuals(oracle_lib.default_opts(max_num_residuals=INT_MAX), raw, q, t, t_last)
    ctx = srl.Context(0)
    try:
        ctx.map_upload(keys, counts, xyz)
        ctx.sweep_upload(raw)
        ctx.set_taps(1)
        neq, rc = ctx.build_residuals(capi.make_frame(q, t, t_last), srl.default_opts(max_num_residuals=INT_MAX))
        ids, status, ncand = ctx.fetch_neighbors(K=20)
        ctx.set_taps(0)
    finally:
        ctx.close()
    return o, neq, ids, status, ncand


@pytest.mark.parametrize("layers", [1, 6])
def test_sparse_maps_cross_the_upper_end_of_the_band(oracle_lib, oracle_backend, layers):
    """layers = 1: a single sheet of voxels with three points each -- nine occupied voxels around a keypoint, 27 candidates, the
    pair selection (<= 12 voxels) with the 20th neighbour at 1.2 ... 1.7 m; layers = 6: all 27 voxels occupied with 1-3 points
    (the single-keypoint / general paths, many keypoints with more than 64 survivors)."""
    rng = np.random.default_rng(77)
    pts = []
    for ix in range(-12, 12):
        for iy in range(-12, 12):
            for iz in range(-(layers // 2), layers - layers // 2):
                for _ in range(3 if layers == 1 else int(rng.integers(1, 4))):
                    pts.append(np.array([ix, iy, iz]) + rng.uniform(0.02, 0.98, 3))
    pts = np.array(pts, np.float32)
    keys, counts, xyz = _voxelise(pts)
    zr = 0.45 if layers == 1 else 2.0
    raw = rng.uniform([-10, -10, 0.5 - zr if layers == 1 else -zr], [10, 10, 0.5 + zr if layers == 1 else zr], size=(4096, 3))
    o, neq, ids, status, ncand = _pass(oracle_lib, oracle_backend, keys, counts, xyz, raw)
    assert np.array_equal(ids, o["ids"]) and np.array_equal(status, o["status"])
    assert neq.sum_candidates == o["neq"].sum_candidates
    if layers == 1:
        assert neq.num_fallback == 0 and np.all(ncand >= 27)            # everybody stayed on the fast (pair) path (keys truncate toward zero: the voxels at coordinate 0 are twice as wide)
    # the scene does what it is meant to: 20th-neighbour squared distances on both sides of 2 m^2
    flat = xyz.reshape(-1, 3).astype(np.float64)
    full = ids.min(1) >= 0
    d2_k = ((flat[ids[full][:, 19]] - raw[full]) ** 2).sum(1)
    assert (d2_k > 2.0).sum() > 100 and (d2_k < 2.0).sum() > 100, (int((d2_k > 2.0).sum()), int((d2_k < 2.0).sum()))


def test_tiny_distances_sit_below_the_lower_end_of_the_band(oracle_lib, oracle_backend):
    rng = np.random.default_rng(78)
    clouds, queries = [], []
    for i in range(512):
        c = np.array([(i % 32) * 3 + 0.5, (i // 32) * 3 + 0.5, 0.5])
        P = c + rng.normal(0, 0.0012, size=(20, 3))                     # 20 points inside ~4 mm
        clouds.append(P)
        queries.append(c + rng.normal(0, 0.0005, 3))
        if i % 2:                                                       # every other keypoint: a sparse neighbour voxel as well
            clouds.append(c + np.array([1.0, 0, 0]) + rng.uniform(-0.3, 0.3, size=(3, 3)))
    keys, counts, xyz = _voxelise(np.concatenate(clouds).astype(np.float32))
    raw = np.array(queries)
    o, neq, ids, status, ncand = _pass(oracle_lib, oracle_backend, keys, counts, xyz, raw)
    assert np.array_equal(ids, o["ids"]) and np.array_equal(status, o["status"])
    flat = xyz.reshape(-1, 3).astype(np.float64)
    d2_k = ((flat[ids[:, 19]] - raw) ** 2).sum(1)
    assert np.mean(d2_k < 2.0 ** -15) > 0.9                               # (nearly) the whole selection happens under the clamp


def test_dense_and_sparse_keypoints_share_a_pair(oracle_lib, oracle_backend):
    rng = np.random.default_rng(79)
    dense = np.array([0.5, 0.5, 0.5]) + rng.uniform(-0.45, 0.45, size=(20, 3))
    pts = [dense]
    for ix in range(20, 44):
        for iy in range(-4, 4):
            for iz in range(-2, 2):
                pts.append((np.array([ix, iy, iz]) + rng.uniform(0.05, 0.95, 3))[None])
    keys, counts, xyz = _voxelise(np.concatenate(pts).astype(np.float32))
    n = 1024
    raw = np.empty((n, 3))
    raw[0::2] = np.array([0.5, 0.5, 0.5]) + rng.uniform(-0.3, 0.3, size=(n // 2, 3))          # dense: 20 candidates within 1 m
    raw[1::2] = rng.uniform([24, -2, -1], [40, 2, 1], size=(n // 2, 3))                       # sparse: one point per voxel
    o, neq, ids, status, ncand = _pass(oracle_lib, oracle_backend, keys, counts, xyz, raw)
    assert np.array_equal(ids, o["ids"]) and np.array_equal(status, o["status"])
    assert neq.sum_candidates == o["neq"].sum_candidates
