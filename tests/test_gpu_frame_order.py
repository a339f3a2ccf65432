"""Keypoint ORDER computed on the device (srl_frame_select_keypoints; sr_livo_amd/csrc/host/tr1_relation.h: the iteration order of
gridSampling's std::tr1::unordered_map, utility.cpp:167-201, as a pairwise relation) against the host replay of the container's moves
(csrc/host/tr1_order.h) and against the oracle's gridSampling (a real std::tr1::unordered_map): the same permutation, the same resident
sweep, on random frames, at the rehash boundaries, with adversarial keys (an overfull bucket falls back to the host replay) and for
frames beyond the one-launch scan (two launches there, still on the device)."""
import numpy as np
import pytest

import sr_livo_amd as srl
from sr_livo_amd import capi, synth

pytestmark = pytest.mark.gpu

INT_MAX = 2**31 - 1
Q_ID = np.array([1.0, 0.0, 0.0, 0.0])
T0 = np.zeros(3)


def points_of_keys(keys):
    """one point per voxel key (sample size 1, identity pose): truncation toward zero maps the point back to its key"""
    keys = np.asarray(keys, dtype=np.int64)
    return keys.astype(np.float64) + np.where(keys >= 0, 0.5, -0.5)


def both_orders(ctx, raw, q, t, size, R_il=None, t_il=None):
    ctx.set_frame_order_mode(0)
    ctx.frame_upload(raw)
    dev = ctx.frame_select_keypoints(q, t, size, R_il, t_il)
    used = ctx.frame_order_used()
    ctx.set_frame_order_mode(1)
    ctx.frame_upload(raw)
    host = ctx.frame_select_keypoints(q, t, size, R_il, t_il)
    assert ctx.frame_order_used() == 2
    ctx.set_frame_order_mode(0)
    return dev, host, used


@pytest.mark.parametrize("n_points,size", [(1, 1.0), (7, 1.0), (300, 0.8), (5_000, 0.5), (24_000, 1.0), (24_000, 0.3), (65_536, 1.5), (120_000, 0.6)])
def test_device_order_equals_host_replay_and_gridsampling(oracle_lib, oracle_backend, n_points, size):
    rng = np.random.default_rng(n_points)
    raw = rng.normal(size=(n_points, 3)) * np.array([25.0, 25.0, 4.0])
    q = synth.quat_from_rotvec([0.03, -0.02, 0.4]) * 1.0003          # un-normalised: transformPoint uses q as is
    t = np.array([3.0, -2.0, 0.5])
    R_il = synth.quat_to_rot(synth.quat_from_rotvec([0.02, 0.01, -0.04])); t_il = np.array([0.05, 0.02, -0.03])
    ctx = srl.Context(0)
    try:
        dev, host, used = both_orders(ctx, raw, q, t, size, R_il, t_il)
        assert used == 1, "the frame should have been ordered on the device"
        world = oracle_lib.transform_points(raw, q, t, R_il, t_il, backend=oracle_backend)
        want = oracle_lib.grid_sampling(world, size, backend=oracle_backend)
        assert np.array_equal(host, want)
        assert np.array_equal(dev, want), "device order differs from gridSampling"
    finally:
        ctx.close()


@pytest.mark.parametrize("n", [10, 11, 12, 23, 24, 47, 48, 97, 98, 199, 200, 409, 410, 823, 824, 1741, 1742, 3739, 3740, 7517, 7518, 15173, 15174])
def test_device_order_at_rehash_boundaries(oracle_lib, oracle_backend, n):
    """voxel counts either side of every rehash of the container (the insertion that triggers one is linked into the new table)"""
    rng = np.random.default_rng(n)
    keys = np.unique(rng.integers(-200, 200, size=(int(n * 1.4) + 8, 3)), axis=0)
    rng.shuffle(keys, axis=0)
    keys = keys[:n]
    assert len(keys) == n
    raw = points_of_keys(keys)
    # every voxel twice (the second point must lose against the first), in a shuffled tail
    raw = np.vstack([raw, raw[rng.permutation(n)] + 0.01])
    ctx = srl.Context(0)
    try:
        dev, host, used = both_orders(ctx, raw, Q_ID, T0, 1.0)
        want = oracle_lib.grid_sampling(raw, 1.0, backend=oracle_backend)
        assert used == 1 and len(want) == n
        assert np.array_equal(dev, want) and np.array_equal(host, want)
    finally:
        ctx.close()


def test_shared_buckets_and_overfull_bucket_fall_back(oracle_lib, oracle_backend):
    ctx = srl.Context(0)
    try:
        # (a) keys that share buckets at several levels (multiples of 11 * 23 along x): chains of a few voxels, ranked on the device
        rng = np.random.default_rng(5)
        xs = rng.choice(np.arange(-120, 120), size=150, replace=False) * 253
        keys = np.column_stack([xs, rng.integers(-2, 3, 150), np.zeros(150, int)])
        raw = points_of_keys(keys)
        dev, host, used = both_orders(ctx, raw, Q_ID, T0, 1.0)
        want = oracle_lib.grid_sampling(raw, 1.0, backend=oracle_backend)
        assert np.array_equal(dev, want) and np.array_equal(host, want)
        # (b) 100 voxels -> 199 buckets at the end; x = 199 k: every voxel in bucket 0 -> more than the device ranks in place:
        #     the selection must notice and run the host replay (order_used == 3), same result
        xs = rng.choice(np.arange(-160, 160), size=100, replace=False) * 199
        keys = np.column_stack([xs, np.zeros(100, int), np.zeros(100, int)])
        raw = points_of_keys(keys)
        dev, host, used = both_orders(ctx, raw, Q_ID, T0, 1.0)
        want = oracle_lib.grid_sampling(raw, 1.0, backend=oracle_backend)
        assert used == 3, "an overfull bucket must send the frame to the host replay"
        assert np.array_equal(dev, want) and np.array_equal(host, want)
        # ... and the next (ordinary) frame is ordered on the device again: the overflow mark does not stick
        raw = np.random.default_rng(6).normal(size=(4000, 3)) * 20.0
        dev, host, used = both_orders(ctx, raw, Q_ID, T0, 1.0)
        assert used == 1 and np.array_equal(dev, host)
    finally:
        ctx.close()


@pytest.mark.parametrize("n_points,size", [(140_000, 0.7), (262_144, 0.5), (262_144, 0.05), (600_000, 0.4)])
def test_frames_beyond_the_one_launch_scan_are_ordered_on_the_device_too(oracle_lib, oracle_backend, n_points, size):
    """a frame of more than 131 072 points (BASELINE config 4's sweep has 262 144) used to fall back to the host replay of
    the container and to the library's scan.  The scans run in two launches there (tile sums, then the same kernel summing the sums in
    front of its tile): same device chain, same order as gridSampling's std::tr1::unordered_map -- also where nearly every point is a voxel
    of its own (size 0.05: ~260 000 voxels, a bucket table of ~410 000)."""
    raw = np.random.default_rng(9 + n_points).normal(size=(n_points, 3)) * np.array([40.0, 40.0, 5.0])
    ctx = srl.Context(0)
    try:
        ctx.frame_upload(raw)
        got = ctx.frame_select_keypoints(Q_ID, T0, size)
        assert ctx.frame_order_used() == 1, "the frame should have been ordered on the device"
        want = oracle_lib.grid_sampling(raw, size, backend=oracle_backend)
        assert len(got) == len(want) and np.array_equal(got, want)
        if n_points == 262_144 and size == 0.5:
            ctx.set_frame_order_mode(1)                        # ... and the host replay (still there for bucket overflows) agrees
            ctx.frame_upload(raw)
            host = ctx.frame_select_keypoints(Q_ID, T0, size)
            assert ctx.frame_order_used() == 2 and np.array_equal(host, want)
    finally:
        ctx.close()


def test_resident_sweep_of_the_device_order_gives_the_same_normal_equations():
    """the device ordering also gathers the raw points into the resident sweep: the pass behind it must produce the normal equations of
    the host-ordered selection and of uploading those keypoints, bit for bit -- also when selections of different sizes alternate"""
    pts, L = synth.map_candidates(2201, 60_000)
    sw = synth.make_sweep(2202, 20_000, L)
    sw2 = synth.make_sweep(2203, 9_000, L)
    opts = srl.default_opts(max_num_residuals=INT_MAX)
    ctx = srl.Context(0); ctx2 = srl.Context(0)
    try:
        for c in (ctx, ctx2):
            c.map_insert(pts)
        for s, size in ((sw, 0.08), (sw2, 0.05), (sw, 0.12), (sw2, 0.1)):
            raw, q, t = s["raw"], s["q_pred"], s["t_pred"]
            f = capi.make_frame(q, t, s["t_last"])
            ctx.set_frame_order_mode(0)
            ctx.frame_upload(raw)
            kd = ctx.frame_select_keypoints(q, t, size)
            assert ctx.frame_order_used() == 1
            n_dev, _ = ctx.build_residuals(f, opts)
            ctx.set_frame_order_mode(1)
            ctx.frame_upload(raw)
            kh = ctx.frame_select_keypoints(q, t, size)
            n_host, _ = ctx.build_residuals(f, opts)
            assert len(kd) == len(kh) and len(kd) > 300, (len(kd), len(kh))
            assert np.array_equal(kd, kh)
            ctx2.sweep_upload(raw[kh])
            n_up, _ = ctx2.build_residuals(f, opts)
            for a in (n_host, n_up):
                assert n_dev.num_residuals == a.num_residuals > 300
                assert np.array_equal(np.array(n_dev.HtH), np.array(a.HtH)) and np.array_equal(np.array(n_dev.Hth), np.array(a.Hth))
    finally:
        ctx.close(); ctx2.close()


@pytest.mark.parametrize("n,bits", [(1, 10), (63, 5), (64, 9), (1000, 10), (1024, 11), (1025, 12), (5000, 14), (24_000, 16), (24_000, 9), (65_536, 17),
                                    (100_000, 18), (131_072, 18), (131_073, 19), (262_144, 19), (300_001, 20), (1_048_576, 21), (1_048_577, 22),
                                    (3_000_000, 23), (3_000_000, 27)])
def test_own_radix_sort_is_a_stable_sort(n, bits):
    """srl_frame_commit groups a frame's points by scratch-table slot with radix passes of our own (srl_frame_scratch.h) instead of the
    library sort: (key, position) pairs must come out exactly as a stable sort leaves them.  One launch per pass up to 131 072 pairs,
    two (tile histograms, then the pass) up to 1 M, five beyond (histograms, their scan, the pass): every form, and three passes (27 bits)."""
    ctx = srl.Context(0)
    try:
        for seed, kind in ((0, "uniform"), (1, "few"), (2, "one"), (3, "high_bits_set"), (4, "sorted_desc")):
            rng = np.random.default_rng(1000 * n + seed)
            if kind == "uniform":
                keys = rng.integers(0, 1 << bits, n, dtype=np.uint32)
            elif kind == "few":
                keys = rng.choice(rng.integers(0, 1 << bits, 7, dtype=np.uint32), n)
            elif kind == "one":
                keys = np.full(n, (1 << bits) - 1, dtype=np.uint32)
            elif kind == "high_bits_set":                    # bits above `bits` must be ignored by the order and carried along
                keys = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
            else:
                keys = np.sort(rng.integers(0, 1 << bits, n, dtype=np.uint32))[::-1].copy()
            ks, ps = ctx.radix_sort_pairs(keys, bits)
            order = np.argsort(keys & np.uint32((1 << bits) - 1), kind="stable")
            assert np.array_equal(ps, order.astype(np.uint32)), (kind, n, bits)
            assert np.array_equal(ks, keys[order]), (kind, n, bits)
    finally:
        ctx.close()
