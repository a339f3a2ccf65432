"""Randomised parity of one pass through the C-ABI against the live oracle: scene size, sweep size and pattern, search radius (init mode),
K, budget, selection path and a pose error larger than the generator's all drawn per case.  Same bars as tests/test_gpu_parity.py:
neighbour ids and status bit-exact, counts equal, residual fields and normal equations to 1e-9 relative.

The default run draws SRL_FUZZ_CASES = 24 cases (seconds); `SRL_FUZZ_CASES=400 python -m pytest tests/test_gpu_fuzz.py -m gpu` is the
longer form whose output is kept under profiles/."""
import os

import numpy as np
import pytest

import sr_livo_amd as srl
from sr_livo_amd import synth

from test_gpu_parity import INT_MAX, check_pass_against, gpu_pass

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("SRL_FUZZ_CASES", "24"))


def _draw(case):
    rng = np.random.default_rng(9000 + case)
    return dict(
        map_pts=int(rng.choice([3_000, 12_000, 40_000, 90_000])),
        n=int(rng.choice([1, 7, 63, 64, 65, 500, 1_500, 4_096, 6_001])),
        pattern=str(rng.choice(["livox", "ouster16"])),
        frame_id=int(rng.choice([5, 100])),                                   # 5: init mode (r = 2), 100: r = 1
        K=int(rng.choice([5, 20, 20, 32])),
        max_res=int(rng.choice([INT_MAX, INT_MAX, 600, 37, 5, 0, -1])),
        select_mode=int(rng.choice([0, 0, 0, 1, 2])),
        extra_rot=rng.normal(0, 0.01, 3), extra_t=rng.normal(0, 0.05, 3),
        seed=int(rng.integers(1, 2**31 - 1)),
    )


@pytest.mark.parametrize("case", range(CASES))
def test_random_pass_against_the_oracle(oracle_lib, oracle_backend, case):
    p = _draw(case)
    pts, L = synth.map_candidates(p["seed"], p["map_pts"])
    sw = synth.make_sweep(p["seed"] + 1, p["n"], L, pattern=p["pattern"])
    q = synth.quat_mul(synth.quat_from_rotvec(p["extra_rot"]), sw["q_pred"])
    t = sw["t_pred"] + p["extra_t"]
    m = oracle_lib.Map(oracle_backend)
    m.add_points(pts)
    ctx = srl.Context(0)
    try:
        ctx.map_upload(*m.export())
        kw = dict(max_num_residuals=p["max_res"], max_number_neighbors=p["K"])
        g = gpu_pass(ctx, sw["raw"], q, t, sw["t_last"], frame_id=p["frame_id"], select_mode=p["select_mode"], **kw)
        o = m.build_plane_residuals(oracle_lib.default_opts(**kw), sw["raw"], q, t, sw["t_last"], frame_id=p["frame_id"])
        if o["neq"].nan_error:
            pytest.skip("the oracle met a NaN planarity in this draw (covered by the dedicated NaN tests)")
        ref = {f"x_one_{k}": v for k, v in o.items() if isinstance(v, np.ndarray)}
        ref.update(x_one_num_ties=o["neq"].num_ties, x_one_num_residuals=o["neq"].num_residuals, x_one_success=o["neq"].success, x_one_loss=o["neq"].loss_sum)
        try:
            check_pass_against(g, ref, "x")
            assert g["neq"].last_visited == o["neq"].num_visited - 1
            if p["max_res"] == INT_MAX:
                assert g["neq"].sum_candidates == o["neq"].sum_candidates
        except AssertionError as e:
            raise AssertionError(f"case {case}: {p}: {e}") from e
    finally:
        ctx.close()


SOLVES = max(CASES // 2, 1)


@pytest.mark.parametrize("case", range(SOLVES))
def test_random_full_solve_against_the_oracle(oracle_lib, oracle_backend, case):
    """The whole ESIKF loop (armed launches, host update) on a random scene against the oracle's updateIEKF: same iteration count, same
    number of residuals, state to 1e-9, covariance to 1e-8 -- for a random budget, K and a pose error up to several iterations' worth."""
    from test_gpu_parity import rel, state16
    rng = np.random.default_rng(77_000 + case)
    map_pts = int(rng.choice([8_000, 30_000, 80_000]))
    n = int(rng.choice([200, 1_000, 3_000, 5_000]))
    pattern = str(rng.choice(["livox", "ouster16"]))
    max_res = int(rng.choice([INT_MAX, 600, 150]))
    K = int(rng.choice([10, 20, 20, 32]))
    frame_id = int(rng.choice([5, 100]))
    seed = int(rng.integers(1, 2**31 - 1))
    pts, L = synth.map_candidates(seed, map_pts)
    sw = synth.make_sweep(seed + 1, n, L, pattern=pattern)
    sw["q_pred"] = synth.quat_mul(synth.quat_from_rotvec(rng.normal(0, 0.004, 3)), sw["q_pred"])
    sw["t_pred"] = sw["t_pred"] + rng.normal(0, 0.03, 3)
    m = oracle_lib.Map(oracle_backend)
    m.add_points(pts)
    lio = srl.Lio(0)
    try:
        lio.ctx.map_upload(*m.export())
        e = oracle_lib.Eskf(oracle_backend)
        synth.eskf_prior(e, sw["q_pred"], sw["t_pred"], sw["vel"])
        lio.eskf_set_state(e.get_state()); lio.eskf_set_cov(e.get_cov())
        st = state16(sw)
        opts = srl.default_opts(max_num_residuals=max_res, max_number_neighbors=K)
        what = f"case {case}: map {map_pts}, n {n}, {pattern}, max_res {max_res}, K {K}, frame_id {frame_id}, seed {seed}"
        try:
            r = lio.update_iekf(opts, sw["raw"], st, sw["t_last"], frame_id=frame_id)
        except srl.SrlError as err:                                         # NaN planarity: the reference throws
            r = dict(rc=err.status)
        u = oracle_lib.update_iekf(m, e, oracle_lib.opts_from_product(opts), sw["raw"], st, sw["t_last"], frame_id=frame_id)
        if u["rc"] < 0 or r["rc"] != 0:
            assert u["rc"] < 0 and r["rc"] != 0, (what, u["rc"], r["rc"])     # a failed solve (too few residuals / NaN planarity) fails on both sides
            return
        assert r["rc"] == 0 and r["iters"] == u["rc"] and r["num_residuals"] == u["num_residuals"], (what, r["rc"], r["iters"], u["rc"])
        assert rel(r["state"], u["state"]) < 1e-9, what
        assert rel(lio.eskf_get_cov(), e.get_cov()) < 1e-8, what
    finally:
        lio.close()


@pytest.mark.parametrize("case", range(SOLVES))
def test_random_frame_pipeline_against_the_oracle(oracle_lib, oracle_backend, case):
    """upload -> device keypoint selection -> commit (deferred or not) on random frames: the keypoints equal gridSampling's in set AND order
    (std::tr1::unordered_map iteration order), the world points equal transformPoint's bit for bit, the map equals the sequential
    addPointsToMap -- for random frame sizes, sampling voxel sizes, extrinsics, un-normalised quaternions and two frames in a row."""
    rng = np.random.default_rng(55_000 + case)
    map_pts = int(rng.choice([5_000, 40_000]))
    seed = int(rng.integers(1, 2**31 - 1))
    pts, L = synth.map_candidates(seed, map_pts)
    m = oracle_lib.Map(oracle_backend)
    m.add_points(pts[: map_pts // 2])
    ctx = srl.Context(0)
    try:
        ctx.map_insert(pts[: map_pts // 2])
        for k in range(2):
            n = int(rng.choice([1, 2, 100, 1_023, 1_024, 1_025, 7_000, 20_000, 40_000]))
            size = float(rng.choice([0.2, 0.5, 1.0, 1.5, 3.0]))
            sw = synth.make_sweep(seed + 10 + k, n, L, pattern=str(rng.choice(["livox", "ouster16"])))
            R_il = synth.quat_to_rot(synth.quat_from_rotvec(rng.normal(0, 0.05, 3))); t_il = rng.normal(0, 0.05, 3)
            raw = (sw["raw"] - t_il) @ R_il
            q = sw["q_pred"] * float(rng.uniform(0.999, 1.001)); t = sw["t_pred"]
            what = f"case {case}.{k}: map {map_pts}, n {n}, size {size}, seed {seed}"
            world = oracle_lib.transform_points(raw, q, t, R_il, t_il, backend=oracle_backend)
            want = oracle_lib.grid_sampling(world, size, backend=oracle_backend)
            ctx.frame_upload(raw)
            got = ctx.frame_select_keypoints(q, t, size, R_il, t_il)
            assert np.array_equal(got, want), what
            q2 = sw["q_gt"]; t2 = sw["t_gt"]
            deferred = bool(rng.integers(0, 2))
            world2, added = ctx.frame_commit(q2, t2, R_il=R_il, t_il=t_il, want_added=not deferred)
            assert np.array_equal(world2, oracle_lib.transform_points(raw, q2, t2, R_il, t_il, backend=oracle_backend)), what
            before = m.size()
            m.add_points(world2)
            assert ctx.map_size()[0] == m.size() and (deferred or added == m.size() - before), what
        kg, cg, xg = ctx.map_download(); ko, co, xo = m.export()
        assert np.array_equal(kg, ko) and np.array_equal(cg, co) and np.array_equal(xg, xo)
    finally:
        ctx.close()
