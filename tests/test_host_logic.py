"""Host-side logic of the product that runs without a GPU: the C++ mirror of eskfEstimator and of the
ESIKF update (sr_livo_amd/csrc/host), driven through the srl_lio handles, against the CPU oracle."""
import numpy as np
import pytest

import sr_livo_amd as srl
from oracle import pyoracle as po
from sr_livo_amd import capi, synth

INT_MAX = 2**31 - 1


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


class EskfAdapter:
    def __init__(self, lio): self.lio = lio
    def set_noise(self, *a): self.lio.eskf_set_noise(*a)
    def scale_init_cov(self): self.lio.eskf_scale_init_cov()
    def init_imu(self, a, g): self.lio.eskf_init_imu(a, g)
    def predict(self, dt, a, g): self.lio.eskf_predict(dt, a, g)
    def get_state(self): return self.lio.eskf_get_state()
    def set_state(self, s): self.lio.eskf_set_state(s)


def oracle_provider(m, raw, opts_o, shard=None, comm=None):
    """normal-equation provider backed by the oracle's buildPlaneResiduals (tests only)."""
    def provider(frame, opts, out):
        q = np.array(frame.q); t = np.array(frame.t); tl = np.array(frame.t_last)
        o = m.build_plane_residuals(opts_o, raw, q, t, tl, R_il=np.array(frame.R_il).reshape(3, 3), t_il=np.array(frame.t_il),
                                    frame_id=frame.frame_id, full=False)
        out.HtH[:] = list(o["HtH"].ravel()); out.Hth[:] = list(o["Hth"])
        out.loss_sum = o["neq"].loss_sum; out.num_residuals = o["neq"].num_residuals; out.success = o["neq"].success
        return 0
    return provider


def test_host_only_handle_refuses_the_device_path():
    lio = srl.Lio(-1)
    with pytest.raises(srl.SrlError) as ei:
        lio.update_iekf(srl.default_opts(), np.zeros((10, 3)), np.r_[1, np.zeros(15)], np.zeros(3))
    assert ei.value.status == capi.SRL_ERR_NO_DEVICE
    with pytest.raises(srl.SrlError):
        lio.add_points_to_map(np.zeros((3, 3)))
    with pytest.raises(srl.SrlError):
        lio.search_neighbors([0, 0, 0])


def test_eskf_mirror_matches_oracle_predict_and_observe():
    lio = srl.Lio(-1)
    e = po.Eskf()
    for obj in (EskfAdapter(lio), e):
        obj.set_noise(0.1, 0.1, 1e-4, 1e-4); obj.scale_init_cov()
        s = obj.get_state(); s[3:7] = synth.quat_from_rotvec([0.1, -0.2, 0.05]); s[7:10] = [0.3, -0.1, 0.05]
        s[10:13] = [0.01, 0.02, -0.01]; s[13:16] = [1e-3, -2e-3, 5e-4]; obj.set_state(s)
        obj.init_imu([0.1, 0.2, 9.7], [0.01, -0.02, 0.03])
    rng = np.random.default_rng(0)
    for _ in range(25):
        acc, gyr = np.array([0, 0, 9.81]) + rng.normal(0, 0.3, 3), rng.normal(0, 0.05, 3)
        lio.eskf_predict(0.01, acc, gyr); e.predict(0.01, acc, gyr)
    assert rel(lio.eskf_get_state(), e.get_state()) < 1e-13
    assert rel(lio.eskf_get_cov(), e.get_cov()) < 1e-12
    dx = rng.normal(0, 1e-2, 17)
    lio.eskf_observe(dx); e.observe(dx)
    assert rel(lio.eskf_get_state(), e.get_state()) < 1e-13
    dx_small = np.r_[0, 0, 0, 1e-6, -2e-6, 1e-6, np.zeros(11)]          # THETA_THRESHOLD branch
    lio.eskf_observe(dx_small); e.observe(dx_small)
    assert rel(lio.eskf_get_state(), e.get_state()) < 1e-13


@pytest.mark.parametrize("frame_id,max_res,iters_icp", [(100, INT_MAX, 5), (100, 600, 5), (1, INT_MAX, 3), (5, 300, 5)])
def test_update_iekf_host_algebra_matches_oracle(small_scene, frame_id, max_res, iters_icp):
    """The product's updateIEKF (17-dim algebra, convergence logic, covariance update) fed with the oracle's
    normal equations reproduces the oracle's full solve: per-iteration d_x, final state and covariance."""
    m, sw = small_scene["map"], small_scene["sweep"]
    raw = sw["raw"][:700]
    opts_p = srl.default_opts(max_num_residuals=max_res, num_iters_icp=iters_icp)
    opts_o = po.opts_from_product(opts_p)
    lio = srl.Lio(-1)
    e = po.Eskf()
    synth.eskf_prior(e, sw["q_pred"], sw["t_pred"], sw["vel"])
    lio.eskf_set_state(e.get_state()); lio.eskf_set_cov(e.get_cov())
    st = np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)])
    u = po.update_iekf(m, e, opts_o, raw, st, sw["t_last"], frame_id=frame_id, log_iters=20)
    g = lio.update_iekf_provided(opts_p, oracle_provider(m, raw, opts_o), len(raw), st, sw["t_last"], frame_id=frame_id, log_iters=20)
    assert g["rc"] == 0 and g["iters"] == u["rc"]
    if frame_id <= 1:
        # convergence test disabled (optimize.cpp:265) and init mode: max(15, num_iters_icp) (optimize.cpp:135-136), i = -1 .. 14
        assert g["iters"] == max(15, iters_icp) + 1
    assert rel(g["log"][:, 42:59], u["log"][:, 42:59]) < 1e-10
    assert rel(g["state"], u["state"]) < 1e-12
    assert rel(lio.eskf_get_state(), e.get_state()) < 1e-12
    assert rel(lio.eskf_get_cov(), e.get_cov()) < 1e-10


def test_update_iekf_failure_paths(small_scene):
    m, sw = small_scene["map"], small_scene["sweep"]
    raw = sw["raw"][:300]
    lio = srl.Lio(-1)
    st = np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)])
    opts_p = srl.default_opts(max_num_residuals=-1)
    g = lio.update_iekf_provided(opts_p, oracle_provider(m, raw, po.opts_from_product(opts_p)), len(raw), st, sw["t_last"])
    assert g["rc"] == capi.SRL_ERR_NOT_ENOUGH_RESIDUALS and np.array_equal(g["state"], st)

    def nan_provider(frame, opts, out):
        return capi.SRL_ERR_NAN_PLANARITY
    g = lio.update_iekf_provided(srl.default_opts(), nan_provider, 10, st, sw["t_last"], allow=(capi.SRL_ERR_NAN_PLANARITY,))
    assert g["rc"] == capi.SRL_ERR_NAN_PLANARITY            # optimize.cpp:348-350 -> std::runtime_error("error")


def test_grid_sampling_keeps_first_point_per_voxel():
    rng = np.random.default_rng(5)
    pts = rng.uniform(-20, 20, (5000, 3))
    idx = srl.grid_sampling(pts, 1.5)
    keys = np.trunc(pts / 1.5).astype(np.int64)
    first = {}
    for i, k in enumerate(map(tuple, keys)):
        first.setdefault(k, i)
    assert sorted(idx.tolist()) == sorted(first.values())          # one point per voxel: the first in frame order
    assert np.array_equal(idx, srl.grid_sampling(pts, 1.5))          # deterministic (tr1 iteration order)
    assert len(srl.grid_sampling(np.zeros((0, 3)), 1.0)) == 0
