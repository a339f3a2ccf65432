"""Host-side logic of the product that runs without a GPU: the C++ mirror of eskfEstimator and of the
ESIKF update (sr_livo_amd/csrc/host), driven through the srl_lio handles, against the CPU oracle."""
import os
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


# ----------------------------------------------------------------------------- row f3: tryInit / stateInitialization
def _imu_batches(rng, n_batches, per_batch, dt, sigma_g, sigma_a, tilt):
    g_dir = np.array([np.sin(tilt), 0.0, np.cos(tilt)])
    out = []
    t = 100.0
    for _ in range(n_batches):
        ts = t + dt * np.arange(1, per_batch + 1)
        t = ts[-1]
        out.append((ts, np.array([0.002, -0.001, 0.0005]) + rng.normal(0, sigma_g, (per_batch, 3)),
                    9.79 * g_dir + rng.normal(0, sigma_a, (per_batch, 3))))
    return out


@pytest.mark.parametrize("sigma_g,sigma_a,want", [(0.01, 0.05, 1), (0.6, 0.05, -1), (0.01, 0.7, -2)])
def test_try_init_matches_oracle_and_numpy(sigma_g, sigma_a, want):
    import np_reference as npr
    rng = np.random.default_rng(5)
    batches = _imu_batches(rng, 8, 100, 0.005, sigma_g, sigma_a, 0.05)      # 0.5 s per batch: initialises in batch 7
    lio = srl.Lio(-1)
    lio.set_initial_flag(False)
    e = po.Eskf()
    lio.eskf_set_noise(0.1, 0.2, 1e-4, 2e-4); e.set_noise(0.1, 0.2, 1e-4, 2e-4)
    ref = npr.try_init_stats(batches)
    codes = []
    for bi, (t, g, a) in enumerate(batches):
        r_h = lio.eskf_try_init(t, g, a); r_o = e.try_init(t, g, a)
        assert r_h == r_o
        codes.append(r_h)
        sh, so = lio.eskf_init_stats(), e.init_stats()
        for k in ("mean_gyr", "mean_acc", "gyr_cov", "acc_cov"):
            assert rel(sh[k], so[k]) < 1e-15, k
        assert sh["num_init_meas"] == so["num_init_meas"] and sh["initial_flag"] == so["initial_flag"]
        if r_h != 0:
            break
    assert codes[-1] == want == ref["code"] and len(codes) - 1 == ref["at"] == 6
    if want == 1:
        assert all(c == 0 for c in codes[:-1])
        s = lio.eskf_get_state()
        assert rel(s[13:16], ref["bg"]) < 1e-12 and rel(s[16:19], ref["gravity"]) < 1e-12
        assert rel(lio.eskf_get_state(), e.get_state()) < 1e-15 and np.array_equal(lio.eskf_get_cov(), e.get_cov())
        P = lio.eskf_get_cov()
        assert np.allclose(np.diag(P)[9:17], [1e-3] * 3 + [1e-4] * 3 + [1e-5] * 2)
        # the filters predict identically afterwards (noise = *_cov_scale, last IMU sample = last of the batch)
        acc, gyr = np.array([0.1, 0.0, 9.8]), np.array([0.01, 0.0, -0.01])
        lio.eskf_predict(0.005, acc, gyr); e.predict(0.005, acc, gyr)
        assert rel(lio.eskf_get_cov(), e.get_cov()) < 1e-14 and rel(lio.eskf_get_state(), e.get_state()) < 1e-14
    else:
        st = lio.eskf_init_stats()
        assert not st["initial_flag"]
        assert rel(st["gyr_cov"], ref["gyr_cov"]) < 1e-10 and rel(st["acc_cov"], ref["acc_cov"]) < 1e-10
    lio.set_initial_flag(False)


def test_state_initialization_matches_oracle_and_rotation_algebra():
    import np_reference as npr
    lio = srl.Lio(-1)
    q2 = synth.quat_from_rotvec([0.02, -0.3, 0.5]); t2 = np.array([1.0, 2.0, 0.1])
    q1 = synth.quat_from_rotvec([0.05, -0.28, 0.61]); t1 = np.array([1.4, 2.3, 0.12])
    p2, p1 = np.r_[q2, t2], np.r_[q1, t1]
    es = lio.eskf_get_state(); es[0:3] = [7.0, 8.0, 9.0]; es[3:7] = synth.quat_from_rotvec([0.3, 0.2, 0.1]); lio.eskf_set_state(es)
    for index_frame in (1, 2, 3, 4, 50):
        for init in (0, 1, 7):
            for flag in (False, True):
                lio.set_initial_flag(flag)
                q, t = lio.state_initialization(index_frame, init, p2, p1)
                qo, to = po.state_initialization(index_frame, init, flag, p2, p1, es[3:7], es[0:3])
                assert np.array_equal(q, qo) and np.array_equal(t, to)
                if index_frame <= 2:
                    assert np.array_equal(q, [1, 0, 0, 0]) and np.array_equal(t, [0, 0, 0])
                elif init == 1 or (init == 0 and not flag):
                    R, tt = npr.state_initialization_const_velocity(q2, t2, q1, t1)
                    assert np.allclose(npr.quat_to_rot(q), R, atol=1e-14) and np.allclose(t, tt, atol=1e-14)
                elif init == 0:
                    assert np.array_equal(q, es[3:7]) and np.array_equal(t, es[0:3])
                else:
                    assert np.array_equal(q, q1) and np.array_equal(t, t1)
    lio.set_initial_flag(False)


def test_inverse_cols_equals_the_columns_of_the_full_inverse_bitwise(tmp_path):
    """updateIEKF's second 17x17 inverse is only read through temp_inv.block<17,6>(0,0) (optimize.cpp:237-242); the host mirror
    solves those six columns alone (srl::inverse_cols<17,6>).  Compiled check on the header itself: same bits as the full inverse."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    src = tmp_path / "inv.cpp"
    src.write_text(r'''
#include "sr_livo_amd/csrc/host/srl_la.h"
#include <cstdio>
#include <cstring>
#include <random>
int main() {
    std::mt19937_64 rng(7);
    std::normal_distribution<double> nd(0.0, 1.0);
    for (int trial = 0; trial < 200; trial++) {
        srl::Mat<17, 17> B, A, full;
        for (int i = 0; i < 17; i++) for (int j = 0; j < 17; j++) B(i, j) = nd(rng);
        for (int i = 0; i < 17; i++) for (int j = 0; j < 17; j++) { double s = (i == j) ? 1e-3 * (1 + trial) : 0.0; for (int k = 0; k < 17; k++) s += B(i, k) * B(j, k); A(i, j) = s; }
        if (trial % 3 == 0) { for (int j = 0; j < 17; j++) { double t = A(2, j); A(2, j) = A(11, j); A(11, j) = t; } }   // force row pivoting
        srl::Mat<17, 6> cols;
        if (!srl::inverse<17>(A, full) || !srl::inverse_cols<17, 6>(A, cols)) return 2;
        for (int i = 0; i < 17; i++) for (int c = 0; c < 6; c++) { double a = full(i, c), b = cols(i, c); if (std::memcmp(&a, &b, 8) != 0) { std::printf("mismatch %d %d %d\n", trial, i, c); return 1; } }
    }
    return 0;
}
''')
    exe = tmp_path / "inv"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I", root, str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout


def test_the_product_tree_never_reaches_for_the_oracle():
    """oracle/ is test infrastructure: nothing under sr_livo_amd/ (library sources, host mirror, Python layer), include/ or integration/
    may import, link, dlopen or spawn anything from it (comments that NAME the oracle or its Makefile are not reaches), and bench.py may
    only touch it behind its timed region (cpu_baseline / parity legs)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    reach = re.compile(r"(from\s+oracle\b|import\s+oracle\b|pyoracle|pyref|liboracle|libref_path|libref_node|dlopen\([^)]*oracle)")
    bad = []
    for top in ("sr_livo_amd", "include", "integration"):
        for dp, _dn, fns in os.walk(os.path.join(root, top)):
            if "__pycache__" in dp or os.sep + "build" in dp:
                continue
            for fn in fns:
                if not fn.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                    continue
                path = os.path.join(dp, fn)
                for ln, line in enumerate(open(path, errors="replace"), 1):
                    code = line.split("//")[0].split("#")[0] if not fn.endswith(".py") else line.split("#")[0]
                    if reach.search(code):
                        bad.append(f"{os.path.relpath(path, root)}:{ln}: {line.strip()[:100]}")
    assert not bad, bad
    # bench.py: every mention of the oracle sits behind the timed region's closing barrier
    src = open(os.path.join(root, "bench.py")).read()
    main = src[src.index("def main("):]
    region_end = main.index("elapsed = time.perf_counter() - t1")
    before = main[:region_end]
    assert not re.search(r"\bpo\.(Map|Eskf|update_iekf|build_plane_residuals)\(|\bpr\.", before), "bench.py touches the oracle before its timed region has ended"
