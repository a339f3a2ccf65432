"""ctypes binding of the CPU ORACLE (oracle/liboracle.so or oracle/_ref/liboracle_tsl.so).

TEST INFRASTRUCTURE ONLY -- import from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never from the product package (see oracle/srl_oracle.h).  Pinned bitwise against the reference's own translation
units (oracle/pyref.py, tests/test_reference_tu.py).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORC_LIB_PLAIN / ORC_LIB_TSL: another build of the same sources (tests/test_sanitizers.py loads the ASan + UBSan build this way)
LIB_PLAIN = os.environ.get("ORC_LIB_PLAIN") or os.path.join(_HERE, "liboracle.so")
LIB_TSL = os.environ.get("ORC_LIB_TSL") or os.path.join(_HERE, "_ref", "liboracle_tsl.so")


class OrcOpts(C.Structure):
    _fields_ = [
        ("threshold_voxel_occupancy", C.c_int), ("init_num_frames", C.c_int), ("size_voxel_map", C.c_double),
        ("num_iters_icp", C.c_int), ("min_number_neighbors", C.c_int), ("voxel_neighborhood", C.c_int),
        ("power_planarity", C.c_double), ("estimate_normal_from_neighborhood", C.c_int),
        ("max_number_neighbors", C.c_int), ("max_dist_to_plane_icp", C.c_double),
        ("threshold_orientation_norm", C.c_double), ("threshold_translation_norm", C.c_double),
        ("max_num_residuals", C.c_int), ("weight_alpha", C.c_double), ("weight_neighborhood", C.c_double),
    ]


class OrcResidualOut(C.Structure):
    _fields_ = [("status", C.c_void_p), ("ids", C.c_void_p), ("tie", C.c_void_p), ("point_world", C.c_void_p),
                ("normal", C.c_void_p), ("a2D", C.c_void_p), ("weight", C.c_void_p), ("norm_offset", C.c_void_p),
                ("distance", C.c_void_p), ("jacobian", C.c_void_p)]


class OrcNormalEq(C.Structure):
    _fields_ = [("HtH", C.c_double * 36), ("Hth", C.c_double * 6), ("loss_sum", C.c_double),
                ("num_residuals", C.c_int32), ("success", C.c_int32), ("sum_candidates", C.c_int64),
                ("num_visited", C.c_int32), ("num_ties", C.c_int32), ("nan_error", C.c_int32)]


_libs = {}


def load(backend="plain"):
    path = LIB_TSL if backend == "tsl" else LIB_PLAIN
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(f"oracle library missing: {path} (run `make -C oracle`)")
    lib = C.CDLL(path)
    p, dp = C.c_void_p, C.POINTER(C.c_double)
    lib.orc_map_create.restype = p
    lib.orc_map_destroy.argtypes = [p]
    lib.orc_map_add_points.argtypes = [p, p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int]
    lib.orc_map_add_points.restype = C.c_int
    lib.orc_map_size.argtypes = [p]; lib.orc_map_size.restype = C.c_size_t
    lib.orc_map_num_voxels.argtypes = [p]; lib.orc_map_num_voxels.restype = C.c_int
    lib.orc_map_export.argtypes = [p, C.c_int, p, p, p]
    lib.orc_map_import.argtypes = [p, p, p, p, C.c_int, C.c_int]; lib.orc_map_import.restype = C.c_int
    lib.orc_voxel_hash.argtypes = [C.c_int16, C.c_int16, C.c_int16]; lib.orc_voxel_hash.restype = C.c_uint64
    lib.orc_voxel_coord.argtypes = [C.c_double, C.c_double]; lib.orc_voxel_coord.restype = C.c_int16
    lib.orc_map_backend.restype = C.c_char_p
    lib.orc_icp_opts_default.argtypes = [C.POINTER(OrcOpts)]
    lib.orc_search_neighbors.argtypes = [p, dp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, p, p, p,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.orc_search_neighbors.restype = C.c_int
    lib.orc_neighborhood.argtypes = [p, C.c_int, dp, dp, dp, dp, dp]; lib.orc_neighborhood.restype = C.c_int
    lib.orc_build_plane_residuals.argtypes = [p, C.POINTER(OrcOpts), p, C.c_int, dp, dp, dp, dp, dp, C.c_int, C.c_int,
                                              C.POINTER(OrcResidualOut), C.POINTER(OrcNormalEq)]
    lib.orc_build_plane_residuals.restype = C.c_int
    lib.orc_eskf_create.restype = p
    lib.orc_eskf_destroy.argtypes = [p]
    for name in ("orc_eskf_get_state", "orc_eskf_set_state", "orc_eskf_get_cov", "orc_eskf_set_cov", "orc_eskf_observe"):
        getattr(lib, name).argtypes = [p, dp]
    lib.orc_eskf_set_noise.argtypes = [p, C.c_double, C.c_double, C.c_double, C.c_double]
    lib.orc_eskf_init_imu.argtypes = [p, dp, dp]
    lib.orc_eskf_scale_init_cov.argtypes = [p]
    lib.orc_eskf_predict.argtypes = [p, C.c_double, dp, dp]
    lib.orc_eskf_try_init.argtypes = [p, dp, dp, dp, C.c_int]; lib.orc_eskf_try_init.restype = C.c_int
    lib.orc_eskf_get_init_stats.argtypes = [p, dp]
    lib.orc_eskf_set_g_norm.argtypes = [p, C.c_double]
    lib.orc_state_initialization.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp]
    lib.orc_update_iekf.argtypes = [p, p, C.POINTER(OrcOpts), p, C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, C.c_double,
                                    p, C.c_int, C.POINTER(C.c_int)]
    lib.orc_update_iekf.restype = C.c_int
    lib.orc_quat_to_rot.argtypes = [dp, dp]; lib.orc_rot_to_quat.argtypes = [dp, dp]
    lib.orc_so3_to_rot.argtypes = [dp, dp]; lib.orc_so3_to_quat.argtypes = [dp, dp]; lib.orc_rot_to_so3.argtypes = [dp, dp]
    lib.orc_angular_distance_so3.argtypes = [dp]; lib.orc_angular_distance_so3.restype = C.c_double
    lib.orc_derivative_s2.argtypes = [dp, dp]
    lib.orc_inverse17.argtypes = [dp, dp]; lib.orc_inverse17.restype = C.c_int
    lib.orc_eig3.argtypes = [dp, dp, dp]
    lib.orc_eig3_solver.argtypes = [C.c_int, dp, dp, dp]; lib.orc_eig3_solver.restype = C.c_int
    lib.orc_set_eig_solver.argtypes = [C.c_int]
    lib.orc_get_eig_solver.restype = C.c_int
    lib.orc_set_threads.argtypes = [C.c_int]
    lib.orc_get_threads.restype = C.c_int
    lib.orc_heap_topk.argtypes = [p, C.c_int, C.c_int, p]; lib.orc_heap_topk.restype = C.c_int
    lib.orc_distort_frame_by_constant.argtypes = [p, p, C.c_int, p, C.c_int, C.c_double, dp, dp, p]
    lib.orc_distort_frame_by_imu.argtypes = [p, p, C.c_int, p, C.c_int, C.c_double, dp, dp, p]
    lib.orc_distort_frame_by_imu.restype = C.c_int
    lib.orc_transform_all_imu_point.argtypes = [p, C.c_int, p, C.c_int, dp, dp, p]
    lib.orc_build_frame_order.argtypes = [p, C.c_int, C.c_double, C.c_int, p]; lib.orc_build_frame_order.restype = C.c_int
    lib.orc_make_point_timestamp.argtypes = [p, C.c_int, C.c_double, C.c_double, C.c_int, p, p, p]
    lib.orc_make_point_timestamp.restype = C.c_int
    lib.orc_mt19937_64_nth.argtypes = [C.c_int]; lib.orc_mt19937_64_nth.restype = C.c_uint64
    lib.orc_transform_points.argtypes = [p, C.c_int, dp, dp, dp, dp, p]
    lib.orc_grid_sampling.argtypes = [p, C.c_int, C.c_double, p]; lib.orc_grid_sampling.restype = C.c_int
    _libs[path] = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a if shape is None else a.reshape(shape)


def default_opts(lib=None, **kw):
    lib = lib or load()
    o = OrcOpts()
    lib.orc_icp_opts_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def opts_from_product(po, lib=None):
    """Copy a product IcpOpts (sr_livo_amd.IcpOpts) into the oracle's option struct."""
    o = default_opts(lib)
    for name, _ in OrcOpts._fields_:
        if hasattr(po, name):
            setattr(o, name, getattr(po, name))
    return o


class Map:
    def __init__(self, backend="plain"):
        self.lib = load(backend)
        self.h = C.c_void_p(self.lib.orc_map_create())

    def __del__(self):
        try:
            if self.h:
                self.lib.orc_map_destroy(self.h)
        except Exception:
            pass

    def add_points(self, xyz, voxel_size=1.0, cap=20, min_dist=0.15, min_num_points=0):
        x = _f64(xyz, (-1, 3))
        return self.lib.orc_map_add_points(self.h, _vp(x), len(x), voxel_size, cap, min_dist, min_num_points)

    def size(self):
        return int(self.lib.orc_map_size(self.h))

    def num_voxels(self):
        return int(self.lib.orc_map_num_voxels(self.h))

    def export(self, cap=20):
        V = self.num_voxels()
        keys = np.zeros((V, 3), dtype=np.int16); counts = np.zeros(V, dtype=np.int32); xyz = np.zeros((V, cap, 3), dtype=np.float32)
        self.lib.orc_map_export(self.h, cap, _vp(keys), _vp(counts), _vp(xyz))
        return keys, counts, xyz

    def import_(self, keys, counts, xyz, cap=20):
        keys = np.ascontiguousarray(keys, dtype=np.int16); counts = np.ascontiguousarray(counts, dtype=np.int32)
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        if self.lib.orc_map_import(self.h, _vp(keys), _vp(counts), _vp(xyz), len(counts), cap) != 0:
            raise ValueError("orc_map_import: duplicate key or count out of range")

    def search_neighbors(self, p, nb=1, size=1.0, K=20, thr=1, cap=20):
        p = _f64(p)
        xyz = np.zeros((K, 3)); ids = np.full(K, -1, dtype=np.int32); dist = np.zeros(K)
        tie, nc = C.c_int(), C.c_int()
        n = self.lib.orc_search_neighbors(self.h, _dp(p), nb, size, K, thr, cap, _vp(xyz), _vp(ids), _vp(dist), C.byref(tie), C.byref(nc))
        return dict(n=n, xyz=xyz[:n], ids=ids[:n], dist=dist[:n], tie=bool(tie.value), num_candidates=nc.value)

    def build_plane_residuals(self, opts, raw, q, t, t_last, R_il=None, t_il=None, frame_id=100, cap=20, full=True):
        raw = _f64(raw, (-1, 3)); n = len(raw); K = opts.max_number_neighbors
        R_il = _f64(np.eye(3) if R_il is None else R_il).ravel(); t_il = _f64(np.zeros(3) if t_il is None else t_il)
        out = OrcResidualOut(); arrs = {}
        if full:
            arrs = dict(status=np.zeros(n, np.uint8), ids=np.zeros((n, K), np.int32), tie=np.zeros(n, np.uint8),
                        point_world=np.zeros((n, 3)), normal=np.zeros((n, 3)), a2D=np.zeros(n), weight=np.zeros(n),
                        norm_offset=np.zeros(n), distance=np.zeros(n), jacobian=np.zeros((n, 6)))
            for k, v in arrs.items():
                setattr(out, k, v.ctypes.data)
        neq = OrcNormalEq()
        rc = self.lib.orc_build_plane_residuals(self.h, C.byref(opts), _vp(raw), n, _dp(_f64(q)), _dp(_f64(t)), _dp(_f64(t_last)),
                                                _dp(R_il), _dp(t_il), frame_id, cap, C.byref(out) if full else None, C.byref(neq))
        arrs["rc"] = rc; arrs["neq"] = neq
        arrs["HtH"] = np.array(neq.HtH).reshape(6, 6); arrs["Hth"] = np.array(neq.Hth)
        return arrs


class Eskf:
    def __init__(self, backend="plain"):
        self.lib = load(backend)
        self.h = C.c_void_p(self.lib.orc_eskf_create())

    def __del__(self):
        try:
            if self.h:
                self.lib.orc_eskf_destroy(self.h)
        except Exception:
            pass

    def get_state(self):
        s = np.empty(19); self.lib.orc_eskf_get_state(self.h, _dp(s)); return s

    def set_state(self, s):
        self.lib.orc_eskf_set_state(self.h, _dp(_f64(s)))

    def get_cov(self):
        P = np.empty(289); self.lib.orc_eskf_get_cov(self.h, _dp(P)); return P.reshape(17, 17)

    def set_cov(self, P):
        self.lib.orc_eskf_set_cov(self.h, _dp(_f64(P).ravel()))

    def set_noise(self, a, g, ba, bg):
        self.lib.orc_eskf_set_noise(self.h, a, g, ba, bg)

    def init_imu(self, acc0, gyr0):
        self.lib.orc_eskf_init_imu(self.h, _dp(_f64(acc0)), _dp(_f64(gyr0)))

    def scale_init_cov(self):
        self.lib.orc_eskf_scale_init_cov(self.h)

    def try_init(self, t, gyr, acc):
        t = _f64(t); g = _f64(gyr, (-1, 3)); a = _f64(acc, (-1, 3))
        return self.lib.orc_eskf_try_init(self.h, _dp(t), _dp(g), _dp(a), len(t))

    def init_stats(self):
        o = np.zeros(14)
        self.lib.orc_eskf_get_init_stats(self.h, _dp(o))
        return dict(mean_gyr=o[0:3].copy(), mean_acc=o[3:6].copy(), gyr_cov=o[6:9].copy(), acc_cov=o[9:12].copy(),
                    num_init_meas=int(o[12]), initial_flag=bool(o[13]))

    def predict(self, dt, acc1, gyr1):
        self.lib.orc_eskf_predict(self.h, dt, _dp(_f64(acc1)), _dp(_f64(gyr1)))

    def observe(self, dx):
        self.lib.orc_eskf_observe(self.h, _dp(_f64(dx)))


def update_iekf(m, e, opts, raw, state, t_last, R_il=None, t_il=None, frame_id=100, cap=20, laser_point_cov=0.001, log_iters=0):
    raw = _f64(raw, (-1, 3)); st = _f64(state).copy()
    R_il = _f64(np.eye(3) if R_il is None else R_il).ravel(); t_il = _f64(np.zeros(3) if t_il is None else t_il)
    log = np.zeros((max(log_iters, 1), 61)) if log_iters else None
    nres = C.c_int()
    rc = m.lib.orc_update_iekf(m.h, e.h, C.byref(opts), _vp(raw), len(raw), _dp(st), _dp(_f64(t_last)), _dp(R_il), _dp(t_il),
                               frame_id, cap, laser_point_cov, _vp(log) if log is not None else None, log_iters, C.byref(nres))
    return dict(rc=rc, state=st, num_residuals=nres.value, log=None if log is None else log[: max(rc, 0)])


def transform_points(raw, q, t, R_il=None, t_il=None, backend="plain"):
    """transformPoint over a frame (utility.cpp:314-318)."""
    lib = load(backend)
    r = _f64(raw, (-1, 3))
    out = np.empty_like(r)
    R_il = _f64(np.eye(3) if R_il is None else R_il).ravel()
    t_il = _f64(np.zeros(3) if t_il is None else t_il)
    lib.orc_transform_points(_vp(r), len(r), _dp(_f64(q)), _dp(_f64(t)), _dp(R_il), _dp(t_il), _vp(out))
    return out


def grid_sampling(world, size_voxel, backend="plain"):
    """gridSampling (utility.cpp:188-201): frame indices of the keypoints, in keypoint order."""
    lib = load(backend)
    w = _f64(world, (-1, 3))
    idx = np.empty(max(len(w), 1), dtype=np.int32)
    m = lib.orc_grid_sampling(_vp(w), len(w), float(size_voxel), _vp(idx))
    return idx[:m].copy()


def state_initialization(index_frame, initialization, initial_flag, prev2, prev1, eskf_q=(1, 0, 0, 0), eskf_t=(0, 0, 0), backend="plain"):
    """stateInitialization (lioOptimization.cpp:895-990) -> (q wxyz, t)."""
    lib = load(backend)
    out = np.zeros(7)
    lib.orc_state_initialization(int(index_frame), int(initialization), int(bool(initial_flag)), _dp(_f64(prev2)), _dp(_f64(prev1)),
                                 _dp(_f64(eskf_q)), _dp(_f64(eskf_t)), _dp(out))
    return out[0:4].copy(), out[4:7].copy()


def _ext(R_il, t_il):
    return _f64(np.eye(3) if R_il is None else R_il).ravel(), _f64(np.zeros(3) if t_il is None else t_il)


def distort_frame(raw, relative_time_ms, imu_states, time_frame_begin, mode, R_il=None, t_il=None, imu_point_in=None, backend="plain"):
    """distortFrameByConstant (mode 1) / distortFrameByImu (mode 0) (utility.cpp:203-306) -> imu_point, number written."""
    lib = load(backend)
    r = _f64(raw, (-1, 3)); rel = _f64(relative_time_ms); st = _f64(imu_states, (-1, 17))
    R, t = _ext(R_il, t_il)
    imu = np.zeros_like(r) if imu_point_in is None else _f64(imu_point_in, (-1, 3)).copy()
    if mode == 1:
        lib.orc_distort_frame_by_constant(_vp(r), _vp(rel), len(r), _vp(st), len(st), float(time_frame_begin), _dp(R), _dp(t), _vp(imu))
        return imu, len(r)
    if mode == 0:
        k = lib.orc_distort_frame_by_imu(_vp(r), _vp(rel), len(r), _vp(st), len(st), float(time_frame_begin), _dp(R), _dp(t), _vp(imu))
        return imu, k
    return imu, 0


def transform_all_imu_point(imu_point, imu_states, R_il=None, t_il=None, backend="plain"):
    lib = load(backend)
    p = _f64(imu_point, (-1, 3)); st = _f64(imu_states, (-1, 17))
    R, t = _ext(R_il, t_il)
    out = np.empty_like(p)
    lib.orc_transform_all_imu_point(_vp(p), len(p), _vp(st), len(st), _dp(R), _dp(t), _vp(out))
    return out


def build_frame_order(point_xyz, sample_size, do_subsample=True, backend="plain"):
    lib = load(backend)
    p = _f64(point_xyz, (-1, 3))
    order = np.empty(max(len(p), 1), dtype=np.int32)
    m = lib.orc_build_frame_order(_vp(p), len(p), float(sample_size), int(bool(do_subsample)), _vp(order))
    return order[:m].copy()


def make_point_timestamp(timestamp, time_begin, time_end, point_time_enable=True, backend="plain"):
    lib = load(backend)
    ts = _f64(timestamp)
    rel = np.zeros_like(ts); alpha = np.zeros_like(ts); keep = np.zeros(len(ts), dtype=np.uint8)
    lib.orc_make_point_timestamp(_vp(ts), len(ts), float(time_begin), float(time_end), int(bool(point_time_enable)), _vp(rel), _vp(alpha), _vp(keep))
    return rel, alpha, keep.astype(bool)


def mt19937_64_nth(n, backend="plain"):
    return int(load(backend).orc_mt19937_64_nth(int(n)))


EIG_EIGEN_QL, EIG_JACOBI = 0, 1


def eig3(A, solver=EIG_EIGEN_QL, backend="plain"):
    """SelfAdjointEigenSolver<Matrix3d> restated: (eigenvalues ascending, eigenvectors as columns, converged)."""
    lib = load(backend)
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(9)
    ev = np.empty(3); V = np.empty(9)
    rc = lib.orc_eig3_solver(int(solver), _dp(A), _dp(ev), _dp(V))
    return ev, V.reshape(3, 3), rc == 0


class eig_solver:
    """with eig_solver(EIG_JACOBI): ...  -- switches the solver inside computeNeighborhoodDistribution (both backends)."""

    def __init__(self, solver):
        self.solver = solver

    def __enter__(self):
        self.libs = [load("plain")] + ([load("tsl")] if os.path.exists(LIB_TSL) else [])
        self.old = [lib.orc_get_eig_solver() for lib in self.libs]
        for lib in self.libs:
            lib.orc_set_eig_solver(int(self.solver))

    def __exit__(self, *exc):
        for lib, o in zip(self.libs, self.old):
            lib.orc_set_eig_solver(o)


def heap_topk(distances, K, backend="plain"):
    """optimize.cpp:394-404,411-422 with the real std::priority_queue on a list of distances: read-out order of indices."""
    d = np.ascontiguousarray(distances, dtype=np.float64)
    out = np.empty(K, dtype=np.int32)
    n = load(backend).orc_heap_topk(_vp(d), len(d), int(K), _vp(out))
    return out[:n].copy()


class threads:
    """with threads(n): ...  -- the all-cores CPU baseline: keypoint blocks visited in parallel, committed in order."""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        self.libs = [load("plain")] + ([load("tsl")] if os.path.exists(LIB_TSL) else [])
        self.old = [lib.orc_get_threads() for lib in self.libs]
        for lib in self.libs:
            lib.orc_set_threads(self.n)

    def __exit__(self, *exc):
        for lib, o in zip(self.libs, self.old):
            lib.orc_set_threads(o)
