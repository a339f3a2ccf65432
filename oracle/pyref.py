"""ctypes binding of oracle/_ref/libref_path.so -- the REFERENCE'S OWN translation units behind a C ABI.

TEST INFRASTRUCTURE ONLY (same rule as oracle/pyoracle.py: tests/, tools/ and the golden generators may import it, the
product package may not).  The library is built by `make -C oracle refpath` from /root/reference/src/{optimize,
lioOptimization,eskfEstimator,utility,state,cloudMap,parameters}.cpp compiled where they lie, against the stand-in
third-party headers of oracle/ref_shim/ (see oracle/ref_harness.cpp).  It exists in the build container and travels to the GPU box as a prebuilt
file; where it is missing, `available()` is False and the tests that need it skip.
"""
import ctypes as C
import os

import numpy as np

from .pyoracle import OrcOpts, _dp, _f64, _vp

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_REF = os.path.join(_HERE, "_ref", "libref_path.so")
_lib = None


def available():
    return os.path.exists(LIB_REF)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not available():
        raise RuntimeError(f"reference-path library missing: {LIB_REF} (run `make -C oracle refpath` where /root/reference exists)")
    lib = C.CDLL(LIB_REF)
    p, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
    lib.ref_describe.restype = C.c_char_p
    lib.ref_map_create.restype = p
    lib.ref_map_destroy.argtypes = [p]
    lib.ref_map_import.argtypes = [p, p, p, p, C.c_int, C.c_int]; lib.ref_map_import.restype = C.c_int
    lib.ref_map_num_voxels.argtypes = [p]; lib.ref_map_num_voxels.restype = C.c_int
    lib.ref_voxel_hash.argtypes = [C.c_int16, C.c_int16, C.c_int16]; lib.ref_voxel_hash.restype = C.c_uint64
    lib.ref_search_neighbors.argtypes = [p, dp, C.c_int, C.c_double, C.c_int, C.c_int, p, p]; lib.ref_search_neighbors.restype = C.c_int
    lib.ref_neighborhood.argtypes = [p, C.c_int, dp, dp, dp, dp]; lib.ref_neighborhood.restype = C.c_int
    lib.ref_build_plane_residuals.argtypes = [p, C.POINTER(OrcOpts), p, C.c_int, dp, dp, dp, dp, dp, C.c_int, p, p, ip, ip, dp]
    lib.ref_build_plane_residuals.restype = C.c_int
    lib.ref_eskf_create.restype = p
    lib.ref_eskf_destroy.argtypes = [p]
    for name in ("ref_eskf_get_state", "ref_eskf_set_state", "ref_eskf_get_cov", "ref_eskf_set_cov", "ref_eskf_observe"):
        getattr(lib, name).argtypes = [p, dp]
    lib.ref_eskf_set_noise.argtypes = [p, C.c_double, C.c_double, C.c_double, C.c_double]
    lib.ref_eskf_set_cov_scales.argtypes = [p, C.c_double, C.c_double, C.c_double, C.c_double]
    lib.ref_eskf_init_imu.argtypes = [p, dp, dp]
    lib.ref_eskf_predict.argtypes = [p, C.c_double, dp, dp]
    lib.ref_eskf_try_init.argtypes = [p, dp, dp, dp, C.c_int, C.c_double, dp]; lib.ref_eskf_try_init.restype = C.c_int
    lib.ref_update_iekf.argtypes = [p, p, C.POINTER(OrcOpts), p, C.c_int, dp, dp, dp, dp, C.c_int, C.c_double, ip, p]
    lib.ref_update_iekf.restype = C.c_int
    lib.ref_optimize.argtypes = [p, p, C.POINTER(OrcOpts), p, p, C.c_int, C.c_double, dp, dp, dp, dp, C.c_int, C.c_double, ip, p]
    lib.ref_optimize.restype = C.c_int
    lib.ref_transform_points.argtypes = [p, C.c_int, dp, dp, dp, dp, p]
    lib.ref_grid_sampling.argtypes = [p, C.c_int, C.c_double, p]; lib.ref_grid_sampling.restype = C.c_int
    lib.ref_distort_frame_by_constant.argtypes = [p, p, C.c_int, p, C.c_int, C.c_double, dp, dp, p]
    lib.ref_distort_frame_by_imu.argtypes = [p, p, C.c_int, p, C.c_int, C.c_double, dp, dp, p]
    lib.ref_transform_all_imu_point.argtypes = [p, C.c_int, p, C.c_int, dp, dp, p]
    lib.ref_angular_distance_so3.argtypes = [dp]; lib.ref_angular_distance_so3.restype = C.c_double
    lib.ref_so3_to_rot.argtypes = [dp, dp]; lib.ref_so3_to_quat.argtypes = [dp, dp]; lib.ref_rot_to_so3.argtypes = [dp, dp]
    lib.ref_derivative_s2.argtypes = [dp, dp]
    lib.ref_param_set_num.argtypes = [C.c_char_p, dp, C.c_int]
    lib.ref_param_set_str.argtypes = [C.c_char_p, C.c_char_p]
    lib.ref_node_create.argtypes = [C.c_int]; lib.ref_node_create.restype = p
    lib.ref_node_destroy.argtypes = [p]
    lib.ref_node_push_imu.argtypes = [p, C.c_double, dp, dp]
    lib.ref_node_push_image_time.argtypes = [p, C.c_double]
    lib.ref_node_push_points.argtypes = [p, p, p, C.c_int]
    lib.ref_node_run.argtypes = [p, dp]; lib.ref_node_run.restype = C.c_int
    lib.ref_node_last_frame_info.argtypes = [p, dp, dp, ip]; lib.ref_node_last_frame_info.restype = C.c_int
    lib.ref_node_last_frame_points.argtypes = [p, p, p, p, p, p, p]
    lib.ref_node_eskf_get.argtypes = [p, dp, dp]
    lib.ref_node_map_num_voxels.argtypes = [p]; lib.ref_node_map_num_voxels.restype = C.c_int
    lib.ref_node_map_export.argtypes = [p, C.c_int, p, p, p]; lib.ref_node_map_export.restype = C.c_int
    lib.ref_node_add_points_to_map.argtypes = [p, p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int]; lib.ref_node_add_points_to_map.restype = C.c_int
    lib.ref_node_state_initialization.argtypes = [p, C.c_int, C.c_int, dp, dp, dp, dp, dp]
    lib.ref_node_make_point_timestamp.argtypes = [p, p, C.c_int, C.c_double, C.c_double, p, p, p]; lib.ref_node_make_point_timestamp.restype = C.c_int
    _lib = lib
    return lib


def _ext(R_il, t_il):
    return _f64(np.eye(3) if R_il is None else R_il).ravel(), _f64(np.zeros(3) if t_il is None else t_il)


class Map:
    """tsl::robin_map<voxel, voxelBlock> of rgbPoint (include/cloudMap.h), filled from exported arrays."""

    def __init__(self, keys, counts, xyz, cap=20):
        self.lib = load()
        self.h = C.c_void_p(self.lib.ref_map_create())
        keys = np.ascontiguousarray(keys, dtype=np.int16); counts = np.ascontiguousarray(counts, dtype=np.int32)
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        if self.lib.ref_map_import(self.h, _vp(keys), _vp(counts), _vp(xyz), len(counts), cap) != 0:
            raise ValueError("ref_map_import: duplicate voxel key")

    @classmethod
    def from_oracle(cls, orc_map, cap=20):
        return cls(*orc_map.export(cap), cap=cap)

    def __del__(self):
        try:
            if self.h:
                self.lib.ref_map_destroy(self.h)
        except Exception:
            pass

    def num_voxels(self):
        return int(self.lib.ref_map_num_voxels(self.h))

    def search_neighbors(self, p, nb=1, size=1.0, K=20, thr=1):
        xyz = np.zeros((K, 3)); vox = np.zeros((K, 3), dtype=np.int16)
        n = self.lib.ref_search_neighbors(self.h, _dp(_f64(p)), nb, size, K, thr, _vp(xyz), _vp(vox))
        return dict(n=n, xyz=xyz[:n], voxels=vox[:n])

    def build_plane_residuals(self, opts, raw, q, t, t_last, R_il=None, t_il=None, frame_id=100):
        raw = _f64(raw, (-1, 3)); n = len(raw)
        R, tl = _ext(R_il, t_il)
        pw = np.zeros((n, 3)); res = np.zeros((max(n, 1), 15))
        succ, nres = C.c_int(), C.c_int(); loss = C.c_double()
        m = self.lib.ref_build_plane_residuals(self.h, C.byref(opts), _vp(raw), n, _dp(_f64(q)), _dp(_f64(t)), _dp(_f64(t_last)), _dp(R), _dp(tl),
                                               frame_id, _vp(pw), _vp(res), C.byref(succ), C.byref(nres), C.byref(loss))
        if m < 0:
            return dict(rc=m)
        r = res[:m]
        return dict(rc=m, point_world=pw, location=r[:, 0:3].copy(), normal=r[:, 3:6].copy(), jacobian=r[:, 6:12].copy(), norm_offset=r[:, 12].copy(),
                    distance=r[:, 13].copy(), weight=r[:, 14].copy(), success=succ.value, num_residuals=nres.value, loss=loss.value)


def neighborhood(pts):
    lib = load()
    p = _f64(pts, (-1, 3))
    c = np.zeros(3); nrm = np.zeros(3); cov = np.zeros(9); a2d = C.c_double()
    rc = lib.ref_neighborhood(_vp(p), len(p), _dp(c), _dp(nrm), _dp(cov), C.byref(a2d))
    return dict(rc=rc, center=c, normal=nrm, cov=cov.reshape(3, 3), a2D=a2d.value)


class Eskf:
    """eskfEstimator of the reference (src/eskfEstimator.cpp); same accessors as pyoracle.Eskf."""

    def __init__(self):
        self.lib = load()
        self.h = C.c_void_p(self.lib.ref_eskf_create())

    def __del__(self):
        try:
            if self.h:
                self.lib.ref_eskf_destroy(self.h)
        except Exception:
            pass

    def get_state(self):
        s = np.empty(19); self.lib.ref_eskf_get_state(self.h, _dp(s)); return s

    def set_state(self, s):
        self.lib.ref_eskf_set_state(self.h, _dp(_f64(s)))

    def get_cov(self):
        P = np.empty(289); self.lib.ref_eskf_get_cov(self.h, _dp(P)); return P.reshape(17, 17)

    def set_cov(self, P):
        self.lib.ref_eskf_set_cov(self.h, _dp(_f64(P).ravel()))

    def set_noise(self, a, g, ba, bg):
        self.lib.ref_eskf_set_noise(self.h, a, g, ba, bg)

    def set_cov_scales(self, a, g, ba, bg):
        self.lib.ref_eskf_set_cov_scales(self.h, a, g, ba, bg)

    def init_imu(self, acc0, gyr0):
        self.lib.ref_eskf_init_imu(self.h, _dp(_f64(acc0)), _dp(_f64(gyr0)))

    def predict(self, dt, acc1, gyr1):
        self.lib.ref_eskf_predict(self.h, dt, _dp(_f64(acc1)), _dp(_f64(gyr1)))

    def observe(self, dx):
        self.lib.ref_eskf_observe(self.h, _dp(_f64(dx)))

    def try_init(self, t, gyr, acc, g_norm):
        t = _f64(t); g = _f64(gyr, (-1, 3)); a = _f64(acc, (-1, 3))
        o = np.zeros(14)
        rc = self.lib.ref_eskf_try_init(self.h, _dp(t), _dp(g), _dp(a), len(t), float(g_norm), _dp(o))
        return rc, dict(mean_gyr=o[0:3].copy(), mean_acc=o[3:6].copy(), gyr_cov=o[6:9].copy(), acc_cov=o[9:12].copy(),
                        num_init_meas=int(o[12]), initial_flag=bool(o[13]))


def reset_globals():
    load().ref_reset_globals()


def update_iekf(m, e, opts, raw, state, t_last, R_il=None, t_il=None, frame_id=100, laser_point_cov=0.001):
    raw = _f64(raw, (-1, 3)); st = _f64(state).copy()
    R, tl = _ext(R_il, t_il)
    nres = C.c_int(); pw = np.zeros((len(raw), 3))
    rc = m.lib.ref_update_iekf(m.h, e.h, C.byref(opts), _vp(raw), len(raw), _dp(st), _dp(_f64(t_last)), _dp(R), _dp(tl), frame_id,
                               laser_point_cov, C.byref(nres), _vp(pw))
    return dict(rc=rc, state=st, num_residuals=nres.value, point_world=pw)


def optimize(m, e, opts, frame_raw, frame_point, sample_voxel_size, state, t_last, R_il=None, t_il=None, frame_id=100, laser_point_cov=0.001):
    fr = _f64(frame_raw, (-1, 3)); fp = _f64(frame_point, (-1, 3)); st = _f64(state).copy()
    R, tl = _ext(R_il, t_il)
    nres = C.c_int(); out = np.zeros_like(fp)
    rc = m.lib.ref_optimize(m.h, e.h, C.byref(opts), _vp(fr), _vp(fp), len(fr), float(sample_voxel_size), _dp(st), _dp(_f64(t_last)), _dp(R), _dp(tl),
                            frame_id, laser_point_cov, C.byref(nres), _vp(out))
    return dict(rc=rc, state=st, num_residuals=nres.value, frame_point=out)


def transform_points(raw, q, t, R_il=None, t_il=None):
    lib = load()
    r = _f64(raw, (-1, 3)); out = np.empty_like(r)
    R, tl = _ext(R_il, t_il)
    lib.ref_transform_points(_vp(r), len(r), _dp(_f64(q)), _dp(_f64(t)), _dp(R), _dp(tl), _vp(out))
    return out


def grid_sampling(world, size_voxel):
    lib = load()
    w = _f64(world, (-1, 3))
    idx = np.empty(max(len(w), 1), dtype=np.int32)
    m = lib.ref_grid_sampling(_vp(w), len(w), float(size_voxel), _vp(idx))
    return idx[:m].copy()


def distort_frame(raw, relative_time_ms, imu_states, time_frame_begin, mode, R_il=None, t_il=None, imu_point_in=None):
    lib = load()
    r = _f64(raw, (-1, 3)); rel = _f64(relative_time_ms); st = _f64(imu_states, (-1, 17))
    R, t = _ext(R_il, t_il)
    imu = np.zeros_like(r) if imu_point_in is None else _f64(imu_point_in, (-1, 3)).copy()
    fn = lib.ref_distort_frame_by_constant if mode == 1 else lib.ref_distort_frame_by_imu
    fn(_vp(r), _vp(rel), len(r), _vp(st), len(st), float(time_frame_begin), _dp(R), _dp(t), _vp(imu))
    return imu


def transform_all_imu_point(imu_point, imu_states, R_il=None, t_il=None):
    lib = load()
    p = _f64(imu_point, (-1, 3)); st = _f64(imu_states, (-1, 17))
    R, t = _ext(R_il, t_il)
    out = np.empty_like(p)
    lib.ref_transform_all_imu_point(_vp(p), len(p), _vp(st), len(st), _dp(R), _dp(t), _vp(out))
    return out


def angular_distance_so3(w):
    return float(load().ref_angular_distance_so3(_dp(_f64(w))))


def so3_to_rot(w):
    R = np.zeros(9); load().ref_so3_to_rot(_dp(_f64(w)), _dp(R)); return R.reshape(3, 3)


def so3_to_quat(w):
    q = np.zeros(4); load().ref_so3_to_quat(_dp(_f64(w)), _dp(q)); return q


def rot_to_so3(R):
    w = np.zeros(3); load().ref_rot_to_so3(_dp(_f64(R).ravel()), _dp(w)); return w


def derivative_s2(g):
    B = np.zeros(6); load().ref_derivative_s2(_dp(_f64(g)), _dp(B)); return B.reshape(3, 2)


# ---------------------------------------------------------------------------------------------------------------------
# the node (src/lioOptimization.cpp): its own constructor / readParameters / imuHandler / getMeasurements / run / process
MOTION_COMPENSATION = {0: "IMU", 1: "CONSTANT_VELOCITY"}                 # utility.h:82-86 <- readParameters' strings
INITIALIZATION = {0: "INIT_IMU", 1: "INIT_CONSTANT_VELOCITY"}            # utility.h:88-92


def set_params(num=None, strs=None, clear=True):
    """Fill the stand-in parameter server the node's readParameters() (src/lioOptimization.cpp:249-349) reads."""
    lib = load()
    if clear:
        lib.ref_param_clear()
    for k, v in (num or {}).items():
        a = _f64(np.atleast_1d(v))
        lib.ref_param_set_num(k.encode(), _dp(a), len(a))
    for k, v in (strs or {}).items():
        lib.ref_param_set_str(k.encode(), str(v).encode())


def params_from_options(oo, icp, gravity=(0.0, 0.0, 9.81)):
    """odometry options dict (as tests/replay_reference.py takes it) + OrcOpts -> parameter names of the reference's yaml."""
    num = {"common/gravity_acc": gravity,
           "imu_parameter/acc_cov": oo["acc_cov"], "imu_parameter/gyr_cov": oo["gyr_cov"],
           "imu_parameter/b_acc_cov": oo["b_acc_cov"], "imu_parameter/b_gyr_cov": oo["b_gyr_cov"]}
    for k in ("init_voxel_size", "init_sample_voxel_size", "init_num_frames", "voxel_size", "sample_voxel_size", "max_num_points_in_voxel",
              "min_distance_points"):
        num["odometry_options/" + k] = oo[k]
    for k in ("threshold_voxel_occupancy", "size_voxel_map", "num_iters_icp", "min_number_neighbors", "voxel_neighborhood", "power_planarity",
              "max_number_neighbors", "max_dist_to_plane_icp", "threshold_orientation_norm", "threshold_translation_norm", "max_num_residuals",
              "weight_alpha", "weight_neighborhood"):
        num["icp_options/" + k] = getattr(icp, k)
    num["icp_options/debug_print"] = 0
    strs = {"odometry_options/motion_compensation": MOTION_COMPENSATION[oo["motion_compensation"]],
            "odometry_options/initialization": INITIALIZATION.get(oo["initialization"], "NONE")}
    return num, strs


class Node:
    def __init__(self, point_time_enable=True):
        self.lib = load()
        reset_globals()
        self.h = C.c_void_p(self.lib.ref_node_create(int(bool(point_time_enable))))

    def close(self):
        if self.h:
            self.lib.ref_node_destroy(self.h)
            self.h = None
        reset_globals()

    def push_imu(self, t, acc, gyr):
        for ti, a, g in zip(_f64(t), _f64(acc, (-1, 3)), _f64(gyr, (-1, 3))):
            self.lib.ref_node_push_imu(self.h, float(ti), _dp(np.ascontiguousarray(a)), _dp(np.ascontiguousarray(g)))

    def push_image_time(self, t):
        self.lib.ref_node_push_image_time(self.h, float(t))

    def push_points(self, raw, timestamp):
        r = _f64(raw, (-1, 3)); ts = _f64(timestamp)
        self.lib.ref_node_push_points(self.h, _vp(r), _vp(ts), len(r))

    def run(self):
        info = np.zeros(8)
        rc = self.lib.ref_node_run(self.h, _dp(info))
        return dict(rc=rc, index_frame=int(info[0]), initial_flag=bool(info[1]), window=int(info[2]), last_frame_id=int(info[3]),
                    points_left=int(info[4]), map_points=int(info[5]), map_voxels=int(info[6]), current_time=float(info[7]))

    def last_frame(self):
        st = np.zeros(16); tm = np.zeros(2); fid = C.c_int()
        n = self.lib.ref_node_last_frame_info(self.h, _dp(st), _dp(tm), C.byref(fid))
        if n < 0:
            return None
        raw = np.zeros((n, 3)); pt = np.zeros((n, 3)); imu = np.zeros((n, 3)); al = np.zeros(n); rel = np.zeros(n); ts = np.zeros(n)
        if n:
            self.lib.ref_node_last_frame_points(self.h, _vp(raw), _vp(pt), _vp(imu), _vp(al), _vp(rel), _vp(ts))
        return dict(state=st, time_sweep_begin=tm[0], time_sweep_end=tm[1], frame_id=fid.value, raw_point=raw, point=pt, imu_point=imu,
                    alpha_time=al, relative_time=rel, timestamp=ts)

    def eskf(self):
        s = np.zeros(19); P = np.zeros(289)
        self.lib.ref_node_eskf_get(self.h, _dp(s), _dp(P))
        return s, P.reshape(17, 17)

    def map_export(self, cap=20):
        V = self.lib.ref_node_map_num_voxels(self.h)
        keys = np.zeros((V, 3), dtype=np.int16); counts = np.zeros(V, dtype=np.int32); xyz = np.zeros((V, cap, 3), dtype=np.float32)
        self.lib.ref_node_map_export(self.h, cap, _vp(keys), _vp(counts), _vp(xyz))
        return keys, counts, xyz

    def add_points_to_map(self, world, voxel_size=1.0, cap=20, min_dist=0.15, min_num_points=0):
        w = _f64(world, (-1, 3))
        return self.lib.ref_node_add_points_to_map(self.h, _vp(w), len(w), voxel_size, cap, min_dist, min_num_points)

    def state_initialization(self, index_frame, initial_flag, prev2, prev1, eskf_q=(1, 0, 0, 0), eskf_t=(0, 0, 0)):
        out = np.zeros(7)
        self.lib.ref_node_state_initialization(self.h, int(index_frame), int(bool(initial_flag)), _dp(_f64(prev2)), _dp(_f64(prev1)),
                                               _dp(_f64(eskf_q)), _dp(_f64(eskf_t)), _dp(out))
        return out[0:4].copy(), out[4:7].copy()

    def make_point_timestamp(self, timestamp, time_begin, time_end):
        ts = _f64(timestamp)
        rel = np.zeros_like(ts); alpha = np.zeros_like(ts); keep = np.zeros(len(ts), dtype=np.int32)
        m = self.lib.ref_node_make_point_timestamp(self.h, _vp(ts), len(ts), float(time_begin), float(time_end), _vp(rel), _vp(alpha), _vp(keep))
        return rel[:m], alpha[:m], keep[:m]


def map_as_dict(keys, counts, xyz):
    """{voxel key: points in slot order} -- container iteration order does not matter for equality."""
    return {tuple(int(c) for c in k): xyz[i, : counts[i]].copy() for i, k in enumerate(keys)}
