"""ctypes binding of oracle/_ref/libref_path.so -- the REFERENCE'S OWN translation units behind a C ABI.

TEST INFRASTRUCTURE ONLY (same rule as oracle/pyoracle.py: tests/, tools/ and the golden generators may import it, the
product package may not).  The library is built by `make -C oracle refpath` from /root/reference/src/{optimize,
eskfEstimator,utility,state,cloudMap}.cpp compiled where they lie, against the stand-in third-party headers of
oracle/ref_shim/ (see oracle/ref_harness.cpp).  It exists in the build container and travels to the GPU box as a prebuilt
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
