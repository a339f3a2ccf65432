"""ctypes bindings of libsrlivo_hip.so (include/srlivo_hip.h + include/srlivo_host.h).

Mirrors the C declarations one to one; numpy arrays are passed as plain pointers.  Nothing here
computes: if the shared library is missing, or no HIP device exists, the calls raise SrlError.
"""
import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRL_LIB_PATH") or os.path.join(_HERE, "libsrlivo_hip.so")   # env override: A/B builds only
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

SRL_OK = 0
SRL_ERR_NO_DEVICE = -1
SRL_ERR_COMM = -7
SRL_ERR_NAN_PLANARITY = -8
SRL_ERR_NOT_ENOUGH_RESIDUALS = -9
SRL_COMM_ID_BYTES = 128


SRL_PEER_HANDLE_BYTES = 64


class SrlError(RuntimeError):
    def __init__(self, status, what, detail=""):
        self.status = status
        super().__init__(f"{what}: status {status} {detail}".strip())


class IcpOpts(C.Structure):
    _fields_ = [
        ("threshold_voxel_occupancy", C.c_int32), ("init_num_frames", C.c_int32),
        ("size_voxel_map", C.c_double), ("num_iters_icp", C.c_int32),
        ("min_number_neighbors", C.c_int32), ("voxel_neighborhood", C.c_int32),
        ("power_planarity", C.c_double), ("max_number_neighbors", C.c_int32),
        ("max_dist_to_plane_icp", C.c_double), ("threshold_orientation_norm", C.c_double),
        ("threshold_translation_norm", C.c_double), ("max_num_residuals", C.c_int32),
        ("weight_alpha", C.c_double), ("weight_neighborhood", C.c_double),
    ]
    # test hook, NOT part of the struct (srl_icp_opts = the reference's icpOptions fields only): default_opts(select_mode=...) keeps the
    # wish as a Python attribute and the wrappers hand it to srl_debug_set_select_mode before the call
    select_mode = 0


class Frame(C.Structure):
    _fields_ = [("q", C.c_double * 4), ("t", C.c_double * 3), ("t_last", C.c_double * 3),
                ("R_il", C.c_double * 9), ("t_il", C.c_double * 3), ("frame_id", C.c_int32)]


class NormalEq(C.Structure):
    _fields_ = [("HtH", C.c_double * 36), ("Hth", C.c_double * 6), ("loss_sum", C.c_double),
                ("num_residuals", C.c_int32), ("success", C.c_int32), ("sum_candidates", C.c_int64),
                ("last_visited", C.c_int64), ("nan_error", C.c_int32), ("num_fallback", C.c_int32)]



class Timing(C.Structure):
    _fields_ = [("assoc_ms", C.c_float), ("reduce_ms", C.c_float), ("total_ms", C.c_float), ("calls", C.c_int32),
                ("algorithmic_bytes", C.c_int64), ("sum_assoc_ms", C.c_double), ("sum_reduce_ms", C.c_double),
                ("sum_total_ms", C.c_double), ("sum_algorithmic_bytes", C.c_int64), ("sum_keypoints", C.c_int64),
                ("sum_host_launch_us", C.c_double), ("sum_host_wait_us", C.c_double), ("sum_host_total_us", C.c_double),
                ("sum_passes", C.c_int64)]


class ImuState(C.Structure):
    _fields_ = [("timestamp", C.c_double), ("un_acc", C.c_double * 3), ("un_gyr", C.c_double * 3), ("trans", C.c_double * 3),
                ("quat", C.c_double * 4), ("vel", C.c_double * 3)]


class OdometryOpts(C.Structure):
    _fields_ = [("init_voxel_size", C.c_double), ("init_sample_voxel_size", C.c_double), ("init_num_frames", C.c_int),
                ("num_for_initialization", C.c_int), ("voxel_size", C.c_double), ("sample_voxel_size", C.c_double),
                ("max_num_points_in_voxel", C.c_int), ("min_distance_points", C.c_double), ("motion_compensation", C.c_int),
                ("initialization", C.c_int), ("point_time_enable", C.c_int), ("acc_cov", C.c_double), ("gyr_cov", C.c_double),
                ("b_acc_cov", C.c_double), ("b_gyr_cov", C.c_double), ("icp", IcpOpts)]


class ReplayResult(C.Structure):
    _fields_ = [("processed", C.c_int), ("initialized", C.c_int), ("index_frame", C.c_int), ("success", C.c_int),
                ("num_residuals_used", C.c_int), ("iterations", C.c_int), ("frame_points", C.c_int), ("keypoints", C.c_int),
                ("points_added", C.c_int), ("state", C.c_double * 16)]


MC_IMU, MC_CONSTANT_VELOCITY, MC_NONE = 0, 1, 2

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p)
PROVIDER_FN = C.CFUNCTYPE(C.c_int, C.POINTER(Frame), C.POINTER(IcpOpts), C.POINTER(NormalEq), C.c_void_p)

_lib = None


def load_library():
    """Load libsrlivo_hip.so; raises SrlError (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SrlError(-1, "libsrlivo_hip.so not built",
                       f"(expected {LIB_PATH}; run `python -c 'import __graft_entry__ as g; g.build()'`)")
    lib = C.CDLL(LIB_PATH)
    p = C.c_void_p
    dp = C.POINTER(C.c_double)
    sig = {
        "srl_device_count": ([C.POINTER(C.c_int)], C.c_int),
        "srl_ctx_create": ([C.c_int, C.POINTER(p)], C.c_int),
        "srl_ctx_destroy": ([p], C.c_int),
        "srl_last_error": ([p], C.c_char_p),
        "srl_status_str": ([C.c_int], C.c_char_p),
        "srl_icp_opts_default": ([C.POINTER(IcpOpts)], None),
        "srl_map_upload": ([p, p, p, p, C.c_int, C.c_int], C.c_int),
        "srl_map_insert": ([p, p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int)], C.c_int),
        "srl_map_size": ([p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)], C.c_int),
        "srl_map_download": ([p, p, p, p, C.c_int], C.c_int),
        "srl_map_probe_checksum": ([p, p, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_uint64)], C.c_int),
        "srl_sweep_upload": ([p, p, C.c_int], C.c_int),
        "srl_sweep_shard": ([p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
        "srl_sweep_prefetch": ([p, p, C.c_int], C.c_int),
        "srl_sweep_swap": ([p], C.c_int),
        "srl_sweep_wait": ([p], C.c_int),
        "srl_thread_pin_to_gpu_numa": ([p, C.POINTER(C.c_int)], C.c_int),
        "srl_pinned_alloc": ([C.c_size_t, C.POINTER(p)], C.c_int),
        "srl_pinned_free": ([p], C.c_int),
        "srl_host_register": ([p, C.c_size_t], C.c_int),
        "srl_host_unregister": ([p], C.c_int),
        "srl_comm_backend_info": ([C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
        "srl_build_residuals": ([p, C.POINTER(Frame), C.POINTER(IcpOpts), C.POINTER(NormalEq)], C.c_int),
        "srl_build_residuals_overlap": ([p, C.POINTER(Frame), C.POINTER(IcpOpts), C.POINTER(NormalEq), p, p], C.c_int),
        "srl_set_taps": ([p, C.c_int], C.c_int),
        "srl_fetch_neighbors": ([p, p, p, p], C.c_int),
        "srl_fetch_residuals": ([p, p, p, p, p, p, p], C.c_int),
        "srl_search_neighbors": ([p, p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, p, p, p], C.c_int),
        "srl_transform_points": ([p, p, C.c_int, dp, dp, dp, dp, p], C.c_int),
        "srl_frame_upload": ([p, p, C.c_int], C.c_int),
        "srl_frame_undistort": ([p, p, p, p, C.c_int, p, C.c_int, C.c_double, C.c_int, dp, dp, p, p], C.c_int),
        "srl_frame_take": ([p, p, C.c_int], C.c_int),
        "srl_frame_size": ([p, C.POINTER(C.c_int)], C.c_int),
        "srl_frame_select_keypoints": ([p, dp, dp, dp, dp, C.c_double, p, C.POINTER(C.c_int)], C.c_int),
        "srl_frame_commit": ([p, dp, dp, dp, dp, C.c_double, C.c_int, C.c_double, C.c_int, p, C.POINTER(C.c_int)], C.c_int),
        "srl_comm_unique_id": ([p], C.c_int),
        "srl_comm_init_rank": ([p, C.c_int, C.c_int, p], C.c_int),
        "srl_comm_destroy": ([p], C.c_int),
        "srl_comm_set_library": ([C.c_char_p], C.c_int),
        "srl_comm_suspend": ([p, C.c_int], C.c_int),
        "srl_comm_info": ([p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64)], C.c_int),
        "srl_comm_set_host_callbacks": ([p, C.c_int, C.c_int, ALLREDUCE_FN, ALLGATHER_FN, p], C.c_int),
        "srl_debug_set_gather_counts": ([p, C.c_int, C.c_int, C.POINTER(C.c_int64)], C.c_int),
        "srl_peer_export": ([p, p, C.POINTER(p)], C.c_int),
        "srl_peer_attach": ([p, C.c_int, C.c_int, p, C.POINTER(p)], C.c_int),
        "srl_peer_set_deadline_ms": ([p, C.c_int], C.c_int),
        "srl_peer_stats": ([p, C.POINTER(C.c_int64), C.POINTER(C.c_int)], C.c_int),
        "srl_peer_detach": ([p], C.c_int),
        "srl_shard_range": ([C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)], None),
        "srl_shard_budget": ([C.c_int, C.POINTER(C.c_int64), C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int)], None),
        "srl_get_timing": ([p, C.POINTER(Timing)], C.c_int),
        "srl_debug_block_times": ([p, p, C.c_int, C.POINTER(C.c_int)], C.c_int),
        "srl_debug_set_ablate": ([p, C.c_int], C.c_int),
        "srl_debug_set_fused_reduce": ([p, C.c_int], C.c_int),
        "srl_debug_set_bound_culling": ([p, C.c_int], C.c_int),
        "srl_debug_set_pose_box": ([p, C.c_int], C.c_int),
        "srl_debug_set_arm_linger": ([p, C.c_double, C.c_double], C.c_int),
        "srl_debug_pass_stamps": ([p, C.c_int, p, p], C.c_int),
        "srl_debug_frame_timing": ([p, C.c_int, p], C.c_int),
        "srl_debug_set_frame_epoch": ([p, C.c_int], C.c_int),
        "srl_set_armed_launch": ([p, C.c_int], C.c_int),
        "srl_disarm": ([p], C.c_int),
        "srl_solve_end": ([p], C.c_int),
        "srl_get_arm_stats": ([p, C.POINTER(C.c_uint64)], C.c_int),
        "srl_debug_set_launch_shape": ([p, C.c_int, C.c_int], C.c_int),
        "srl_debug_set_search_select_mode": ([p, C.c_int], C.c_int),
        "srl_debug_set_select_mode": ([p, C.c_int], C.c_int),
        "srl_debug_set_frame_order_mode": ([p, C.c_int], C.c_int),
        "srl_debug_radix_sort_pairs": ([p, p, C.c_int, C.c_int, p, p], C.c_int),
        "srl_debug_frame_order_used": ([p, C.POINTER(C.c_int)], C.c_int),
        "srl_debug_heap_topk": ([p, C.c_int, C.c_int, p], C.c_int),
        "srl_debug_device_sqrt": ([p, p, C.c_int, p], C.c_int),
        "srl_set_profiling": ([p, C.c_int], C.c_int),
        "srl_set_profiling_period": ([p, C.c_int], C.c_int),
        "srl_timing_mark": ([p], C.c_int),
        # host mirror handles
        "srl_lio_create": ([C.c_int, C.POINTER(p)], C.c_int),
        "srl_lio_destroy": ([p], C.c_int),
        "srl_lio_ctx": ([p], p),
        "srl_lio_last_error": ([p], C.c_char_p),
        "srl_lio_set_extrinsics": ([p, dp, dp], C.c_int),
        "srl_lio_set_laser_point_cov": ([p, C.c_double], C.c_int),
        "srl_lio_last_solve_launches": ([p, C.POINTER(C.c_int)], C.c_int),
        "srl_lio_eskf_get_state": ([p, dp], C.c_int),
        "srl_lio_eskf_set_state": ([p, dp], C.c_int),
        "srl_lio_eskf_get_cov": ([p, dp], C.c_int),
        "srl_lio_eskf_set_cov": ([p, dp], C.c_int),
        "srl_lio_eskf_set_noise": ([p, C.c_double, C.c_double, C.c_double, C.c_double], C.c_int),
        "srl_lio_eskf_init_imu": ([p, dp, dp], C.c_int),
        "srl_lio_eskf_scale_init_cov": ([p], C.c_int),
        "srl_lio_eskf_predict": ([p, C.c_double, dp, dp], C.c_int),
        "srl_lio_eskf_observe": ([p, dp], C.c_int),
        "srl_lio_add_points_to_map": ([p, p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int], C.c_int),
        "srl_lio_map_size": ([p, C.POINTER(C.c_int64)], C.c_int),
        "srl_lio_probe_checksum_of_committed_frame": ([p, C.c_int, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)], C.c_int),
        "srl_lio_resident_sweep": ([p, p, C.c_int], C.c_int),
        "srl_lio_prefetch_sweep": ([p, p, C.c_int], C.c_int),
        "srl_lio_swap_sweep": ([p], C.c_int),
        "srl_lio_prefetch_sweep_during_solve": ([p, p, C.c_int], C.c_int),
        "srl_lio_update_iekf": ([p, C.POINTER(IcpOpts), p, C.c_int, dp, dp, C.c_int, p, C.c_int,
                                 C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
        "srl_lio_stream_step": ([p, C.POINTER(IcpOpts), dp, dp, C.c_int, dp, dp, C.c_int, p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
        "srl_lio_update_iekf_provided": ([p, C.POINTER(IcpOpts), PROVIDER_FN, p, C.c_int, dp, dp, C.c_int, p, C.c_int,
                                          C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
        "srl_lio_optimize": ([p, C.POINTER(IcpOpts), C.c_double, p, p, C.c_int, dp, dp, C.c_int, p,
                              C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
        "srl_lio_eskf_try_init": ([p, dp, dp, dp, C.c_int, C.POINTER(C.c_int)], C.c_int),
        "srl_lio_eskf_get_init_stats": ([p, dp], C.c_int),
        "srl_lio_set_initial_flag": ([p, C.c_int], C.c_int),
        "srl_lio_state_initialization": ([p, C.c_int, C.c_int, dp, dp, dp], C.c_int),
        "srl_lio_set_odometry_options": ([p, C.POINTER(OdometryOpts)], C.c_int),
        "srl_lio_run_measurement": ([p, C.c_double, p, p, p, C.c_int, p, p, C.c_int, C.c_double, C.c_double,
                                     C.POINTER(ReplayResult)], C.c_int),
        "srl_lio_last_frame": ([p, C.c_int, p, p, p, C.POINTER(C.c_int)], C.c_int),
        "srl_lio_optimize_resident": ([p, C.POINTER(IcpOpts), C.c_double, p, C.c_int, dp, dp, C.c_int, p,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
        "srl_lio_commit_frame": ([p, dp, C.c_double, C.c_int, C.c_double, C.c_int, p, C.POINTER(C.c_int)], C.c_int),
        "srl_lio_search_neighbors": ([p, dp, C.c_int, C.c_double, C.c_int, C.c_int, p, p, C.POINTER(C.c_int)], C.c_int),
        "srl_lio_neighborhood": ([p, p, C.c_int, dp, dp, dp, dp], C.c_int),
        "srl_lio_build_plane_residuals": ([p, C.POINTER(IcpOpts), p, C.c_int, dp, dp, C.c_int, p, C.c_int,
                                           C.POINTER(C.c_int), dp, C.POINTER(C.c_int), p], C.c_int),
        "srl_grid_sampling": ([p, C.c_int, C.c_double, p, C.POINTER(C.c_int)], C.c_int),
        "srl_debug_tr1_order": ([p, C.c_int, p], C.c_int),
        "srl_debug_tr1_order_by_relation": ([p, C.c_int, p], C.c_int),
    }
    for name, (argtypes, restype) in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def declared_symbols():
    """Every function name declared in include/srlivo_hip.h and include/srlivo_host.h."""
    names = []
    for hdr in ("srlivo_hip.h", "srlivo_hip_debug.h", "srlivo_host.h"):
        text = open(os.path.join(INCLUDE_DIR, hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(srl_[a-z0-9_]+)\s*\(", text)
    skip = {"srl_allreduce_fn", "srl_allgather_i64_fn", "srl_normal_eq_provider", "srl_probe_mix"}      # (types, and the inline helper of the header)
    return sorted(set(n for n in names if n not in skip))


def library_symbols():
    lib = load_library()
    return [n for n in declared_symbols() if hasattr(lib, n)]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def default_opts(**kw):
    o = IcpOpts()
    load_library().srl_icp_opts_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def heap_topk(distances, K):
    """The device kernels' libstdc++-heap restatement (csrc/srl_heap.h) run on the host: read-out order of candidate indices."""
    d = np.ascontiguousarray(distances, dtype=np.float64)
    out = np.empty(K, dtype=np.int32)
    n = load_library().srl_debug_heap_topk(_ptr(d), len(d), int(K), _ptr(out))
    if n < 0:
        raise SrlError(n, "srl_debug_heap_topk")
    return out[:n].copy()


class PinnedArray:
    """A float64 numpy array in page-locked host memory (srl_pinned_alloc): srl_sweep_upload DMAs straight out of it."""

    def __init__(self, shape):
        self.lib = load_library()
        n = int(np.prod(shape))
        self.ptr = C.c_void_p()
        rc = self.lib.srl_pinned_alloc(max(n, 1) * 8, C.byref(self.ptr))
        if rc:
            raise SrlError(rc, "srl_pinned_alloc")
        buf = (C.c_double * max(n, 1)).from_address(self.ptr.value)
        self.array = np.frombuffer(buf, dtype=np.float64, count=n).reshape(shape)

    def close(self):
        if self.ptr:
            self.array = None
            self.lib.srl_pinned_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

def comm_set_library(path):
    """debug / test hook (srl_comm_set_library): resolve the nccl* entry points from `path`; only before the first communicator call"""
    rc = load_library().srl_comm_set_library(os.fsencode(path))
    if rc:
        raise SrlError(rc, "srl_comm_set_library", "a communicator call has already resolved this process's RCCL")


def comm_backend_info():
    """(path of the RCCL shared object in use, ncclGetVersion, found-already-loaded-in-the-process) or raises."""
    buf = C.create_string_buffer(512)
    ver, pre = C.c_int(), C.c_int()
    rc = load_library().srl_comm_backend_info(buf, 512, C.byref(ver), C.byref(pre))
    if rc:
        raise SrlError(rc, "srl_comm_backend_info", buf.value.decode())
    return buf.value.decode(), ver.value, bool(pre.value)


def shard_range(n, nranks, rank):
    b, c = C.c_int(), C.c_int()
    load_library().srl_shard_range(n, nranks, rank, C.byref(b), C.byref(c))
    return b.value, c.value


def shard_budget(max_num_residuals, accepted_per_rank, rank):
    arr = np.ascontiguousarray(accepted_per_rank, dtype=np.int64)
    b, m = C.c_int64(), C.c_int()
    load_library().srl_shard_budget(int(max_num_residuals), arr.ctypes.data_as(C.POINTER(C.c_int64)), len(arr), rank,
                                    C.byref(b), C.byref(m))
    return b.value, m.value


def grid_sampling(world_xyz, size_voxel):
    w = _f64(world_xyz, (-1, 3))
    idx = np.empty(len(w), dtype=np.int32)
    n = C.c_int()
    rc = load_library().srl_grid_sampling(_ptr(w), len(w), float(size_voxel), _ptr(idx), C.byref(n))
    if rc:
        raise SrlError(rc, "srl_grid_sampling")
    return idx[: n.value].copy()


def tr1_order(keys_xyz, by_relation=False):
    """Iteration order of std::tr1::unordered_map<voxel, ...> after inserting the distinct int16 keys in order: the flat replay of
    host/tr1_order.h, or (by_relation) the pairwise relation of host/tr1_relation.h the device ordering evaluates."""
    k = np.ascontiguousarray(keys_xyz, dtype=np.int16).reshape(-1, 3)
    out = np.empty(len(k), dtype=np.int32)
    name = "srl_debug_tr1_order_by_relation" if by_relation else "srl_debug_tr1_order"
    rc = getattr(load_library(), name)(_ptr(k), len(k), _ptr(out))
    if rc:
        raise SrlError(rc, name)
    return out


def make_frame(q, t, t_last, R_il=None, t_il=None, frame_id=100):
    f = Frame()
    f.q[:] = list(np.asarray(q, dtype=np.float64))
    f.t[:] = list(np.asarray(t, dtype=np.float64))
    f.t_last[:] = list(np.asarray(t_last, dtype=np.float64))
    f.R_il[:] = list(np.eye(3).ravel() if R_il is None else np.asarray(R_il, dtype=np.float64).ravel())
    f.t_il[:] = list(np.zeros(3) if t_il is None else np.asarray(t_il, dtype=np.float64))
    f.frame_id = int(frame_id)
    return f


class Context:
    """Kernel-level C-ABI (srl_ctx)."""

    def __init__(self, device=0, handle=None):
        self.lib = load_library()
        self._own = handle is None
        if handle is None:
            h = C.c_void_p()
            rc = self.lib.srl_ctx_create(device, C.byref(h))
            if rc:
                raise SrlError(rc, "srl_ctx_create", self.lib.srl_status_str(rc).decode())
            handle = h
        self.h = handle
        self._cb = None
        if os.environ.get("SRL_ABLATE"):      # profiling tools only (tools/*.py, tools/*.sh): the library itself reads no env
            self.lib.srl_debug_set_ablate(self.h, int(os.environ["SRL_ABLATE"]))

    def close(self):
        if self.h and self._own:
            self.lib.srl_ctx_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what, ok=()):
        if rc and rc not in ok:
            raise SrlError(rc, what, (self.lib.srl_last_error(self.h) or b"").decode())
        return rc

    def map_upload(self, keys, counts, xyz, cap=20):
        keys = np.ascontiguousarray(keys, dtype=np.int16).reshape(-1, 3)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(len(counts), cap, 3)
        self._chk(self.lib.srl_map_upload(self.h, _ptr(keys), _ptr(counts), _ptr(xyz), len(counts), cap), "srl_map_upload")

    def map_insert(self, world_xyz, voxel_size=1.0, cap=20, min_dist=0.15, min_num_points=0):
        w = _f64(world_xyz, (-1, 3))
        added = C.c_int()
        self._chk(self.lib.srl_map_insert(self.h, _ptr(w), len(w), voxel_size, cap, min_dist, min_num_points, C.byref(added)), "srl_map_insert")
        return added.value

    def map_size(self):
        npnt, nv = C.c_int64(), C.c_int32()
        self._chk(self.lib.srl_map_size(self.h, C.byref(npnt), C.byref(nv)), "srl_map_size")
        return npnt.value, nv.value

    def map_probe_checksum(self, world_xyz=None, stride=1, voxel_size=1.0):
        """srl_map_probe_checksum: world_xyz None = the world points of the last committed frame"""
        out = C.c_uint64()
        if world_xyz is None:
            rc = self.lib.srl_map_probe_checksum(self.h, None, 0, int(stride), float(voxel_size), C.byref(out))
        else:
            w = _f64(world_xyz, (-1, 3))
            rc = self.lib.srl_map_probe_checksum(self.h, _ptr(w), len(w), int(stride), float(voxel_size), C.byref(out))
        self._chk(rc, "srl_map_probe_checksum")
        return out.value

    def map_download(self, cap=20):
        _, nv = self.map_size()
        keys = np.zeros((nv, 3), dtype=np.int16)
        counts = np.zeros(nv, dtype=np.int32)
        xyz = np.zeros((nv, cap, 3), dtype=np.float32)
        self._chk(self.lib.srl_map_download(self.h, _ptr(keys), _ptr(counts), _ptr(xyz), nv), "srl_map_download")
        return keys, counts, xyz

    def sweep_upload(self, raw_xyz):
        r = _f64(raw_xyz, (-1, 3))
        self._chk(self.lib.srl_sweep_upload(self.h, _ptr(r), len(r)), "srl_sweep_upload")

    def sweep_prefetch(self, raw_xyz):
        r = _f64(raw_xyz, (-1, 3))
        self._keep_prefetch = r                   # a page-locked source must outlive the copy
        self._chk(self.lib.srl_sweep_prefetch(self.h, _ptr(r), len(r)), "srl_sweep_prefetch")

    def sweep_wait(self):
        self._chk(self.lib.srl_sweep_wait(self.h), "srl_sweep_wait")

    def sweep_swap(self):
        self._chk(self.lib.srl_sweep_swap(self.h), "srl_sweep_swap")

    def sweep_shard(self):
        b, c, t = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.lib.srl_sweep_shard(self.h, C.byref(b), C.byref(c), C.byref(t)), "srl_sweep_shard")
        return b.value, c.value, t.value

    def set_taps(self, on):
        self._chk(self.lib.srl_set_taps(self.h, int(on)), "srl_set_taps")

    def set_profiling(self, on):
        self._chk(self.lib.srl_set_profiling(self.h, int(on)), "srl_set_profiling")

    def timing_mark(self):
        """the light profiling's sums start here (no read-back, an armed launch stays armed): srl_timing_mark"""
        self._chk(self.lib.srl_timing_mark(self.h), "srl_timing_mark")

    def set_profiling_period(self, period):
        """light profiling (mode 2): time every period-th association launch only (2 event records per period launches)"""
        self._chk(self.lib.srl_set_profiling_period(self.h, int(period)), "srl_set_profiling_period")

    def timing(self):
        t = Timing()
        self._chk(self.lib.srl_get_timing(self.h, C.byref(t)), "srl_get_timing")
        return t

    def _select(self, opts):
        """test hook: the selection path wished for with default_opts(select_mode=...) (srl_debug_set_select_mode; sticky per context)"""
        mode = int(getattr(opts, "select_mode", 0) or 0)
        if mode != getattr(self, "_select_mode", 0):
            self._chk(self.lib.srl_debug_set_select_mode(self.h, mode), "srl_debug_set_select_mode")
            self._select_mode = mode

    def build_residuals(self, frame, opts, allow=(SRL_ERR_NAN_PLANARITY,)):
        self._select(opts)
        out = NormalEq()
        rc = self._chk(self.lib.srl_build_residuals(self.h, C.byref(frame), C.byref(opts), C.byref(out)), "srl_build_residuals", ok=allow)
        return out, rc

    def build_residuals_overlap(self, frame, opts, fn, allow=(SRL_ERR_NAN_PLANARITY,)):
        """srl_build_residuals with a host callback (a Python callable without arguments) run while the kernels are in flight."""
        self._select(opts)
        out = NormalEq()
        cb = C.CFUNCTYPE(None, C.c_void_p)(lambda _u: fn())
        rc = self._chk(self.lib.srl_build_residuals_overlap(self.h, C.byref(frame), C.byref(opts), C.byref(out), cb, None),
                       "srl_build_residuals_overlap", ok=allow)
        return out, rc

    def pin_thread_to_gpu_numa(self):
        """restrict the calling thread to the CPUs of the GPU's NUMA node; returns the node, or None when the topology is unreadable"""
        node = C.c_int(-1)
        rc = self.lib.srl_thread_pin_to_gpu_numa(self.h, C.byref(node))
        return node.value if rc == 0 else None

    def fetch_neighbors(self, K=20):
        _, n, _ = self.sweep_shard()
        ids = np.empty((n, K), dtype=np.int32)
        status = np.empty(n, dtype=np.uint8)
        ncand = np.empty(n, dtype=np.int32)
        self._chk(self.lib.srl_fetch_neighbors(self.h, _ptr(ids), _ptr(status), _ptr(ncand)), "srl_fetch_neighbors")
        return ids, status, ncand

    def fetch_residuals(self):
        _, n, _ = self.sweep_shard()
        normal = np.empty((n, 3)); a2d = np.empty(n); w = np.empty(n); off = np.empty(n); d = np.empty(n); J = np.empty((n, 6))
        self._chk(self.lib.srl_fetch_residuals(self.h, _ptr(normal), _ptr(a2d), _ptr(w), _ptr(off), _ptr(d), _ptr(J)), "srl_fetch_residuals")
        return dict(normal=normal, a2D=a2d, weight=w, norm_offset=off, distance=d, jacobian=J)

    def search_neighbors(self, world_xyz, nb=1, size=1.0, K=20, thr=1):
        q = _f64(world_xyz, (-1, 3))
        ids = np.empty((len(q), K), dtype=np.int32)
        xyz = np.empty((len(q), K, 3), dtype=np.float32)
        nf = np.empty(len(q), dtype=np.int32)
        self._chk(self.lib.srl_search_neighbors(self.h, _ptr(q), len(q), nb, size, K, thr, _ptr(ids), _ptr(xyz), _ptr(nf)), "srl_search_neighbors")
        return ids, xyz, nf

    def set_launch_shape(self, kpw, wpb):
        self._chk(self.lib.srl_debug_set_launch_shape(self.h, int(kpw), int(wpb)), "srl_debug_set_launch_shape")

    def set_fused_reduce(self, on):
        self._chk(self.lib.srl_debug_set_fused_reduce(self.h, 1 if on else 0), "srl_debug_set_fused_reduce")

    def set_bound_culling(self, on):
        """test hook: 0 = every pass visits every found voxel (no use of the previous pass's neighbourhood bounds)"""
        self._chk(self.lib.srl_debug_set_bound_culling(self.h, 1 if on else 0), "srl_debug_set_bound_culling")

    def set_search_select_mode(self, mode):
        self._chk(self.lib.srl_debug_set_search_select_mode(self.h, int(mode)), "srl_debug_set_search_select_mode")

    def set_armed_launch(self, on):
        """armed launches (the next pass's kernel enqueued while the current one runs): on by default"""
        mode = int(on) if on in (0, 1, 2) and not isinstance(on, bool) else (1 if on else 0)      # 2: armed behind every eligible pass (no policy)
        self._chk(self.lib.srl_set_armed_launch(self.h, mode), "srl_set_armed_launch")

    def solve_end(self):
        """the caller's ESIKF loop on the current sweep has ended (srl_solve_end)"""
        self._chk(self.lib.srl_solve_end(self.h), "srl_solve_end")

    def disarm(self):
        self._chk(self.lib.srl_disarm(self.h), "srl_disarm")

    def arm_stats(self):
        out = (C.c_uint64 * 4)()
        self._chk(self.lib.srl_get_arm_stats(self.h, out), "srl_get_arm_stats")
        return dict(zip(("armed", "fired", "cancelled", "expired"), (int(v) for v in out)))

    def set_pose_box(self, kind):
        """0: pinned host memory + device relay; 1: fine-grained device memory written through the BAR (raises if not CPU-visible)"""
        self._chk(self.lib.srl_debug_set_pose_box(self.h, int(kind)), "srl_debug_set_pose_box")

    FRAME_STAGES = ("upload", "select_group", "select_download", "select_host_order", "select_gather", "commit_transform", "commit_download",
                    "insert_sort", "insert_segments", "insert_lookup_create", "insert_replay")

    def set_frame_epoch(self, frames_to_wrap):
        """test hook: the epoch of the frame pipeline's scratch tables wraps `frames_to_wrap` frames from now (srl_debug_set_frame_epoch)"""
        self._chk(self.lib.srl_debug_set_frame_epoch(self.h, int(frames_to_wrap)), "srl_debug_set_frame_epoch")

    def frame_timing(self, enable=True):
        """accumulated stage times (us) of the frame pipeline since the last call (which clears them); see srl_debug_frame_timing"""
        out = np.zeros(16)
        self._chk(self.lib.srl_debug_frame_timing(self.h, 1 if enable else 0, _ptr(out)), "srl_debug_frame_timing")
        return dict(zip(self.FRAME_STAGES, (float(v) for v in out)))

    def pass_stamps(self, enable=True, read=True):
        """(gpu[64, 16] device-clock ticks of 10 ns, host[64, 4] steady-clock ns) of the last 64 passes; see srl_debug_pass_stamps"""
        g = np.zeros((64, 32), dtype=np.int64); h = np.zeros((64, 4), dtype=np.int64)
        self._chk(self.lib.srl_debug_pass_stamps(self.h, 1 if enable else 0, _ptr(g) if read else None, _ptr(h) if read else None), "srl_debug_pass_stamps")
        return g, h

    def set_arm_linger(self, host_linger_us=150.0, kernel_linger_us=300.0):
        self._chk(self.lib.srl_debug_set_arm_linger(self.h, float(host_linger_us), float(kernel_linger_us)), "srl_debug_set_arm_linger")

    def device_sqrt(self, x):
        x = _f64(x).ravel()
        out = np.empty_like(x)
        self._chk(self.lib.srl_debug_device_sqrt(self.h, _ptr(x), len(x), _ptr(out)), "srl_debug_device_sqrt")
        return out

    def transform_points(self, raw_xyz, q, t, R_il=None, t_il=None):
        r = _f64(raw_xyz, (-1, 3))
        out = np.empty_like(r)
        R_il = _f64(np.eye(3) if R_il is None else R_il).ravel()
        t_il = _f64(np.zeros(3) if t_il is None else t_il)
        self._chk(self.lib.srl_transform_points(self.h, _ptr(r), len(r), _dptr(_f64(q)), _dptr(_f64(t)), _dptr(R_il), _dptr(t_il), _ptr(out)), "srl_transform_points")
        return out

    # --- frame-resident pipeline
    def frame_upload(self, raw_xyz):
        r = _f64(raw_xyz, (-1, 3))
        self._chk(self.lib.srl_frame_upload(self.h, _ptr(r), len(r)), "srl_frame_upload")

    def frame_undistort(self, raw_xyz, relative_time_ms, imu_states, time_frame_begin, mode, R_il=None, t_il=None, imu_point_in=None):
        """imu_states: (S, 17) array = timestamp, un_acc, un_gyr, trans, quat wxyz, vel.  Returns (imu_point, raw_point)."""
        r = _f64(raw_xyz, (-1, 3)); rel = _f64(relative_time_ms)
        st = _f64(imu_states, (-1, 17))
        R_il = _f64(np.eye(3) if R_il is None else R_il).ravel()
        t_il = _f64(np.zeros(3) if t_il is None else t_il)
        pin = None if imu_point_in is None else _f64(imu_point_in, (-1, 3))
        imu = np.empty_like(r); out = np.empty_like(r)
        self._chk(self.lib.srl_frame_undistort(self.h, _ptr(r), _ptr(rel), None if pin is None else _ptr(pin), len(r), _ptr(st), len(st),
                                               float(time_frame_begin), int(mode), _dptr(R_il), _dptr(t_il), _ptr(imu), _ptr(out)),
                  "srl_frame_undistort")
        return imu, out

    def frame_take(self, index):
        idx = np.ascontiguousarray(index, dtype=np.int32)
        self._chk(self.lib.srl_frame_take(self.h, _ptr(idx), len(idx)), "srl_frame_take")

    def frame_size(self):
        n = C.c_int()
        self._chk(self.lib.srl_frame_size(self.h, C.byref(n)), "srl_frame_size")
        return n.value

    def radix_sort_pairs(self, keys, bits):
        """test hook: the frame path's stable (key, position) sort over the low `bits` bits -> (keys_sorted, positions_sorted)"""
        k = np.ascontiguousarray(keys, dtype=np.uint32)
        ks = np.empty_like(k); ps = np.empty_like(k)
        self._chk(self.lib.srl_debug_radix_sort_pairs(self.h, _ptr(k), len(k), int(bits), _ptr(ks), _ptr(ps)), "srl_debug_radix_sort_pairs")
        return ks, ps

    def set_frame_order_mode(self, mode):
        """test hook: 0 = keypoint order on the device where it applies (default), 1 = always the host replay (srl_debug_set_frame_order_mode)"""
        self._chk(self.lib.srl_debug_set_frame_order_mode(self.h, int(mode)), "srl_debug_set_frame_order_mode")

    def frame_order_used(self):
        """1 = the last selection ordered on the device, 2 = host replay, 3 = device order overflowed a bucket and the host replay ran"""
        u = C.c_int()
        self._chk(self.lib.srl_debug_frame_order_used(self.h, C.byref(u)), "srl_debug_frame_order_used")
        return u.value

    def frame_select_keypoints(self, q, t, sample_voxel_size, R_il=None, t_il=None, want_index=True):
        """want_index=False: keypoint_index = NULL (the selection stays on the device as the resident sweep; only the count comes back)"""
        R_il = _f64(np.eye(3) if R_il is None else R_il).ravel()
        t_il = _f64(np.zeros(3) if t_il is None else t_il)
        m = C.c_int()
        if not want_index:
            self._chk(self.lib.srl_frame_select_keypoints(self.h, _dptr(_f64(q)), _dptr(_f64(t)), _dptr(R_il), _dptr(t_il),
                                                          float(sample_voxel_size), None, C.byref(m)), "srl_frame_select_keypoints")
            return m.value
        idx = np.empty(max(self.frame_size(), 1), dtype=np.int32)
        self._chk(self.lib.srl_frame_select_keypoints(self.h, _dptr(_f64(q)), _dptr(_f64(t)), _dptr(R_il), _dptr(t_il),
                                                      float(sample_voxel_size), _ptr(idx), C.byref(m)), "srl_frame_select_keypoints")
        return idx[: m.value].copy()

    def frame_commit(self, q, t, voxel_size=1.0, cap=20, min_dist=0.15, min_num_points=0, R_il=None, t_il=None, want_world=True, want_added=True,
                     world_out=None):
        """want_added=False: num_added = NULL (the insertion is only enqueued, see include/srlivo_hip.h); world_out: a caller-owned
        (n, 3) float64 array to receive the world points (page-locked for the fastest path) instead of a fresh one"""
        R_il = _f64(np.eye(3) if R_il is None else R_il).ravel()
        t_il = _f64(np.zeros(3) if t_il is None else t_il)
        world = None
        if want_world:
            world = world_out if world_out is not None else np.empty((self.frame_size(), 3))
            assert world.dtype == np.float64 and world.flags.c_contiguous and world.shape == (self.frame_size(), 3)
        added = C.c_int()
        self._chk(self.lib.srl_frame_commit(self.h, _dptr(_f64(q)), _dptr(_f64(t)), _dptr(R_il), _dptr(t_il), float(voxel_size), cap,
                                            float(min_dist), min_num_points, _ptr(world) if want_world else None,
                                            C.byref(added) if want_added else None),
                  "srl_frame_commit")
        return world, (added.value if want_added else None)

    # --- multi-GPU
    @staticmethod
    def comm_unique_id():
        buf = (C.c_ubyte * SRL_COMM_ID_BYTES)()
        rc = load_library().srl_comm_unique_id(buf)
        if rc:
            raise SrlError(rc, "srl_comm_unique_id")
        return bytes(buf)

    def comm_init_rank(self, nranks, rank, uid):
        buf = (C.c_ubyte * SRL_COMM_ID_BYTES).from_buffer_copy(uid)
        self._chk(self.lib.srl_comm_init_rank(self.h, nranks, rank, buf), "srl_comm_init_rank")

    def comm_suspend(self, suspend=True):
        self._chk(self.lib.srl_comm_suspend(self.h, int(bool(suspend))), "srl_comm_suspend")

    def comm_destroy(self):
        self._chk(self.lib.srl_comm_destroy(self.h), "srl_comm_destroy")

    def set_gather_counts(self, nranks=1, rank=0, counts=None):
        """srl_debug_set_gather_counts (counts=None: off)"""
        if counts is None:
            self._chk(self.lib.srl_debug_set_gather_counts(self.h, 1, 0, None), "set_gather_counts")
            return
        c = (C.c_int64 * int(nranks))(*[int(x) for x in counts])
        self._chk(self.lib.srl_debug_set_gather_counts(self.h, int(nranks), int(rank), c), "set_gather_counts")

    def peer_export(self):
        """srl_peer_export -> (64-byte HIP IPC handle for peers in other processes, device pointer for peers in this process)"""
        h = (C.c_ubyte * SRL_PEER_HANDLE_BYTES)()
        ptr = C.c_void_p()
        self._chk(self.lib.srl_peer_export(self.h, h, C.byref(ptr)), "srl_peer_export")
        return bytes(h), int(ptr.value)

    def peer_attach(self, nranks, rank, handles=None, local_ptrs=None):
        """srl_peer_attach: handles = list of nranks 64-byte handles (or None), local_ptrs = list of nranks device pointers (0 / None
        entries fall back to the handle).  Upload the sweep afterwards."""
        hb = None
        if handles is not None:
            assert len(handles) == nranks and all(len(x) == SRL_PEER_HANDLE_BYTES for x in handles)
            hb = (C.c_ubyte * (SRL_PEER_HANDLE_BYTES * nranks)).from_buffer_copy(b"".join(handles))
        lp = None
        if local_ptrs is not None:
            lp = (C.c_void_p * nranks)(*[C.c_void_p(int(x) if x else None) for x in local_ptrs])
        self._chk(self.lib.srl_peer_attach(self.h, int(nranks), int(rank), hb, lp), "srl_peer_attach")

    def peer_set_deadline_ms(self, ms):
        """how long a pass keeps re-polling for a late rank's row before the session is given up on every rank (srl_peer_set_deadline_ms)"""
        self._chk(self.lib.srl_peer_set_deadline_ms(self.h, int(ms)), "srl_peer_set_deadline_ms")

    def peer_stats(self):
        """(passes repeated because a row had not arrived within one kernel's spin, session failed)"""
        rep, failed = C.c_int64(), C.c_int()
        self._chk(self.lib.srl_peer_stats(self.h, C.byref(rep), C.byref(failed)), "srl_peer_stats")
        return rep.value, bool(failed.value)

    def comm_info(self):
        """srl_comm_info: what the sharded path of this context runs on"""
        tr, nr, rk, seen, armed = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int64()
        self._chk(self.lib.srl_comm_info(self.h, C.byref(tr), C.byref(nr), C.byref(rk), C.byref(seen), C.byref(armed)), "srl_comm_info")
        return dict(transport_used=("none", "rccl", "peer", "host-callbacks")[tr.value], nranks=nr.value, rank=rk.value, ranks_seen=seen.value,
                    passes_armed=armed.value)

    def peer_detach(self):
        self._chk(self.lib.srl_peer_detach(self.h), "srl_peer_detach")

    def comm_set_host_callbacks(self, nranks, rank, allreduce, allgather):
        """allreduce(np.ndarray float64) -> in place sum; allgather(int) -> list of ints."""
        def _ar(buf, count, _user):
            a = np.ctypeslib.as_array(buf, shape=(count,))
            allreduce(a)
            return 0

        def _ag(mine, allp, _user):
            vals = allgather(int(mine[0]))
            for i, v in enumerate(vals):
                allp[i] = int(v)
            return 0
        self._cb = (ALLREDUCE_FN(_ar), ALLGATHER_FN(_ag))
        self._chk(self.lib.srl_comm_set_host_callbacks(self.h, nranks, rank, self._cb[0], self._cb[1], None), "srl_comm_set_host_callbacks")


class Lio:
    """C++ host mirror (lioOptimization + eskfEstimator) through the srl_lio handles."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.srl_lio_create(device, C.byref(h))
        if rc:
            raise SrlError(rc, "srl_lio_create", self.lib.srl_status_str(rc).decode())
        self.h = h
        self._provider = None
        ctxp = self.lib.srl_lio_ctx(self.h)
        self.ctx = Context(handle=C.c_void_p(ctxp)) if ctxp else None

    def close(self):
        if self.h:
            self.lib.srl_lio_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what, ok=()):
        if rc and rc not in ok:
            raise SrlError(rc, what, (self.lib.srl_lio_last_error(self.h) or b"").decode())
        return rc

    def set_extrinsics(self, R_il, t_il):
        self._chk(self.lib.srl_lio_set_extrinsics(self.h, _dptr(_f64(R_il).ravel()), _dptr(_f64(t_il))), "set_extrinsics")

    def last_solve_launches(self):
        n = C.c_int()
        self._chk(self.lib.srl_lio_last_solve_launches(self.h, C.byref(n)), "last_solve_launches")
        return n.value

    def set_laser_point_cov(self, c):
        self._chk(self.lib.srl_lio_set_laser_point_cov(self.h, float(c)), "set_laser_point_cov")

    def eskf_get_state(self):
        s = np.empty(19)
        self._chk(self.lib.srl_lio_eskf_get_state(self.h, _dptr(s)), "eskf_get_state")
        return s

    def eskf_set_state(self, s):
        self._chk(self.lib.srl_lio_eskf_set_state(self.h, _dptr(_f64(s))), "eskf_set_state")

    def eskf_get_cov(self):
        P = np.empty(289)
        self._chk(self.lib.srl_lio_eskf_get_cov(self.h, _dptr(P)), "eskf_get_cov")
        return P.reshape(17, 17)

    def eskf_set_cov(self, P):
        self._chk(self.lib.srl_lio_eskf_set_cov(self.h, _dptr(_f64(P).ravel())), "eskf_set_cov")

    def eskf_set_noise(self, acc, gyr, bacc, bgyr):
        self._chk(self.lib.srl_lio_eskf_set_noise(self.h, acc, gyr, bacc, bgyr), "eskf_set_noise")

    def eskf_init_imu(self, acc0, gyr0):
        self._chk(self.lib.srl_lio_eskf_init_imu(self.h, _dptr(_f64(acc0)), _dptr(_f64(gyr0))), "eskf_init_imu")

    def eskf_try_init(self, t, gyr, acc):
        t = _f64(t); g = _f64(gyr, (-1, 3)); a = _f64(acc, (-1, 3))
        r = C.c_int()
        self._chk(self.lib.srl_lio_eskf_try_init(self.h, _dptr(t), _dptr(g), _dptr(a), len(t), C.byref(r)), "eskf_try_init")
        return r.value

    def eskf_init_stats(self):
        o = np.zeros(14)
        self._chk(self.lib.srl_lio_eskf_get_init_stats(self.h, _dptr(o)), "eskf_get_init_stats")
        return dict(mean_gyr=o[0:3].copy(), mean_acc=o[3:6].copy(), gyr_cov=o[6:9].copy(), acc_cov=o[9:12].copy(),
                    num_init_meas=int(o[12]), initial_flag=bool(o[13]))

    def set_initial_flag(self, flag):
        self._chk(self.lib.srl_lio_set_initial_flag(self.h, int(bool(flag))), "set_initial_flag")

    def state_initialization(self, index_frame, initialization, prev2, prev1):
        out = np.zeros(7)
        self._chk(self.lib.srl_lio_state_initialization(self.h, int(index_frame), int(initialization), _dptr(_f64(prev2)),
                                                        _dptr(_f64(prev1)), _dptr(out)), "state_initialization")
        return out[0:4].copy(), out[4:7].copy()

    def eskf_scale_init_cov(self):
        self._chk(self.lib.srl_lio_eskf_scale_init_cov(self.h), "eskf_scale_init_cov")

    def eskf_predict(self, dt, acc1, gyr1):
        self._chk(self.lib.srl_lio_eskf_predict(self.h, float(dt), _dptr(_f64(acc1)), _dptr(_f64(gyr1))), "eskf_predict")

    def eskf_observe(self, dx):
        self._chk(self.lib.srl_lio_eskf_observe(self.h, _dptr(_f64(dx))), "eskf_observe")

    def add_points_to_map(self, world_xyz, voxel_size=1.0, cap=20, min_dist=0.15, min_num_points=0):
        w = _f64(world_xyz, (-1, 3))
        self._chk(self.lib.srl_lio_add_points_to_map(self.h, _ptr(w), len(w), voxel_size, cap, min_dist, min_num_points), "add_points_to_map")

    def map_size(self):
        n = C.c_int64()
        self._chk(self.lib.srl_lio_map_size(self.h, C.byref(n)), "map_size")
        return n.value

    def resident_sweep(self, raw_xyz):
        r = _f64(raw_xyz, (-1, 3))
        self._chk(self.lib.srl_lio_resident_sweep(self.h, _ptr(r), len(r)), "resident_sweep")
        return len(r)

    def prefetch_sweep(self, raw_xyz):
        r = _f64(raw_xyz, (-1, 3))
        self._keep_prefetch = r
        self._chk(self.lib.srl_lio_prefetch_sweep(self.h, _ptr(r), len(r)), "prefetch_sweep")

    def swap_sweep(self):
        self._chk(self.lib.srl_lio_swap_sweep(self.h), "swap_sweep")

    def prefetch_sweep_during_solve(self, raw_xyz):
        """the upload of the NEXT sweep, issued by the next update_iekf beside the kernel of its first pass (srl_lio_prefetch_sweep_during_solve)"""
        r = _f64(raw_xyz, (-1, 3))
        self._keep_prefetch = r
        self._chk(self.lib.srl_lio_prefetch_sweep_during_solve(self.h, _ptr(r), len(r)), "prefetch_sweep_during_solve")

    def update_iekf(self, opts, raw_xyz, state, t_last, frame_id=100, log_iters=0, n_resident=None,
                    allow=(SRL_ERR_NOT_ENOUGH_RESIDUALS,)):
        """raw_xyz=None uses the sweep pinned by resident_sweep (n_resident = its size)."""
        if self.ctx is not None:
            self.ctx._select(opts)
        st = _f64(state).copy()
        log = np.zeros((max(log_iters, 1), 61)) if log_iters else None
        iters, nres = C.c_int(), C.c_int()
        if raw_xyz is None:
            rp, n = None, int(n_resident)
        else:
            r = _f64(raw_xyz, (-1, 3))
            rp, n = _ptr(r), len(r)
        rc = self._chk(self.lib.srl_lio_update_iekf(self.h, C.byref(opts), rp, n, _dptr(st), _dptr(_f64(t_last)), int(frame_id),
                                                    _ptr(log) if log is not None else None, log_iters, C.byref(iters), C.byref(nres)),
                       "update_iekf", ok=allow)
        return dict(rc=rc, state=st, iters=iters.value, num_residuals=nres.value, log=None if log is None else log[: iters.value])

    def bound_solver(self, opts, eskf_state, eskf_cov, state, t_last, frame_id, n_resident):
        """A zero-allocation closure for repeated solves of the resident sweep from the same prior (bench / replay
        loops): every argument is converted once; each call = srl_lio_eskf_set_state + _set_cov + srl_lio_update_iekf.
        Returns (status, iterations, residuals); the solved state is left in `solver.state`."""
        if self.ctx is not None:
            self.ctx._select(opts)
        es = _f64(eskf_state).copy(); ec = _f64(eskf_cov).ravel().copy()
        st0 = _f64(state).copy(); st = st0.copy(); tl = _f64(t_last).copy()
        iters, nres = C.c_int(), C.c_int()
        lib, h = self.lib, self.h
        p_es, p_ec, p_st, p_tl = _dptr(es), _dptr(ec), _dptr(st), _dptr(tl)
        o = C.byref(opts); bi, bn = C.byref(iters), C.byref(nres)
        set_state, set_cov, upd = lib.srl_lio_eskf_set_state, lib.srl_lio_eskf_set_cov, lib.srl_lio_update_iekf
        fid, n = int(frame_id), int(n_resident)

        def solve():
            st[:] = st0
            rc = set_state(h, p_es) or set_cov(h, p_ec) or upd(h, o, None, n, p_st, p_tl, fid, None, 0, bi, bn)
            return rc, iters.value, nres.value
        solve.state = st
        solve._keep = (es, ec, st0, tl, opts)
        return solve

    def bound_stream_step(self, opts, eskf_state, eskf_cov, state, t_last, frame_id, n_resident, next_pinned):
        """The node's loop body over a stream of sweeps as ONE C call per step (srl_lio_stream_step): prior reset, the upload of the NEXT
        sweep (`next_pinned`: a page-locked (n, 3) array) registered with the solve, the solve of the resident sweep, the swap.  Every argument
        is converted once.  Returns (status, iterations, residuals); the solved state is left in `step.state`."""
        if self.ctx is not None:
            self.ctx._select(opts)
        es = _f64(eskf_state).copy(); ec = _f64(eskf_cov).ravel().copy()
        st0 = _f64(state).copy(); st = st0.copy(); tl = _f64(t_last).copy()
        iters, nres = C.c_int(), C.c_int()
        lib, h = self.lib, self.h
        p_es, p_ec, p_st, p_tl = _dptr(es), _dptr(ec), _dptr(st), _dptr(tl)
        o = C.byref(opts); bi, bn = C.byref(iters), C.byref(nres)
        fn = lib.srl_lio_stream_step
        fid, n = int(frame_id), int(n_resident)
        nx_ptr, nx_n = next_pinned.ctypes.data_as(C.c_void_p), int(len(next_pinned))
        copy_state = np.copyto

        def step():
            copy_state(st, st0)
            return fn(h, o, p_es, p_ec, n, p_st, p_tl, fid, nx_ptr, nx_n, bi, bn), iters.value, nres.value
        step.state = st
        step._keep = (es, ec, st0, tl, opts, next_pinned)
        return step

    def update_iekf_provided(self, opts, provider, n, state, t_last, frame_id=100, log_iters=0,
                             allow=(SRL_ERR_NOT_ENOUGH_RESIDUALS,)):
        """provider(frame: Frame, opts: IcpOpts, out: NormalEq) -> int status."""
        if self.ctx is not None:
            self.ctx._select(opts)
        def _p(fp, op, outp, _user):
            return int(provider(fp.contents, op.contents, outp.contents))
        self._provider = PROVIDER_FN(_p)
        st = _f64(state).copy()
        log = np.zeros((max(log_iters, 1), 61)) if log_iters else None
        iters, nres = C.c_int(), C.c_int()
        rc = self._chk(self.lib.srl_lio_update_iekf_provided(self.h, C.byref(opts), self._provider, None, int(n), _dptr(st),
                                                             _dptr(_f64(t_last)), int(frame_id), _ptr(log) if log is not None else None,
                                                             log_iters, C.byref(iters), C.byref(nres)), "update_iekf_provided", ok=allow)
        return dict(rc=rc, state=st, iters=iters.value, num_residuals=nres.value, log=None if log is None else log[: iters.value])

    def optimize(self, opts, sample_voxel_size, frame_raw, frame_world, state, t_last, frame_id=100,
                 allow=(SRL_ERR_NOT_ENOUGH_RESIDUALS,)):
        if self.ctx is not None:
            self.ctx._select(opts)
        raw = _f64(frame_raw, (-1, 3))
        world = _f64(frame_world, (-1, 3)).copy()
        st = _f64(state).copy()
        kidx = np.empty(len(raw), dtype=np.int32)
        nk, iters, nres = C.c_int(), C.c_int(), C.c_int()
        rc = self._chk(self.lib.srl_lio_optimize(self.h, C.byref(opts), float(sample_voxel_size), _ptr(raw), _ptr(world), len(raw), _dptr(st),
                                                 _dptr(_f64(t_last)), int(frame_id), _ptr(kidx), C.byref(nk), C.byref(iters), C.byref(nres)),
                       "optimize", ok=allow)
        return dict(rc=rc, state=st, world=world, keypoint_index=kidx[: nk.value].copy(), iters=iters.value, num_residuals=nres.value)

    def set_odometry_options(self, **kw):
        o = OdometryOpts()
        d = dict(init_voxel_size=0.2, init_sample_voxel_size=1.0, init_num_frames=20, num_for_initialization=10, voxel_size=0.5,
                 sample_voxel_size=1.5, max_num_points_in_voxel=20, min_distance_points=0.1, motion_compensation=MC_CONSTANT_VELOCITY,
                 initialization=0, point_time_enable=1, acc_cov=0.1, gyr_cov=0.1, b_acc_cov=0.0001, b_gyr_cov=0.0001)
        icp = kw.pop("icp", None) or default_opts()
        d.update(kw)
        for k, v in d.items():
            setattr(o, k, v)
        o.icp = icp
        self._chk(self.lib.srl_lio_set_odometry_options(self.h, C.byref(o)), "set_odometry_options")
        return o

    def run_measurement(self, time_frame, imu_t, imu_acc, imu_gyr, pts_raw, pts_timestamp, time_sweep_begin, time_sweep_offset,
                        allow=(SRL_ERR_NOT_ENOUGH_RESIDUALS,)):
        it = _f64(imu_t); ia = _f64(imu_acc, (-1, 3)); ig = _f64(imu_gyr, (-1, 3))
        pr = _f64(pts_raw, (-1, 3)); pt = _f64(pts_timestamp)
        out = ReplayResult()
        rc = self._chk(self.lib.srl_lio_run_measurement(self.h, float(time_frame), _ptr(it), _ptr(ia), _ptr(ig), len(it), _ptr(pr), _ptr(pt),
                                                        len(pr), float(time_sweep_begin), float(time_sweep_offset), C.byref(out)),
                       "run_measurement", ok=allow)
        return dict(rc=rc, processed=bool(out.processed), initialized=bool(out.initialized), index_frame=out.index_frame,
                    success=bool(out.success), num_residuals=out.num_residuals_used, iters=out.iterations, frame_points=out.frame_points,
                    keypoints=out.keypoints, points_added=out.points_added, state=np.array(out.state))

    def last_frame(self):
        n = C.c_int()
        self._chk(self.lib.srl_lio_last_frame(self.h, 0, None, None, None, C.byref(n)), "last_frame")
        raw = np.empty((n.value, 3)); pt = np.empty((n.value, 3)); imu = np.empty((n.value, 3))
        self._chk(self.lib.srl_lio_last_frame(self.h, n.value, _ptr(raw), _ptr(pt), _ptr(imu), C.byref(n)), "last_frame")
        return dict(raw_point=raw, point=pt, imu_point=imu)

    def optimize_resident(self, opts, sample_voxel_size, frame_raw, state, t_last, frame_id=100,
                          allow=(SRL_ERR_NOT_ENOUGH_RESIDUALS,)):
        if self.ctx is not None:
            self.ctx._select(opts)
        raw = _f64(frame_raw, (-1, 3))
        st = _f64(state).copy()
        kidx = np.empty(max(len(raw), 1), dtype=np.int32)
        nk, iters, nres = C.c_int(), C.c_int(), C.c_int()
        rc = self._chk(self.lib.srl_lio_optimize_resident(self.h, C.byref(opts), float(sample_voxel_size), _ptr(raw), len(raw), _dptr(st),
                                                          _dptr(_f64(t_last)), int(frame_id), _ptr(kidx), C.byref(nk), C.byref(iters),
                                                          C.byref(nres)), "optimize_resident", ok=allow)
        return dict(rc=rc, state=st, keypoint_index=kidx[: nk.value].copy(), iters=iters.value, num_residuals=nres.value)

    def commit_frame(self, state, voxel_size=1.0, cap=20, min_dist=0.15, min_num_points=0, want_world=True, want_added=True):
        world = np.empty((self.ctx.frame_size(), 3)) if want_world else None
        added = C.c_int()
        self._chk(self.lib.srl_lio_commit_frame(self.h, _dptr(_f64(state)), float(voxel_size), cap, float(min_dist), min_num_points,
                                                _ptr(world) if want_world else None, C.byref(added) if want_added else None), "commit_frame")
        return world, (added.value if want_added else None)

    def search_neighbors(self, point, nb=1, size=1.0, K=20, thr=1):
        out = np.zeros((K, 3)); vox = np.zeros((K, 3), dtype=np.int16); nf = C.c_int()
        self._chk(self.lib.srl_lio_search_neighbors(self.h, _dptr(_f64(point)), nb, size, K, thr, _ptr(out), _ptr(vox), C.byref(nf)), "search_neighbors")
        return out[: nf.value], vox[: nf.value]

    def neighborhood(self, pts):
        P = _f64(pts, (-1, 3))
        c = np.empty(3); n = np.empty(3); cov = np.empty(9); a = C.c_double()
        self._chk(self.lib.srl_lio_neighborhood(self.h, _ptr(P), len(P), _dptr(c), _dptr(n), _dptr(cov), C.byref(a)), "neighborhood")
        return dict(center=c, normal=n, covariance=cov.reshape(3, 3), a2D=a.value)

    def build_plane_residuals(self, opts, raw_xyz, state, t_last, frame_id=100):
        if self.ctx is not None:
            self.ctx._select(opts)
        r = _f64(raw_xyz, (-1, 3))
        rows = np.zeros((len(r), 15)); world = np.zeros((len(r), 3))
        nout, succ = C.c_int(), C.c_int(); loss = C.c_double()
        self._chk(self.lib.srl_lio_build_plane_residuals(self.h, C.byref(opts), _ptr(r), len(r), _dptr(_f64(state)), _dptr(_f64(t_last)), int(frame_id),
                                                         _ptr(rows), len(r), C.byref(nout), C.byref(loss), C.byref(succ), _ptr(world)),
                  "build_plane_residuals")
        return dict(rows=rows[: nout.value], loss_sum=loss.value, success=bool(succ.value), keypoint_world=world)
