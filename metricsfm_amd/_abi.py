"""ctypes mirrors of the structs in include/msfm.h (layout only, no behaviour)."""
import ctypes as C

import numpy as np

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int32)
c_u8_p = C.POINTER(C.c_uint8)

MSFM_OK = 0
MSFM_E_INVAL, MSFM_E_NOMEM, MSFM_E_DEVICE, MSFM_E_NUMERIC = -1, -2, -3, -4
MSFM_MATCH_GOOD = 0x40000000
MSFM_MATCH_NOT_ALL = 0x20000000
MSFM_MATCH_ID_MASK = 0x1FFFFFFF
MSFM_MAX_KERNEL_STATS = 32

TERMINATION = {1: "CONVERGENCE_FUNCTION", 2: "CONVERGENCE_GRADIENT", 3: "CONVERGENCE_PARAMETER",
               4: "NO_CONVERGENCE", 5: "FAILURE", 6: "MIN_RADIUS"}


class BaProblem(C.Structure):
    _fields_ = [("n_cams", C.c_int), ("n_models", C.c_int), ("n_points", C.c_int), ("n_obs", C.c_int),
                ("cam_pose", c_double_p), ("cam_model", c_double_p), ("cam_model_of_cam", c_int_p),
                ("point", c_double_p), ("obs_cam", c_int_p), ("obs_pt", c_int_p), ("obs_xy", c_double_p),
                ("pt_weight", c_double_p), ("cam_mutable", c_u8_p), ("model_mutable", c_u8_p),
                ("pt_mutable", c_u8_p), ("gps_xyz", c_double_p), ("gps_weight", C.c_double)]


class BaLayout(C.Structure):
    """msfm_ba_layout (include/msfm.h)."""
    _fields_ = [("reduced_order", C.c_int), ("system_order", C.c_int), ("n_domains", C.c_int), ("domain_cols", C.c_int * 8),
                ("separator_cols", C.c_int), ("panel_launches", C.c_int), ("n_levels", C.c_int), ("level_nodes", C.c_int * 3),
                ("level_begin", C.c_int * 3), ("root_cols", C.c_int),
                ("cc_entries", C.c_longlong), ("cc_entries_folded", C.c_longlong), ("fold_slots", C.c_int), ("fold_passes", C.c_int),
                ("mc_entries", C.c_longlong), ("mc_entries_folded", C.c_longlong), ("fold_mc_slots", C.c_int), ("reserved_", C.c_int)]


class FransacOptions(C.Structure):
    """msfm_fransac_options (include/msfm.h)."""
    _fields_ = [("threshold", C.c_double), ("confidence", C.c_double), ("max_iterations", C.c_int),
                ("min_points", C.c_int), ("min_inliers", C.c_int), ("seed", C.c_uint64)]


class SlamMatchOptions(C.Structure):
    """msfm_slam_match_options (include/msfm.h)."""
    _fields_ = [("th_first_second_ratio", C.c_float), ("th_epipolar", C.c_float), ("th_distance", C.c_float)]


class BaOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int), ("num_threads", C.c_int), ("progress_to_stdout", C.c_int),
                ("huber_delta", C.c_double), ("function_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("max_num_consecutive_invalid_steps", C.c_int), ("jacobi_scaling", C.c_int)]


class BaIteration(C.Structure):
    _fields_ = [("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double),
                ("step_norm", C.c_double), ("relative_decrease", C.c_double),
                ("trust_region_radius", C.c_double), ("step_is_valid", C.c_int32),
                ("step_is_successful", C.c_int32)]


class BaSummary(C.Structure):
    _fields_ = [("termination", C.c_int), ("num_iterations", C.c_int), ("num_successful_steps", C.c_int),
                ("num_unsuccessful_steps", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("num_residuals", C.c_int), ("num_reduced_params", C.c_int),
                ("iterations", C.POINTER(BaIteration)), ("iterations_capacity", C.c_int),
                ("solve_ms", C.c_double), ("setup_ms", C.c_double)]


class Tracks(C.Structure):
    _fields_ = [("n_tracks", C.c_int), ("n_cams", C.c_int), ("track_off", c_int_p), ("track_cam", c_int_p),
                ("track_xy", c_double_p), ("cam_R", c_double_p), ("cam_t", c_double_p), ("cam_c", c_double_p),
                ("cam_fk", c_double_p)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


ITER_DTYPE = np.dtype([("cost", "f8"), ("cost_change", "f8"), ("gradient_max_norm", "f8"), ("step_norm", "f8"),
                       ("relative_decrease", "f8"), ("trust_region_radius", "f8"), ("step_is_valid", "i4"),
                       ("step_is_successful", "i4")])


def ptr(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def as_c(a, dtype):
    """C-contiguous array of the given dtype (copy only if needed); None passes through."""
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


class BaArrays:
    """Owns contiguous copies of a BA problem's arrays and the ctypes struct over them."""

    def __init__(self, cam_pose, cam_model, cam_model_of_cam, point, obs_cam, obs_pt, obs_xy, pt_weight=None,
                 cam_mutable=None, model_mutable=None, pt_mutable=None, gps_xyz=None, gps_weight=0.0):
        self.cam_pose = np.array(cam_pose, dtype=np.float64, order="C").reshape(-1, 6)
        self.cam_model = np.array(cam_model, dtype=np.float64, order="C").reshape(-1, 3)
        self.point = np.array(point, dtype=np.float64, order="C").reshape(-1, 3)
        self.cam_model_of_cam = as_c(cam_model_of_cam, np.int32)
        self.obs_cam = as_c(obs_cam, np.int32)
        self.obs_pt = as_c(obs_pt, np.int32)
        self.obs_xy = as_c(obs_xy, np.float64)
        self.pt_weight = as_c(np.ones(len(self.point)) if pt_weight is None else pt_weight, np.float64)
        self.cam_mutable = as_c(cam_mutable, np.uint8)
        self.model_mutable = as_c(model_mutable, np.uint8)
        self.pt_mutable = as_c(pt_mutable, np.uint8)
        self.gps_xyz = as_c(gps_xyz, np.float64)
        s = BaProblem()
        s.n_cams, s.n_models, s.n_points, s.n_obs = len(self.cam_pose), len(self.cam_model), len(self.point), len(self.obs_cam)
        s.cam_pose, s.cam_model, s.point = ptr(self.cam_pose, c_double_p), ptr(self.cam_model, c_double_p), ptr(self.point, c_double_p)
        s.cam_model_of_cam, s.obs_cam, s.obs_pt = ptr(self.cam_model_of_cam, c_int_p), ptr(self.obs_cam, c_int_p), ptr(self.obs_pt, c_int_p)
        s.obs_xy, s.pt_weight = ptr(self.obs_xy, c_double_p), ptr(self.pt_weight, c_double_p)
        s.cam_mutable, s.model_mutable, s.pt_mutable = ptr(self.cam_mutable, c_u8_p), ptr(self.model_mutable, c_u8_p), ptr(self.pt_mutable, c_u8_p)
        s.gps_xyz, s.gps_weight = ptr(self.gps_xyz, c_double_p), float(gps_weight)
        self.struct = s

    @classmethod
    def from_scene(cls, sc, **kw):
        return cls(sc.cam_pose, sc.cam_model, sc.cam_model_of_cam, sc.point, sc.obs_cam, sc.obs_pt, sc.obs_xy,
                   sc.pt_weight, **kw)


class SummaryBuf:
    def __init__(self, capacity=512):
        self.rows = np.zeros(capacity, dtype=ITER_DTYPE)
        self.struct = BaSummary()
        self.struct.iterations = self.rows.ctypes.data_as(C.POINTER(BaIteration))
        self.struct.iterations_capacity = capacity

    def result(self):
        s = self.struct
        n = min(s.num_iterations + 1, len(self.rows))
        return dict(termination=TERMINATION.get(s.termination, s.termination), num_iterations=s.num_iterations,
                    num_successful_steps=s.num_successful_steps, num_unsuccessful_steps=s.num_unsuccessful_steps,
                    initial_cost=s.initial_cost, final_cost=s.final_cost, num_residuals=s.num_residuals,
                    num_reduced_params=s.num_reduced_params, solve_ms=s.solve_ms, setup_ms=s.setup_ms,
                    iterations=self.rows[:n].copy())


class TrackArrays:
    def __init__(self, track_off, track_cam, track_xy, cam_R, cam_t, cam_c, cam_fk):
        self.track_off = as_c(track_off, np.int32)
        self.track_cam = as_c(track_cam, np.int32)
        self.track_xy = as_c(track_xy, np.float64)
        self.cam_R, self.cam_t = as_c(cam_R, np.float64), as_c(cam_t, np.float64)
        self.cam_c, self.cam_fk = as_c(cam_c, np.float64), as_c(cam_fk, np.float64)
        s = Tracks()
        s.n_tracks, s.n_cams = len(self.track_off) - 1, len(self.cam_t)
        s.track_off, s.track_cam = ptr(self.track_off, c_int_p), ptr(self.track_cam, c_int_p)
        s.track_xy, s.cam_R, s.cam_t = ptr(self.track_xy, c_double_p), ptr(self.cam_R, c_double_p), ptr(self.cam_t, c_double_p)
        s.cam_c, s.cam_fk = ptr(self.cam_c, c_double_p), ptr(self.cam_fk, c_double_p)
        self.struct = s
