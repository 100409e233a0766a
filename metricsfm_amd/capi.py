"""ctypes binding of libmsfm.so (include/msfm.h).  There is no CPU fallback: if the HIP
library is missing or no GPU is visible, every entry point fails loudly."""
import ctypes as C
import os

import numpy as np

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSFM_LIB") or os.path.join(_HERE, "libmsfm.so")   # (MSFM_LIB: a differently built libmsfm, for A/B timing)
_lib = None

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)

# every symbol include/msfm.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = [
    "msfm_version", "msfm_ctx_create", "msfm_ctx_destroy", "msfm_last_error", "msfm_ctx_stream",
    "msfm_ctx_synchronize", "msfm_ctx_profile_enable", "msfm_ctx_profile_reset", "msfm_ctx_profile_get",
    "msfm_knn2_f32", "msfm_descset_create", "msfm_descset_upload", "msfm_descset_count", "msfm_descset_destroy",
    "msfm_match_pairs", "msfm_match_result_counts", "msfm_match_result_fetch", "msfm_match_result_stats", "msfm_match_result_destroy",
    "msfm_match_pairs_rerun", "msfm_descset_upload_keypoints", "msfm_slam_match_default_options", "msfm_match_pairs_slam",
    "msfm_chain_create", "msfm_chain_verify", "msfm_chain_matches", "msfm_chain_fetch_matches", "msfm_chain_build_tracks",
    "msfm_chain_fetch_tracks", "msfm_chain_triangulate", "msfm_chain_fetch_points", "msfm_chain_ba_create", "msfm_chain_fetch_point_tracks",
    "msfm_chain_destroy",
    "msfm_ba_options_default", "msfm_ba_solve", "msfm_ba_create", "msfm_ba_run",
    "msfm_ba_upload_params", "msfm_ba_download_params", "msfm_ba_destroy", "msfm_ba_get_layout", "msfm_camera_graph_dissection", "msfm_ctx_set_allreduce",
    "msfm_triangulate_midpoint_batch", "msfm_triangulate_dlt_batch", "msfm_reproject_mse_batch",
    "msfm_epipolar_filter", "msfm_fransac_default_options", "msfm_fundamental_ransac_batch",
    "msfm_epipolar_filter_batch", "msfm_tracks_build", "msfm_tracks_build_device", "msfm_track_set_size", "msfm_track_set_fetch", "msfm_track_set_destroy",
    "msfm_epnp_ransac_batch", "msfm_relpose_5pt_batch", "msfm_rccl_get_unique_id", "msfm_ctx_init_rccl", "msfm_ctx_allreduce",
    "msfm_ctx_create_multi", "msfm_multi_destroy", "msfm_multi_size", "msfm_multi_ctx", "msfm_multi_last_error", "msfm_multi_ba_solve",
    "msfm_multi_triangulate_midpoint_batch", "msfm_multi_triangulate_dlt_batch", "msfm_multi_reproject_mse_batch", "msfm_multi_match_pairs",
]


class MsfmError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libmsfm error %d: %s" % (code, text))
        self.code = code


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` or "
                          "`make -C metricsfm_amd/csrc` (no CPU fallback exists)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i, d, f = C.c_void_p, C.c_int, C.c_double, C.c_float
    L.msfm_version.restype = i
    L.msfm_ctx_create.argtypes = [i, C.POINTER(vp)]
    L.msfm_ctx_destroy.argtypes = [vp]
    L.msfm_ctx_destroy.restype = None
    L.msfm_last_error.argtypes = [vp]
    L.msfm_last_error.restype = C.c_char_p
    L.msfm_ctx_stream.argtypes = [vp]
    L.msfm_ctx_stream.restype = vp
    L.msfm_ctx_synchronize.argtypes = [vp]
    L.msfm_ctx_profile_enable.argtypes = [vp, i]
    L.msfm_ctx_profile_reset.argtypes = [vp]
    L.msfm_ctx_profile_get.argtypes = [vp, C.POINTER(A.KernelStat), i, C.POINTER(i)]
    L.msfm_knn2_f32.argtypes = [vp, A.c_float_p, i, A.c_float_p, i, i, A.c_int_p, A.c_float_p]
    L.msfm_descset_create.argtypes = [vp, i, i, C.POINTER(vp)]
    L.msfm_descset_upload.argtypes = [vp, i, A.c_float_p, i]
    L.msfm_descset_count.argtypes = [vp, i]
    L.msfm_descset_destroy.argtypes = [vp]
    L.msfm_descset_destroy.restype = None
    L.msfm_match_pairs.argtypes = [vp, A.c_int_p, i, f, f, i, C.POINTER(vp)]
    L.msfm_match_pairs_rerun.argtypes = [vp, vp]
    L.msfm_descset_upload_keypoints.argtypes = [vp, i, A.c_float_p, i]
    L.msfm_slam_match_default_options.argtypes = [C.POINTER(A.SlamMatchOptions)]
    L.msfm_slam_match_default_options.restype = None
    L.msfm_match_pairs_slam.argtypes = [vp, A.c_int_p, i, A.c_double_p, A.c_double_p, C.POINTER(A.SlamMatchOptions), i, C.POINTER(vp)]
    L.msfm_match_result_counts.argtypes = [vp, A.c_int_p, A.c_int_p]
    L.msfm_match_result_fetch.argtypes = [vp, i, A.c_int_p, A.c_int_p, A.c_float_p]
    L.msfm_match_result_stats.argtypes = [vp, A.c_int_p, A.c_int_p]
    L.msfm_match_result_destroy.argtypes = [vp]
    L.msfm_match_result_destroy.restype = None
    L.msfm_ba_options_default.argtypes = [C.POINTER(A.BaOptions)]
    L.msfm_ba_options_default.restype = None
    L.msfm_ba_solve.argtypes = [vp, C.POINTER(A.BaProblem), C.POINTER(A.BaOptions), C.POINTER(A.BaSummary)]
    L.msfm_ba_create.argtypes = [vp, C.POINTER(A.BaProblem), C.POINTER(vp)]
    L.msfm_ba_run.argtypes = [vp, C.POINTER(A.BaOptions), C.POINTER(A.BaSummary)]
    L.msfm_ba_upload_params.argtypes = [vp, A.c_double_p, A.c_double_p, A.c_double_p]
    L.msfm_ba_download_params.argtypes = [vp, A.c_double_p, A.c_double_p, A.c_double_p]
    L.msfm_ba_destroy.argtypes = [vp]
    L.msfm_ba_destroy.restype = None
    L.msfm_ctx_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp, i, i]
    for fn in (L.msfm_triangulate_midpoint_batch, L.msfm_triangulate_dlt_batch):
        fn.argtypes = [vp, C.POINTER(A.Tracks), d, d, A.c_double_p, A.c_double_p, A.c_u8_p]
    L.msfm_reproject_mse_batch.argtypes = [vp, C.POINTER(A.Tracks), A.c_double_p, A.c_double_p]
    L.msfm_epipolar_filter.argtypes = [vp, A.c_float_p, A.c_float_p, i, A.c_double_p, d, A.c_u8_p]
    L.msfm_ba_get_layout.argtypes = [vp, C.POINTER(A.BaLayout)]
    L.msfm_camera_graph_dissection.argtypes = [i, vp, i, i, vp, C.POINTER(i), C.POINTER(i)]
    L.msfm_fransac_default_options.argtypes = [C.POINTER(A.FransacOptions)]
    L.msfm_fransac_default_options.restype = None
    L.msfm_fundamental_ransac_batch.argtypes = [vp, i, A.c_int_p, A.c_float_p, A.c_float_p, C.POINTER(A.FransacOptions),
                                                A.c_double_p, A.c_u8_p, A.c_int_p, A.c_u8_p]
    L.msfm_epipolar_filter_batch.argtypes = [vp, i, A.c_int_p, A.c_float_p, A.c_float_p, A.c_double_p, A.c_u8_p, d, A.c_u8_p]
    L.msfm_epnp_ransac_batch.argtypes = [vp, i, A.c_int_p, A.c_double_p, A.c_double_p, A.c_double_p, i, C.c_uint64, A.c_double_p,
                                         A.c_double_p, A.c_double_p, A.c_double_p, A.c_int_p]
    L.msfm_relpose_5pt_batch.argtypes = [vp, i, A.c_int_p, A.c_double_p, A.c_double_p, A.c_double_p, A.c_double_p, i, C.c_uint64,
                                         A.c_double_p, A.c_double_p, A.c_double_p, A.c_u8_p, A.c_int_p]
    L.msfm_rccl_get_unique_id.argtypes = [vp, C.POINTER(C.c_ubyte)]
    L.msfm_ctx_init_rccl.argtypes = [vp, C.POINTER(C.c_ubyte), i, i]
    L.msfm_ctx_allreduce.argtypes = [vp, vp, C.c_size_t, i]
    L.msfm_tracks_build.argtypes = [i, A.c_int_p, i, A.c_int_p, A.c_int_p, A.c_int_p, C.POINTER(vp)]
    L.msfm_tracks_build_device.argtypes = [vp, i, A.c_int_p, i, A.c_int_p, A.c_int_p, A.c_int_p, C.POINTER(vp)]
    L.msfm_track_set_size.argtypes = [vp, A.c_int_p, A.c_int_p]
    L.msfm_track_set_fetch.argtypes = [vp, A.c_int_p, A.c_int_p, A.c_int_p]
    L.msfm_track_set_destroy.argtypes = [vp]
    L.msfm_track_set_destroy.restype = None
    L.msfm_chain_create.argtypes = [vp, C.POINTER(vp)]
    L.msfm_chain_verify.argtypes = [vp, vp, C.POINTER(A.FransacOptions), d]
    L.msfm_chain_matches.argtypes = [vp, A.c_int_p, A.c_u8_p, A.c_double_p]
    L.msfm_chain_fetch_matches.argtypes = [vp, i, A.c_int_p]
    L.msfm_chain_build_tracks.argtypes = [vp, A.c_int_p, A.c_int_p]
    L.msfm_chain_fetch_tracks.argtypes = [vp, A.c_int_p, A.c_int_p, A.c_int_p]
    L.msfm_chain_triangulate.argtypes = [vp, i, A.c_double_p, A.c_double_p, A.c_double_p, A.c_double_p, d, d, A.c_int_p]
    L.msfm_chain_fetch_points.argtypes = [vp, A.c_double_p, A.c_double_p, A.c_u8_p]
    L.msfm_chain_ba_create.argtypes = [vp, i, i, A.c_double_p, A.c_double_p, A.c_int_p, i, d, C.POINTER(vp), A.c_int_p, A.c_int_p]
    L.msfm_chain_fetch_point_tracks.argtypes = [vp, A.c_int_p]
    L.msfm_chain_destroy.argtypes = [vp]
    L.msfm_chain_destroy.restype = None
    L.msfm_ctx_create_multi.argtypes = [i, A.c_int_p, C.POINTER(vp)]
    L.msfm_multi_destroy.argtypes = [vp]
    L.msfm_multi_destroy.restype = None
    L.msfm_multi_size.argtypes = [vp]
    L.msfm_multi_ctx.argtypes = [vp, i]
    L.msfm_multi_ctx.restype = vp
    L.msfm_multi_last_error.argtypes = [vp]
    L.msfm_multi_last_error.restype = C.c_char_p
    L.msfm_multi_ba_solve.argtypes = [vp, C.POINTER(A.BaProblem), C.POINTER(A.BaOptions), C.POINTER(A.BaSummary)]
    L.msfm_multi_triangulate_midpoint_batch.argtypes = [vp, C.POINTER(A.Tracks), d, d, A.c_double_p, A.c_double_p, A.c_u8_p]
    L.msfm_multi_triangulate_dlt_batch.argtypes = [vp, C.POINTER(A.Tracks), d, d, A.c_double_p, A.c_double_p, A.c_u8_p]
    L.msfm_multi_reproject_mse_batch.argtypes = [vp, C.POINTER(A.Tracks), A.c_double_p, A.c_double_p]
    L.msfm_multi_match_pairs.argtypes = [vp, i, C.POINTER(A.c_float_p), A.c_int_p, i, A.c_int_p, i, f, f, C.POINTER(A.c_int_p), A.c_int_p, A.c_int_p]
    _lib = L
    return L


def default_options(**kw):
    o = A.BaOptions()
    lib().msfm_ba_options_default(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def _flatten_matches(n_features, pairs, matches_per_pair):
    nf = A.as_c(np.asarray(n_features, dtype=np.int32), np.int32)
    pr = A.as_c(np.asarray(pairs, dtype=np.int32).reshape(-1, 2), np.int32)
    lens = np.array([len(m) for m in matches_per_pair], dtype=np.int64)
    off = np.zeros(len(pr) + 1, dtype=np.int32)
    off[1:] = np.cumsum(lens)
    nonempty = [np.asarray(m, dtype=np.int32).reshape(-1, 2) for m in matches_per_pair if len(m)]
    flat = np.concatenate(nonempty) if nonempty else np.zeros((1, 2), dtype=np.int32)
    return nf, pr, off, A.as_c(flat, np.int32)


def _fetch_track_set(h):
    try:
        nt, no = C.c_int32(), C.c_int32()
        lib().msfm_track_set_size(h, C.byref(nt), C.byref(no))
        toff = np.zeros(nt.value + 1, dtype=np.int32)
        oi, of = np.zeros(max(1, no.value), dtype=np.int32), np.zeros(max(1, no.value), dtype=np.int32)
        lib().msfm_track_set_fetch(h, A.ptr(toff, A.c_int_p), A.ptr(oi, A.c_int_p), A.ptr(of, A.c_int_p))
    finally:
        lib().msfm_track_set_destroy(h)
    return toff, oi[:no.value], of[:no.value]


def build_tracks(n_features, pairs, matches_per_pair):
    """SLAMGPS::Triangulation's data association (slam_gps.cc:565-635), the walk over the match lists on the host as the
    reference does it.  n_features[image]; pairs [(idx1, idx2)] in visiting order; matches_per_pair[p] = int array [m][2]
    (feature in idx1, feature in idx2).
    Returns CSR tracks: track_off, obs_image, obs_feature (observations in ascending image order).
    `Context.build_tracks` returns the same from the GPU."""
    nf, pr, off, flat = _flatten_matches(n_features, pairs, matches_per_pair)
    h = C.c_void_p()
    rc = lib().msfm_tracks_build(len(nf), A.ptr(nf, A.c_int_p), len(pr), A.ptr(pr, A.c_int_p), A.ptr(off, A.c_int_p),
                                 A.ptr(flat, A.c_int_p), C.byref(h))
    if rc != 0:
        raise MsfmError(rc, "msfm_tracks_build: invalid input")
    return _fetch_track_set(h)


def build_tracks_flat(n_features, pairs, match_off, matches):
    """`build_tracks` on flat int32 arrays (pairs [P][2], match_off [P+1], matches [M][2])."""
    nf, pr, off = (A.as_c(np.asarray(x, dtype=np.int32), np.int32) for x in (n_features, pairs, match_off))
    fl = A.as_c(np.asarray(matches, dtype=np.int32).reshape(-1, 2), np.int32)
    h = C.c_void_p()
    rc = lib().msfm_tracks_build(len(nf), A.ptr(nf, A.c_int_p), len(pr), A.ptr(pr, A.c_int_p), A.ptr(off, A.c_int_p),
                                 A.ptr(fl, A.c_int_p), C.byref(h))
    if rc != 0:
        raise MsfmError(rc, "msfm_tracks_build: invalid input")
    return _fetch_track_set(h)


def fransac_options(**kw):
    o = A.FransacOptions()
    lib().msfm_fransac_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


class Context:
    """One per GPU (one process per GPU)."""

    def __init__(self, device=-1):
        self._h = C.c_void_p()
        rc = lib().msfm_ctx_create(device, C.byref(self._h))
        if rc != 0:
            raise MsfmError(rc, "msfm_ctx_create failed (no visible GPU? libmsfm has no CPU fallback)")
        self._cb = None

    def close(self):
        if self._h:
            lib().msfm_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise MsfmError(rc, lib().msfm_last_error(self._h).decode())

    @property
    def stream(self):
        return lib().msfm_ctx_stream(self._h)

    def synchronize(self):
        self.check(lib().msfm_ctx_synchronize(self._h))

    # -- profiling --
    def profile(self, enable=True):
        self.check(lib().msfm_ctx_profile_enable(self._h, int(enable)))

    def profile_reset(self):
        self.check(lib().msfm_ctx_profile_reset(self._h))

    def profile_get(self):
        arr = (A.KernelStat * A.MSFM_MAX_KERNEL_STATS)()
        n = C.c_int()
        self.check(lib().msfm_ctx_profile_get(self._h, arr, A.MSFM_MAX_KERNEL_STATS, C.byref(n)))
        return {arr[k].name.decode(): dict(launches=int(arr[k].launches), total_ms=float(arr[k].total_ms))
                for k in range(n.value)}

    def set_allreduce(self, fn, rank, world_size):
        """fn(buf_ptr:int, count:int, op:int, stream_ptr:int) -> int (0 ok); op 0 = sum, 1 = max."""
        if fn is None:
            self._cb = ALLREDUCE_FN()
        else:
            def tramp(user, buf, count, op, stream):
                try:
                    return int(fn(buf, count, op, stream) or 0)
                except Exception:  # never let an exception cross the C boundary
                    import traceback
                    traceback.print_exc()
                    return -1
            self._cb = ALLREDUCE_FN(tramp)
        self.check(lib().msfm_ctx_set_allreduce(self._h, self._cb, None, rank, world_size))

    def rccl_unique_id(self):
        """128 opaque bytes (ncclUniqueId) from rank 0, to be handed to every rank's init_rccl."""
        buf = (C.c_ubyte * 128)()
        self.check(lib().msfm_rccl_get_unique_id(self._h, buf))
        return bytes(buf)

    def init_rccl(self, unique_id, rank, world_size):
        """Collective: the library creates its own RCCL communicator and reduces with ncclAllReduce on its stream -
        no Python in the LM loop (msfm_ctx_init_rccl)."""
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        self.check(lib().msfm_ctx_init_rccl(self._h, buf, rank, world_size))

    def allreduce(self, dev_ptr, count, op=0):
        """msfm_ctx_allreduce on a device buffer of doubles (op 0 = sum, 1 = max), ordered on the context's stream."""
        self.check(lib().msfm_ctx_allreduce(self._h, dev_ptr, count, op))

    # -- matching --
    def knn2(self, train, query):
        """fine_matching_graph.cc:99 shaped call: returns ids [nq,2] i32, sqdists [nq,2] f32."""
        train, query = A.as_c(train, np.float32), A.as_c(query, np.float32)
        if train.ndim != 2 or query.ndim != 2 or train.shape[1] != query.shape[1]:
            raise ValueError("train/query must be [n, dim] with equal dim")
        ids = np.zeros((len(query), 2), dtype=np.int32)
        d = np.zeros((len(query), 2), dtype=np.float32)
        self.check(lib().msfm_knn2_f32(self._h, A.ptr(train, A.c_float_p), len(train), A.ptr(query, A.c_float_p),
                                       len(query), train.shape[1], A.ptr(ids, A.c_int_p), A.ptr(d, A.c_float_p)))
        return ids, d

    def descset(self, descs, keypoints=None):
        return DescSet(self, descs, keypoints)

    # -- bundle adjustment --
    def ba_solve(self, arrays: A.BaArrays, options=None, capacity=512):
        """msfm_ba_solve: optimises `arrays` IN PLACE; returns the summary dict."""
        options = options or default_options()
        buf = A.SummaryBuf(capacity)
        self.check(lib().msfm_ba_solve(self._h, C.byref(arrays.struct), C.byref(options), C.byref(buf.struct)))
        return buf.result()

    def ba(self, arrays: A.BaArrays):
        return BaResident(self, arrays)

    # -- triangulation --
    def _tri(self, fn, tracks, th_error, th_angle, X0):
        n = tracks.struct.n_tracks
        X = np.zeros((n, 3)) if X0 is None else np.array(X0, dtype=np.float64, order="C")
        mse, ok = np.zeros(n), np.zeros(n, dtype=np.uint8)
        self.check(fn(self._h, C.byref(tracks.struct), th_error, th_angle, A.ptr(X, A.c_double_p),
                      A.ptr(mse, A.c_double_p), A.ptr(ok, A.c_u8_p)))
        return X, mse, ok

    def build_tracks(self, n_features, pairs, matches_per_pair, flat=None):
        """`build_tracks` on the GPU (msfm_tracks_build_device): identical output.  `flat` = (n_features, pairs [p][2],
        match_off [p+1], matches [m][2]) int32 arrays skips the per-pair Python lists."""
        nf, pr, off, fl = flat if flat is not None else _flatten_matches(n_features, pairs, matches_per_pair)
        h = C.c_void_p()
        self.check(lib().msfm_tracks_build_device(self._h, len(nf), A.ptr(nf, A.c_int_p), len(pr), A.ptr(pr, A.c_int_p),
                                                  A.ptr(off, A.c_int_p), A.ptr(fl, A.c_int_p), C.byref(h)))
        return _fetch_track_set(h)

    def triangulate_midpoint(self, tracks, th_error, th_angle, X0=None):
        return self._tri(lib().msfm_triangulate_midpoint_batch, tracks, th_error, th_angle, X0)

    def triangulate_dlt(self, tracks, th_error, th_angle, X0=None):
        return self._tri(lib().msfm_triangulate_dlt_batch, tracks, th_error, th_angle, X0)

    def reproject_mse(self, tracks, X):
        X = A.as_c(X, np.float64)
        mse = np.zeros(tracks.struct.n_tracks)
        self.check(lib().msfm_reproject_mse_batch(self._h, C.byref(tracks.struct), A.ptr(X, A.c_double_p),
                                                  A.ptr(mse, A.c_double_p)))
        return mse

    def epipolar_filter(self, pt1, pt2, F, th=3.0):
        pt1, pt2 = A.as_c(pt1, np.float32), A.as_c(pt2, np.float32)
        F = A.as_c(np.asarray(F, dtype=np.float64).reshape(9), np.float64)
        out = np.zeros(len(pt1), dtype=np.uint8)
        self.check(lib().msfm_epipolar_filter(self._h, A.ptr(pt1, A.c_float_p), A.ptr(pt2, A.c_float_p), len(pt1),
                                              A.ptr(F, A.c_double_p), th, A.ptr(out, A.c_u8_p)))
        return out

    def fundamental_ransac(self, offsets, pt1, pt2, **opts):
        """GeoVerification::GeoVerificationFundamental for a batch of pairs (geo_verification.cc:30-58).
        offsets[n_pairs+1] delimit each pair's matches in pt1/pt2 (float32 [total][2]).
        Returns F [n_pairs][3][3], inlier mask [total], n_inliers [n_pairs], ok [n_pairs]."""
        offsets = A.as_c(offsets, np.int32)
        pt1, pt2 = A.as_c(np.asarray(pt1, dtype=np.float32).reshape(-1, 2), np.float32), A.as_c(np.asarray(pt2, dtype=np.float32).reshape(-1, 2), np.float32)
        n_pairs = len(offsets) - 1
        o = fransac_options(**opts)
        F = np.zeros((n_pairs, 3, 3), dtype=np.float64)
        inl = np.zeros(max(1, len(pt1)), dtype=np.uint8)
        nin = np.zeros(max(1, n_pairs), dtype=np.int32)
        ok = np.zeros(max(1, n_pairs), dtype=np.uint8)
        self.check(lib().msfm_fundamental_ransac_batch(self._h, n_pairs, A.ptr(offsets, A.c_int_p), A.ptr(pt1, A.c_float_p),
                                                       A.ptr(pt2, A.c_float_p), C.byref(o), A.ptr(F, A.c_double_p),
                                                       A.ptr(inl, A.c_u8_p), A.ptr(nin, A.c_int_p), A.ptr(ok, A.c_u8_p)))
        return F, inl[:len(pt1)], nin[:n_pairs], ok[:n_pairs]

    def epnp_ransac(self, offsets, pts_w, pts_2d, f, max_iter=200, seed=0x4D53464D50):
        """AbsolutePoseEstimation::AbsolutePoseWithFocalLength for a batch of images (absolute_pose_estimation.cc:42-58):
        EPnP RANSAC over 4-point samples + the reprojection errors of all correspondences.
        Returns R [n][3][3], t [n][3], errors [total], avg_error [n], best_iter [n]."""
        offsets = A.as_c(offsets, np.int32)
        pts_w = A.as_c(np.asarray(pts_w, dtype=np.float64).reshape(-1, 3), np.float64)
        pts_2d = A.as_c(np.asarray(pts_2d, dtype=np.float64).reshape(-1, 2), np.float64)
        n = len(offsets) - 1
        f = A.as_c(np.broadcast_to(np.asarray(f, dtype=np.float64), (n,)).copy(), np.float64)
        R = np.zeros((max(1, n), 3, 3)); t = np.zeros((max(1, n), 3)); err = np.zeros(max(1, len(pts_w))); avg = np.zeros(max(1, n))
        best = np.zeros(max(1, n), dtype=np.int32)
        self.check(lib().msfm_epnp_ransac_batch(self._h, n, A.ptr(offsets, A.c_int_p), A.ptr(pts_w, A.c_double_p), A.ptr(pts_2d, A.c_double_p),
                                                A.ptr(f, A.c_double_p), max_iter, seed, A.ptr(R, A.c_double_p), A.ptr(t, A.c_double_p),
                                                A.ptr(err, A.c_double_p), A.ptr(avg, A.c_double_p), A.ptr(best, A.c_int_p)))
        return R[:n], t[:n], err[:len(pts_w)], avg[:n], best[:n]

    def relpose_5pt(self, offsets, pts_ref, pts_cur, f_ref, f_cur, ransac_times=100, seed=0x4D53464D45):
        """RelativePoseEstimation::RelativePoseWithFocalLength for a batch of image pairs (relative_pose_estimation.cc:91-120):
        five-point RANSAC + decomposition of the best essential matrix.
        Returns E [n][3][3], R [n][3][3], t [n][3], ok [n], n_candidates [n]."""
        offsets = A.as_c(offsets, np.int32)
        pts_ref = A.as_c(np.asarray(pts_ref, dtype=np.float64).reshape(-1, 2), np.float64)
        pts_cur = A.as_c(np.asarray(pts_cur, dtype=np.float64).reshape(-1, 2), np.float64)
        n = len(offsets) - 1
        f_ref = A.as_c(np.broadcast_to(np.asarray(f_ref, dtype=np.float64), (n,)).copy(), np.float64)
        f_cur = A.as_c(np.broadcast_to(np.asarray(f_cur, dtype=np.float64), (n,)).copy(), np.float64)
        E = np.zeros((max(1, n), 3, 3)); R = np.zeros((max(1, n), 3, 3)); t = np.zeros((max(1, n), 3))
        ok = np.zeros(max(1, n), dtype=np.uint8); nc = np.zeros(max(1, n), dtype=np.int32)
        self.check(lib().msfm_relpose_5pt_batch(self._h, n, A.ptr(offsets, A.c_int_p), A.ptr(pts_ref, A.c_double_p), A.ptr(pts_cur, A.c_double_p),
                                                A.ptr(f_ref, A.c_double_p), A.ptr(f_cur, A.c_double_p), ransac_times, seed,
                                                A.ptr(E, A.c_double_p), A.ptr(R, A.c_double_p), A.ptr(t, A.c_double_p), A.ptr(ok, A.c_u8_p),
                                                A.ptr(nc, A.c_int_p)))
        return E[:n], R[:n], t[:n], ok[:n], nc[:n]

    def epipolar_filter_batch(self, offsets, pt1, pt2, F, ok=None, th=3.0):
        offsets = A.as_c(offsets, np.int32)
        pt1, pt2 = A.as_c(np.asarray(pt1, dtype=np.float32).reshape(-1, 2), np.float32), A.as_c(np.asarray(pt2, dtype=np.float32).reshape(-1, 2), np.float32)
        F = A.as_c(np.asarray(F, dtype=np.float64).reshape(-1, 9), np.float64)
        okp = None
        if ok is not None:
            ok = A.as_c(ok, np.uint8)
            okp = A.ptr(ok, A.c_u8_p)
        out = np.zeros(max(1, len(pt1)), dtype=np.uint8)
        self.check(lib().msfm_epipolar_filter_batch(self._h, len(offsets) - 1, A.ptr(offsets, A.c_int_p), A.ptr(pt1, A.c_float_p),
                                                    A.ptr(pt2, A.c_float_p), A.ptr(F, A.c_double_p), okp, th, A.ptr(out, A.c_u8_p)))
        return out[:len(pt1)]


class DescSet:
    """Device-resident descriptors of a set of images (msfm_descset)."""

    def __init__(self, ctx: Context, descs, keypoints=None):
        self.ctx = ctx
        self._h = C.c_void_p()
        dim = descs[0].shape[1]
        ctx.check(lib().msfm_descset_create(ctx._h, len(descs), dim, C.byref(self._h)))
        for i, d in enumerate(descs):
            d = A.as_c(d, np.float32)
            ctx.check(lib().msfm_descset_upload(self._h, i, A.ptr(d, A.c_float_p), len(d)))
        self.counts = [len(d) for d in descs]
        if keypoints is not None:
            for i, xy in enumerate(keypoints):
                if xy is not None:
                    self.upload_keypoints(i, xy)

    def upload_keypoints(self, image, xy):
        """msfm_descset_upload_keypoints: [count][2] float positions of the image's features (cv::Point2f)."""
        xy = A.as_c(np.asarray(xy).reshape(-1, 2), np.float32)
        self.ctx.check(lib().msfm_descset_upload_keypoints(self._h, image, A.ptr(xy, A.c_float_p), len(xy)))

    def match_pairs(self, pairs, ratio_good=0.6, ratio_all=0.85, keep_knn=False):
        return MatchResult(self, pairs, ratio_good, ratio_all, keep_knn)

    def match_pairs_slam(self, pairs, F, H, keep_knn=False, **opts):
        """msfm_match_pairs_slam (slam_gps.cc:455-503): ratio test `> th`, then the prior F / H gates; F, H [n_pairs][3][3]."""
        return MatchResult(self, pairs, None, None, keep_knn, slam=(F, H, opts))

    def close(self):
        if self._h:
            lib().msfm_descset_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MatchResult:
    def __init__(self, ds: DescSet, pairs, ratio_good, ratio_all, keep_knn, slam=None):
        self.ds, self.ctx = ds, ds.ctx
        self.pairs = A.as_c(np.asarray(pairs).reshape(-1, 2), np.int32)
        self.keep_knn = keep_knn
        self._h = C.c_void_p()
        if slam is not None:
            F, H, kw = slam
            F = A.as_c(np.asarray(F, dtype=np.float64).reshape(-1, 9), np.float64)
            H = A.as_c(np.asarray(H, dtype=np.float64).reshape(-1, 9), np.float64)
            if len(F) != len(self.pairs) or len(H) != len(self.pairs):
                raise ValueError("one F and one H per pair")
            o = A.SlamMatchOptions()
            lib().msfm_slam_match_default_options(C.byref(o))
            for k, v in kw.items():
                if not hasattr(o, k):
                    raise AttributeError(k)
                setattr(o, k, v)
            self.ctx.check(lib().msfm_match_pairs_slam(ds._h, A.ptr(self.pairs, A.c_int_p), len(self.pairs), A.ptr(F, A.c_double_p),
                                                       A.ptr(H, A.c_double_p), C.byref(o), int(keep_knn), C.byref(self._h)))
            return
        self.ctx.check(lib().msfm_match_pairs(ds._h, A.ptr(self.pairs, A.c_int_p), len(self.pairs), ratio_good,
                                              ratio_all, int(keep_knn), C.byref(self._h)))

    def rerun(self):
        self.ctx.check(lib().msfm_match_pairs_rerun(self.ds._h, self._h))

    def counts(self):
        na, ng = np.zeros(len(self.pairs), np.int32), np.zeros(len(self.pairs), np.int32)
        self.ctx.check(lib().msfm_match_result_counts(self._h, A.ptr(na, A.c_int_p), A.ptr(ng, A.c_int_p)))
        return na, ng

    def stats(self):
        nq, ns = C.c_int32(), C.c_int32()
        self.ctx.check(lib().msfm_match_result_stats(self._h, C.byref(nq), C.byref(ns)))
        return dict(queries=nq.value, slow_path=ns.value)

    def fetch(self, pair):
        nq = self.ds.counts[self.pairs[pair, 1]]
        code = np.zeros(nq, np.int32)
        ids = np.zeros((nq, 2), np.int32) if self.keep_knn else None
        d = np.zeros((nq, 2), np.float32) if self.keep_knn else None
        self.ctx.check(lib().msfm_match_result_fetch(self._h, pair, A.ptr(code, A.c_int_p), A.ptr(ids, A.c_int_p),
                                                     A.ptr(d, A.c_float_p)))
        return code, ids, d

    def close(self):
        if self._h:
            lib().msfm_match_result_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Chain:
    """msfm_chain: match codes -> verification -> tracks -> triangulation -> bundle adjustment, everything resident
    (include/msfm.h).  Built from a MatchResult whose DescSet holds the keypoints."""

    def __init__(self, res):
        self.ctx, self.res = res.ctx, res
        self.n_pairs = len(res.pairs)
        self._h = C.c_void_p()
        self.ctx.check(lib().msfm_chain_create(res._h, C.byref(self._h)))

    def verify(self, th_filter=3.0, **opts):
        o = fransac_options(**opts)
        self.ctx.check(lib().msfm_chain_verify(self._h, self.res._h, C.byref(o), th_filter))
        n = np.zeros(max(1, self.n_pairs), np.int32)
        ok = np.zeros(max(1, self.n_pairs), np.uint8)
        F = np.zeros((max(1, self.n_pairs), 3, 3))
        self.ctx.check(lib().msfm_chain_matches(self._h, A.ptr(n, A.c_int_p), A.ptr(ok, A.c_u8_p), A.ptr(F, A.c_double_p)))
        self.n_matches = n[:self.n_pairs]
        return self.n_matches, ok[:self.n_pairs], F[:self.n_pairs]

    def fetch_matches(self, pair):
        m = np.zeros((int(self.n_matches[pair]), 2), np.int32)
        self.ctx.check(lib().msfm_chain_fetch_matches(self._h, pair, A.ptr(m, A.c_int_p)))
        return m

    def build_tracks(self):
        nt, no = C.c_int32(), C.c_int32()
        self.ctx.check(lib().msfm_chain_build_tracks(self._h, C.byref(nt), C.byref(no)))
        self.n_tracks, self.n_obs = nt.value, no.value
        return self.n_tracks, self.n_obs

    def fetch_tracks(self):
        off = np.zeros(self.n_tracks + 1, np.int32)
        img, feat = np.zeros(max(1, self.n_obs), np.int32), np.zeros(max(1, self.n_obs), np.int32)
        self.ctx.check(lib().msfm_chain_fetch_tracks(self._h, A.ptr(off, A.c_int_p), A.ptr(img, A.c_int_p), A.ptr(feat, A.c_int_p)))
        return off, img[:self.n_obs], feat[:self.n_obs]

    def triangulate(self, R, t, c, fk, th_error, th_angle):
        R, t, c, fk = (A.as_c(np.asarray(x, dtype=np.float64), np.float64) for x in (R, t, c, fk))
        n = C.c_int32()
        self.ctx.check(lib().msfm_chain_triangulate(self._h, len(t), A.ptr(R, A.c_double_p), A.ptr(t, A.c_double_p), A.ptr(c, A.c_double_p),
                                                    A.ptr(fk, A.c_double_p), th_error, th_angle, C.byref(n)))
        return n.value

    def fetch_points(self):
        X, mse, ok = np.zeros((max(1, self.n_tracks), 3)), np.zeros(max(1, self.n_tracks)), np.zeros(max(1, self.n_tracks), np.uint8)
        self.ctx.check(lib().msfm_chain_fetch_points(self._h, A.ptr(X, A.c_double_p), A.ptr(mse, A.c_double_p), A.ptr(ok, A.c_u8_p)))
        return X[:self.n_tracks], mse[:self.n_tracks], ok[:self.n_tracks]

    def ba_create(self, cam_pose, cam_model, cam_model_of_cam, min_views=3, weight_ge3=1.0):
        """msfm_chain_ba_create: returns a BaResident on the accepted tracks (its `.arrays` hold only the camera side: the
        points live on the device; `download()` returns them in `track_of_point` order)."""
        cam_pose, cam_model = A.as_c(np.array(cam_pose, dtype=np.float64), np.float64), A.as_c(np.array(cam_model, dtype=np.float64), np.float64)
        moc = A.as_c(np.asarray(cam_model_of_cam, dtype=np.int32), np.int32)
        h, npt, nob = C.c_void_p(), C.c_int32(), C.c_int32()
        self.ctx.check(lib().msfm_chain_ba_create(self._h, len(cam_pose), len(cam_model), A.ptr(cam_pose, A.c_double_p), A.ptr(cam_model, A.c_double_p),
                                                  A.ptr(moc, A.c_int_p), min_views, weight_ge3, C.byref(h), C.byref(npt), C.byref(nob)))
        top = np.zeros(npt.value, np.int32)
        self.ctx.check(lib().msfm_chain_fetch_point_tracks(self._h, A.ptr(top, A.c_int_p)))

        class _Shapes:   # what BaResident.download needs to size its buffers
            pass
        sh = _Shapes()
        sh.cam_pose, sh.cam_model, sh.point = cam_pose, cam_model, np.zeros((npt.value, 3))
        ba = BaResident.__new__(BaResident)
        ba.ctx, ba.arrays, ba._h = self.ctx, sh, h
        ba.track_of_point, ba.n_obs = top, nob.value
        return ba

    def close(self):
        if self._h:
            lib().msfm_chain_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BaResident:
    """msfm_ba_create / run / upload / download: a BA problem kept in HBM across solves."""

    def __init__(self, ctx: Context, arrays: A.BaArrays):
        self.ctx, self.arrays = ctx, arrays
        self._h = C.c_void_p()
        ctx.check(lib().msfm_ba_create(ctx._h, C.byref(arrays.struct), C.byref(self._h)))

    def run(self, options=None, capacity=512):
        options = options or default_options()
        buf = A.SummaryBuf(capacity)
        self.ctx.check(lib().msfm_ba_run(self._h, C.byref(options), C.byref(buf.struct)))
        return buf.result()

    def layout(self):
        lay = A.BaLayout()
        self.ctx.check(lib().msfm_ba_get_layout(self._h, C.byref(lay)))
        return dict(reduced_order=lay.reduced_order, system_order=lay.system_order, n_domains=lay.n_domains,
                    domain_cols=list(lay.domain_cols)[:max(0, lay.n_domains if lay.n_domains > 1 else 0)],
                    separator_cols=lay.separator_cols, panel_launches=lay.panel_launches, n_levels=lay.n_levels,
                    level_nodes=list(lay.level_nodes)[:lay.n_levels], level_begin=list(lay.level_begin)[:lay.n_levels], root_cols=lay.root_cols,
                    fold=dict(cc_entries=lay.cc_entries, cc_entries_folded=lay.cc_entries_folded, slots=lay.fold_slots, passes=lay.fold_passes,
                              mc_entries=lay.mc_entries, mc_entries_folded=lay.mc_entries_folded, mc_slots=lay.fold_mc_slots))

    def upload(self, cam_pose=None, cam_model=None, point=None):
        cp, cm, pt = A.as_c(cam_pose, np.float64), A.as_c(cam_model, np.float64), A.as_c(point, np.float64)
        self.ctx.check(lib().msfm_ba_upload_params(self._h, A.ptr(cp, A.c_double_p), A.ptr(cm, A.c_double_p),
                                                   A.ptr(pt, A.c_double_p)))

    def download(self):
        a = self.arrays
        cp, cm, pt = np.zeros_like(a.cam_pose), np.zeros_like(a.cam_model), np.zeros_like(a.point)
        self.ctx.check(lib().msfm_ba_download_params(self._h, A.ptr(cp, A.c_double_p), A.ptr(cm, A.c_double_p),
                                                     A.ptr(pt, A.c_double_p)))
        return cp, cm, pt

    def close(self):
        if self._h:
            lib().msfm_ba_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiContext:
    """msfm_ctx_create_multi: one process, several GPUs - the library owns a context and a host thread per device and the
    communicator between them (ncclCommInitAll over xGMI; an in-process reduction when `devices` names one device several
    times, which is how the path runs on a one-GPU box).  The calls take the whole problem; the split is inside."""

    def __init__(self, n_gpus, devices=None):
        self._h = C.c_void_p()
        dev = None if devices is None else A.as_c(np.asarray(devices, dtype=np.int32), np.int32)
        rc = lib().msfm_ctx_create_multi(n_gpus, None if dev is None else A.ptr(dev, A.c_int_p), C.byref(self._h))
        if rc != 0:
            raise MsfmError(rc, "msfm_ctx_create_multi failed")
        self.n = lib().msfm_multi_size(self._h)

    def check(self, rc):
        if rc != 0:
            raise MsfmError(rc, lib().msfm_multi_last_error(self._h).decode())

    def ba_solve(self, arrays: A.BaArrays, options=None, capacity=512):
        options = options or default_options()
        buf = A.SummaryBuf(capacity)
        self.check(lib().msfm_multi_ba_solve(self._h, C.byref(arrays.struct), C.byref(options), C.byref(buf.struct)))
        return buf.result()

    def _tri(self, fn, tracks, th_error, th_angle):
        n = tracks.struct.n_tracks
        X, mse, ok = np.zeros((n, 3)), np.zeros(n), np.zeros(n, dtype=np.uint8)
        self.check(fn(self._h, C.byref(tracks.struct), th_error, th_angle, A.ptr(X, A.c_double_p), A.ptr(mse, A.c_double_p), A.ptr(ok, A.c_u8_p)))
        return X, mse, ok

    def triangulate_midpoint(self, tracks, th_error, th_angle):
        return self._tri(lib().msfm_multi_triangulate_midpoint_batch, tracks, th_error, th_angle)

    def triangulate_dlt(self, tracks, th_error, th_angle):
        return self._tri(lib().msfm_multi_triangulate_dlt_batch, tracks, th_error, th_angle)

    def reproject_mse(self, tracks, X):
        X = A.as_c(X, np.float64)
        mse = np.zeros(tracks.struct.n_tracks)
        self.check(lib().msfm_multi_reproject_mse_batch(self._h, C.byref(tracks.struct), A.ptr(X, A.c_double_p), A.ptr(mse, A.c_double_p)))
        return mse

    def match_pairs(self, descs, pairs, ratio_good=0.6, ratio_all=0.85):
        """codes per pair, n_all, n_good - what DescSet.match_pairs + fetch give, with the pair list split over the contexts."""
        descs = [A.as_c(d, np.float32) for d in descs]
        pairs = A.as_c(np.asarray(pairs, dtype=np.int32).reshape(-1, 2), np.int32)
        count = np.array([len(d) for d in descs], dtype=np.int32)
        dp = (A.c_float_p * len(descs))(*[A.ptr(d, A.c_float_p) if len(d) else None for d in descs])
        codes = [np.zeros(max(1, int(count[j])), np.int32) for _, j in pairs]
        cp = (A.c_int_p * max(1, len(pairs)))(*[A.ptr(c, A.c_int_p) for c in codes])
        na, ng = np.zeros(max(1, len(pairs)), np.int32), np.zeros(max(1, len(pairs)), np.int32)
        self.check(lib().msfm_multi_match_pairs(self._h, len(descs), dp, A.ptr(count, A.c_int_p), 128, A.ptr(pairs, A.c_int_p), len(pairs),
                                                ratio_good, ratio_all, cp, A.ptr(na, A.c_int_p), A.ptr(ng, A.c_int_p)))
        return [c[:count[j]] for c, (_, j) in zip(codes, pairs)], na[:len(pairs)], ng[:len(pairs)]

    def close(self):
        if self._h:
            lib().msfm_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def camera_graph_dissection(adjacency, tail_cols=4, force_depth=-1):
    """msfm_camera_graph_dissection (host only: works without a GPU): labels per camera (leaf id, or -(d + 1) for a separator cut
    at depth d), number of leaves (0: dense order kept) and the critical path in 64-column panel steps."""
    import numpy as np
    adj = np.ascontiguousarray(adjacency, dtype=np.uint8)
    n = adj.shape[0]
    assert adj.shape == (n, n)
    label = np.zeros(n, np.int32)
    nl, steps = C.c_int(0), C.c_int(0)
    rc = lib().msfm_camera_graph_dissection(n, adj.ctypes.data, int(tail_cols), int(force_depth), label.ctypes.data, C.byref(nl), C.byref(steps))
    if rc != 0:
        raise MsfmError(rc, "msfm_camera_graph_dissection")
    return label, nl.value, steps.value
