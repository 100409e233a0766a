"""ctypes front end of the CPU oracle (oracle/msfm_oracle.cpp).

TEST INFRASTRUCTURE ONLY — PARITY UNPINNED (see the header of msfm_oracle.cpp).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from metricsfm_amd import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmsfm_oracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, "msfm_oracle.cpp"), os.path.join(_HERE, "pose_oracle.cpp"),
            os.path.join(_HERE, "..", "include", "msfm.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(f) for f in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_ba_solve.argtypes = [C.POINTER(A.BaProblem), C.POINTER(A.BaOptions), C.POINTER(A.BaSummary)]
        _lib.orc_ba_options_default.argtypes = [C.POINTER(A.BaOptions)]
        _lib.orc_ba_reduced_system.argtypes = [C.POINTER(A.BaProblem), C.POINTER(A.BaOptions), C.c_double,
                                               A.c_double_p, A.c_double_p, C.c_int, A.c_double_p, A.c_double_p]
        for f in (_lib.orc_triangulate_midpoint_batch, _lib.orc_triangulate_dlt_batch):
            f.argtypes = [C.POINTER(A.Tracks), C.c_double, C.c_double, A.c_double_p, A.c_double_p, A.c_u8_p]
        _lib.orc_reproject_mse_batch.argtypes = [C.POINTER(A.Tracks), A.c_double_p, A.c_double_p]
        _lib.orc_epipolar_filter.argtypes = [A.c_float_p, A.c_float_p, C.c_int, A.c_double_p, C.c_double, A.c_u8_p]
        for f in (_lib.orc_knn2_f32, _lib.orc_knn2_f32_fast):
            f.argtypes = [A.c_float_p, C.c_int, A.c_float_p, C.c_int, C.c_int, A.c_int_p, A.c_float_p]
        _lib.orc_ratio_codes.argtypes = [A.c_int_p, A.c_float_p, C.c_int, C.c_float, C.c_float, A.c_int_p,
                                         A.c_int_p, A.c_int_p]
        for f in (_lib.orc_reproj_dual, _lib.orc_reproj_analytic):
            f.argtypes = [A.c_double_p] * 4 + [C.c_double, A.c_double_p, A.c_double_p]
            f.restype = None
        _lib.orc_huber.argtypes = [C.c_double, C.c_double, A.c_double_p]
        _lib.orc_huber.restype = None
        for f in (_lib.orc_angle_axis_to_R, _lib.orc_R_to_angle_axis):
            f.argtypes = [A.c_double_p, A.c_double_p]
            f.restype = None
        _lib.orc_angle_axis_rotate_point.argtypes = [A.c_double_p] * 3
        _lib.orc_angle_axis_rotate_point.restype = None
        _lib.orc_set_num_threads.argtypes = [C.c_int]
        _lib.orc_set_num_threads.restype = None
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def set_num_threads(n):
    """Host threads of the kNN timing leg (the BA takes `num_threads` from its options).  Results do not depend on it."""
    lib().orc_set_num_threads(int(n))


def host_cores():
    """Cores this process may run on (the GPU box gives a 1-GPU job a share of the host, not all of it)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for f in ("/sys/fs/cgroup/cpu.max",):          # cgroup v2 quota: "<quota> <period>" or "max <period>"
        try:
            q, p = open(f).read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(q) // int(p)))
        except Exception:
            pass
    try:                                           # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // p))
    except Exception:
        pass
    if os.environ.get("MSFM_BENCH_CORES"):
        n = max(1, int(os.environ["MSFM_BENCH_CORES"]))
    return n


def default_options(**kw):
    o = A.BaOptions()
    lib().orc_ba_options_default(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def ba_solve(arrays: A.BaArrays, options=None, capacity=512):
    """Runs the LM on `arrays` IN PLACE (like Ceres on the data blocks); returns the summary dict."""
    options = options or default_options()
    buf = A.SummaryBuf(capacity)
    rc = lib().orc_ba_solve(C.byref(arrays.struct), C.byref(options), C.byref(buf.struct))
    if rc != 0:
        raise RuntimeError("orc_ba_solve failed: %d" % rc)
    return buf.result()


def ba_reduced_system(arrays: A.BaArrays, radius=1e4, options=None):
    options = options or default_options()
    n = 6 * arrays.struct.n_cams + 3 * arrays.struct.n_models
    S = np.zeros((n, n))
    rhs = np.zeros(n)
    cost, gmax = C.c_double(), C.c_double()
    m = lib().orc_ba_reduced_system(C.byref(arrays.struct), C.byref(options), radius, A.ptr(S, A.c_double_p),
                                    A.ptr(rhs, A.c_double_p), n, C.byref(cost), C.byref(gmax))
    if m < 0:
        raise RuntimeError("capacity")
    S = S.reshape(-1)[: m * m].reshape(m, m)
    return S, rhs[:m], cost.value, gmax.value


def reproj(pose, cam, xyz, obs, weight=1.0, dual=False):
    pose, cam, xyz, obs = (np.ascontiguousarray(v, dtype=np.float64) for v in (pose, cam, xyz, obs))
    r, J = np.zeros(2), np.zeros((2, 12))
    f = lib().orc_reproj_dual if dual else lib().orc_reproj_analytic
    f(A.ptr(pose, A.c_double_p), A.ptr(cam, A.c_double_p), A.ptr(xyz, A.c_double_p), A.ptr(obs, A.c_double_p),
      weight, A.ptr(r, A.c_double_p), A.ptr(J, A.c_double_p))
    return r, J


def huber(a, s):
    rho = np.zeros(3)
    lib().orc_huber(a, s, A.ptr(rho, A.c_double_p))
    return rho


def angle_axis_to_R(aa):
    aa = np.ascontiguousarray(aa, dtype=np.float64)
    R = np.zeros(9)
    lib().orc_angle_axis_to_R(A.ptr(aa, A.c_double_p), A.ptr(R, A.c_double_p))
    return R.reshape(3, 3)


def R_to_angle_axis(R):
    R = np.ascontiguousarray(R, dtype=np.float64).reshape(9)
    aa = np.zeros(3)
    lib().orc_R_to_angle_axis(A.ptr(R, A.c_double_p), A.ptr(aa, A.c_double_p))
    return aa


def rotate_point(aa, pt):
    aa, pt = np.ascontiguousarray(aa, dtype=np.float64), np.ascontiguousarray(pt, dtype=np.float64)
    out = np.zeros(3)
    lib().orc_angle_axis_rotate_point(A.ptr(aa, A.c_double_p), A.ptr(pt, A.c_double_p), A.ptr(out, A.c_double_p))
    return out


def _tri(fn, tracks: A.TrackArrays, th_error, th_angle, X0=None):
    n = tracks.struct.n_tracks
    X = np.zeros((n, 3)) if X0 is None else np.array(X0, dtype=np.float64, order="C")
    mse, ok = np.zeros(n), np.zeros(n, dtype=np.uint8)
    fn(C.byref(tracks.struct), th_error, th_angle, A.ptr(X, A.c_double_p), A.ptr(mse, A.c_double_p), A.ptr(ok, A.c_u8_p))
    return X, mse, ok


def triangulate_midpoint(tracks, th_error, th_angle, X0=None):
    return _tri(lib().orc_triangulate_midpoint_batch, tracks, th_error, th_angle, X0)


def triangulate_dlt(tracks, th_error, th_angle, X0=None):
    return _tri(lib().orc_triangulate_dlt_batch, tracks, th_error, th_angle, X0)


def reproject_mse(tracks, X):
    X = np.ascontiguousarray(X, dtype=np.float64)
    mse = np.zeros(tracks.struct.n_tracks)
    lib().orc_reproject_mse_batch(C.byref(tracks.struct), A.ptr(X, A.c_double_p), A.ptr(mse, A.c_double_p))
    return mse


def epipolar_filter(pt1, pt2, F, th=3.0):
    pt1, pt2 = np.ascontiguousarray(pt1, dtype=np.float32), np.ascontiguousarray(pt2, dtype=np.float32)
    F = np.ascontiguousarray(F, dtype=np.float64).reshape(9)
    out = np.zeros(len(pt1), dtype=np.uint8)
    lib().orc_epipolar_filter(A.ptr(pt1, A.c_float_p), A.ptr(pt2, A.c_float_p), len(pt1), A.ptr(F, A.c_double_p), th,
                              A.ptr(out, A.c_u8_p))
    return out


def fundamental_ransac(offsets, pt1, pt2, threshold=3.0, confidence=0.99, max_iterations=2000, min_points=30, min_inliers=30,
                       seed=0x4D53464D46):
    """Sequential restatement of cv::findFundamentalMat(FM_RANSAC, 3.0) + the 30-inlier gate (geo_verification.cc:30-58)."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    pt1 = np.ascontiguousarray(np.asarray(pt1, dtype=np.float32).reshape(-1, 2))
    pt2 = np.ascontiguousarray(np.asarray(pt2, dtype=np.float32).reshape(-1, 2))
    n = len(offsets) - 1
    F = np.zeros((n, 3, 3), dtype=np.float64)
    inl = np.zeros(max(1, len(pt1)), dtype=np.uint8)
    nin = np.zeros(max(1, n), dtype=np.int32)
    ok = np.zeros(max(1, n), dtype=np.uint8)
    f = lib().orc_fundamental_ransac
    f.argtypes = [C.c_int, A.c_int_p, A.c_float_p, A.c_float_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_uint64,
                  A.c_double_p, A.c_u8_p, A.c_int_p, A.c_u8_p]
    rc = f(n, A.ptr(offsets, A.c_int_p), A.ptr(pt1, A.c_float_p), A.ptr(pt2, A.c_float_p), threshold, confidence, max_iterations,
           min_points, min_inliers, seed, A.ptr(F, A.c_double_p), A.ptr(inl, A.c_u8_p), A.ptr(nin, A.c_int_p), A.ptr(ok, A.c_u8_p))
    if rc != 0:
        raise ValueError("orc_fundamental_ransac rc=%d" % rc)
    return F, inl[:len(pt1)], nin[:n], ok[:n]


def build_tracks(pairs, matches_per_pair, idx_max_per_image=1000000):
    """SLAMGPS::Triangulation's data association, slam_gps.cc:565-635, restated literally: one map from the global
    feature id (local + image * idx_max_per_image) to a point, points hold {image: feature} maps filled with
    insert-if-absent (std::map::insert).  Pure Python: for small cases only.
    Returns CSR tracks (track_off, obs_image, obs_feature), observations in ascending image (= map key) order."""
    pts_points_map = {}
    pts = []   # each: dict image -> feature (Point3D::cams_ / pts2d_ are keyed by the image id here, slam_gps.cc:600-603)
    for (id_img1, id_img2), matches in zip(pairs, matches_per_pair):
        for id_pt1_local, id_pt2_local in matches:
            g1 = int(id_pt1_local) + int(id_img1) * idx_max_per_image
            g2 = int(id_pt2_local) + int(id_img2) * idx_max_per_image
            if g1 in pts_points_map:
                id_pt = pts_points_map[g1]
                pts[id_pt].setdefault(int(id_img2), int(id_pt2_local))
                pts_points_map.setdefault(g2, id_pt)
            elif g2 in pts_points_map:
                id_pt = pts_points_map[g2]
                pts[id_pt].setdefault(int(id_img1), int(id_pt1_local))
                pts_points_map.setdefault(g1, id_pt)
            else:
                pt = {}
                pt.setdefault(int(id_img1), int(id_pt1_local))
                pt.setdefault(int(id_img2), int(id_pt2_local))
                pts.append(pt)
                pts_points_map.setdefault(g1, len(pts) - 1)
                pts_points_map.setdefault(g2, len(pts) - 1)
    off, img, feat = [0], [], []
    for pt in pts:
        for k in sorted(pt):
            img.append(k)
            feat.append(pt[k])
        off.append(len(img))
    return np.array(off, np.int32), np.array(img, np.int32), np.array(feat, np.int32)


def generate_new_points(cam1, visible_cams, matches_per_cam, done1, done2_per_cam, keypoints, cam_R, cam_t, cam_c, cam_fk,
                        th_mse_reprojection=3.0, th_angle_small=3.0 / 180.0 * 3.1415, th_angle_large=5.0 / 180.0 * 3.1415):
    """IncrementalSfM::GenerateNew3DPoints, sfm_incremental.cc:755-915, restated literally: one Trianglate2 per candidate."""
    out = []
    for k, cam2 in enumerate(visible_cams):
        if cam2 == cam1:
            continue
        matches = np.asarray(matches_per_cam[k]).reshape(-1, 2)
        th = th_angle_large if len(matches) > 500 else th_angle_small
        for f1, f2 in matches:
            if done1[f1] or done2_per_cam[k][f2]:
                continue
            xy = np.array([keypoints[cam1][f1], keypoints[cam2][f2]], dtype=np.float64)
            tr = A.TrackArrays(np.array([0, 2], np.int32), np.array([cam1, cam2], np.int32), xy, cam_R, cam_t, cam_c, cam_fk)
            X, mse, ok = triangulate_midpoint(tr, th_mse_reprojection, th)
            if ok[0]:
                out.append((int(mse[0]), X[0].copy(), float(mse[0]), int(cam2), int(f1), int(f2)))   # pair<Point3DNew*, int>(.., mse_)
    out.sort(key=lambda r: r[0])   # stable
    if not out:
        z = np.zeros(0, dtype=np.int64)
        return np.zeros((0, 3)), np.zeros(0), z, z, z
    return (np.array([r[1] for r in out]), np.array([r[2] for r in out]), np.array([r[3] for r in out]), np.array([r[4] for r in out]),
            np.array([r[5] for r in out]))


def knn2(train, query, fast=False):
    train, query = np.ascontiguousarray(train, dtype=np.float32), np.ascontiguousarray(query, dtype=np.float32)
    ids = np.zeros((len(query), 2), dtype=np.int32)
    d = np.zeros((len(query), 2), dtype=np.float32)
    f = lib().orc_knn2_f32_fast if fast else lib().orc_knn2_f32
    rc = f(A.ptr(train, A.c_float_p), len(train), A.ptr(query, A.c_float_p), len(query), train.shape[1],
           A.ptr(ids, A.c_int_p), A.ptr(d, A.c_float_p))
    if rc != 0:
        raise ValueError("orc_knn2 rc=%d" % rc)
    return ids, d


def ratio_codes(ids, sqd, ratio_good=0.6, ratio_all=0.85):
    ids, sqd = np.ascontiguousarray(ids, dtype=np.int32), np.ascontiguousarray(sqd, dtype=np.float32)
    code = np.zeros(len(ids), dtype=np.int32)
    na, ng = C.c_int32(), C.c_int32()
    lib().orc_ratio_codes(A.ptr(ids, A.c_int_p), A.ptr(sqd, A.c_float_p), len(ids), ratio_good, ratio_all,
                          A.ptr(code, A.c_int_p), C.byref(na), C.byref(ng))
    return code, na.value, ng.value


def slam_gate(ids, sqd, kp1, kp2, F, H, th_first_second_ratio=0.80, th_epipolar=2.0, th_distance=5.0):
    """The three checks of SLAMGPS::FeatureMatching step 2 (slam_gps.cc:466-503) on one pair.  Returns code [n_query]
    (train index or -1), the survivors of the ratio check and the number of matches kept."""
    ids, sqd = np.ascontiguousarray(ids, dtype=np.int32), np.ascontiguousarray(sqd, dtype=np.float32)
    kp1, kp2 = np.ascontiguousarray(kp1, dtype=np.float32), np.ascontiguousarray(kp2, dtype=np.float32)
    F, H = np.ascontiguousarray(F, dtype=np.float64).reshape(9), np.ascontiguousarray(H, dtype=np.float64).reshape(9)
    code = np.zeros(len(ids), dtype=np.int32)
    nr, nk = C.c_int32(), C.c_int32()
    f = lib().orc_slam_gate
    f.restype = None
    f.argtypes = [A.c_int_p, A.c_float_p, C.c_int, A.c_float_p, A.c_float_p, A.c_double_p, A.c_double_p, C.c_float, C.c_float, C.c_float,
                  A.c_int_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    f(A.ptr(ids, A.c_int_p), A.ptr(sqd, A.c_float_p), len(ids), A.ptr(kp1, A.c_float_p), A.ptr(kp2, A.c_float_p), A.ptr(F, A.c_double_p),
      A.ptr(H, A.c_double_p), th_first_second_ratio, th_epipolar, th_distance, A.ptr(code, A.c_int_p), C.byref(nr), C.byref(nk))
    return code, nr.value, nk.value


def epnp_ransac(offsets, pts_w, pts_2d, f, max_iter=200, seed=0x4D53464D50):
    """AbsolutePoseEstimation::AbsolutePoseWithFocalLength (absolute_pose_estimation.cc:42-58) per image of a batch:
    EPNPRansac (absolute_pose_via_epnp.cc:103-139) + Error (:67-103).  Returns R [n,3,3], t [n,3], errors [total],
    avg_error [n], best_iter [n]."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    pts_w = np.ascontiguousarray(np.asarray(pts_w, dtype=np.float64).reshape(-1, 3))
    pts_2d = np.ascontiguousarray(np.asarray(pts_2d, dtype=np.float64).reshape(-1, 2))
    n = len(offsets) - 1
    f = np.ascontiguousarray(np.broadcast_to(np.asarray(f, dtype=np.float64), (n,)))
    R = np.zeros((max(1, n), 3, 3)); t = np.zeros((max(1, n), 3)); err = np.zeros(max(1, len(pts_w))); avg = np.zeros(max(1, n))
    best = np.zeros(max(1, n), dtype=np.int32)
    fn = lib().orc_epnp_ransac
    fn.argtypes = [C.c_int, A.c_int_p, A.c_double_p, A.c_double_p, A.c_double_p, C.c_int, C.c_uint64, A.c_double_p, A.c_double_p,
                   A.c_double_p, A.c_double_p, A.c_int_p]
    rc = fn(n, A.ptr(offsets, A.c_int_p), A.ptr(pts_w, A.c_double_p), A.ptr(pts_2d, A.c_double_p), A.ptr(f, A.c_double_p), max_iter, seed,
            A.ptr(R, A.c_double_p), A.ptr(t, A.c_double_p), A.ptr(err, A.c_double_p), A.ptr(avg, A.c_double_p), A.ptr(best, A.c_int_p))
    if rc != 0:
        raise ValueError("orc_epnp_ransac rc=%d" % rc)
    return R[:n], t[:n], err[:len(pts_w)], avg[:n], best[:n]


def relpose_5pt(offsets, pts_ref, pts_cur, f_ref, f_cur, ransac_times=100, seed=0x4D53464D45):
    """RelativePoseEstimation::RelativePoseWithFocalLength (relative_pose_estimation.cc:91-120) per image pair of a batch.
    Returns E [n,3,3] (x_cur^T E x_ref = 0 on pixel / f), R [n,3,3], t [n,3], ok [n], n_candidates [n]."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    pts_ref = np.ascontiguousarray(np.asarray(pts_ref, dtype=np.float64).reshape(-1, 2))
    pts_cur = np.ascontiguousarray(np.asarray(pts_cur, dtype=np.float64).reshape(-1, 2))
    n = len(offsets) - 1
    f_ref = np.ascontiguousarray(np.broadcast_to(np.asarray(f_ref, dtype=np.float64), (n,)))
    f_cur = np.ascontiguousarray(np.broadcast_to(np.asarray(f_cur, dtype=np.float64), (n,)))
    E = np.zeros((max(1, n), 3, 3)); R = np.zeros((max(1, n), 3, 3)); t = np.zeros((max(1, n), 3))
    ok = np.zeros(max(1, n), dtype=np.uint8); nc = np.zeros(max(1, n), dtype=np.int32)
    fn = lib().orc_relpose_5pt
    fn.argtypes = [C.c_int, A.c_int_p, A.c_double_p, A.c_double_p, A.c_double_p, A.c_double_p, C.c_int, C.c_uint64, A.c_double_p,
                   A.c_double_p, A.c_double_p, A.c_u8_p, A.c_int_p]
    rc = fn(n, A.ptr(offsets, A.c_int_p), A.ptr(pts_ref, A.c_double_p), A.ptr(pts_cur, A.c_double_p), A.ptr(f_ref, A.c_double_p),
            A.ptr(f_cur, A.c_double_p), ransac_times, seed, A.ptr(E, A.c_double_p), A.ptr(R, A.c_double_p), A.ptr(t, A.c_double_p),
            A.ptr(ok, A.c_u8_p), A.ptr(nc, A.c_int_p))
    if rc != 0:
        raise ValueError("orc_relpose_5pt rc=%d" % rc)
    return E[:n], R[:n], t[:n], ok[:n], nc[:n]


def _test_jacobi_svd(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    m, n = a.shape
    W = np.zeros(n); Ut = np.zeros((n, m)); Vt = np.zeros((n, n))
    fn = lib().orc_test_jacobi_svd
    fn.argtypes = [A.c_double_p, C.c_int, C.c_int, A.c_double_p, A.c_double_p, A.c_double_p]
    fn.restype = None
    fn(A.ptr(a, A.c_double_p), m, n, A.ptr(W, A.c_double_p), A.ptr(Ut, A.c_double_p), A.ptr(Vt, A.c_double_p))
    return W, Ut, Vt


def _test_eig10(a):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(10, 10)
    wr = np.zeros(10); wi = np.zeros(10)
    fn = lib().orc_test_eig10
    fn.argtypes = [A.c_double_p, A.c_double_p, A.c_double_p]
    ok = fn(A.ptr(a, A.c_double_p), A.ptr(wr, A.c_double_p), A.ptr(wi, A.c_double_p))
    return bool(ok), wr, wi


def _test_five_point(x1, x2):
    x1 = np.ascontiguousarray(x1, dtype=np.float64).reshape(-1, 2)
    x2 = np.ascontiguousarray(x2, dtype=np.float64).reshape(-1, 2)
    Es = np.zeros((10, 9))
    fn = lib().orc_test_five_point
    fn.argtypes = [A.c_double_p, A.c_double_p, C.c_int, A.c_double_p]
    c = fn(A.ptr(x1, A.c_double_p), A.ptr(x2, A.c_double_p), len(x1), A.ptr(Es, A.c_double_p))
    return Es[:c].reshape(c, 3, 3).transpose(0, 2, 1)   # column-major entries -> E[r, c]


def _test_epnp4(Xw, x2d, f):
    Xw = np.ascontiguousarray(Xw, dtype=np.float64).reshape(4, 3)
    x2d = np.ascontiguousarray(x2d, dtype=np.float64).reshape(4, 2)
    R = np.zeros((3, 3)); t = np.zeros(3); e = C.c_double(0)
    fn = lib().orc_test_epnp4
    fn.argtypes = [A.c_double_p, A.c_double_p, C.c_double, A.c_double_p, A.c_double_p, C.POINTER(C.c_double)]
    fn.restype = None
    fn(A.ptr(Xw, A.c_double_p), A.ptr(x2d, A.c_double_p), f, A.ptr(R, A.c_double_p), A.ptr(t, A.c_double_p), C.byref(e))
    return R, t, e.value


def _test_epnp_n(Xw, x2d, f):
    """EPnP on all given correspondences (n >= 4); returns R, t, mean reprojection error."""
    Xw = np.ascontiguousarray(Xw, dtype=np.float64).reshape(-1, 3)
    x2d = np.ascontiguousarray(x2d, dtype=np.float64).reshape(-1, 2)
    R = np.zeros((3, 3)); t = np.zeros(3)
    fn = lib().orc_test_epnp_n
    fn.argtypes = [A.c_double_p, A.c_double_p, C.c_int, C.c_double, A.c_double_p, A.c_double_p]
    fn.restype = C.c_double
    e = fn(A.ptr(Xw, A.c_double_p), A.ptr(x2d, A.c_double_p), len(Xw), f, A.ptr(R, A.c_double_p), A.ptr(t, A.c_double_p))
    return R, t, e
