"""Host window selection (metricsfm_amd/window.py) against a literal object-graph walk of the reference's loops
(sfm_incremental.cc:448-506, :917-945, :1865-1903; optimizer.cc:59-129; slam_gps.cc:824) and, through the oracle,
the functor selection the masks imply."""
import numpy as np

from metricsfm_amd import _abi as A
from metricsfm_amd import scene, window


class _Pt:
    def __init__(self, pid):
        self.id, self.cams, self.mutable, self.bad = pid, {}, True, False


class _Cam:
    def __init__(self, cid, model):
        self.id, self.model, self.pts, self.visible, self.mutable = cid, model, {}, [], True


def _graph(sc, bad=None):
    cams = [_Cam(c, int(sc.cam_model_of_cam[c])) for c in range(sc.n_cams)]
    pts = [_Pt(p) for p in range(sc.n_points)]
    for o in range(sc.n_obs):
        c, p = int(sc.obs_cam[o]), int(sc.obs_pt[o])
        key = o + 1000000 * c                        # global feature id, basic_structs.h:171
        pts[p].cams[key] = cams[c]
        cams[c].pts[key] = pts[p]
    if bad is not None:
        for p in np.nonzero(bad)[0]:
            pts[p].bad = True
    return cams, pts


def _literal_visible(cams, new):
    """FindImageToLocalize's count (:486-506) + UpdateVisibleGraph (:1895-1903) for an already attached camera."""
    vis = [new.id]
    for other in cams:
        if other is new:
            continue
        count = 0
        for key, pt in other.pts.items():            # matches (new <-> other) whose feature in `other` has a 3-D point
            if not pt.bad and any(c is new for c in pt.cams.values()):
                count += 1
        if count > 5:
            vis.append(other.id)
    return vis


def _literal_partial(cams, pts, idx):
    for c in cams:                                   # ImmutableCamsPoints
        c.mutable = False
        for pt in c.pts.values():
            pt.mutable = False
    for c in cams:                                   # cam_model_->idx_cams_
        if c.model == cams[idx].model:
            c.mutable = True
            for pt in c.pts.values():
                if not pt.bad:
                    pt.mutable = True
    for v in cams[idx].visible:
        cams[v].mutable = True
        for pt in cams[v].pts.values():
            if not pt.bad:
                pt.mutable = True


def test_visible_cameras_and_partial_masks_match_the_literal_walk():
    sc = scene.make_aerial_scene(30, 1500, seed=41, n_models=30)     # one CameraModel per camera (use_same_camera = false)
    rng = np.random.default_rng(1)
    bad = rng.random(sc.n_points) < 0.05
    cams, pts = _graph(sc, bad)
    for idx in (29, 0, 13):
        vis = window.visible_cameras(sc.obs_cam, sc.obs_pt, sc.n_cams, idx, bad)
        assert list(vis) == _literal_visible(cams, cams[idx])
        cams[idx].visible = list(vis)
        _literal_partial(cams, pts, idx)
        cm, pm = window.partial_ba_masks(sc.obs_cam, sc.obs_pt, sc.n_cams, sc.n_points, sc.cam_model_of_cam, idx, vis, bad)
        assert [c.mutable for c in cams] == list(cm != 0)
        assert [p.mutable for p in pts] == list(pm != 0)
        assert 1 < cm.sum() < sc.n_cams and not pm[bad].any()
    # one shared model (UAV mode): idx_cams_ holds every camera, the "window" is everything that is not bad
    s1 = scene.make_aerial_scene(12, 400, seed=42)
    vis = window.visible_cameras(s1.obs_cam, s1.obs_pt, s1.n_cams, 11)
    cm, pm = window.partial_ba_masks(s1.obs_cam, s1.obs_pt, s1.n_cams, s1.n_points, s1.cam_model_of_cam, 11, vis)
    assert cm.all() and pm.all()


def test_gather_weights_bad_points_and_gps_weight():
    sc = scene.make_aerial_scene(16, 800, seed=43, n_models=16, gps_sigma=0.5)
    bad = np.zeros(sc.n_points, bool)
    bad[::7] = True
    arr, info = window.partial_bundle_adjustment_problem(sc, 15, bad=bad, gps=True)
    kept = info["kept"]
    assert len(arr.point) == int((~bad).sum()) and (arr.point == sc.point[kept]).all()
    k = np.bincount(sc.obs_pt, minlength=sc.n_points)[kept]
    assert ((arr.pt_weight == 2.0) == (k >= 3)).all() and ((arr.pt_weight == 1.0) == (k < 3)).all()
    assert (np.diff(arr.obs_pt) >= 0).all() and arr.obs_pt.max() == len(kept) - 1
    # slam_gps.cc:824: count1 / cams_.size() with count1 = residual blocks actually added (both-frozen rows add none)
    cm, pm = info["cam_mutable"], info["pt_mutable"][kept]
    count1 = int(((cm[arr.obs_cam] != 0) | (pm[arr.obs_pt] != 0)).sum())
    assert arr.struct.gps_weight == float(count1 // sc.n_cams) and 0 < count1 < len(arr.obs_cam)
    assert window.gps_weight(7, 2) == 3.0


def test_window_problem_residual_count_through_the_oracle(oracle):
    """The masks select the functors of optimizer.cc:86-125: every observation with a free point or a free camera is one
    residual block (2 rows), plus 3 GPS rows per free camera; the reduced system has 6 + 3 columns per window camera."""
    sc = scene.make_aerial_scene(20, 900, seed=44, n_models=20, gps_sigma=0.5, rot_sigma=0.02, trans_sigma=0.2, point_sigma=0.2)
    arr, info = window.partial_bundle_adjustment_problem(sc, 19, gps=True)
    cm, pm = info["cam_mutable"], info["pt_mutable"]
    r = oracle.ba_solve(arr, oracle.default_options(max_num_iterations=3))
    active = (cm[sc.obs_cam] != 0) | (pm[sc.obs_pt] != 0)
    assert r["num_residuals"] == 2 * int(active.sum()) + 3 * int(cm.sum())
    assert r["num_reduced_params"] == 9 * int(cm.sum())
    frozen_c, frozen_p = cm == 0, pm == 0
    assert frozen_c.any() and frozen_p.any()
    assert (arr.cam_pose[frozen_c] == sc.cam_pose[frozen_c]).all() and (arr.point[frozen_p] == sc.point[frozen_p]).all()
    assert (arr.cam_model[frozen_c] == sc.cam_model[frozen_c]).all()
    assert r["iterations"]["cost"][-1] < r["iterations"]["cost"][0]


def test_compact_gather_is_the_same_problem(oracle):
    """Handing over only the rows that make residual blocks (and only the points that have one) is the problem the
    reference builds (optimizer.cc:86-125); the full hand-over relies on the solver to drop the rest.  Same trajectory, same
    result, through the oracle."""
    sc = scene.make_aerial_scene(24, 1200, seed=45, n_models=24, gps_sigma=0.5, rot_sigma=0.02, trans_sigma=0.2, point_sigma=0.2)
    bad = np.zeros(sc.n_points, bool)
    bad[5::11] = True
    full, fi = window.partial_bundle_adjustment_problem(sc, 23, bad=bad, gps=True)
    comp, ci = window.partial_bundle_adjustment_problem(sc, 23, bad=bad, gps=True, compact=True)
    assert len(comp.obs_cam) < len(full.obs_cam) and len(comp.point) < len(full.point)
    assert comp.struct.gps_weight == full.struct.gps_weight
    assert set(ci["kept"]) <= set(fi["kept"])
    # every point the compact form leaves out is frozen and seen by frozen cameras only
    left_out = np.setdiff1d(fi["kept"], ci["kept"])
    assert not fi["pt_mutable"][left_out].any()
    opt = oracle.default_options(max_num_iterations=4)
    rf, rc = oracle.ba_solve(full, opt), oracle.ba_solve(comp, opt)
    assert rf["num_residuals"] == rc["num_residuals"] and rf["num_reduced_params"] == rc["num_reduced_params"]
    np.testing.assert_allclose(rc["iterations"]["cost"], rf["iterations"]["cost"], rtol=1e-12)
    np.testing.assert_allclose(comp.cam_pose, full.cam_pose, rtol=0, atol=1e-11)
    pos = {p: i for i, p in enumerate(fi["kept"])}
    sel = np.array([pos[p] for p in ci["kept"]])
    np.testing.assert_allclose(comp.point, full.point[sel], rtol=0, atol=1e-11)
