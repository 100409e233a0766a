"""Parity of the batched triangulation / reprojection kernels with the CPU oracle."""
import numpy as np
import pytest

from metricsfm_amd import _abi as A
from metricsfm_amd import scene

pytestmark = pytest.mark.gpu


def _tracks(sc, pose=None, keep=None):
    R, t, c, fk = scene.cameras_for_tracks(sc, pose)
    off = sc.track_offsets()
    return A.TrackArrays(off, sc.obs_cam, sc.obs_xy, R, t, c, fk)


def test_midpoint_dlt_reproject(ctx, oracle):
    sc = scene.make_aerial_scene(30, 5000, seed=77)
    tr = _tracks(sc)
    th_err, th_ang = 7.0, np.deg2rad(3.0)  # th_mse_reprojection, sfm_incremental.cc:780-784
    for name in ("triangulate_midpoint", "triangulate_dlt"):
        Xr, mr, okr = getattr(oracle, name)(tr, th_err, th_ang)
        Xg, mg, okg = getattr(ctx, name)(tr, th_err, th_ang)
        np.testing.assert_array_equal(okg, okr)
        np.testing.assert_allclose(Xg, Xr, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(mg, mr, rtol=1e-7, atol=1e-9)
        assert okr.mean() > 0.9
        assert np.abs(Xr - sc.point_gt)[okr > 0].max() < 1.0  # recovers the scene
    mse_r = oracle.reproject_mse(tr, sc.point_gt)
    mse_g = ctx.reproject_mse(tr, sc.point_gt)
    np.testing.assert_allclose(mse_g, mse_r, rtol=1e-12)


def test_edge_cases(ctx, oracle):
    sc = scene.make_ring_scene(5, 40, seed=3, noise_px=0.0)
    R, t, c, fk = scene.cameras_for_tracks(sc)
    # noise-free tracks of 2, 3, 5 views; a single-view track; an empty track; a point behind a camera
    off, cams, xy = [0], [], []
    o = sc.track_offsets()
    for p, k in [(0, 2), (1, 3), (2, 5), (3, 1), (4, 0)]:
        cams += list(sc.obs_cam[o[p]:o[p] + k]); xy += list(sc.obs_xy[o[p]:o[p] + k]); off.append(len(cams))
    # mirrored observation -> negative depth -> mse = 1e5 (structure.cc:280-284)
    cams += list(sc.obs_cam[o[5]:o[5] + 2]); xy += [-v for v in sc.obs_xy[o[5]:o[5] + 2]]; off.append(len(cams))
    tr = A.TrackArrays(np.array(off), np.array(cams), np.array(xy).reshape(-1, 2), R, t, c, fk)
    X0 = np.full((len(off) - 1, 3), 7.0)
    for name in ("triangulate_midpoint", "triangulate_dlt"):
        Xr, mr, okr = getattr(oracle, name)(tr, 3.0, np.deg2rad(3.0), X0)
        Xg, mg, okg = getattr(ctx, name)(tr, 3.0, np.deg2rad(3.0), X0)
        np.testing.assert_array_equal(okg, okr)
        np.testing.assert_allclose(Xg, Xr, rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(mg, mr, rtol=1e-7, atol=1e-12)
        assert okr[:3].all() and not okr[3] and not okr[4]
        np.testing.assert_allclose(Xr[:3], sc.point_gt[:3], atol=1e-7)
        assert (Xg[4] == 7.0).all()  # untouched (LLT failure / < 2 views)
    # midpoint on a single ray fails its LLT and leaves X alone (structure.cc:248-251)
    Xg, _, _ = ctx.triangulate_midpoint(tr, 3.0, 0.05, X0)
    assert (Xg[3] == 7.0).all()


def test_epipolar_filter(ctx, oracle):
    rng = np.random.default_rng(4)
    F = rng.standard_normal((3, 3)); F[2, 2] = 1.0
    p1 = (rng.standard_normal((5000, 2)) * 800).astype(np.float32)
    p2 = (rng.standard_normal((5000, 2)) * 800).astype(np.float32)
    np.testing.assert_array_equal(ctx.epipolar_filter(p1, p2, F, 3.0), oracle.epipolar_filter(p1, p2, F, 3.0))
    assert ctx.epipolar_filter(p1[:0], p2[:0], F).shape == (0,)


def test_tracks_to_triangulated_points(ctx, oracle):
    """Match graph -> tracks (slam_gps.cc:565-635) -> one batched Trianglate2 call with the 3-view rule (:637-648)."""
    from metricsfm_amd import capi, tracks
    sc = scene.make_aerial_scene(14, 600, seed=9)
    n_feat, keyp, pairs, matches = tracks.matches_from_scene(sc)
    off, img, feat = capi.build_tracks(n_feat, pairs, matches)
    o_off, o_img, o_feat = oracle.build_tracks(pairs, matches)
    np.testing.assert_array_equal(off, o_off); np.testing.assert_array_equal(img, o_img); np.testing.assert_array_equal(feat, o_feat)
    assert len(off) - 1 == sc.n_points                      # consistent matches: one track per scene point
    R, t, c, fk = scene.cameras_for_tracks(sc)
    X, mse, bad = tracks.triangulate_tracks(ctx, off, img, feat, keyp, R, t, c, fk, th_outlier=7.0)
    xy = np.concatenate([keyp[i][f][None] for i, f in zip(img, feat)])
    Xo, mo, oko = oracle.triangulate_midpoint(A.TrackArrays(off, img, xy, R, t, c, fk), 7.0, np.deg2rad(3.0))
    np.testing.assert_array_equal(bad, (oko == 0) | (np.diff(off) < 3))
    np.testing.assert_allclose(X, Xo, rtol=1e-9, atol=1e-9)
    # every track is one scene point: its first observation names it
    first_obs = {}
    for o in range(sc.n_obs):
        first_obs.setdefault((int(sc.obs_cam[o]), tuple(sc.obs_xy[o])), int(sc.obs_pt[o]))
    pid = np.array([first_obs[(int(img[off[k]]), tuple(keyp[img[off[k]]][feat[off[k]]]))] for k in range(len(off) - 1)])
    good = ~bad
    assert good.mean() > 0.5 and np.abs(X[good] - sc.point_gt[pid[good]]).max() < 1.0


def test_generate_new_points_matches_literal_loop(ctx, oracle):
    """IncrementalSfM::GenerateNew3DPoints (sfm_incremental.cc:755-915): two batched calls against one Trianglate2 per candidate."""
    from metricsfm_amd import tracks
    rng = np.random.default_rng(4)
    sc = scene.make_aerial_scene(16, 6000, seed=12)
    n_feat, keyp, pairs, matches = tracks.matches_from_scene(sc)
    cam1 = 15
    vis = [j for (i, j) in pairs if i == cam1] + [cam1]          # visible_cams_ contains the camera itself (:1895-1903)
    mlist = [matches[pairs.index((cam1, j))] if j != cam1 else np.zeros((0, 2), np.int32) for j in vis]
    # a few wrong matches, and some features already triangulated
    for m in mlist:
        if len(m):
            bad = rng.random(len(m)) < 0.05
            m[bad, 1] = rng.integers(0, 50, bad.sum())
    done1 = rng.random(n_feat[cam1]) < 0.3
    done2 = [rng.random(n_feat[j]) < 0.3 for j in vis]
    assert any(len(m) > 500 for m in mlist) and any(0 < len(m) <= 500 for m in mlist)   # both angle thresholds are exercised
    R, t, c, fk = scene.cameras_for_tracks(sc)
    got = tracks.generate_new_points(ctx, cam1, vis, mlist, done1, done2, keyp, R, t, c, fk, th_mse_reprojection=7.0)
    want = oracle.generate_new_points(cam1, vis, mlist, done1, done2, keyp, R, t, c, fk, th_mse_reprojection=7.0)
    assert len(got[0]) == len(want[0]) > 100
    for g, w in zip(got[2:], want[2:]):
        np.testing.assert_array_equal(g, w)                      # same candidates, same order
    np.testing.assert_allclose(got[0], want[0], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(got[1], want[1], rtol=1e-7, atol=1e-9)
    assert (np.diff(np.trunc(got[1])) >= 0).all()
