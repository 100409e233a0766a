"""The device-resident chain (msfm_chain: match codes -> verification -> tracks -> triangulation -> bundle adjustment)
against the same steps through the host-array entry points: every intermediate result identical."""
import numpy as np
import pytest

from metricsfm_amd import _abi as A
from metricsfm_amd import capi, matchfiles, scene

pytestmark = pytest.mark.gpu


def _host_path(ctx, sc, res, pairs, kps, n_img, seed):
    """fine_matching_graph.cc:116-186 + slam_gps.cc:557-648 + optimizer.cc:59-133 through host arrays."""
    good_l, all_l = [], []
    for p in range(len(pairs)):
        code, _, _ = res.fetch(p)
        g, a = matchfiles.codes_to_matches(code)
        good_l.append(np.asarray(g, np.int32).reshape(-1, 2)); all_l.append(np.asarray(a, np.int32).reshape(-1, 2))
    off_g = np.concatenate([[0], np.cumsum([len(g) for g in good_l])]).astype(np.int32)
    off_a = np.concatenate([[0], np.cumsum([len(a) for a in all_l])]).astype(np.int32)
    g1 = np.concatenate([kps[i][g[:, 0]] for (i, j), g in zip(pairs, good_l)]); g2 = np.concatenate([kps[j][g[:, 1]] for (i, j), g in zip(pairs, good_l)])
    a1 = np.concatenate([kps[i][a[:, 0]] for (i, j), a in zip(pairs, all_l)]); a2 = np.concatenate([kps[j][a[:, 1]] for (i, j), a in zip(pairs, all_l)])
    F, _, _, ok = ctx.fundamental_ransac(off_g, g1, g2, seed=seed)
    in_a = ctx.epipolar_filter_batch(off_a, a1, a2, F, ok, 3.0)
    fin = [all_l[p][in_a[off_a[p]:off_a[p + 1]] != 0] if ok[p] else np.zeros((0, 2), np.int32) for p in range(len(pairs))]
    n_feat = np.array([len(k) for k in kps], np.int32)
    moff = np.concatenate([[0], np.cumsum([len(m) for m in fin])]).astype(np.int32)
    flat = (n_feat, A.as_c(pairs, np.int32), moff, A.as_c(np.concatenate(fin).reshape(-1, 2), np.int32))
    off, img, feat = ctx.build_tracks(None, None, None, flat=flat)
    return F, ok, fin, (off, img, feat)


def test_config2_chain_matches_the_host_array_path(ctx):
    """BASELINE config 2 end to end: 50 images x 4096 features uploaded once (descriptors + keypoints), every ordered pair
    matched, verified, associated, triangulated and bundle-adjusted on the device; one download of the parameters at the end.
    The host-array path runs the same steps on the fetched results of each stage."""
    sc = scene.add_features(scene.config_scene(2), 4096)
    n_img = sc.n_cams
    kps = [np.ascontiguousarray(k, np.float32) for k in sc.kp_xy]
    ds = ctx.descset(sc.desc, keypoints=kps)
    pairs = scene.all_pairs(n_img)
    res = ds.match_pairs(pairs, 0.6, 0.85)
    seed = 0x4D53464D
    ch = capi.Chain(res)
    n_m, ok, F = ch.verify(3.0, seed=seed)
    nt, no = ch.build_tracks()
    R, t, c, fk = scene.cameras_for_tracks(sc)          # triangulate with the true cameras (the model is "already oriented")
    n_acc = ch.triangulate(R, t, c, fk, 7.0, np.deg2rad(3.0))
    assert ok.sum() > 300 and nt > 10000 and n_acc > 0.8 * nt

    F_h, ok_h, fin_h, (off_h, img_h, feat_h) = _host_path(ctx, sc, res, pairs, kps, n_img, seed)
    np.testing.assert_array_equal(ok, ok_h)
    np.testing.assert_array_equal(F, F_h)
    np.testing.assert_array_equal(n_m, [len(m) for m in fin_h])
    for p in np.nonzero(n_m)[0][::37]:
        np.testing.assert_array_equal(ch.fetch_matches(int(p)), fin_h[p])
    off, img, feat = ch.fetch_tracks()
    np.testing.assert_array_equal(off, off_h); np.testing.assert_array_equal(img, img_h); np.testing.assert_array_equal(feat, feat_h)
    # triangulation through the host-array entry point on the same tracks
    xy = np.array([kps[i][f] for i, f in zip(img_h, feat_h)], dtype=np.float64)
    tr = A.TrackArrays(off_h, img_h, xy, R, t, c, fk)
    X_h, mse_h, tok_h = ctx.triangulate_midpoint(tr, 7.0, np.deg2rad(3.0))
    X, mse, tok = ch.fetch_points()
    np.testing.assert_array_equal(tok, tok_h); np.testing.assert_array_equal(X, X_h); np.testing.assert_array_equal(mse, mse_h)
    # the tracks are the scene's points: a track's features belong to one 3-D point
    pid = np.array([sc.feat_point[i][f] for i, f in zip(img_h, feat_h)])
    same = np.array([len(set(pid[off_h[k]:off_h[k + 1]])) == 1 for k in range(0, nt, 50)])
    assert same.mean() > 0.95

    # bundle adjustment: created on the device from the accepted tracks with >= 3 views
    opts = capi.default_options(max_num_iterations=8)
    ba = ch.ba_create(sc.cam_pose, sc.cam_model, sc.cam_model_of_cam, min_views=3, weight_ge3=1.0)
    r = ba.run(opts)
    cam_d, model_d, pt_d = ba.download()
    keep = (tok_h != 0) & (np.diff(off_h) >= 3)
    kept = np.nonzero(keep)[0]
    np.testing.assert_array_equal(ba.track_of_point, kept)
    obs_sel = np.repeat(keep, np.diff(off_h))
    new_pt = np.cumsum(keep) - 1
    arrays = A.BaArrays(sc.cam_pose.copy(), sc.cam_model.copy(), sc.cam_model_of_cam, X_h[keep].copy(), img_h[obs_sel],
                        np.repeat(new_pt, np.diff(off_h))[obs_sel].astype(np.int32), xy[obs_sel], np.ones(len(kept)))
    r_h = ctx.ba_solve(arrays, opts)
    assert r["num_iterations"] == r_h["num_iterations"] and r["num_residuals"] == r_h["num_residuals"] == 2 * int(obs_sel.sum())
    np.testing.assert_array_equal(r["iterations"]["cost"], r_h["iterations"]["cost"])
    np.testing.assert_array_equal(cam_d, arrays.cam_pose); np.testing.assert_array_equal(pt_d, arrays.point); np.testing.assert_array_equal(model_d, arrays.cam_model)
    assert r["final_cost"] < r["initial_cost"]
    ba.close(); ch.close(); res.close(); ds.close()


def test_chain_refuses_what_it_cannot_do(ctx):
    rng = np.random.default_rng(2)
    descs = [np.rint(rng.uniform(0, 255, (80, 128))).astype(np.float32) for _ in range(3)]
    ds = ctx.descset(descs)                                    # no keypoints
    res = ds.match_pairs(np.array([[0, 1], [1, 2]], np.int32))
    with pytest.raises(capi.MsfmError) as e:
        capi.Chain(res)
    assert e.value.code == A.MSFM_E_INVAL
    for i in range(3):
        ds.upload_keypoints(i, rng.uniform(-100, 100, (80, 2)))
    ch = capi.Chain(res)
    with pytest.raises(capi.MsfmError):
        ch.build_tracks()                                      # before verify
    n_m, ok, _ = ch.verify()
    assert ok.sum() == 0 and n_m.sum() == 0                    # random descriptors: nothing to verify (< 30 matches)
    assert ch.build_tracks() == (0, 0)
    ch.close(); res.close(); ds.close()


def test_chain_owns_its_keypoints(ctx):
    """The chain copies the keypoints when it is created: positions uploaded again afterwards (or a descriptor set destroyed
    before the triangulation) do not reach it - same matches, tracks and points as the undisturbed chain."""
    sc = scene.add_features(scene.config_scene(1), 1500)
    kps = [np.ascontiguousarray(k, np.float32) for k in sc.kp_xy]
    pairs = scene.all_pairs(sc.n_cams)
    R, t, c, fk = scene.cameras_for_tracks(sc)

    def run(disturb):
        ds = ctx.descset(sc.desc, keypoints=kps)
        res = ds.match_pairs(pairs, 0.6, 0.85)
        ch = capi.Chain(res)
        if disturb:
            for i in range(sc.n_cams):
                ds.upload_keypoints(i, np.full_like(kps[i], 1.0e6))     # the old blocks go back to the pool ...
            junk = [ctx.descset([np.zeros((len(k), 128), np.float32)], keypoints=[np.full_like(k, -7.0)]) for k in kps[:4]]   # ... and are reused
        n_m, ok, F = ch.verify(3.0, seed=5)
        nt, no = ch.build_tracks()
        if disturb:
            res.close(); ds.close()                                      # the set is gone before the triangulation
            for j in junk:
                j.close()
        n_acc = ch.triangulate(R, t, c, fk, 7.0, np.deg2rad(3.0))
        out = (n_m.copy(), ok.copy(), F.copy(), nt, no, n_acc) + tuple(x.copy() for x in ch.fetch_points())
        ch.close()
        if not disturb:
            res.close(); ds.close()
        return out

    a, b = run(False), run(True)
    assert a[3] > 500 and a[5] > 0.8 * a[3]
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
