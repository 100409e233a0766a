"""The C++ host mirror (host/objectsfm.{h,cc}) and the test_sfm driver run end to end on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_test_sfm_driver():
    exe = os.path.join(ROOT, "host", "test_sfm")
    assert os.path.exists(exe), "host/test_sfm not built (run __graft_entry__.build())"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "test_sfm ok" in out.stdout


def test_window_driver_matches_the_python_host(tmp_path, ctx):
    """host/test_window builds the reference's object graph from a scene file and runs VisibleCameras / UpdateVisibleGraph /
    PartialBundleAdjustment (+ GPS rows) / RemovePointOutliers / SLAMGPS::FullBundleAdjustment.  The Python host
    (metricsfm_amd/window.py) makes the same selections on flat arrays; both drive the same library, so every array must
    agree bit for bit."""
    import numpy as np
    from metricsfm_amd import _abi as A, capi, scene, window
    exe = os.path.join(ROOT, "host", "test_window")
    assert os.path.exists(exe), "host/test_window not built (run __graft_entry__.build())"
    sc = scene.make_aerial_scene(60, 6000, seed=61, n_models=60, gps_sigma=0.5, rot_sigma=2e-4, trans_sigma=0.01, point_sigma=0.02)
    idx = 59
    scene.perturb_camera(sc, idx)
    bad = np.zeros(sc.n_points, np.uint8)
    bad[::23] = 1
    sc.point[5::97] += 3.0                       # a few points the outlier sweep has to catch
    f = tmp_path / "scene.bin"
    with open(f, "wb") as fh:
        np.array([sc.n_cams, len(sc.cam_model), sc.n_points, sc.n_obs, idx, 1], np.int32).tofile(fh)
        for a, dt in ((sc.cam_pose, "f8"), (sc.cam_model, "f8"), (sc.cam_model_of_cam, "i4"), (sc.point, "f8"), (sc.obs_cam, "i4"),
                      (sc.obs_pt, "i4"), (sc.obs_xy, "f8"), (bad, "u1"), (sc.gps_xyz, "f8")):
            np.ascontiguousarray(a, dtype=dt).tofile(fh)
    out = tmp_path / "result.bin"
    run = subprocess.run([exe, str(f), str(out)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "test_window ok" in run.stdout, run.stdout + run.stderr
    raw = open(out, "rb").read()
    pos = [0]

    def take(dt, n):
        a = np.frombuffer(raw, dtype=dt, count=n, offset=pos[0])
        pos[0] += a.nbytes
        return a

    Nc, Nm, Np = sc.n_cams, len(sc.cam_model), sc.n_points

    def take_state():
        return dict(cam=take("f8", 6 * Nc).reshape(Nc, 6), model=take("f8", 3 * Nm).reshape(Nm, 3), point=take("f8", 3 * Np).reshape(Np, 3),
                    it=take("i4", 2), cost=take("f8", 2))

    nv = int(take("i4", 1)[0])
    vis_c = take("i4", nv)
    cmut_c, pmut_c = take("u1", Nc), take("u1", Np)
    st1 = take_state()
    bad_after, mse_c = take("u1", Np), take("f8", Np)
    st2 = take_state()
    assert pos[0] == len(raw)
    # ---- the same three stages through the Python host ----
    arr, info = window.partial_bundle_adjustment_problem(sc, idx, bad=bad != 0, gps=True)
    np.testing.assert_array_equal(vis_c, info["visible"])
    np.testing.assert_array_equal(cmut_c, info["cam_mutable"])
    np.testing.assert_array_equal(pmut_c, info["pt_mutable"])
    assert 3 < nv < Nc and pmut_c.sum() < Np - bad.sum()
    r1 = ctx.ba_solve(arr, capi.default_options(max_num_iterations=20))
    kept = info["kept"]
    pt1 = sc.point.copy()
    pt1[kept] = arr.point
    np.testing.assert_array_equal(st1["cam"], arr.cam_pose)
    np.testing.assert_array_equal(st1["model"], arr.cam_model)
    np.testing.assert_array_equal(st1["point"], pt1)
    assert st1["it"][0] == r1["num_iterations"] and st1["cost"][1] == r1["final_cost"] and st1["cost"][0] == r1["initial_cost"]
    # RemovePointOutliers (sfm_incremental.cc:1831-1863): mse of every live point, bad iff sqrt(mse) > 3
    R, t, c, fk = scene.cameras_for_tracks(sc, st1["cam"], st1["model"])
    tr = A.TrackArrays(sc.track_offsets(), sc.obs_cam, sc.obs_xy, R, t, c, fk)
    mse = ctx.reproject_mse(tr, pt1)
    live = bad == 0
    np.testing.assert_allclose(mse_c[live], mse[live], rtol=1e-9)   # R from the C++ Rodrigues vs numpy's: last-bit differences
    np.testing.assert_array_equal(bad_after != 0, (bad != 0) | (live & (np.sqrt(mse) > 3.0)))
    assert (bad_after != 0).sum() > (bad != 0).sum()
    # SLAMGPS::FullBundleAdjustment (slam_gps.cc:675-863): everything free, the points' weights as they stand, GPS rows
    sc2 = scene.Scene(sc.name, sc.cam_pose_gt, sc.cam_model_gt, sc.point_gt, st1["cam"].copy(), st1["model"].copy(), pt1.copy(),
                      sc.cam_model_of_cam, sc.obs_cam, sc.obs_pt, sc.obs_xy, sc.pt_weight, sc.gps_xyz)
    arr2, kept2 = window.gather(sc2, weight=window.PARTIAL_WEIGHT, bad=bad_after != 0, gps=True)
    r2 = ctx.ba_solve(arr2, capi.default_options(max_num_iterations=200))
    pt2 = pt1.copy()
    pt2[kept2] = arr2.point
    np.testing.assert_array_equal(st2["cam"], arr2.cam_pose)
    np.testing.assert_array_equal(st2["model"], arr2.cam_model)
    np.testing.assert_array_equal(st2["point"], pt2)
    assert st2["it"][0] == r2["num_iterations"] and st2["cost"][1] == r2["final_cost"] < r2["initial_cost"]


def test_test_sfm_driver_with_several_contexts_in_one_process():
    """SURVEY 8b (threading row): `msfm_ctx_create_multi(n_gpus)` - the reference's entry point stays ONE process
    (test_sfm.cc:22-70), the library owns a context and a host thread per device and the communicator, and matching,
    triangulation / reprojection and the bundle adjustment split inside it.  On the one-GPU test box the contexts share the
    device (in-process reduction in rank order instead of ncclCommInitAll): `host/test_sfm --gpus N --share-device` with
    N = 2 and 3 reproduces the one-context run - matching, verification, tracks and triangulation line for line, the bundle
    adjustment to 1e-9 (its sums are formed in another order)."""
    import re
    exe = os.path.join(ROOT, "host", "test_sfm")
    assert os.path.exists(exe), "host/test_sfm not built (run __graft_entry__.build())"

    def run(n):
        args = [exe] if n == 1 else [exe, "--gpus", str(n), "--share-device"]
        out = subprocess.run(args, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "test_sfm ok" in out.stdout, out.stdout + out.stderr
        lines = out.stdout.splitlines()
        final = [float(x) for x in next(l for l in lines if l.startswith("ba_final")).split()[1:]]
        stages = [l for l in lines if re.match(r"(matching|verification|tracks|triangulation|seed pair|localisation):", l)]
        return final, stages, out.stdout

    ref, stages_ref, _ = run(1)
    for n in (2, 3):
        got, stages, text = run(n)
        assert "contexts: %d (one device)" % n in text
        assert stages == stages_ref                     # everything without a collective is bit for bit what one context gives
        assert got[5] == ref[5]                         # the same number of LM iterations
        for a, b in zip(got[:5], ref[:5]):
            assert abs(a - b) <= 1e-9 * max(1.0, abs(b)), (n, got, ref)


def test_test_sfm_driver_on_distinct_gpus():
    """`host/test_sfm --gpus N` WITHOUT --share-device: one context per GPU, ncclCommInitAll between them (skipped on a one-GPU
    box).  Same bars as the shared-device run above."""
    import re
    import torch
    n = min(torch.cuda.device_count(), 4)
    if n < 2:
        pytest.skip("needs 2 GPUs")
    exe = os.path.join(ROOT, "host", "test_sfm")

    def run(args):
        out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "test_sfm ok" in out.stdout, out.stdout + out.stderr
        lines = out.stdout.splitlines()
        final = [float(x) for x in next(l for l in lines if l.startswith("ba_final")).split()[1:]]
        stages = [l for l in lines if re.match(r"(matching|verification|tracks|triangulation|seed pair|localisation):", l)]
        return final, stages

    ref, stages_ref = run([])
    got, stages = run(["--gpus", str(n)])
    assert stages == stages_ref and got[5] == ref[5]
    for a, b in zip(got[:5], ref[:5]):
        assert abs(a - b) <= 1e-9 * max(1.0, abs(b)), (got, ref)
