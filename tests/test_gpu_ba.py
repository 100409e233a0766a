"""Parity of the HIP bundle adjustment (through the C ABI) with the CPU oracle."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from metricsfm_amd import _abi as A
from metricsfm_amd import scene

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


def check_parity(ctx, O, make_arrays, opts_kw=None, tol_param=1e-7, tol_cost=1e-9):
    opts_kw = opts_kw or {}
    a_ref, a_gpu = make_arrays(), make_arrays()
    from metricsfm_amd import capi
    r_ref = O.ba_solve(a_ref, O.default_options(**opts_kw))
    r_gpu = ctx.ba_solve(a_gpu, capi.default_options(**opts_kw))
    assert r_gpu["num_residuals"] == r_ref["num_residuals"]
    assert r_gpu["num_reduced_params"] == r_ref["num_reduced_params"]
    assert r_gpu["termination"] == r_ref["termination"], (r_gpu["termination"], r_ref["termination"])
    assert r_gpu["num_iterations"] == r_ref["num_iterations"]
    ig, ir = r_gpu["iterations"], r_ref["iterations"]
    # same accept / reject sequence, same trajectory
    np.testing.assert_array_equal(ig["step_is_successful"], ir["step_is_successful"])
    np.testing.assert_array_equal(ig["step_is_valid"], ir["step_is_valid"])
    np.testing.assert_allclose(ig["cost"], ir["cost"], rtol=tol_cost)
    np.testing.assert_allclose(ig["trust_region_radius"], ir["trust_region_radius"], rtol=1e-6)
    np.testing.assert_allclose(ig["gradient_max_norm"], ir["gradient_max_norm"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(ig["step_norm"], ir["step_norm"], rtol=1e-5, atol=1e-12)
    assert abs(r_gpu["final_cost"] - r_ref["final_cost"]) <= tol_cost * abs(r_ref["final_cost"])
    # north_star: poses and points within 1e-5 relative; we hold a tighter bar
    assert _rel(a_gpu.cam_pose, a_ref.cam_pose) < tol_param
    assert _rel(a_gpu.cam_model, a_ref.cam_model) < tol_param
    assert _rel(a_gpu.point, a_ref.point) < tol_param
    return r_gpu, r_ref, a_gpu


def test_ba_config1_full(ctx, oracle):
    sc = scene.config_scene(1)
    r, _, a = check_parity(ctx, oracle, lambda: A.BaArrays.from_scene(sc), dict(max_num_iterations=50))
    assert r["termination"].startswith("CONVERGENCE")
    uv, _ = scene.project(a.cam_pose[sc.obs_cam], a.cam_model[sc.cam_model_of_cam[sc.obs_cam]], a.point[sc.obs_pt])
    assert np.linalg.norm(uv - sc.obs_xy, axis=1).mean() < 1.0  # converged to the noise floor


def test_ba_config2_full(ctx, oracle):
    sc = scene.config_scene(2)
    check_parity(ctx, oracle, lambda: A.BaArrays.from_scene(sc), dict(max_num_iterations=50))


def test_ba_iteration_cap_and_rejections(ctx, oracle):
    # a tiny trust region forces short early steps and hits the iteration cap; a large one makes the
    # reduced system (7 gauge freedoms, nothing fixed: sfm_incremental.cc:1016-1026) ill-conditioned,
    # cond ~ radius, so the comparison tolerance scales with it
    sc = scene.make_ring_scene(6, 300, seed=11)
    check_parity(ctx, oracle, lambda: A.BaArrays.from_scene(sc),
                 dict(max_num_iterations=12, initial_trust_region_radius=1e-2))
    check_parity(ctx, oracle, lambda: A.BaArrays.from_scene(sc),
                 dict(max_num_iterations=12, initial_trust_region_radius=1e7), tol_param=1e-5, tol_cost=1e-6)


def test_ba_window_masks(ctx, oracle):
    """PartialBundleAdjustment-style windows (sfm_incremental.cc:917-1014): frozen cameras give
    ReprojectionErrorXYZ rows, frozen points give ReprojectionErrorPoseCam rows, both frozen
    give no residual (optimizer.cc:86-125)."""
    sc = scene.make_aerial_scene(24, 3000, seed=5, rot_sigma=0.02, trans_sigma=0.2, point_sigma=0.2)
    rng = np.random.default_rng(3)
    cam_mut = (np.arange(sc.n_cams) % 3 != 0).astype(np.uint8)
    pt_mut = (rng.random(sc.n_points) > 0.2).astype(np.uint8)
    w = np.where(np.diff(sc.track_offsets()) >= 3, 2.0, 1.0)  # optimizer.cc:69-78 with weight = 2
    mk = lambda: A.BaArrays(sc.cam_pose, sc.cam_model, sc.cam_model_of_cam, sc.point, sc.obs_cam, sc.obs_pt,
                            sc.obs_xy, w, cam_mutable=cam_mut, pt_mutable=pt_mut)
    check_parity(ctx, oracle, mk, dict(max_num_iterations=25))
    # intrinsics frozen too -> ReprojectionErrorPoseXYZ / ReprojectionErrorPose
    mk2 = lambda: A.BaArrays(sc.cam_pose, sc.cam_model, sc.cam_model_of_cam, sc.point, sc.obs_cam, sc.obs_pt,
                             sc.obs_xy, w, cam_mutable=cam_mut, pt_mutable=pt_mut,
                             model_mutable=np.zeros(1, np.uint8))
    check_parity(ctx, oracle, mk2, dict(max_num_iterations=25))


def test_ba_multiple_models_and_no_points(ctx, oracle):
    sc = scene.make_aerial_scene(16, 2000, seed=9, n_models=3, rot_sigma=0.02, trans_sigma=0.2, point_sigma=0.2)
    check_parity(ctx, oracle, lambda: A.BaArrays.from_scene(sc), dict(max_num_iterations=20))
    # all points frozen: no Schur elimination at all, only camera blocks
    mk = lambda: A.BaArrays.from_scene(sc, pt_mutable=np.zeros(sc.n_points, np.uint8))
    check_parity(ctx, oracle, mk, dict(max_num_iterations=20))


def test_ba_gps(ctx, oracle):
    """Absolute GPS residuals on pose[3:6] (gps_error_pose_absolute.h:31-44, slam_gps.cc:818-830)."""
    sc = scene.make_aerial_scene(20, 2500, seed=21, gps_sigma=0.5, rot_sigma=0.02, trans_sigma=0.3, point_sigma=0.2)
    wgt = float(sc.n_obs // sc.n_cams)  # integer division, slam_gps.cc:824
    mk = lambda: A.BaArrays.from_scene(sc, gps_xyz=sc.gps_xyz, gps_weight=wgt)
    check_parity(ctx, oracle, mk, dict(max_num_iterations=30))


def test_ba_rejects_bad_input(ctx):
    from metricsfm_amd import capi
    sc = scene.make_ring_scene(4, 50, seed=2)
    a = A.BaArrays.from_scene(sc)
    a.obs_pt[:] = a.obs_pt[::-1].copy()  # decreasing
    with pytest.raises(capi.MsfmError) as e:
        ctx.ba_solve(a)
    assert e.value.code == A.MSFM_E_INVAL
    a = A.BaArrays.from_scene(sc)
    a.obs_cam[3] = 99
    with pytest.raises(capi.MsfmError):
        ctx.ba_solve(a)


def test_ba_degenerate_structures(ctx, oracle):
    """Structures the gather can produce: a camera nobody observes, a track that sees the same camera
    twice (two features of one image in one track), a single free camera, and nothing free at all."""
    from metricsfm_amd import capi
    sc = scene.make_ring_scene(6, 200, seed=17, rot_sigma=0.02, trans_sigma=0.2, point_sigma=0.2)
    # (a) camera 5 loses all its observations -> it is not a parameter block (Ceres drops unused blocks)
    keep = sc.obs_cam != 5
    mk = lambda: A.BaArrays(sc.cam_pose, sc.cam_model, sc.cam_model_of_cam, sc.point, sc.obs_cam[keep], sc.obs_pt[keep],
                            sc.obs_xy[keep], sc.pt_weight)
    r, _, a = check_parity(ctx, oracle, mk, dict(max_num_iterations=15))
    assert r["num_reduced_params"] == 5 * 6 + 3 and (a.cam_pose[5] == sc.cam_pose[5]).all()
    # (b) duplicate camera inside a track: observation 1 of every 10th point is re-pointed at the camera of observation 0
    oc = sc.obs_cam.copy()
    off = sc.track_offsets()
    oc[off[:-1][::10] + 1] = oc[off[:-1][::10]]
    mk = lambda: A.BaArrays(sc.cam_pose, sc.cam_model, sc.cam_model_of_cam, sc.point, oc, sc.obs_pt, sc.obs_xy, sc.pt_weight)
    check_parity(ctx, oracle, mk, dict(max_num_iterations=10), tol_param=1e-6, tol_cost=1e-8)
    # (c) only camera 2 and the points are free
    cm = np.zeros(6, np.uint8); cm[2] = 1
    mk = lambda: A.BaArrays.from_scene(sc, cam_mutable=cm)
    check_parity(ctx, oracle, mk, dict(max_num_iterations=15))
    # (d) nothing is free: no residual blocks at all -> zero cost, gradient tolerance fires at iteration 0
    mk = lambda: A.BaArrays.from_scene(sc, cam_mutable=np.zeros(6, np.uint8), pt_mutable=np.zeros(sc.n_points, np.uint8))
    a_ref, a_gpu = mk(), mk()
    r_ref = oracle.ba_solve(a_ref, oracle.default_options())
    r_gpu = ctx.ba_solve(a_gpu, capi.default_options())
    assert r_gpu["num_residuals"] == r_ref["num_residuals"] == 0
    assert r_gpu["termination"] == r_ref["termination"] and r_gpu["num_iterations"] == r_ref["num_iterations"] == 0
    assert (a_gpu.point == sc.point).all() and (a_gpu.cam_pose == sc.cam_pose).all()


def _panel_launches(ctx, arrays, opts):
    ctx.profile(True)
    ctx.profile_reset()
    r = ctx.ba_solve(arrays, opts)
    st = ctx.profile_get()
    ctx.profile(False)
    return r, st["chol_panel_mfma"]["launches"]


@pytest.mark.parametrize("depth", ["1", "2"])
def test_ba_domain_parallel_factorisation(ctx, oracle, monkeypatch, depth):
    """>= 128 cameras: the camera graph is bisected (depth 1 / 2) and the domains' panel chains share launches.
    Same answers as the dense elimination order and as the oracle, with fewer panel launches."""
    from metricsfm_amd import capi
    sc = scene.make_aerial_scene(168, 6000, seed=5, gps_sigma=0.5)
    kw = dict(gps_xyz=sc.gps_xyz, gps_weight=float(sc.n_obs // sc.n_cams))
    okw = dict(max_num_iterations=12)
    monkeypatch.setenv("MSFM_CHOL_DOMAINS", "0")
    a_dense = A.BaArrays.from_scene(sc, **kw)
    r_dense, n_dense = _panel_launches(ctx, a_dense, capi.default_options(**okw))
    monkeypatch.setenv("MSFM_CHOL_DOMAINS", depth)
    a_dom = A.BaArrays.from_scene(sc, **kw)
    lay = ctx.ba(A.BaArrays.from_scene(sc, **kw)).layout()
    assert lay["n_domains"] == 2 ** int(depth) and all(c % 64 == 0 for c in lay["domain_cols"])
    assert sum(lay["domain_cols"]) + lay["separator_cols"] == lay["system_order"] >= lay["reduced_order"] == 6 * 168 + 3
    r_dom, n_dom = _panel_launches(ctx, a_dom, capi.default_options(**okw))
    assert n_dom < n_dense, (n_dom, n_dense)           # the chains really ran side by side
    assert r_dom["num_iterations"] == r_dense["num_iterations"]
    np.testing.assert_array_equal(r_dom["iterations"]["step_is_successful"], r_dense["iterations"]["step_is_successful"])
    np.testing.assert_allclose(r_dom["iterations"]["cost"], r_dense["iterations"]["cost"], rtol=1e-10)
    for name in ("cam_pose", "cam_model", "point"):
        assert _rel(getattr(a_dom, name), getattr(a_dense, name)) < 1e-8, name
    # and against the CPU oracle (dense Schur complement, camera order as given)
    check_parity(ctx, oracle, lambda: A.BaArrays.from_scene(sc, **kw), okw)


def test_ba_disconnected_camera_graph(ctx, oracle, monkeypatch):
    """Two surveys that share nothing but the camera model: the camera graph has two components, the separator of the
    bisection is empty and only the intrinsics block (+ rhs) is left for the single chain."""
    from metricsfm_amd import capi
    s1 = scene.make_aerial_scene(70, 2500, seed=31)
    s2 = scene.make_aerial_scene(66, 2200, seed=32)

    def arrays():
        return A.BaArrays(np.concatenate([s1.cam_pose, s2.cam_pose]), s1.cam_model.copy(),
                          np.concatenate([s1.cam_model_of_cam, s2.cam_model_of_cam]), np.concatenate([s1.point, s2.point]),
                          np.concatenate([s1.obs_cam, s2.obs_cam + s1.n_cams]), np.concatenate([s1.obs_pt, s2.obs_pt + s1.n_points]),
                          np.concatenate([s1.obs_xy, s2.obs_xy]), np.concatenate([s1.pt_weight, s2.pt_weight]))

    monkeypatch.setenv("MSFM_CHOL_DOMAINS", "1")
    lay = ctx.ba(arrays()).layout()
    assert lay["n_domains"] == 2 and lay["separator_cols"] == 3          # nothing but f, k1, k2 couples the two surveys
    assert sorted(lay["domain_cols"]) == [64 * -(-6 * 66 // 64), 64 * -(-6 * 70 // 64)]
    check_parity(ctx, oracle, arrays, dict(max_num_iterations=10))


def test_ba_full_size_properties(ctx, monkeypatch):
    """BASELINE config 3 (500 cameras / 200k points / 1.2M observations) is too large for the CPU oracle inside a test:
    size-independent properties instead - monotone cost over accepted steps, the noise floor reached, and the two
    elimination orders (camera domains vs dense) giving the same trajectory."""
    from metricsfm_amd import capi
    sc = scene.config_scene(3)
    opts = dict(max_num_iterations=25)
    runs = {}
    for order in ("0", "2"):
        monkeypatch.setenv("MSFM_CHOL_DOMAINS", order)
        a = A.BaArrays.from_scene(sc)
        runs[order] = (ctx.ba_solve(a, capi.default_options(**opts)), a)
    r, a = runs["2"]
    it = r["iterations"]
    cost = it["cost"][it["step_is_successful"] > 0]
    assert (np.diff(cost) <= 0).all() and cost[-1] < 1e-3 * cost[0]
    assert r["termination"].startswith("CONVERGENCE")
    uv, _ = scene.project(a.cam_pose[sc.obs_cam], a.cam_model[sc.cam_model_of_cam[sc.obs_cam]], a.point[sc.obs_pt])
    assert np.linalg.norm(uv - sc.obs_xy, axis=1).mean() < 1.0          # observations carry 0.5 px noise
    assert 2 * r["final_cost"] / (2 * sc.n_obs) < 1.0                   # mean squared residual at the noise floor
    r0, a0 = runs["0"]
    assert r0["num_iterations"] == r["num_iterations"]
    np.testing.assert_allclose(r0["iterations"]["cost"], it["cost"], rtol=1e-9)
    assert _rel(a0.cam_pose, a.cam_pose) < 1e-7 and _rel(a0.point, a.point) < 1e-7


def test_ba_two_runs_are_bitwise_identical(ctx):
    """Determinism: every reduction has a fixed shape, so two solves of the headline configuration - fold
    tables on, pair kernels on the second stream - give bitwise the same trajectory and parameters."""
    from metricsfm_amd import capi
    sc = scene.config_scene(3)
    runs = []
    for _ in range(2):
        a = A.BaArrays.from_scene(sc)
        r = ctx.ba_solve(a, capi.default_options(max_num_iterations=6))
        runs.append((r, a))
    (r0, a0), (r1, a1) = runs
    np.testing.assert_array_equal(r0["iterations"]["cost"], r1["iterations"]["cost"])
    np.testing.assert_array_equal(r0["iterations"]["gradient_max_norm"], r1["iterations"]["gradient_max_norm"])
    np.testing.assert_array_equal(a0.cam_pose, a1.cam_pose)
    np.testing.assert_array_equal(a0.cam_model, a1.cam_model)
    np.testing.assert_array_equal(a0.point, a1.point)


def test_ba_config3_full_size_trajectory_parity(ctx, oracle):
    """The headline configuration itself (BASELINE config 3: 500 cameras / 200 000 points / 1.2 M observations) against the
    OpenMP oracle on every core of the box - bit-identical to its 1-thread run (tests/test_oracle.py) - for 12 LM iterations:
    same accept / reject sequence, costs within 1e-9, parameters within 1e-7 (north_star asks 1e-5)."""
    sc = scene.config_scene(3)
    # stopping rules off (negative tolerances never fire): the run would otherwise converge after 9 iterations
    r, r_ref, _ = check_parity(ctx, oracle, lambda: A.BaArrays.from_scene(sc),
                               dict(max_num_iterations=12, num_threads=oracle.host_cores(), function_tolerance=-1.0, gradient_tolerance=-1.0,
                                    parameter_tolerance=-1.0))
    assert r["num_iterations"] == 12 and r["num_successful_steps"] >= 8
    assert r["num_reduced_params"] == 3003 and r["num_residuals"] == 2 * sc.n_obs


def test_ba_one_model_per_camera_beyond_the_corner_table(ctx):
    """use_same_camera = false with thousands of cameras: 3 intrinsics columns per camera put more than 128 blocks behind
    the leaf level, which the deferred corner update cannot address - the plan must fall back (dense order) when it is
    chosen, not fail in the middle of the first factorisation."""
    from metricsfm_amd import capi
    sc = scene.make_aerial_scene(2750, 30000, seed=77, n_models=2750, rot_sigma=1e-3, trans_sigma=0.02, point_sigma=0.05)
    ba = ctx.ba(A.BaArrays.from_scene(sc))
    lay = ba.layout()
    assert lay["reduced_order"] == 9 * 2750 and lay["n_domains"] <= 1 and lay["n_levels"] == 0
    r = ba.run(capi.default_options(max_num_iterations=2))
    ba.close()
    assert r["num_iterations"] == 2 and r["num_successful_steps"] >= 1 and r["final_cost"] < r["initial_cost"]
    # the same cameras sharing one CameraModel do get an elimination tree
    sc1 = scene.make_aerial_scene(2750, 30000, seed=77)
    ba = ctx.ba(A.BaArrays.from_scene(sc1))
    lay1 = ba.layout()
    ba.close()
    assert lay1["n_domains"] >= 2 and lay1["n_levels"] >= 1


def test_ba_folded_schur_products_match_the_oracle_and_the_gather_path(oracle):
    """The camera x camera Schur products formed inside k_point (FoldTables, ba.hip) instead of by the gather kernel: forced on
    for small problems (MSFM_FOLD_MIN=0) - long tracks that need the second-round record area, frozen cameras / points, several
    intrinsics blocks, GPS rows - each against the oracle, and against the gather path (MSFM_NO_FOLD=1) to rounding."""
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from metricsfm_amd import _abi as A, capi, scene\n"
            "from oracle import oracle as O\n"
            "ctx = capi.Context(0)\n"
            "out = []\n"
            "def run(sc, **kw):\n"
            "    a, r = A.BaArrays.from_scene(sc, **kw), A.BaArrays.from_scene(sc, **kw)\n"
            "    g = ctx.ba_solve(a, capi.default_options(max_num_iterations=10)); o = O.ba_solve(r, O.default_options(max_num_iterations=10))\n"
            "    assert g['num_iterations'] == o['num_iterations'] and (g['iterations']['step_is_successful'] == o['iterations']['step_is_successful']).all()\n"
            "    np.testing.assert_allclose(g['iterations']['cost'], o['iterations']['cost'], rtol=1e-9)\n"
            "    assert np.abs(a.cam_pose - r.cam_pose).max() <= 1e-7 * np.abs(r.cam_pose).max() and np.abs(a.point - r.point).max() <= 1e-7 * np.abs(r.point).max()\n"
            "    out.append([float(c) for c in g['iterations']['cost']])\n"
            "run(scene.make_aerial_scene(60, 5000, seed=71))\n"
            "sc = scene.make_ring_scene(14, 900, seed=72)           # every point seen by all 14 cameras: second-round records\n"
            "run(sc)\n"
            "run(scene.make_ring_scene(20, 700, seed=74))           # 20 rows per point: no workgroup folds, everything stays on the gather lists\n"
            "sc = scene.make_aerial_scene(90, 9000, seed=75)        # mixed: a few long tracks among short ones -> some workgroups fold, some do not\n"
            "run(sc)\n"
            "sc = scene.make_aerial_scene(40, 4000, seed=73, n_models=5, gps_sigma=0.5)\n"
            "rng = np.random.default_rng(3); cm = (np.arange(40) %% 6 != 0).astype(np.uint8); pm = (rng.random(4000) > 0.15).astype(np.uint8)\n"
            "run(sc, cam_mutable=cm, pt_mutable=pm, gps_xyz=sc.gps_xyz, gps_weight=50.0)\n"
            "print(repr(out))\n") % ROOT
    outs = {}
    for name, env in (("fold", dict(MSFM_FOLD_MIN="0")), ("gather", dict(MSFM_NO_FOLD="1")), ("fold_separate_launches", dict(MSFM_FOLD_MIN="0", MSFM_FUSED_SUMS="0")),
                      ("fold_all_T_stored", dict(MSFM_FOLD_MIN="0", MSFM_KEEP_T="1", MSFM_TU_DIRECT="0"))):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[name] = eval(r.stdout.strip().splitlines()[-1])
    for a, b in zip(outs["fold"], outs["gather"]):
        np.testing.assert_allclose(a, b, rtol=1e-8)   # (the two paths sum in different orders: equal only to rounding; both
                                                      #  already matched the oracle to 1e-9 above, rejected trial steps included)
    # round 4: per-camera sums, pair-list residue and zero fill as ONE launch (k_sums) against the separate launches on two streams:
    # the same arithmetic per chunk, so every cost of every trajectory is the same number
    assert outs["fold"] == outs["fold_separate_launches"]
    # round 4: a folding workgroup does not store its T records (their only readers are pair-list entries that did not fold, and a
    # workgroup folds all of its entries or none) and sends T.u out without the lane exchange: storing everything as before
    # must give the same numbers - a reader of an unstored record would show here
    assert outs["fold"] == outs["fold_all_T_stored"]


def test_ba_domains_with_window_masks(ctx, oracle, monkeypatch):
    """Frozen cameras and frozen points (PartialBundleAdjustment masks, sfm_incremental.cc:917-1014) on a problem large enough
    for the camera-domain order: frozen cameras have no block, observations of frozen points only touch camera diagonals."""
    monkeypatch.setenv("MSFM_CHOL_DOMAINS", "2")
    sc = scene.make_aerial_scene(150, 4000, seed=17)
    cam_mut = np.ones(sc.n_cams, np.uint8); cam_mut[::13] = 0
    pt_mut = (np.arange(sc.n_points) % 11 != 0).astype(np.uint8)

    def arrays():
        return A.BaArrays.from_scene(sc, cam_mutable=cam_mut, pt_mutable=pt_mut)

    lay = ctx.ba(arrays()).layout()
    assert lay["n_domains"] == 4 and lay["reduced_order"] == 6 * int(cam_mut.sum()) + 3
    # the trajectory has rejected trial steps with costs 300x the accepted ones: their cost is compared a little looser
    _, _, a = check_parity(ctx, oracle, arrays, dict(max_num_iterations=10), tol_cost=1e-8)
    np.testing.assert_array_equal(a.cam_pose[cam_mut == 0], sc.cam_pose[cam_mut == 0])     # frozen blocks untouched
    np.testing.assert_array_equal(a.point[pt_mut == 0], sc.point[pt_mut == 0])


def test_host_thread_count_does_not_change_the_result():
    """msfm_ba_create builds its index structures with a few host threads (counting sorts that keep the input order):
    1 thread and 8 threads must give the same structure, hence bitwise the same solve."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from metricsfm_amd import _abi as A, capi, scene\n"
            "sc = scene.make_aerial_scene(140, 6000, seed=11)\n"
            "a = A.BaArrays.from_scene(sc)\n"
            "r = capi.Context(0).ba_solve(a, capi.default_options(max_num_iterations=8))\n"
            "print(repr(r['final_cost']), r['num_iterations'], repr(float(a.cam_pose.sum())), repr(float(a.point.sum())))\n") % ROOT
    outs = []
    for n in ("1", "8"):
        env = dict(os.environ, MSFM_HOST_THREADS=n, MSFM_CHOL_DOMAINS="1")
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300))
        assert outs[-1].returncode == 0, outs[-1].stderr[-2000:]
    assert outs[0].stdout == outs[1].stdout and outs[0].stdout.strip()


def test_ba_partial_window_with_gps(ctx, oracle):
    """IncrementalSfM::PartialBundleAdjustment(idx) (sfm_incremental.cc:917-1014) as the host selects it
    (metricsfm_amd/window.py: visible_cams_ > 5 shared matches, weight 2.0, bad points skipped) with the SLAMGPS rows
    (slam_gps.cc:818-830) on the window's cameras: same trajectory as the oracle, frozen blocks bit-untouched."""
    from metricsfm_amd import window
    sc = scene.make_aerial_scene(40, 4000, seed=51, n_models=40, gps_sigma=0.5, rot_sigma=0.02, trans_sigma=0.3, point_sigma=0.2)
    bad = np.zeros(sc.n_points, bool)
    bad[::17] = True
    for idx in (39, 7):
        _, info = window.partial_bundle_adjustment_problem(sc, idx, bad=bad, gps=True)
        cm, pm, kept = info["cam_mutable"], info["pt_mutable"], info["kept"]
        assert 2 < cm.sum() < sc.n_cams and 0 < pm.sum() < sc.n_points
        mk = lambda: window.partial_bundle_adjustment_problem(sc, idx, bad=bad, gps=True)[0]
        r, _, a = check_parity(ctx, oracle, mk, dict(max_num_iterations=20))
        assert r["num_reduced_params"] == 9 * int(cm.sum())
        np.testing.assert_array_equal(a.cam_pose[cm == 0], sc.cam_pose[cm == 0])
        np.testing.assert_array_equal(a.cam_model[cm == 0], sc.cam_model[cm == 0])
        np.testing.assert_array_equal(a.point[pm[kept] == 0], sc.point[kept][pm[kept] == 0])
        assert np.abs(a.cam_pose[cm != 0] - sc.cam_pose[cm != 0]).max() > 0
    # one shared CameraModel (UAV mode): idx_cams_ frees every camera, the window is the whole (non-bad) model
    s1 = scene.make_aerial_scene(24, 2500, seed=52, gps_sigma=0.5, rot_sigma=0.02, trans_sigma=0.3, point_sigma=0.2)
    mk = lambda: window.partial_bundle_adjustment_problem(s1, 23, gps=True)[0]
    r, _, _ = check_parity(ctx, oracle, mk, dict(max_num_iterations=20))
    assert r["num_reduced_params"] == 6 * 24 + 3


_C5 = {}


def _config5_window_scene():
    """BASELINE config 5 at full size, the newest camera straight from EPnP (built once per session: 6 M observations)."""
    if "sc" not in _C5:
        sc = scene.config_scene(5, n_models=2000, rot_sigma=2e-4, trans_sigma=0.01, point_sigma=0.02)
        scene.perturb_camera(sc, sc.n_cams - 1)
        _C5["sc"] = sc
    return _C5["sc"]


def test_ba_config5_window_full_size_oracle_parity(ctx, oracle):
    """BASELINE config 5 AT ITS OWN SIZE against the oracle (round 5; the test below holds the size-independent properties):
    the partial bundle adjustment of the newest of 2000 aerial cameras over 1 M points / 6 M observations with GPS rows
    (sfm_incremental.cc:917-1014 + slam_gps.cc:818-830).  The compact hand-over is exactly the residual blocks the reference's
    loop adds (optimizer.cc:86-125) - 23 free cameras, ~20 k free points - which the CPU oracle solves in seconds: same accept /
    reject sequence and termination within 20 LM iterations, costs to 1e-9, parameters to 1e-7.  The FULL hand-over (all 6 M rows; the library drops
    the frozen x frozen ones itself) must then give the compact one's trajectory on the GPU."""
    from metricsfm_amd import capi, window
    sc = _config5_window_scene()
    assert (sc.n_cams, sc.n_points, sc.n_obs) == (2000, 1000000, 6000000)
    idx = sc.n_cams - 1
    made = {}

    def make():
        arr, info = window.partial_bundle_adjustment_problem(sc, idx, gps=True, compact=True)
        made["info"] = info
        return arr
    opts = dict(max_num_iterations=20, num_threads=oracle.host_cores())
    r, r_ref, comp = check_parity(ctx, oracle, make, opts)
    ci = made["info"]
    n_win = int(ci["cam_mutable"].sum())
    assert 5 < n_win < 200 and r["num_reduced_params"] == 9 * n_win and r["num_iterations"] >= 5 and r["num_successful_steps"] >= 5
    assert len(comp.obs_cam) < sc.n_obs // 10 and len(ci["kept"]) < sc.n_points // 10
    # the full hand-over on the GPU: same residual blocks, same trajectory (point blocks numbered differently: rounding only)
    full, fi = window.partial_bundle_adjustment_problem(sc, idx, gps=True)
    assert len(full.obs_cam) == sc.n_obs
    rf = ctx.ba_solve(full, capi.default_options(**{k: v for k, v in opts.items() if k != "num_threads"}))
    assert rf["num_residuals"] == r["num_residuals"] and rf["num_reduced_params"] == r["num_reduced_params"]
    np.testing.assert_array_equal(rf["iterations"]["step_is_successful"], r["iterations"]["step_is_successful"])
    np.testing.assert_allclose(rf["iterations"]["cost"], r["iterations"]["cost"], rtol=1e-10)
    np.testing.assert_allclose(full.cam_pose, comp.cam_pose, rtol=0, atol=1e-8)
    np.testing.assert_allclose(full.cam_model, comp.cam_model, rtol=0, atol=1e-7)
    np.testing.assert_allclose(full.point[ci["kept"]], comp.point, rtol=0, atol=1e-8)


def test_ba_config5_window_full_size(ctx):
    """BASELINE config 5 at full size (2000 aerial cameras / 1M points / 6M observations, GPS rows, one CameraModel per
    camera as with use_same_camera = false): the partial bundle adjustment of the newest camera.  Too large for the CPU
    oracle, so size-independent properties: the window is what the host selected, weight 2.0 on >= 3-view points, cost
    monotone over accepted steps, every frozen block bit-untouched, and the free part converges to the noise floor."""
    from metricsfm_amd import capi, window
    # the model is in its adjusted state (pixel-level residual error left), the newest camera comes straight from EPnP
    sc = _config5_window_scene()
    assert (sc.n_cams, sc.n_points, sc.n_obs) == (2000, 1000000, 6000000)
    idx = sc.n_cams - 1
    arr, info = window.partial_bundle_adjustment_problem(sc, idx, gps=True)
    cm, pm, vis = info["cam_mutable"], info["pt_mutable"], info["visible"]
    assert vis[0] == idx and 5 < len(vis) < 200 and int(cm.sum()) == len(vis)
    k = np.bincount(sc.obs_pt, minlength=sc.n_points)
    assert ((arr.pt_weight == 2.0) == (k >= 3)).all()
    active = (cm[sc.obs_cam] != 0) | (pm[sc.obs_pt] != 0)
    assert arr.struct.gps_weight == float(int(active.sum()) // sc.n_cams)
    r = ctx.ba_solve(arr, capi.default_options(max_num_iterations=30))
    assert r["num_reduced_params"] == 9 * len(vis)
    assert r["num_residuals"] == 2 * int(active.sum()) + 3 * len(vis)
    it = r["iterations"]
    cost = it["cost"][it["step_is_successful"] > 0]
    assert len(cost) >= 3 and (np.diff(cost) <= 0).all() and cost[-1] < 0.2 * cost[0]
    assert r["termination"].startswith("CONVERGENCE")
    np.testing.assert_array_equal(arr.cam_pose[cm == 0], sc.cam_pose[cm == 0])
    np.testing.assert_array_equal(arr.cam_model[cm == 0], sc.cam_model[cm == 0])
    np.testing.assert_array_equal(arr.point[pm == 0], sc.point[pm == 0])
    assert (arr.cam_pose[cm != 0] != sc.cam_pose[cm != 0]).any(axis=1).all()
    # rows whose camera AND point were free end at the 0.5 px noise floor; the new camera started ~160 px off
    both = (cm[sc.obs_cam] != 0) & (pm[sc.obs_pt] != 0)
    uv, _ = scene.project(arr.cam_pose[sc.obs_cam[both]], arr.cam_model[sc.cam_model_of_cam[sc.obs_cam[both]]], arr.point[sc.obs_pt[both]])
    uv0, _ = scene.project(sc.cam_pose[sc.obs_cam[both]], sc.cam_model[sc.cam_model_of_cam[sc.obs_cam[both]]], sc.point[sc.obs_pt[both]])
    e1, e0 = np.linalg.norm(uv - sc.obs_xy[both], axis=1), np.linalg.norm(uv0 - sc.obs_xy[both], axis=1)
    assert np.median(e1) < 0.7 < np.median(e0)
    new = sc.obs_cam == idx
    uvn, _ = scene.project(arr.cam_pose[sc.obs_cam[new]], arr.cam_model[sc.cam_model_of_cam[sc.obs_cam[new]]], arr.point[sc.obs_pt[new]])
    assert np.median(np.linalg.norm(uvn - sc.obs_xy[new], axis=1)) < 0.7


def test_device_built_structures_match_the_host_build():
    """msfm_ba_create builds its index structures on the device (scans, stable radix sorts, small kernels); MSFM_CREATE_HOST=1
    keeps the first, host-threaded implementation.  Both must produce the same structures, hence bitwise the same solve:
    a domain-ordered problem, window masks + several intrinsics blocks + GPS, and a tiny degenerate one."""
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from metricsfm_amd import _abi as A, capi, scene, window\n"
            "ctx = capi.Context(0)\n"
            "out = []\n"
            "sc = scene.make_aerial_scene(140, 6000, seed=11)\n"
            "a = A.BaArrays.from_scene(sc); r = ctx.ba_solve(a, capi.default_options(max_num_iterations=8))\n"
            "out.append((repr(r['final_cost']), r['num_iterations'], repr(float(a.cam_pose.sum())), repr(float(a.point.sum())), r['num_residuals'], r['num_reduced_params']))\n"
            "sc = scene.make_aerial_scene(40, 4000, seed=51, n_models=7, gps_sigma=0.5)\n"
            "rng = np.random.default_rng(2); cm = (np.arange(40) %% 5 != 0).astype(np.uint8); pm = (rng.random(4000) > 0.2).astype(np.uint8); mm = (np.arange(7) %% 3 != 1).astype(np.uint8)\n"
            "a = A.BaArrays.from_scene(sc, cam_mutable=cm, pt_mutable=pm, model_mutable=mm, gps_xyz=sc.gps_xyz, gps_weight=60.0); r = ctx.ba_solve(a, capi.default_options(max_num_iterations=8))\n"
            "out.append((repr(r['final_cost']), r['num_iterations'], repr(float(a.cam_pose.sum())), repr(float(a.point.sum())), repr(float(a.cam_model.sum())), r['num_residuals'], r['num_reduced_params']))\n"
            "arr, _ = window.partial_bundle_adjustment_problem(sc, 39, gps=True); r = ctx.ba_solve(arr, capi.default_options(max_num_iterations=6))\n"
            "out.append((repr(r['final_cost']), r['num_iterations'], repr(float(arr.cam_pose.sum())), repr(float(arr.point.sum())), r['num_residuals']))\n"
            "sc = scene.make_ring_scene(3, 7, seed=2)\n"
            "a = A.BaArrays.from_scene(sc); r = ctx.ba_solve(a, capi.default_options(max_num_iterations=4))\n"
            "out.append((repr(r['final_cost']), r['num_iterations'], repr(float(a.cam_pose.sum()))))\n"
            "print(out)\n") % ROOT
    outs = []
    for host in ("0", "1"):
        env = dict(os.environ, MSFM_CREATE_HOST=host, MSFM_CHOL_DOMAINS="1")
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600))
        assert outs[-1].returncode == 0, outs[-1].stderr[-3000:]
    assert outs[0].stdout == outs[1].stdout and outs[0].stdout.strip()


def test_compact_window_hand_over_gives_the_same_solution(ctx):
    """window.gather(compact=True) hands over only the rows that make residual blocks and the points that have one (what
    the reference's own loop adds, optimizer.cc:86-125); the library drops the rest of a full hand-over itself.  Same
    accept / reject sequence and costs, parameters equal to rounding (the point blocks are numbered differently)."""
    from metricsfm_amd import capi, window
    sc = scene.make_aerial_scene(120, 30000, seed=91, n_models=120, gps_sigma=0.5, rot_sigma=2e-3, trans_sigma=0.05, point_sigma=0.05)
    scene.perturb_camera(sc, 119)
    full, fi = window.partial_bundle_adjustment_problem(sc, 119, gps=True)
    comp, ci = window.partial_bundle_adjustment_problem(sc, 119, gps=True, compact=True)
    assert len(comp.obs_cam) < len(full.obs_cam) // 2
    opt = capi.default_options(max_num_iterations=8)
    rf, rc = ctx.ba_solve(full, opt), ctx.ba_solve(comp, opt)
    assert rf["num_residuals"] == rc["num_residuals"] and rf["num_reduced_params"] == rc["num_reduced_params"]
    assert rf["num_iterations"] == rc["num_iterations"]
    np.testing.assert_allclose(rc["iterations"]["cost"], rf["iterations"]["cost"], rtol=1e-11)
    np.testing.assert_allclose(comp.cam_pose, full.cam_pose, rtol=0, atol=1e-9)
    pos = {p: i for i, p in enumerate(fi["kept"])}
    sel = np.array([pos[p] for p in ci["kept"]])
    np.testing.assert_allclose(comp.point, full.point[sel], rtol=0, atol=1e-9)


def test_ba_small_angle_branch_cameras(ctx, oracle):
    """Cameras whose angle-axis vector is exactly zero, or so short that |a|^2 <= DBL_EPSILON, take the first-order branch of
    the rotation formula (SfM/src/utils/basic_funcs.cc:122,165) and its exact derivative; the per-camera rotation data
    (msfm_rot_prepare) carries the branch flag.  Iteration 0 is evaluated in that branch, later ones leave it."""
    rng = np.random.default_rng(77)
    n_cams, n_pts = 8, 600
    pose = np.zeros((n_cams, 6))
    pose[:, 3] = -np.linspace(0.0, 7.0, n_cams)                 # t = -R c, cameras in a row along x, looking along +z
    pose[3, :3] = [1e-9, -2e-9, 5e-10]
    pose[4, :3] = [-3e-10, 1e-10, 8e-9]
    pose[5:, :3] = rng.normal(0, 0.05, (3, 3))
    pt = np.column_stack([rng.uniform(-5, 12, n_pts), rng.uniform(-3, 3, n_pts), rng.uniform(8, 14, n_pts)])
    model = np.array([[800.0, 0.0, 0.0]])
    obs_cam = np.tile(np.arange(n_cams, dtype=np.int32), n_pts)
    obs_pt = np.repeat(np.arange(n_pts, dtype=np.int32), n_cams)
    uv, depth = scene.project(pose[obs_cam], model[np.zeros(len(obs_cam), int)], pt[obs_pt])
    assert (depth > 0).all()
    sc = scene.Scene(name="small-angle", cam_pose_gt=pose.copy(), cam_model_gt=model.copy(), point_gt=pt.copy(), cam_pose=pose.copy(),
                     cam_model=model.copy(), point=pt + rng.normal(0, 0.05, pt.shape), cam_model_of_cam=np.zeros(n_cams, np.int32),
                     obs_cam=obs_cam, obs_pt=obs_pt, obs_xy=uv + rng.normal(0, 0.3, uv.shape), pt_weight=np.ones(n_pts))
    assert ((sc.cam_pose[:5, :3] ** 2).sum(axis=1) <= 2.220446049250313e-16).all()
    r, _, a = check_parity(ctx, oracle, lambda: A.BaArrays.from_scene(sc), dict(max_num_iterations=6))
    assert r["iterations"]["cost"][-1] < r["iterations"]["cost"][0]
    assert np.abs(a.cam_pose[:5, :3]).max() > 1e-7             # the adjusted cameras have left the branch


@pytest.mark.parametrize("domains", ["0", "1", "2"])
def test_ba_persistent_panel_chain_is_bit_identical_to_the_launch_chain(ctx, monkeypatch, domains):
    """Round 4: the panel chain of a tree level runs as ONE persistent launch (k_chain: row owners keep their rows in
    registers from step to step, L[t, t-1] travels through a self-tagged hand-off buffer, bulk tiles by ticket).  Same
    arithmetic in the same order as one launch per 64-column panel (MSFM_CHOL_LAUNCHES=1): every cost, gradient norm and
    parameter of the trajectory is bit-identical - dense order, one bisection (2 leaves | root) and two (4 | 2 | root),
    with GPS rows, and twice in a row on the same resident problem (the hand-off buffers alternate from solve to solve)."""
    from metricsfm_amd import capi
    sc = scene.make_aerial_scene(168, 6000, seed=5, gps_sigma=0.5)
    kw = dict(gps_xyz=sc.gps_xyz, gps_weight=float(sc.n_obs // sc.n_cams))
    monkeypatch.setenv("MSFM_CHOL_DOMAINS", domains)
    opts = capi.default_options(max_num_iterations=9)
    out = {}
    for mode in ("launches", "chain"):
        if mode == "launches":
            monkeypatch.setenv("MSFM_CHOL_LAUNCHES", "1")
        else:
            monkeypatch.delenv("MSFM_CHOL_LAUNCHES")
        a = A.BaArrays.from_scene(sc, **kw)
        r = ctx.ba_solve(a, opts)
        out[mode] = (r, a)
    (r0, a0), (r1, a1) = out["launches"], out["chain"]
    assert r0["num_iterations"] == r1["num_iterations"] >= 5
    for key in ("cost", "gradient_max_norm", "step_norm", "step_is_successful"):
        np.testing.assert_array_equal(r0["iterations"][key], r1["iterations"][key])
    for name in ("cam_pose", "cam_model", "point"):
        np.testing.assert_array_equal(getattr(a0, name), getattr(a1, name))
    # a resident problem solved twice: the second run starts from the first one's result and must agree with the launch chain too
    res = {}
    for mode in ("launches", "chain"):
        if mode == "launches":
            monkeypatch.setenv("MSFM_CHOL_LAUNCHES", "1")
        else:
            monkeypatch.delenv("MSFM_CHOL_LAUNCHES")
        ba = ctx.ba(A.BaArrays.from_scene(sc, **kw))
        ra = ba.run(capi.default_options(max_num_iterations=3))
        rb = ba.run(capi.default_options(max_num_iterations=3))
        res[mode] = (ra["iterations"]["cost"], rb["iterations"]["cost"], ba.download())
        ba.close()
    np.testing.assert_array_equal(res["launches"][0], res["chain"][0])
    np.testing.assert_array_equal(res["launches"][1], res["chain"][1])
    for x, y in zip(res["launches"][2], res["chain"][2]):
        np.testing.assert_array_equal(x, y)


def test_chain_barriers_pair_up():
    """A row-owner workgroup of k_chain runs two programs - the pivot wave and three helper waves - between the SAME nine
    barriers per 64-column step (chol.hip: CHAIN_BAR(name)).  libmsfm_barcheck.so is the library with every one of those barriers
    logged per wave and compared behind the barrier (a mismatch -> MSFM_E_DEVICE).  Sweep of system sizes with the chain forced
    (MSFM_CHAIN_FORCE=1): ragged last blocks with 1 .. 63 real columns, one to fifteen blocks, dense order and one bisection -
    every solve must pass the check AND be bit-identical to one launch per panel (MSFM_CHOL_LAUNCHES=1)."""
    import subprocess
    import sys
    lib = os.path.join(ROOT, "metricsfm_amd", "libmsfm_barcheck.so")
    assert os.path.exists(lib), "libmsfm_barcheck.so not built (run __graft_entry__.build())"
    code = ("import os, sys, numpy as np; sys.path.insert(0, %r)\n"
            "from metricsfm_amd import _abi as A, capi, scene\n"
            "ctx = capi.Context(0)\n"
            "for n_cams, dom in ((21, '0'), (22, '0'), (32, '0'), (43, '0'), (64, '0'), (75, '0'), (107, '0'), (150, '0'), (150, '1')):\n"
            "    os.environ['MSFM_CHOL_DOMAINS'] = dom\n"
            "    sc = scene.make_aerial_scene(n_cams, 30 * n_cams, seed=100 + n_cams)\n"
            "    out = []\n"
            "    for launches in (True, False):\n"
            "        if launches: os.environ['MSFM_CHOL_LAUNCHES'] = '1'\n"
            "        else: os.environ.pop('MSFM_CHOL_LAUNCHES', None)\n"
            "        a = A.BaArrays.from_scene(sc)\n"
            "        r = ctx.ba_solve(a, capi.default_options(max_num_iterations=4))\n"
            "        out.append((r['iterations']['cost'].copy(), a.cam_pose.copy(), a.point.copy()))\n"
            "    for x, y in zip(out[0], out[1]): assert np.array_equal(x, y), n_cams\n"
            "print('ok')\n") % ROOT
    env = dict(os.environ, MSFM_LIB=lib, MSFM_CHAIN_FORCE="1", MSFM_CHAIN_TRACE="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    traced = [l for l in r.stderr.splitlines() if l.startswith("k_chain:")]
    sizes = sorted({int(l.split()[2].rstrip(",")) for l in traced})
    assert len(traced) >= 9 * 4 and len(sizes) >= 8, (len(traced), sizes)   # the persistent launch really ran at every size


def test_ba_device_side_verdict_and_launch_ahead_are_bit_identical_to_the_host_order(ctx, monkeypatch):
    """Round 4: the step's verdict (valid / tolerances / accept + radius / reject) is formed on the device and the next
    linearisation - rows of frozen points, GPS rows, k_point - is enqueued behind it before the host has seen the step,
    gated and given its radius by the device.  MSFM_SPEC=0 enqueues it only after the read-back, as before.  Both orders
    must give the same trajectory to the last bit: accepted and rejected steps (frozen cameras left at perturbed poses), the iteration cap,
    convergence by tolerance, frozen cameras / points / intrinsics with GPS rows (the gated small kernels), and a resident
    problem run twice (a launch enqueued ahead of the last step of the first run must leave nothing behind)."""
    from metricsfm_amd import capi
    rng = np.random.default_rng(5)
    sc1 = scene.make_ring_scene(6, 300, seed=11)
    sc2 = scene.make_aerial_scene(24, 3000, seed=5, gps_sigma=0.5, rot_sigma=0.02, trans_sigma=0.2, point_sigma=0.2)
    cam_mut = (np.arange(sc2.n_cams) % 3 != 0).astype(np.uint8)
    pt_mut = (rng.random(sc2.n_points) > 0.2).astype(np.uint8)
    sc3 = scene.make_aerial_scene(14, 2000, seed=17)
    cm3 = np.ones(sc3.n_cams, np.uint8); cm3[::4] = 0
    pm3 = (np.arange(sc3.n_points) % 11 != 0).astype(np.uint8)
    cases = [
        (lambda: A.BaArrays.from_scene(sc3, cam_mutable=cm3, pt_mutable=pm3), dict(max_num_iterations=12)),   # rejected steps
        (lambda: A.BaArrays.from_scene(sc1), dict(max_num_iterations=14, initial_trust_region_radius=1e7)),
        (lambda: A.BaArrays.from_scene(sc1), dict(max_num_iterations=5, initial_trust_region_radius=1e-2)),   # the cap
        (lambda: A.BaArrays.from_scene(sc1), dict(max_num_iterations=60)),                                    # a tolerance stops it
        (lambda: A.BaArrays.from_scene(sc2, gps_xyz=sc2.gps_xyz, gps_weight=float(sc2.n_obs // sc2.n_cams), cam_mutable=cam_mut, pt_mutable=pt_mut),
         dict(max_num_iterations=20)),
    ]
    for mk, kw in cases:
        out = []
        # (third variant: the launches behind the back substitution - model cost change of the remaining rows, cost at the
        #  candidate, its GPS rows - separately instead of as k_tail with one reduction for the four sums)
        for spec, tail in (("0", "1"), ("1", "1"), ("1", "0")):
            monkeypatch.setenv("MSFM_SPEC", spec)
            monkeypatch.setenv("MSFM_FUSED_TAIL", tail)
            a = mk()
            r = ctx.ba_solve(a, capi.default_options(**kw))
            out.append((r, a))
        monkeypatch.delenv("MSFM_FUSED_TAIL")
        (r0, a0) = out[0]
        for (r1, a1) in out[1:]:
            assert r0["termination"] == r1["termination"] and r0["num_iterations"] == r1["num_iterations"]
            for key in ("cost", "gradient_max_norm", "step_norm", "step_is_successful", "step_is_valid", "trust_region_radius", "relative_decrease"):
                np.testing.assert_array_equal(r0["iterations"][key], r1["iterations"][key])
            for name in ("cam_pose", "cam_model", "point"):
                np.testing.assert_array_equal(getattr(a0, name), getattr(a1, name))
    r_rej = ctx.ba_solve(cases[0][0](), capi.default_options(**cases[0][1]))
    ok_steps = r_rej["iterations"]["step_is_successful"][1:]
    assert (ok_steps == 0).sum() >= 3 and (ok_steps == 1).sum() >= 3, "the first case is meant to mix rejected and accepted steps"
    res = []
    for spec in ("0", "1"):
        monkeypatch.setenv("MSFM_SPEC", spec)
        ba = ctx.ba(cases[4][0]())
        ra = ba.run(capi.default_options(max_num_iterations=3))
        rb = ba.run(capi.default_options(max_num_iterations=4))
        res.append((ra["iterations"]["cost"], rb["iterations"]["cost"], ba.download()))
        ba.close()
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    for x, y in zip(res[0][2], res[1][2]):
        np.testing.assert_array_equal(x, y)
