#!/usr/bin/env python3
"""The reference's CALLING PATTERN for bundle adjustment, measured (bench.py's `ba_incremental` leg imports this):
IncrementalSfM::Run adds one image at a time - LocalizeImage -> GenerateNew3DPoints -> PartialBundleAdjustment(new camera)
-> every th_step_full_bundle_adjustment-th image (5, basic_structs.h:183) a FullBundleAdjustment -> RemovePointOutliers
(sfm_incremental.cc:146-190).  Every bundle adjustment there is a fresh `ceres::Solve` on a fresh problem, so every one here is
a fresh msfm_ba_solve: create (upload + index structures) + LM iterations + download.

  sequence(ctx, config=2): cameras of BASELINE config 2 in flight-line order from the seed pair on; per added camera the
      partial bundle adjustment of its window (compact hand-over: the residual blocks the reference's loop adds), every 5th a
      full one over the cameras so far; the model state is carried from call to call.
  windows(ctx, n=20): twenty consecutive windows of BASELINE config 5 (the newest cameras of the 2000, GPS rows).
Reports per kind: calls, total ms, set-up ms (summary.setup_ms: upload + symbolic set-up), LM iterations."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metricsfm_amd import capi, scene, window  # noqa: E402


def prefix_scene(sc, n_cams, pose, point):
    """The model after `n_cams` images: their cameras, the observations they hold and the points at least two of them see
    (GenerateNew3DPoints triangulates from two views on), with the current state of poses and points."""
    sel = sc.obs_cam < n_cams
    k = np.bincount(sc.obs_pt[sel], minlength=sc.n_points)
    keep = k >= 2
    sel &= keep[sc.obs_pt]
    new_id = (np.cumsum(keep) - 1).astype(np.int32)
    sub = scene.Scene(name="%s[:%d]" % (sc.name, n_cams), cam_pose_gt=sc.cam_pose_gt[:n_cams], cam_model_gt=sc.cam_model_gt, point_gt=sc.point_gt[keep],
                      cam_pose=pose[:n_cams].copy(), cam_model=sc.cam_model.copy(), point=point[keep].copy(),
                      cam_model_of_cam=sc.cam_model_of_cam[:n_cams], obs_cam=sc.obs_cam[sel], obs_pt=new_id[sc.obs_pt[sel]], obs_xy=sc.obs_xy[sel],
                      pt_weight=sc.pt_weight[keep], gps_xyz=None if sc.gps_xyz is None else sc.gps_xyz[:n_cams])
    return sub, np.nonzero(keep)[0]


def _acc(stat, r, wall):
    stat["calls"] += 1
    stat["total_ms"] += 1e3 * wall
    stat["setup_ms"] += r["setup_ms"]
    stat["iterations"] += r["num_iterations"]


def _finish(stat):
    stat["setup_share"] = stat["setup_ms"] / max(1e-9, stat["total_ms"])
    stat["ms_per_call"] = stat["total_ms"] / max(1, stat["calls"])
    return stat


def sequence(ctx, config=2, step_full=5, max_iter=100, first=2):
    sc = scene.config_scene(config)
    pose, point = sc.cam_pose.copy(), sc.point.copy()
    part = dict(calls=0, total_ms=0.0, setup_ms=0.0, iterations=0)
    full = dict(calls=0, total_ms=0.0, setup_ms=0.0, iterations=0)
    opts = capi.default_options(max_num_iterations=max_iter)
    added = 0
    for n in range(first + 1, sc.n_cams + 1):          # the (n - 1)-th camera is the new one
        idx = n - 1
        sub, kept = prefix_scene(sc, n, pose, point)
        scene.perturb_camera(sub, idx)                 # straight from LocalizeImage
        arr, info = window.partial_bundle_adjustment_problem(sub, idx, compact=True)
        t0 = time.perf_counter()
        r = ctx.ba_solve(arr, opts)
        _acc(part, r, time.perf_counter() - t0)
        pose[:n] = arr.cam_pose
        point[kept[info["kept"]]] = arr.point
        added += 1
        if added % step_full == 0:
            sub, kept = prefix_scene(sc, n, pose, point)
            arr, kp = window.gather(sub, compact=True)
            t0 = time.perf_counter()
            r = ctx.ba_solve(arr, opts)
            _acc(full, r, time.perf_counter() - t0)
            pose[:n] = arr.cam_pose
            point[kept[kp]] = arr.point
    return dict(workload="config %d: cameras %d..%d added one at a time (sfm_incremental.cc:146-190), partial BA per camera, full BA every %d-th; "
                         "every call a fresh msfm_ba_solve from host arrays" % (config, first, sc.n_cams - 1, step_full),
                partial=_finish(part), full=_finish(full),
                total_ms=part["total_ms"] + full["total_ms"], setup_ms=part["setup_ms"] + full["setup_ms"],
                setup_share=(part["setup_ms"] + full["setup_ms"]) / max(1e-9, part["total_ms"] + full["total_ms"]))


def windows(ctx, n=20, max_iter=20, sc=None):
    sc = sc or scene.config_scene(5, n_models=2000, rot_sigma=2e-4, trans_sigma=0.01, point_sigma=0.02)
    stat = dict(calls=0, total_ms=0.0, setup_ms=0.0, iterations=0)
    opts = capi.default_options(max_num_iterations=max_iter)
    gather_ms = 0.0
    for idx in range(sc.n_cams - n, sc.n_cams):
        scene.perturb_camera(sc, idx)
        t0 = time.perf_counter()
        arr, info = window.partial_bundle_adjustment_problem(sc, idx, gps=True, compact=True)
        gather_ms += 1e3 * (time.perf_counter() - t0)
        t0 = time.perf_counter()
        r = ctx.ba_solve(arr, opts)
        _acc(stat, r, time.perf_counter() - t0)
        sc.cam_pose[:] = arr.cam_pose
        sc.cam_model[:] = arr.cam_model
        sc.point[info["kept"]] = arr.point
    out = _finish(stat)
    out["workload"] = ("config 5: the windows of its %d newest cameras one after the other (PartialBundleAdjustment + GPS rows, compact hand-over), "
                       "each a fresh msfm_ba_solve of at most %d iterations" % (n, max_iter))
    out["host_gather_ms"] = gather_ms
    return out


if __name__ == "__main__":
    ctx = capi.Context(0)
    out = dict(sequence_c2=sequence(ctx, 2))
    if "--c5" in sys.argv:
        out["windows_c5"] = windows(ctx)
    print(json.dumps(out))
