"""Host-side window selection of the incremental loop, on the flat arrays of `msfm_ba_problem`.

Mirrors, without the object graph:
  IncrementalSfM::FindImageToLocalize   SfM/src/sfm_incremental.cc:448-506  (which cameras are "visible" from a new one)
  IncrementalSfM::UpdateVisibleGraph    SfM/src/sfm_incremental.cc:1895-1903
  IncrementalSfM::ImmutableCamsPoints   SfM/src/sfm_incremental.cc:1865-1878
  IncrementalSfM::MutableCamsPoints     SfM/src/sfm_incremental.cc:1880-1893
  IncrementalSfM::PartialBundleAdjustment  SfM/src/sfm_incremental.cc:917-1014
  SLAMGPS::FullBundleAdjustment (GPS rows) SfM/src/slam_gps.cc:714-832

The masks select the reference's functors inside libmsfm (include/msfm.h, msfm_ba_problem): a frozen camera
seen by a free point gives a ReprojectionErrorXYZ row, both frozen gives no residual (optimizer.cc:86-125).
"""
from __future__ import annotations

import numpy as np

from . import _abi as A

TH_VISIBLE = 5          # `count_2d3d_ij > 5`, sfm_incremental.cc:503
PARTIAL_WEIGHT = 2.0    # RunOptimizetion(!found_seed_, 2.0), sfm_incremental.cc:1012
FULL_WEIGHT = 1.0       # sfm_incremental.cc:1024


def shared_point_counts(obs_cam, obs_pt, n_cams, cam, bad=None):
    """count[j] = 2D-3D matches camera `cam` has through camera j: points observed by both whose 3-D point is not
    `is_bad_estimated_` (sfm_incremental.cc:486-499; on a synthetic scene every common point is a match)."""
    obs_cam, obs_pt = np.asarray(obs_cam), np.asarray(obs_pt)
    mine = np.zeros(int(obs_pt.max()) + 1 if len(obs_pt) else 0, dtype=bool)
    mine[obs_pt[obs_cam == cam]] = True
    if bad is not None:
        mine &= ~np.asarray(bad, dtype=bool)[: len(mine)]
    return np.bincount(obs_cam[mine[obs_pt]], minlength=n_cams)


def visible_cameras(obs_cam, obs_pt, n_cams, cam, bad=None, th=TH_VISIBLE):
    """`visible_cams_` of a newly localised camera: itself first (UpdateVisibleGraph pushes idx_new_cam before the list,
    :1897), then every other camera with more than `th` shared 2D-3D matches, ascending (the loop over images at :455)."""
    cnt = shared_point_counts(obs_cam, obs_pt, n_cams, cam, bad)
    cnt[cam] = 0
    return np.concatenate([[cam], np.nonzero(cnt > th)[0]]).astype(np.int32)


def immutable_cams_points(n_cams, n_points):
    """ImmutableCamsPoints (:1865-1878): every camera and every point attached to a camera is frozen."""
    return np.zeros(n_cams, np.uint8), np.zeros(n_points, np.uint8)


def mutable_cams_points(n_cams, n_points):
    """MutableCamsPoints (:1880-1893)."""
    return np.ones(n_cams, np.uint8), np.ones(n_points, np.uint8)


def partial_ba_masks(obs_cam, obs_pt, n_cams, n_points, cam_model_of_cam, idx, visible, bad=None):
    """PartialBundleAdjustment(idx) (:917-945): freeze everything, then free (a) every camera of the new camera's
    CameraModel (`cam_model_->idx_cams_`, :922-933) and (b) its `visible_cams_` (:934-945), each with all of its
    points that are not bad.  With one shared model (use_same_camera, UAV mode) (a) frees every camera."""
    obs_cam, obs_pt = np.asarray(obs_cam), np.asarray(obs_pt)
    cam_mut, pt_mut = immutable_cams_points(n_cams, n_points)
    cam_mut[np.asarray(cam_model_of_cam) == cam_model_of_cam[idx]] = 1
    cam_mut[np.asarray(visible, dtype=np.int64)] = 1
    pt_mut[obs_pt[cam_mut[obs_cam] != 0]] = 1
    if bad is not None:
        pt_mut[np.asarray(bad, dtype=bool)] = 0
    return cam_mut, pt_mut


def point_weights(obs_pt, n_points, weight):
    """optimizer.cc:69-78: two views -> 1.0, three or more -> the caller's weight (other lengths keep 1.0)."""
    k = np.bincount(np.asarray(obs_pt), minlength=n_points)
    w = np.ones(n_points)
    w[k >= 3] = weight
    return w


def gps_weight(n_residual_blocks, n_cams):
    """`double weight = count1 / cams_.size();` (slam_gps.cc:824): integer division of the number of reprojection
    residual blocks added to the problem by the number of cameras."""
    return float(int(n_residual_blocks) // int(n_cams))


def gather(sc, cam_mutable=None, pt_mutable=None, weight=FULL_WEIGHT, bad=None, gps=False, compact=False):
    """BundleAdjuster::RunOptimizetion's gather (optimizer.cc:59-129) on a Scene: bad points are dropped (:64), weights
    follow the view count, masks pass through; with `gps` the SLAMGPS rows are attached (weight from the number of
    residual blocks the masks leave, slam_gps.cc:824).  Returns (BaArrays, kept point indices).

    The reference adds a residual block only where the camera or the point is free (:86-125): with `compact` the rows whose
    camera and point are both frozen, and the points left without any row, are not handed over at all (the library would
    drop them itself - the solution is the same - but for the window of one camera out of 2000 that is 6 M rows of PCIe
    traffic for 0.2 M residual blocks)."""
    good = np.ones(sc.n_points, bool) if bad is None else ~np.asarray(bad, dtype=bool)
    keep = good.copy()
    cm = np.ones(sc.n_cams, bool) if cam_mutable is None else np.asarray(cam_mutable) != 0
    pmf = np.ones(sc.n_points, bool) if pt_mutable is None else np.asarray(pt_mutable) != 0
    w_full = point_weights(sc.obs_pt, sc.n_points, weight)          # view counts are those of the whole track (:69-78)
    sel = keep[sc.obs_pt]
    active = cm[sc.obs_cam] | pmf[sc.obs_pt]                        # both frozen -> no residual block
    if compact:
        sel &= active
        keep = keep & (np.bincount(sc.obs_pt[sel], minlength=sc.n_points) > 0)
    new_id = np.cumsum(keep) - 1
    obs_cam, obs_pt = sc.obs_cam[sel], new_id[sc.obs_pt[sel]].astype(np.int32)
    pm = None if pt_mutable is None else np.asarray(pt_mutable, np.uint8)[keep]
    kw = {}
    if gps:
        n_blocks = int((active & good[sc.obs_pt]).sum())
        kw = dict(gps_xyz=sc.gps_xyz, gps_weight=gps_weight(n_blocks, sc.n_cams))
    arr = A.BaArrays(sc.cam_pose, sc.cam_model, sc.cam_model_of_cam, sc.point[keep], obs_cam, obs_pt, sc.obs_xy[sel],
                     w_full[keep], cam_mutable=cam_mutable, pt_mutable=pm, **kw)
    return arr, np.nonzero(keep)[0]


def partial_bundle_adjustment_problem(sc, idx, bad=None, gps=False, th=TH_VISIBLE, compact=False):
    """The problem PartialBundleAdjustment(idx) hands to the solver for camera `idx` of a Scene."""
    vis = visible_cameras(sc.obs_cam, sc.obs_pt, sc.n_cams, idx, bad, th)
    cam_mut, pt_mut = partial_ba_masks(sc.obs_cam, sc.obs_pt, sc.n_cams, sc.n_points, sc.cam_model_of_cam, idx, vis, bad)
    arr, kept = gather(sc, cam_mut, pt_mut, PARTIAL_WEIGHT, bad, gps, compact)
    return arr, dict(visible=vis, cam_mutable=cam_mut, pt_mutable=pt_mut, kept=kept)
