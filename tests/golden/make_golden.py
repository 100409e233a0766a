#!/usr/bin/env python3
"""Generates the fixtures in this directory.  The reference has no golden vectors and cannot be
built or imported here (C++/MSVC, Ceres/Eigen/OpenCV/FLANN absent — SURVEY.md §8c), so these are
outputs of the repo's own CPU oracle on seeded inputs: they pin the oracle against regressions and
give the GPU tests byte-stable inputs; they are NOT outputs of the reference ("parity unpinned")."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from metricsfm_amd import _abi as A, scene  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    # BA: 8 ring cameras, 120 points, a frozen camera and some frozen points (window semantics)
    sc = scene.make_ring_scene(8, 120, seed=1234)
    cam_mut = np.ones(8, np.uint8); cam_mut[0] = 0
    pt_mut = (np.arange(120) % 7 != 0).astype(np.uint8)
    w = np.full(120, 2.0)
    arr = A.BaArrays(sc.cam_pose, sc.cam_model, sc.cam_model_of_cam, sc.point, sc.obs_cam, sc.obs_pt, sc.obs_xy, w,
                     cam_mutable=cam_mut, pt_mutable=pt_mut)
    max_it = 25
    res = O.ba_solve(arr, O.default_options(max_num_iterations=max_it))
    np.savez_compressed(os.path.join(HERE, "ba_c1_small.npz"), cam_pose0=sc.cam_pose, cam_model0=sc.cam_model,
                        point0=sc.point, cam_model_of_cam=sc.cam_model_of_cam, obs_cam=sc.obs_cam, obs_pt=sc.obs_pt,
                        obs_xy=sc.obs_xy, pt_weight=w, cam_mutable=cam_mut, pt_mutable=pt_mut, max_it=max_it,
                        traj_cost=res["iterations"]["cost"], traj_ok=res["iterations"]["step_is_successful"],
                        cam_pose=arr.cam_pose, cam_model=arr.cam_model, point=arr.point)
    # matching: 96 x 80 integer descriptors with duplicates
    rng = np.random.default_rng(99)
    tr = scene._sift_like(rng, 96).astype(np.float32); qu = scene._sift_like(rng, 80).astype(np.float32)
    tr[70] = tr[2]; qu[5] = tr[2]; qu[6] = tr[33]; qu[6, 3] += 1
    ids, sqd = O.knn2(tr, qu)
    np.savez_compressed(os.path.join(HERE, "knn_small.npz"), train=tr.astype(np.uint8), query=qu.astype(np.uint8), ids=ids, sqd=sqd)
    # triangulation
    sc = scene.make_ring_scene(6, 60, seed=77)
    R, t, c, fk = scene.cameras_for_tracks(sc)
    tra = A.TrackArrays(sc.track_offsets(), sc.obs_cam, sc.obs_xy, R, t, c, fk)
    th = np.deg2rad(3.0)
    Xm, _, okm = O.triangulate_midpoint(tra, 3.0, th)
    Xd, _, okd = O.triangulate_dlt(tra, 3.0, th)
    np.savez_compressed(os.path.join(HERE, "tri_small.npz"), off=tra.track_off, cam=tra.track_cam, xy=tra.track_xy, R=R, t=t,
                        c=c, fk=fk, th_angle=th, X_mid=Xm, ok_mid=okm, X_dlt=Xd, ok_dlt=okd)
    # geometric verification: three pairs (one below the 30-point gate), oracle output of the FM_RANSAC restatement
    sys.path.insert(0, os.path.dirname(HERE))
    from twoview import make_batch  # noqa: E402
    off, p1, p2, _ = make_batch(2024, [150, 25, 60], outlier_frac=0.3)
    F, inl, nin, ok = O.fundamental_ransac(off, p1, p2)
    np.savez_compressed(os.path.join(HERE, "fransac_small.npz"), off=off, pt1=p1, pt2=p2, F=F, inlier=inl, n_inliers=nin, ok=ok)
    # tracks: a small match graph with conflicting matches
    pairs = [(0, 1), (0, 2), (1, 2), (2, 3), (1, 3), (0, 3)]
    rng = np.random.default_rng(5)
    matches = [np.column_stack([rng.permutation(40)[:18], rng.permutation(40)[:18]]).astype(np.int32) for _ in pairs]
    toff, timg, tfeat = O.build_tracks(pairs, matches)
    np.savez_compressed(os.path.join(HERE, "tracks_small.npz"), pairs=np.array(pairs, np.int32), match_off=np.cumsum([0] + [len(m) for m in matches]),
                        matches=np.concatenate(matches), track_off=toff, obs_image=timg, obs_feature=tfeat)
    # pose initialisers: EPnP RANSAC for three images, five-point RANSAC for three pairs (one with 8 matches: the
    # all-points branch), oracle output
    from twoview import make_pnp_batch, make_relpose_batch  # noqa: E402
    seed = 0x4D53
    poff, X, x, _, _ = make_pnp_batch(77, [60, 4, 25], outlier_frac=0.2)
    pf = np.array([4800.0, 4800.0, 4000.0])
    R, t, err, avg, best = O.epnp_ransac(poff, X, x, pf, max_iter=50, seed=seed)
    roff, a, b, _, _ = make_relpose_batch(78, [80, 8, 30], outlier_frac=0.1)
    f1 = np.array([4800.0, 4800.0, 4000.0]); f2 = np.array([4800.0, 4700.0, 4000.0])
    E, R2, t2, ok, nc = O.relpose_5pt(roff, a, b, f1, f2, ransac_times=40, seed=seed)
    np.savez_compressed(os.path.join(HERE, "pose_small.npz"), seed=seed, pnp_off=poff, pnp_X=X, pnp_x=x, pnp_f=pf, pnp_iters=50, pnp_R=R,
                        pnp_t=t, pnp_err=err, pnp_avg=avg, pnp_best=best, rel_off=roff, rel_a=a, rel_b=b, rel_f1=f1, rel_f2=f2,
                        rel_times=40, rel_E=E, rel_R=R2, rel_t=t2, rel_ok=ok, rel_nc=nc)


if __name__ == "__main__":
    main()
