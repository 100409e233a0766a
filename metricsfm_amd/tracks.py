"""Track building and the triangulation of the new tracks, Python host side.

`capi.build_tracks` is SLAMGPS::Triangulation's data association (slam_gps.cc:565-635) through the C ABI;
`triangulate_tracks` is the loop that follows it (:637-648): Trianglate2 with th_tri_angle = 3 degrees for every new
point - one batched GPU call here - and "not ok, or fewer than 3 views" -> is_bad_estimated_."""
import numpy as np

from . import _abi as A
from . import capi


def triangulate_tracks(ctx: capi.Context, track_off, obs_image, obs_feature, keypoints, cam_R, cam_t, cam_c, cam_fk, th_outlier,
                       th_tri_angle=3.0 / 180.0 * np.pi, min_views=3):
    """keypoints[image] = [n_features][2] centred pixels.  Returns X [n][3], mse [n], bad [n] (bool)."""
    track_off = np.asarray(track_off, dtype=np.int32)
    xy = np.zeros((len(obs_image), 2))
    for i in np.unique(obs_image):
        sel = obs_image == i
        xy[sel] = np.asarray(keypoints[i], dtype=np.float64)[obs_feature[sel]]
    tr = A.TrackArrays(track_off, obs_image, xy, cam_R, cam_t, cam_c, cam_fk)
    X, mse, ok = ctx.triangulate_midpoint(tr, th_outlier, th_tri_angle)
    bad = (ok == 0) | (np.diff(track_off) < min_views)   # slam_gps.cc:642-647
    return X, mse, bad


def matches_from_scene(sc, pairs=None):
    """Test / bench helper: every camera's features are its observations (in observation order); the matches of an image
    pair are the scene points both see.  Returns n_features, keypoints, pairs, matches_per_pair."""
    n = sc.n_cams
    feat_of = [dict() for _ in range(n)]
    keyp = [[] for _ in range(n)]
    for o in range(sc.n_obs):
        c, p = int(sc.obs_cam[o]), int(sc.obs_pt[o])
        feat_of[c][p] = len(keyp[c])
        keyp[c].append(sc.obs_xy[o])
    keyp = [np.array(k, dtype=np.float64).reshape(-1, 2) for k in keyp]
    if pairs is None:
        pairs = [(i, j) for i in range(n) for j in range(n) if i != j]
    out_pairs, matches = [], []
    for i, j in pairs:
        common = sorted(set(feat_of[i]) & set(feat_of[j]))
        if not common:
            continue
        out_pairs.append((i, j))
        matches.append(np.array([[feat_of[i][p], feat_of[j][p]] for p in common], dtype=np.int32))
    return [len(k) for k in keyp], keyp, out_pairs, matches
