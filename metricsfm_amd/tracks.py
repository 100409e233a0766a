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


def flat_matches_from_scene(sc, wrong=0.0, seed=0):
    """`matches_from_scene` for every ordered image pair (i, j != i) of a large scene, as flat int32 arrays (vectorised; config
    3: 6 M matches): n_features [n_cams], pairs [P][2] in the reference's visiting order (idx1 ascending, idx2 ascending),
    match_off [P+1], matches [M][2] (points ascending inside a pair).  A fraction `wrong` of the matches has its second
    feature replaced by a random feature of that image (what a real matcher's outliers look like to the track builder)."""
    oc = np.asarray(sc.obs_cam, dtype=np.int64)
    op = np.asarray(sc.obs_pt, dtype=np.int64)
    n, No = sc.n_cams, len(oc)
    assert (np.diff(op) >= 0).all()
    # feature index of an observation = its rank among its camera's observations
    by_cam = np.argsort(oc, kind="stable")
    nf = np.bincount(oc, minlength=n)
    start = np.concatenate([[0], np.cumsum(nf)])
    feat = np.empty(No, dtype=np.int64)
    feat[by_cam] = np.arange(No) - start[oc[by_cam]]
    # every observation paired with the other observations of its point
    k = np.bincount(op, minlength=sc.n_points)
    first = np.concatenate([[0], np.cumsum(k)])
    reps = k[op] - 1
    a = np.repeat(np.arange(No), reps)
    r = np.arange(len(a)) - np.repeat(np.cumsum(reps) - reps, reps)
    pos = np.arange(No) - first[op]
    b = first[op[a]] + r + (r >= pos[a])
    key = oc[a] * n + oc[b]
    order = np.lexsort((op[a], key))
    a, b, key = a[order], b[order], key[order]
    keep = oc[a] != oc[b]                                     # (a point seen twice by one camera makes no pair with itself)
    a, b, key = a[keep], b[keep], key[keep]
    ukey, counts = np.unique(key, return_counts=True)
    pairs = np.column_stack([ukey // n, ukey % n]).astype(np.int32)
    moff = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    m = np.column_stack([feat[a], feat[b]]).astype(np.int32)
    if wrong > 0:
        rng = np.random.default_rng(seed)
        w = np.nonzero(rng.random(len(m)) < wrong)[0]
        m[w, 1] = (rng.random(len(w)) * nf[oc[b[w]]]).astype(np.int32)
    return nf.astype(np.int32), pairs, moff, np.ascontiguousarray(m)


def generate_new_points(ctx: capi.Context, cam1, visible_cams, matches_per_cam, done1, done2_per_cam, keypoints, cam_R, cam_t, cam_c,
                        cam_fk, th_mse_reprojection=3.0, th_angle_small=3.0 / 180.0 * 3.1415, th_angle_large=5.0 / 180.0 * 3.1415):
    """IncrementalSfM::GenerateNew3DPoints (sfm_incremental.cc:755-915) for the newest camera `cam1`: every match with a
    visible camera whose two features are not triangulated yet becomes a two-view candidate; Trianglate2 with
    th_angle_small, or th_angle_large when the pair has more than 500 matches (:780-784); accepted candidates are sorted
    by their mse TRUNCATED to an integer (the reference stores it through a pair<.., int>, :828) - stable here.
    All candidates of all visible cameras go through at most two batched GPU calls (one per angle threshold).

    matches_per_cam[k] = [m][2] (feature in cam1, feature in visible_cams[k]); done1[f] / done2_per_cam[k][f] = already
    triangulated.  Returns X [n][3], mse [n], cam2 [n], feat1 [n], feat2 [n] in the order the reference appends them to pts_."""
    cand = []   # (cam2, f1, f2, large?)
    for k, cam2 in enumerate(visible_cams):
        if cam2 == cam1:
            continue
        m = np.asarray(matches_per_cam[k], dtype=np.int64).reshape(-1, 2)
        large = len(m) > 500
        d2 = np.asarray(done2_per_cam[k], dtype=bool)
        for f1, f2 in m:
            if done1[f1] or d2[f2]:
                continue
            cand.append((cam2, int(f1), int(f2), large))
    if not cand:
        z = np.zeros(0, dtype=np.int64)
        return np.zeros((0, 3)), np.zeros(0), z, z, z
    cand = np.array(cand, dtype=np.int64)
    n = len(cand)
    X, mse, ok = np.zeros((n, 3)), np.zeros(n), np.zeros(n, dtype=np.uint8)
    k1 = np.asarray(keypoints[cam1], dtype=np.float64)
    for large, th in ((0, th_angle_small), (1, th_angle_large)):
        sel = np.nonzero(cand[:, 3] == large)[0]
        if len(sel) == 0:
            continue
        cam = np.column_stack([np.full(len(sel), cam1), cand[sel, 0]]).reshape(-1)
        xy = np.zeros((2 * len(sel), 2))
        xy[0::2] = k1[cand[sel, 1]]
        for c2 in np.unique(cand[sel, 0]):
            s2 = cand[sel, 0] == c2
            xy[1::2][s2] = np.asarray(keypoints[c2], dtype=np.float64)[cand[sel, 2][s2]]
        tr = A.TrackArrays(2 * np.arange(len(sel) + 1, dtype=np.int32), cam.astype(np.int32), xy, cam_R, cam_t, cam_c, cam_fk)
        Xs, ms, oks = ctx.triangulate_midpoint(tr, th_mse_reprojection, th)
        X[sel], mse[sel], ok[sel] = Xs, ms, oks
    keep = np.nonzero(ok)[0]
    order = keep[np.argsort(np.trunc(mse[keep]).astype(np.int64), kind="stable")]
    return X[order], mse[order], cand[order, 0], cand[order, 1], cand[order, 2]
