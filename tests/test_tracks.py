"""Track building: the flat-array restatement in libmsfm (host code, no GPU needed) against the
literal std::map restatement of SLAMGPS::Triangulation's data association (slam_gps.cc:565-635) in the oracle."""
import numpy as np
import pytest

from metricsfm_amd import capi


def _scene_matches(rng, n_img, n_pts, n_feat, p_seen=0.6, p_wrong=0.0):
    """Features of n_pts scene points scattered over the images; matches between every ordered pair i < j (visiting order of
    the reference: idx1 ascending, idx2 ascending), a fraction of them wrong (pointing at a random feature)."""
    feat_pt = [rng.permutation(n_feat)[:n_pts] for _ in range(n_img)]          # feature index of point p in image i
    seen = rng.random((n_img, n_pts)) < p_seen
    pairs, matches = [], []
    for i in range(n_img):
        for j in range(n_img):
            if i == j or rng.random() < 0.3:
                continue
            both = np.nonzero(seen[i] & seen[j])[0]
            if len(both) == 0:
                continue
            m = np.column_stack([feat_pt[i][both], feat_pt[j][both]])
            wrong = rng.random(len(m)) < p_wrong
            m[wrong, 1] = rng.integers(0, n_feat, wrong.sum())
            pairs.append((i, j))
            matches.append(m[rng.permutation(len(m))])
    return pairs, matches


@pytest.mark.parametrize("p_wrong", [0.0, 0.15])
def test_tracks_match_literal_restatement(oracle, p_wrong):
    rng = np.random.default_rng(3)
    n_img, n_feat = 9, 400
    pairs, matches = _scene_matches(rng, n_img, 250, n_feat, p_wrong=p_wrong)
    got = capi.build_tracks([n_feat] * n_img, pairs, matches)
    want = oracle.build_tracks(pairs, matches)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)
    off, img, feat = got
    for t in range(len(off) - 1):
        assert (np.diff(img[off[t]:off[t + 1]]) > 0).all()      # one observation per image, ascending (std::map order)
    if p_wrong == 0.0:
        assert len(off) - 1 <= 250                             # consistent matches never split a scene point's track in two ...
        assert np.diff(off).max() <= n_img


def test_tracks_quirks_and_errors(oracle):
    # two existing points are never merged, and a feature stays with the first point it was mapped to
    pairs = [(0, 1), (2, 3), (1, 2), (0, 2)]
    matches = [np.array([[5, 6]]), np.array([[7, 8]]), np.array([[6, 7]]), np.array([[5, 9]])]
    got = capi.build_tracks([10, 10, 10, 10], pairs, matches)
    want = oracle.build_tracks(pairs, matches)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)
    off, img, feat = got
    assert len(off) - 1 == 2                                   # (1,6)-(2,7) joins nothing: both ends already belong to points
    assert list(img[off[0]:off[1]]) == [0, 1, 2] and list(feat[off[0]:off[1]]) == [5, 6, 7]   # image 2's feature 9 loses to 7
    with pytest.raises(capi.MsfmError):
        capi.build_tracks([10, 10], [(0, 1)], [np.array([[3, 12]])])   # feature index out of range
    assert len(capi.build_tracks([4, 4], [], [])[0]) == 1


def test_tracks_golden_fixture():
    import os
    k = np.load(os.path.join(os.path.dirname(__file__), "golden", "tracks_small.npz"))
    mo = k["match_off"]
    got = capi.build_tracks([40, 40, 40, 40], [tuple(p) for p in k["pairs"]], [k["matches"][mo[i]:mo[i + 1]] for i in range(len(k["pairs"]))])
    for g, name in zip(got, ("track_off", "obs_image", "obs_feature")):
        np.testing.assert_array_equal(g, k[name])
