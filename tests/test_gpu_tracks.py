"""Track building on the GPU (msfm_tracks_build_device): identical - every offset, image and feature -
to the host walk over the match lists (msfm_tracks_build) and, at oracle sizes, to the literal std::map restatement of
SLAMGPS::Triangulation's data association (slam_gps.cc:565-635) in the oracle."""
import os
import time

import numpy as np
import pytest

from metricsfm_amd import capi, scene
from tests.test_tracks import _scene_matches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context()
    yield c
    c.close()


@pytest.mark.parametrize("p_wrong", [0.0, 0.15, 0.6])
def test_device_tracks_match_the_oracle_and_the_host_walk(ctx, oracle, p_wrong):
    rng = np.random.default_rng(11)
    n_img, n_feat = 9, 400
    pairs, matches = _scene_matches(rng, n_img, 250, n_feat, p_wrong=p_wrong)
    got = ctx.build_tracks([n_feat] * n_img, pairs, matches)
    host = capi.build_tracks([n_feat] * n_img, pairs, matches)
    want = oracle.build_tracks(pairs, matches)
    for g, h, w in zip(got, host, want):
        np.testing.assert_array_equal(g, w)
        np.testing.assert_array_equal(g, h)


def test_device_tracks_quirks_errors_and_golden(ctx):
    # two existing points are never merged; a feature stays with the first point it was mapped to; the same match twice;
    # an empty pair in the middle of the list
    pairs = [(0, 1), (2, 3), (1, 2), (3, 0), (0, 2), (0, 1)]
    matches = [np.array([[5, 6]]), np.array([[7, 8]]), np.array([[6, 7]]), np.zeros((0, 2), dtype=np.int32), np.array([[5, 9]]),
               np.array([[5, 6], [4, 6], [5, 3]])]
    got = ctx.build_tracks([10, 10, 10, 10], pairs, matches)
    host = capi.build_tracks([10, 10, 10, 10], pairs, matches)
    for g, h in zip(got, host):
        np.testing.assert_array_equal(g, h)
    off, img, feat = got
    assert list(img[off[0]:off[1]]) == [0, 1, 2] and list(feat[off[0]:off[1]]) == [5, 6, 7]
    with pytest.raises(capi.MsfmError):
        ctx.build_tracks([10, 10], [(0, 1)], [np.array([[3, 12]])])        # feature index out of range
    with pytest.raises(capi.MsfmError):
        ctx.build_tracks([10, 10], [(0, 2)], [np.array([[3, 1]])])         # image out of range
    assert len(ctx.build_tracks([4, 4], [], [])[0]) == 1
    k = np.load(os.path.join(os.path.dirname(__file__), "golden", "tracks_small.npz"))
    mo = k["match_off"]
    got = ctx.build_tracks([40, 40, 40, 40], [tuple(p) for p in k["pairs"]], [k["matches"][mo[i]:mo[i + 1]] for i in range(len(k["pairs"]))])
    for g, name in zip(got, ("track_off", "obs_image", "obs_feature")):
        np.testing.assert_array_equal(g, k[name])


def test_device_tracks_long_chains(ctx):
    # one point seen by 300 images, matched as a chain in an order that makes the "joins" forest deep: image k is matched
    # with image k + 1 only, visited from the far end first
    n = 300
    pairs = [(k, k + 1) for k in range(n - 2, -1, -1)]
    matches = [np.array([[0, 0], [1, 1]]) for _ in pairs]
    got = ctx.build_tracks([2] * n, pairs, matches)
    host = capi.build_tracks([2] * n, pairs, matches)
    for g, h in zip(got, host):
        np.testing.assert_array_equal(g, h)


def test_device_tracks_config2_scale_matches_host(ctx):
    """Config 2's scene (50 cameras / 20 000 points): matches of every ordered image pair as the matcher would report them
    (plus 2 % wrong ones), ~0.8 M matches; device == host, and the consistent part recovers the scene's tracks."""
    from metricsfm_amd import tracks as T
    sc = scene.config_scene(2)
    nf, off_pairs, moff, flat = T.flat_matches_from_scene(sc, wrong=0.02, seed=5)
    t0 = time.perf_counter()
    got = ctx.build_tracks(None, None, None, flat=(nf, off_pairs, moff, flat))
    t1 = time.perf_counter()
    h = capi.build_tracks_flat(nf, off_pairs, moff, flat)
    t2 = time.perf_counter()
    for g, w in zip(got, h):
        np.testing.assert_array_equal(g, w)
    print(f"tracks C2: {len(flat)} matches, {len(got[0]) - 1} tracks; device {1e3 * (t1 - t0):.1f} ms, host walk {1e3 * (t2 - t1):.1f} ms")
