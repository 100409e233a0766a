"""Batched fundamental-matrix RANSAC + epipolar filter on the GPU against the sequential CPU oracle
(reference GeoVerification::GeoVerificationFundamental, geo_verification.cc:30-79).
Same counter-based sampler, arithmetic-only solver, contraction off on both sides: the comparison is exact."""
import numpy as np
import pytest

from metricsfm_amd import capi
from tests.twoview import make_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O(oracle):
    return oracle


def check_equal(g, o):
    Fg, ig, ng, kg = g
    Fo, io, no, ko = o
    np.testing.assert_array_equal(kg, ko)
    np.testing.assert_array_equal(ng, no)
    np.testing.assert_array_equal(ig, io)
    np.testing.assert_array_equal(Fg, Fo)   # bit-exact: + - * / sqrt in one fixed order on both sides


def test_fransac_matches_oracle_mixed_batch(ctx, O):
    sizes = [400, 60, 29, 0, 1500, 31, 2500, 8, 120, 77]   # < 30 -> false; 1500 / 2500 exceed one 1024-match LDS tile
    off, p1, p2, good = make_batch(3, sizes, outlier_frac=0.3)
    g = ctx.fundamental_ransac(off, p1, p2)
    o = O.fundamental_ransac(off, p1, p2)
    check_equal(g, o)
    assert g[3][0] == 1 and g[3][4] == 1 and g[3][2] == 0
    s = slice(off[4], off[5])
    assert g[1][s][good[s]].mean() > 0.9 and g[1][s][~good[s]].mean() < 0.1


@pytest.mark.parametrize("frac,seed", [(0.0, 1), (0.6, 2), (0.8, 3)])
def test_fransac_matches_oracle_outlier_ratios(ctx, O, frac, seed):
    # 80 % outliers keeps the adaptive stop from firing: all 2000 samples are replayed
    off, p1, p2, _ = make_batch(seed, [200] * 6, outlier_frac=frac)
    check_equal(ctx.fundamental_ransac(off, p1, p2, seed=1234 + seed), O.fundamental_ransac(off, p1, p2, seed=1234 + seed))


def test_fransac_options_and_degenerate(ctx, O):
    off, p1, p2, _ = make_batch(9, [90, 90], outlier_frac=0.2)
    kw = dict(threshold=1.5, confidence=0.999, max_iterations=300, min_points=20, min_inliers=50, seed=7)
    check_equal(ctx.fundamental_ransac(off, p1, p2, **kw), O.fundamental_ransac(off, p1, p2, **kw))
    z = np.ones((64, 2), np.float32)
    g = ctx.fundamental_ransac(np.array([0, 64], np.int32), z, z)
    assert g[3][0] == 0 and g[2][0] == 0 and not g[0].any()
    line = np.column_stack([np.arange(64), 2 * np.arange(64)]).astype(np.float32)   # collinear points
    check_equal(ctx.fundamental_ransac(np.array([0, 64], np.int32), line, line + 1), O.fundamental_ransac(np.array([0, 64], np.int32), line, line + 1))
    with pytest.raises(capi.MsfmError):
        ctx.fundamental_ransac(np.array([0, 64], np.int32), z, z, max_iterations=0)


def test_epipolar_filter_batch_matches_oracle(ctx, O):
    off, p1, p2, _ = make_batch(4, [300, 50, 0, 700], outlier_frac=0.3)
    F, _, _, ok = ctx.fundamental_ransac(off, p1, p2)
    ok = ok.copy()
    ok[1] = 0   # a pair whose RANSAC failed keeps nothing (fine_matching_graph.cc:148-150)
    got = ctx.epipolar_filter_batch(off, p1, p2, F, ok, 3.0)
    for p in range(4):
        s = slice(off[p], off[p + 1])
        want = O.epipolar_filter(p1[s], p2[s], F[p], 3.0) if ok[p] else np.zeros(off[p + 1] - off[p], np.uint8)
        np.testing.assert_array_equal(got[s], want)


def test_fransac_golden_fixture(ctx):
    """The committed oracle output (tests/golden/fransac_small.npz) without running the oracle."""
    import os
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "fransac_small.npz"))
    F, inl, nin, ok = ctx.fundamental_ransac(f["off"], f["pt1"], f["pt2"])
    np.testing.assert_array_equal(F, f["F"]); np.testing.assert_array_equal(inl, f["inlier"])
    np.testing.assert_array_equal(nin, f["n_inliers"]); np.testing.assert_array_equal(ok, f["ok"])
