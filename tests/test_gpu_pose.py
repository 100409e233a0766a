"""Batched pose initialisers on the GPU against the sequential CPU oracle (reference
AbsolutePoseEstimation::AbsolutePoseWithFocalLength, absolute_pose_estimation.cc:42-58, and
RelativePoseEstimation::RelativePoseWithFocalLength, relative_pose_estimation.cc:91-120).
Same counter-based sampler, only + - * / sqrt in one fixed order, contraction off on both sides: the comparison is exact."""
import os

import numpy as np
import pytest

from tests.twoview import make_pnp_batch, make_relpose_batch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "pose_small.npz")


@pytest.fixture(scope="module")
def O(oracle):
    return oracle


def equal(g, o):
    for a, b in zip(g, o):
        np.testing.assert_array_equal(a, b)   # NaNs compare equal, everything else bit for bit


def test_epnp_matches_oracle_mixed_batch(ctx, O):
    sizes = [300, 0, 3, 4, 5, 40, 2500, 64]
    off, X, x, R, t = make_pnp_batch(11, sizes, outlier_frac=0.15)
    g = ctx.epnp_ransac(off, X, x, 4800.0)
    o = O.epnp_ransac(off, X, x, 4800.0)
    equal(g, o)
    Rg, tg, err, avg, best = g
    assert avg[1] == 10000.0 and avg[2] == 10000.0 and best[1] == -1 and best[2] == -1
    for p in (0, 5, 6, 7):   # the kept minimal sample explains the inliers to a few pixels
        assert avg[p] < 5.0 and np.abs(Rg[p] - R[p]).max() < 5e-3
        assert (err[off[p]:off[p + 1]] < 10).mean() > 0.8


@pytest.mark.parametrize("seed,iters", [(1, 200), (2, 37), (3, 1000)])
def test_epnp_matches_oracle_options(ctx, O, seed, iters):
    off, X, x, _, _ = make_pnp_batch(20 + seed, [120] * 5, outlier_frac=0.3, noise=1.0)
    f = np.array([4800.0, 4000.0, 5200.0, 4800.0, 3000.0])
    equal(ctx.epnp_ransac(off, X, x, f, max_iter=iters, seed=99 + seed), O.epnp_ransac(off, X, x, f, max_iter=iters, seed=99 + seed))


def test_epnp_degenerate_inputs(ctx, O):
    rng = np.random.default_rng(5)
    off = np.array([0, 30, 60, 90], np.int32)
    X = np.concatenate([np.column_stack([rng.uniform(-40, 40, 30), rng.uniform(-30, 30, 30), np.zeros(30)]),   # coplanar
                        np.tile([[1.0, 2.0, 3.0]], (30, 1)),                                                     # one point
                        np.column_stack([np.arange(30.0), 2 * np.arange(30.0), 3 * np.arange(30.0)])])           # collinear
    x = rng.uniform(-1000, 1000, (90, 2))
    equal(ctx.epnp_ransac(off, X, x, 4800.0, max_iter=64), O.epnp_ransac(off, X, x, 4800.0, max_iter=64))


def test_epnp_is_independent_of_the_batch_split(ctx):
    off, X, x, _, _ = make_pnp_batch(31, [80, 90, 100], outlier_frac=0.2)
    whole = ctx.epnp_ransac(off, X, x, 4800.0)
    # sample `it` of image p depends on (seed, p, it): image 0 alone must reproduce its row of the batch
    one = ctx.epnp_ransac(off[:2], X[:off[1]], x[:off[1]], 4800.0)
    np.testing.assert_array_equal(whole[0][0], one[0][0])
    np.testing.assert_array_equal(whole[2][:off[1]], one[2])


def test_relpose_matches_oracle_mixed_batch(ctx, O):
    sizes = [300, 0, 4, 5, 7, 9, 10, 60, 2000]
    off, a, b, R, t = make_relpose_batch(12, sizes, outlier_frac=0.0)   # verified matches, as the reference feeds this stage
    g = ctx.relpose_5pt(off, a, b, 4800.0, 4800.0)
    o = O.relpose_5pt(off, a, b, 4800.0, 4800.0)
    equal(g, o)
    E, Rg, tg, ok, nc = g
    assert ok[1] == 0 and ok[2] == 0 and not E[1].any() and nc[1] == 0
    for p in (0, 7, 8):
        assert ok[p] == 1 and np.abs(Rg[p] - R[p]).max() < 2e-2
        x1 = np.c_[a[off[p]:off[p + 1]] / 4800.0, np.ones(sizes[p])]
        x2 = np.c_[b[off[p]:off[p + 1]] / 4800.0, np.ones(sizes[p])]
        r = np.abs(np.einsum("ni,ij,nj->n", x2, E[p] / np.linalg.norm(E[p]), x1))
        assert np.median(r) < 1e-3


@pytest.mark.parametrize("seed,times", [(1, 100), (2, 13), (3, 400)])
def test_relpose_matches_oracle_options(ctx, O, seed, times):
    off, a, b, _, _ = make_relpose_batch(40 + seed, [150] * 4, outlier_frac=0.3, noise=1.0)
    f1 = np.array([4800.0, 4000.0, 5200.0, 2500.0])
    f2 = np.array([4800.0, 4100.0, 4800.0, 2500.0])
    equal(ctx.relpose_5pt(off, a, b, f1, f2, ransac_times=times, seed=5 + seed), O.relpose_5pt(off, a, b, f1, f2, ransac_times=times, seed=5 + seed))


def test_relpose_degenerate_inputs(ctx, O):
    rng = np.random.default_rng(6)
    off = np.array([0, 40, 80], np.int32)
    same = np.tile([[100.0, 50.0]], (40, 1))
    line = np.column_stack([np.arange(40.0) * 10, np.arange(40.0) * 20])
    a = np.concatenate([same, line])
    b = np.concatenate([same, line + 3.0])
    equal(ctx.relpose_5pt(off, a, b, 4800.0, 4800.0, ransac_times=32), O.relpose_5pt(off, a, b, 4800.0, 4800.0, ransac_times=32))


def test_pose_golden_fixture(ctx):
    z = np.load(GOLD)
    g = ctx.epnp_ransac(z["pnp_off"], z["pnp_X"], z["pnp_x"], z["pnp_f"], max_iter=int(z["pnp_iters"]), seed=int(z["seed"]))
    for a, k in zip(g, ("pnp_R", "pnp_t", "pnp_err", "pnp_avg", "pnp_best")):
        np.testing.assert_array_equal(a, z[k])
    g = ctx.relpose_5pt(z["rel_off"], z["rel_a"], z["rel_b"], z["rel_f1"], z["rel_f2"], ransac_times=int(z["rel_times"]), seed=int(z["seed"]))
    for a, k in zip(g, ("rel_E", "rel_R", "rel_t", "rel_ok", "rel_nc")):
        np.testing.assert_array_equal(a, z[k])
