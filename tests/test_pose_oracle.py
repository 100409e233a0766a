"""The CPU oracle of the pose initialisers (oracle/pose_oracle.cpp) against numpy and against ground truth:
the restated OpenCV Jacobi SVD and Eigen eigenvalue / LU pieces, the five-point solver on exact data, EPnP on exact
data, the two RANSAC drivers on noisy data with outliers, and the committed fixture."""
import os

import numpy as np
import pytest

from tests.twoview import make_pnp_batch, make_relpose_batch, rodrigues

GOLD = os.path.join(os.path.dirname(__file__), "golden", "pose_small.npz")


@pytest.fixture(scope="module")
def O(oracle):
    return oracle


@pytest.mark.parametrize("m,n", [(3, 3), (6, 3), (6, 4), (6, 5), (9, 9), (12, 12)])
def test_jacobi_svd_against_numpy(O, m, n):
    rng = np.random.default_rng(m * 16 + n)
    a = rng.normal(size=(m, n))
    if m == 12:
        b = rng.normal(size=(8, 12))
        a = b.T @ b          # the rank-8 M^T M of a 4-point EPnP: four singular values at rounding level
    W, Ut, Vt = O._test_jacobi_svd(a)
    np.testing.assert_allclose(W, np.linalg.svd(a, compute_uv=False), atol=1e-13 * max(1.0, W[0]))
    assert (np.diff(W) <= 0).all()
    np.testing.assert_allclose((Ut.T * W) @ Vt, a, atol=1e-13 * max(1.0, W[0]))
    np.testing.assert_allclose(Ut @ Ut.T, np.eye(n), atol=1e-12)
    np.testing.assert_allclose(Vt @ Vt.T, np.eye(n), atol=1e-12)


def test_real_eigenvalues_against_numpy(O):
    rng = np.random.default_rng(3)
    for _ in range(20):
        a = rng.normal(size=(10, 10))
        ok, wr, wi = O._test_eig10(a)
        assert ok
        ev = np.linalg.eigvals(a)
        np.testing.assert_allclose(np.sort_complex(wr + 1j * wi), np.sort_complex(ev), atol=1e-11)
        assert ((wi == 0) == (np.abs(wi) < 1e-300)).all()


def _two_view(rng, n):
    R = rodrigues(np.array([0.05, -0.1, 0.03]))
    t = np.array([1.0, 0.1, -0.05])
    X = np.column_stack([rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(4, 8, n)])
    x1 = X[:, :2] / X[:, 2:]
    Xc = X @ R.T + t
    x2 = Xc[:, :2] / Xc[:, 2:]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    return x1, x2, R, t, tx @ R


@pytest.mark.parametrize("n", [5, 6, 8, 9])
def test_five_point_solutions_are_essential_and_contain_the_truth(O, n):
    x1, x2, R, t, Et = _two_view(np.random.default_rng(1), n)
    Es = O._test_five_point(x1, x2)
    assert 1 <= len(Es) <= 10
    h1, h2 = np.c_[x1, np.ones(n)], np.c_[x2, np.ones(n)]
    best = 1.0
    for E in Es:
        E = E / np.linalg.norm(E)
        sv = np.linalg.svd(E, compute_uv=False)
        assert abs(sv[0] - sv[1]) < 1e-9 and sv[2] < 1e-9                      # two equal singular values, one zero
        if n == 5:   # minimal: every solution puts all five matches on their epipolar lines; with more matches the
            #          four smallest right singular vectors span more than the null space and only the truth does
            assert np.abs(np.einsum("ni,ij,nj->n", h2, E, h1)).max() < 1e-9
        Etn = Et / np.linalg.norm(Et)
        best = min(best, np.abs(E - Etn).max(), np.abs(E + Etn).max())
    assert best < 1e-9


def test_epnp_recovers_an_exact_pose(O):
    rng = np.random.default_rng(2)
    R = rodrigues(np.array([0.2, -0.1, 0.3]))
    t = np.array([0.5, -0.3, 2.0])
    for n in (5, 6, 8, 20):
        for _ in range(10):
            X = np.column_stack([rng.uniform(-20, 20, n), rng.uniform(-15, 15, n), rng.uniform(40, 80, n)])
            Xc = X @ R.T + t
            x = 4800.0 * Xc[:, :2] / Xc[:, 2:]
            Rr, tr, e = O._test_epnp_n(X, x, 4800.0)
            assert e < 1e-8
            np.testing.assert_allclose(Rr, R, atol=1e-9)
            np.testing.assert_allclose(tr, t, atol=1e-7)


def test_epnp_on_four_points_is_what_the_reference_runs(O):
    """EPNPRansac feeds EPnP four correspondences (absolute_pose_via_epnp.cc:113): M^T M then has a four-dimensional null
    space while the solver only seeds one to three of its vectors, so most samples end in a poor local minimum and the
    RANSAC keeps the rare sample that does not.  Whatever comes back is a rotation, and the error is that of the pose."""
    rng = np.random.default_rng(2)
    R = rodrigues(np.array([0.2, -0.1, 0.3]))
    t = np.array([0.5, -0.3, 2.0])
    good = 0
    for _ in range(100):
        X = np.column_stack([rng.uniform(-20, 20, 4), rng.uniform(-15, 15, 4), rng.uniform(40, 80, 4)])
        Xc = X @ R.T + t
        x = 4800.0 * Xc[:, :2] / Xc[:, 2:]
        Rr, tr, e = O._test_epnp4(X, x, 4800.0)
        np.testing.assert_allclose(Rr @ Rr.T, np.eye(3), atol=1e-9)
        Xr = X @ Rr.T + tr
        d = np.linalg.norm(4800.0 * Xr[:, :2] / Xr[:, 2:] - x, axis=1)
        if (d < 10).sum() >= 2:
            assert abs(e - np.sqrt((d[d < 10] ** 2).mean())) < 1e-6 * max(1.0, e)
        else:
            assert e == 100000.0
        good += e < 1e-3
    assert good >= 1


def test_epnp_ransac_on_noisy_data(O):
    off, X, x, R, t = make_pnp_batch(11, [300, 3, 40], outlier_frac=0.15)
    Rr, tr, err, avg, best = O.epnp_ransac(off, X, x, 4800.0)
    assert avg[1] == 10000.0 and best[1] == -1 and not Rr[1].any() and (err[off[1]:off[2]] == 1000.0).all()
    for p in (0, 2):
        assert 0 <= best[p] < 200 and avg[p] < 5.0
        assert np.abs(Rr[p] - R[p]).max() < 5e-3 and np.abs(tr[p] - t[p]).max() < 1.0
        e = err[off[p]:off[p + 1]]
        assert (e < 10).mean() > 0.8 and ((e < 10) | (e == 1000.0)).all()
    # fewer iterations = a prefix of the same sample sequence: the kept sample can only be an earlier one
    _, _, _, avg50, best50 = O.epnp_ransac(off, X, x, 4800.0, max_iter=50)
    assert best50[0] <= best[0] if best[0] < 50 else best50[0] < 50


def test_relpose_on_noisy_data(O):
    sizes = [300, 4, 7, 60]
    # the Sampson sum over ALL matches is not robust: the reference feeds this stage RANSAC-verified matches only
    off, a, b, R, t = make_relpose_batch(12, sizes, outlier_frac=0.0)
    E, Rr, tr, ok, nc = O.relpose_5pt(off, a, b, 4800.0, 4800.0)
    assert ok[1] == 0 and nc[1] == 0 and not E[1].any() and not Rr[1].any()
    assert nc[2] <= 10                      # 5..9 matches: one solve on all of them
    for p in (0, 3):
        assert ok[p] == 1 and nc[p] >= 100
        np.testing.assert_allclose(Rr[p] @ Rr[p].T, np.eye(3), atol=1e-9)
        assert np.linalg.det(Rr[p]) > 0 and abs(np.linalg.norm(tr[p]) - 1) < 1e-9
        assert np.abs(Rr[p] - R[p]).max() < 2e-2
        # the reference returns t = -R^T u with u the left null vector of E: it is parallel to R^T t_true
        d = R[p].T @ t[p] / np.linalg.norm(t[p])
        assert abs(abs(d @ tr[p]) - 1) < 5e-2


def test_pose_fixture_is_current(O):
    z = np.load(GOLD)
    g = O.epnp_ransac(z["pnp_off"], z["pnp_X"], z["pnp_x"], z["pnp_f"], max_iter=int(z["pnp_iters"]), seed=int(z["seed"]))
    for a, k in zip(g, ("pnp_R", "pnp_t", "pnp_err", "pnp_avg", "pnp_best")):
        np.testing.assert_array_equal(a, z[k])
    g = O.relpose_5pt(z["rel_off"], z["rel_a"], z["rel_b"], z["rel_f1"], z["rel_f2"], ransac_times=int(z["rel_times"]), seed=int(z["seed"]))
    for a, k in zip(g, ("rel_E", "rel_R", "rel_t", "rel_ok", "rel_nc")):
        np.testing.assert_array_equal(a, z[k])
