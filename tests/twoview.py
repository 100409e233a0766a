"""Synthetic two-view correspondences for the geometric-verification tests (test helper, not product)."""
import numpy as np


def rodrigues(a):
    th = np.linalg.norm(a)
    if th < 1e-12:
        return np.eye(3)
    k = a / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def make_pair(rng, n, outlier_frac=0.3, noise=0.5, f=4800.0):
    """n matches between two pinhole views (centred pixels, +z forward); returns pt1, pt2 (float32), true-inlier mask, F_true."""
    X = np.column_stack([rng.uniform(-40, 40, n), rng.uniform(-30, 30, n), rng.uniform(80, 120, n)])
    R = rodrigues(rng.normal(0, 0.05, 3))
    t = np.array([10.0, 1.0, 0.5]) + rng.normal(0, 0.5, 3)
    x1 = f * X[:, :2] / X[:, 2:3]
    Xc = X @ R.T + t
    x2 = f * Xc[:, :2] / Xc[:, 2:3]
    x1 = x1 + rng.normal(0, noise, x1.shape)
    x2 = x2 + rng.normal(0, noise, x2.shape)
    good = np.ones(n, bool)
    nout = int(round(outlier_frac * n))
    if nout:
        bad = rng.choice(n, nout, replace=False)
        x2[bad] = np.column_stack([rng.uniform(-2000, 2000, nout), rng.uniform(-1500, 1500, nout)])
        good[bad] = False
    K = np.diag([f, f, 1.0])
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ft = np.linalg.inv(K).T @ tx @ R @ np.linalg.inv(K)
    return x1.astype(np.float32), x2.astype(np.float32), good, Ft


def make_batch(seed, sizes, outlier_frac=0.3, noise=0.5):
    rng = np.random.default_rng(seed)
    p1, p2, good, off = [], [], [], [0]
    for n in sizes:
        a, b, g, _ = make_pair(rng, n, outlier_frac, noise) if n > 0 else (np.zeros((0, 2), np.float32),) * 2 + (np.zeros(0, bool), None)
        p1.append(a); p2.append(b); good.append(g); off.append(off[-1] + n)
    return np.array(off, np.int32), np.concatenate(p1), np.concatenate(p2), np.concatenate(good)


def make_pnp(rng, n, outlier_frac=0.1, noise=0.5, f=4800.0):
    """n 2D-3D correspondences of one camera (Xc = R Xw + t, centred pixels); returns Xw, x2d, R, t."""
    X = np.column_stack([rng.uniform(-40, 40, n), rng.uniform(-30, 30, n), rng.uniform(-5, 5, n)])
    R = rodrigues(np.array([np.pi, 0.0, 0.0]) + rng.normal(0, 0.05, 3))   # nadir view from above
    c = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), 100.0])
    t = -R @ c
    Xc = X @ R.T + t
    x = f * Xc[:, :2] / Xc[:, 2:3] + rng.normal(0, noise, (n, 2))
    nout = int(round(outlier_frac * n))
    if nout:
        bad = rng.choice(n, nout, replace=False)
        x[bad] = np.column_stack([rng.uniform(-2000, 2000, nout), rng.uniform(-1500, 1500, nout)])
    return X, x, R, t


def make_pnp_batch(seed, sizes, outlier_frac=0.1, noise=0.5, f=4800.0):
    rng = np.random.default_rng(seed)
    Xs, xs, Rs, ts, off = [], [], [], [], [0]
    for n in sizes:
        X, x, R, t = make_pnp(rng, n, outlier_frac, noise, f)
        Xs.append(X); xs.append(x); Rs.append(R); ts.append(t); off.append(off[-1] + n)
    return np.array(off, np.int32), np.concatenate(Xs), np.concatenate(xs), np.array(Rs), np.array(ts)


def make_relpose_batch(seed, sizes, outlier_frac=0.1, noise=0.5, f=4800.0):
    """Matches of calibrated pairs as float64 centred pixels; returns off, x_ref, x_cur, R [n], t [n] (x_cur ~ R X + t)."""
    rng = np.random.default_rng(seed)
    a, b, Rs, ts, off = [], [], [], [], [0]
    for n in sizes:
        X = np.column_stack([rng.uniform(-40, 40, n), rng.uniform(-30, 30, n), rng.uniform(80, 120, n)])
        R = rodrigues(rng.normal(0, 0.05, 3))
        t = np.array([10.0, 1.0, 0.5]) + rng.normal(0, 0.5, 3)
        x1 = f * X[:, :2] / X[:, 2:3] + rng.normal(0, noise, (n, 2))
        Xc = X @ R.T + t
        x2 = f * Xc[:, :2] / Xc[:, 2:3] + rng.normal(0, noise, (n, 2))
        nout = int(round(outlier_frac * n))
        if nout:
            bad = rng.choice(n, nout, replace=False)
            x2[bad] = np.column_stack([rng.uniform(-2000, 2000, nout), rng.uniform(-1500, 1500, nout)])
        a.append(x1); b.append(x2); Rs.append(R); ts.append(t); off.append(off[-1] + n)
    return np.array(off, np.int32), np.concatenate(a), np.concatenate(b), np.array(Rs), np.array(ts)
