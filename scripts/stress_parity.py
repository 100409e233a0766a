#!/usr/bin/env python3
"""Randomised parity sweep (GPU path through the C ABI against the CPU oracle) beyond the fixed cases of tests/:
many seeds, ragged and degenerate batches.  Exact comparisons for the RANSAC / pose / matching legs, the tolerances of
the parity gates for bundle adjustment.  usage: stress_parity.py [n_rounds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from metricsfm_amd import _abi as A, capi, scene  # noqa: E402
from oracle import oracle as O  # noqa: E402
from twoview import make_batch, make_pnp_batch, make_relpose_batch  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ctx = capi.Context(0)
rng = np.random.default_rng(2026)
bad = 0
n_masked = n_masked_skipped = 0
t_start = time.time()


def same(name, g, o, seed):
    global bad
    for k, (a, b) in enumerate(zip(g, o)):
        if not np.array_equal(a, b, equal_nan=True):
            bad += 1
            neq = ~((a == b) | (np.isnan(a) & np.isnan(b))) if a.dtype.kind == "f" else (a != b)
            print("MISMATCH %s output %d seed %d: %d entries differ, first at %s" % (name, k, seed, int(neq.sum()), np.argwhere(neq)[:1].tolist()))
            return


for r in range(rounds):
    seed = int(rng.integers(1, 1 << 30))
    sizes = [int(x) for x in rng.choice([0, 3, 4, 5, 6, 9, 10, 17, 40, 130, 700, 1500], size=rng.integers(1, 7))]
    of = float(rng.choice([0.0, 0.1, 0.4, 0.7]))
    noise = float(rng.choice([0.0, 0.5, 3.0]))
    # absolute pose
    off, X, x, _, _ = make_pnp_batch(seed, sizes, outlier_frac=of, noise=noise)
    if r % 5 == 0 and len(X):   # degenerate geometry: all points on one plane / one line
        X = X.copy()
        X[:, 2] = 0.0
        if r % 10 == 0:
            X[:, 1] = 2 * X[:, 0]
    f = rng.choice([800.0, 4800.0, 12000.0], size=len(sizes))
    it = int(rng.choice([1, 17, 64, 200]))
    same("epnp", ctx.epnp_ransac(off, X, x, f, max_iter=it, seed=seed), O.epnp_ransac(off, X, x, f, max_iter=it, seed=seed), seed)
    # relative pose
    off, a, b, _, _ = make_relpose_batch(seed + 1, sizes, outlier_frac=of, noise=noise)
    if r % 7 == 0 and len(a):
        b = a + 5.0     # pure image translation: degenerate for the essential matrix
    tm = int(rng.choice([1, 9, 100]))
    f2 = f[::-1].copy()
    same("relpose", ctx.relpose_5pt(off, a, b, f, f2, ransac_times=tm, seed=seed), O.relpose_5pt(off, a, b, f, f2, ransac_times=tm, seed=seed), seed)
    # fundamental-matrix RANSAC (float32 points)
    sz = [s for s in sizes if s != 1500] or [40]
    off, p1, p2, _ = make_batch(seed + 2, sz, outlier_frac=of, noise=max(noise, 0.1))
    mi = int(rng.choice([50, 500, 2000]))
    same("fransac", ctx.fundamental_ransac(off, p1, p2, seed=seed, max_iterations=mi), O.fundamental_ransac(off, p1, p2, seed=seed, max_iterations=mi), seed)
    # matching on integer descriptors
    n1, n2 = int(rng.integers(2, 700)), int(rng.integers(1, 700))
    d1 = scene._sift_like(np.random.default_rng(seed), n1).astype(np.float32)
    d2 = scene._sift_like(np.random.default_rng(seed + 9), n2).astype(np.float32)
    if r % 4 == 0:
        h = min(n2, n1) // 2
        d2[:h] = d1[:h]   # planted duplicates: ties by index
    same("knn2", ctx.knn2(d1, d2), O.knn2(d1, d2, fast=True), seed)
    # matching on float descriptors (round 5): the codes-only form decides most codes from certified distance intervals, the
    # keep-arrays form evaluates every candidate - both against the oracle's binary64 definition.  Random magnitudes, near
    # duplicates (ratios near 0 and 0 / 0), second-nearest rows planted near the two thresholds, random threshold pairs.
    fr = np.random.default_rng(seed + 17)
    m1, m2 = int(fr.integers(2, 900)), int(fr.integers(1, 600))
    if r % 6 == 5:
        m1, m2 = int(fr.integers(900, 5000)), int(fr.integers(300, 1500))   # several 256-row windows, more than one workgroup of queries
    mag = float(fr.choice([2.0 ** -9, 1.0, 512.0, 3.0e4]))
    f1 = d1[fr.integers(0, n1, m1)] + fr.normal(0, 0.3, (m1, 128))
    f2 = d2[fr.integers(0, n2, m2)] + fr.normal(0, 0.3, (m2, 128))
    f1 = (mag * f1 / np.maximum(1e-9, np.linalg.norm(f1, axis=1, keepdims=True))).astype(np.float32)
    f2 = (mag * f2 / np.maximum(1e-9, np.linalg.norm(f2, axis=1, keepdims=True))).astype(np.float32)
    for k in range(min(m2, m1 - 1, 24)):          # query k near train row k, a planted second row at ratio ~ th
        th = (0.6, 0.85)[k & 1]
        u = fr.standard_normal(128); u *= 0.04 * mag / np.linalg.norm(u)
        w = fr.standard_normal(128); w -= w.dot(u) / u.dot(u) * u; w /= np.linalg.norm(w)
        q = f1[k].astype(np.float64) + u
        f2[k] = q.astype(np.float32)
        f1[m1 - 1 - k] = (q + np.sqrt(u.dot(u) / th * (1 + (k - 12) * 2e-8)) * w).astype(np.float32)
    if r % 3 == 0 and m2 > 30:
        f2[25:25 + min(5, m1)] = f1[:min(5, m1)]    # exact duplicates
    rg_, ra_ = [(0.6, 0.85), (0.99, 0.9), (0.5, 0.5)][r % 3]
    ds = ctx.descset([f1, f2])
    pr = np.array([[0, 1]], np.int32)
    ids_o, d_o = O.knn2(f1, f2)
    code_o, na_o, ng_o = O.ratio_codes(ids_o, d_o, rg_, ra_)
    for keep in (False, True):
        res = ds.match_pairs(pr, rg_, ra_, keep_knn=keep)
        code, ids, dist = res.fetch(0)
        na, ng = res.counts()
        outs_g, outs_o = [code, np.array([na[0], ng[0]])], [code_o, np.array([na_o, ng_o])]
        if keep:
            outs_g += [ids, dist]; outs_o += [ids_o, d_o]
        same("float match keep=%d mag=%g" % (keep, mag), outs_g, outs_o, seed)
        res.close()
    ds.close()
    # bundle adjustment on a small random scene
    if r % 4 == 0:
        sc = scene.make_ring_scene(int(rng.integers(4, 14)), int(rng.integers(50, 400)), seed=seed)
        g, o = A.BaArrays.from_scene(sc), A.BaArrays.from_scene(sc)
        rg = ctx.ba_solve(g, capi.default_options(max_num_iterations=12))
        ro = O.ba_solve(o, O.default_options(max_num_iterations=12))
        ok = (rg["num_iterations"] == ro["num_iterations"] and abs(rg["final_cost"] - ro["final_cost"]) <= 1e-9 * abs(ro["final_cost"])
              and np.abs(g.cam_pose - o.cam_pose).max() <= 1e-5 * np.abs(o.cam_pose).max())
        if not ok:
            bad += 1
            print("MISMATCH ba seed %d: iterations %d / %d, cost %.12e / %.12e" % (seed, rg["num_iterations"], ro["num_iterations"], rg["final_cost"], ro["final_cost"]))
    # bundle adjustment with random windows (frozen cameras / points / intrinsics), several intrinsics blocks, GPS rows
    if r % 8 == 3:
        nm = int(rng.choice([1, 1, 2, 3]))
        gps = bool(rng.random() < 0.3)
        try:
            sc = scene.make_aerial_scene(int(rng.integers(16, 40)), int(rng.integers(300, 2500)), seed=seed, n_models=nm,
                                         gps_sigma=0.5 if gps else None, rot_sigma=0.02, trans_sigma=0.2, point_sigma=0.2)
        except RuntimeError:   # the synthetic-scene generator could not place every point for this camera count: not a solver case
            n_masked_skipped += 1
            continue
        n_masked += 1
        mrng = np.random.default_rng(seed + 5)
        kw = {}
        if mrng.random() < 0.7:
            cm = (mrng.random(sc.n_cams) > 0.3).astype(np.uint8)
            cm[int(mrng.integers(sc.n_cams))] = 1
            kw["cam_mutable"] = cm
        if mrng.random() < 0.7:
            kw["pt_mutable"] = (mrng.random(sc.n_points) > float(mrng.choice([0.1, 0.5, 1.0]))).astype(np.uint8)
        if mrng.random() < 0.3:
            kw["model_mutable"] = (mrng.random(nm) > 0.5).astype(np.uint8)
        if gps:
            kw["gps_xyz"] = sc.gps_xyz
            kw["gps_weight"] = float(sc.n_obs // sc.n_cams)
        g, o = A.BaArrays.from_scene(sc, **kw), A.BaArrays.from_scene(sc, **kw)
        rg = ctx.ba_solve(g, capi.default_options(max_num_iterations=15))
        ro = O.ba_solve(o, O.default_options(max_num_iterations=15))
        scale = max(np.abs(o.cam_pose).max(), 1.0)
        ok = (rg["num_iterations"] == ro["num_iterations"] and abs(rg["final_cost"] - ro["final_cost"]) <= 1e-8 * max(abs(ro["final_cost"]), 1e-30)
              and np.abs(g.cam_pose - o.cam_pose).max() <= 1e-5 * scale and np.abs(g.point - o.point).max() <= 1e-5 * max(np.abs(o.point).max(), 1.0)
              and (rg["iterations"]["step_is_successful"] == ro["iterations"]["step_is_successful"]).all())
        if not ok:
            bad += 1
            print("MISMATCH masked ba seed %d (%s): iterations %d / %d, cost %.12e / %.12e, dpose %.2e" % (
                seed, sorted(kw), rg["num_iterations"], ro["num_iterations"], rg["final_cost"], ro["final_cost"], np.abs(g.cam_pose - o.cam_pose).max()))
    if r % 10 == 9:
        print("round %d done, %.0f s, %d mismatches" % (r + 1, time.time() - t_start, bad), flush=True)
print("masked / GPS / multi-model bundle adjustments compared: %d (%d scenes the generator could not build were skipped)" % (n_masked, n_masked_skipped))
print("stress parity: %d rounds, %d mismatches" % (rounds, bad))
sys.exit(1 if bad else 0)
