#!/usr/bin/env python3
"""Side measurements that are not the headline bench line: the non-integral descriptor path and the
largest BASELINE configuration (config 5: 2000 cameras / 1M points / 6M observations + GPS)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metricsfm_amd import _abi as A, capi, scene  # noqa: E402

ctx = capi.Context(0)
out = {}
# ---- general float descriptors (VLFeat 512*x floats): certified f16 path ----
rng = np.random.default_rng(0)
n_float_images = 24   # 552 ordered pairs: enough workgroups to fill the chip (12 pairs leave it two thirds idle)
d = [(rng.gamma(0.6, 1.0, (4096, 128)) * 40).astype(np.float32) for _ in range(n_float_images)]
ds = ctx.descset(d)
pairs = scene.all_pairs(n_float_images)
res = ds.match_pairs(pairs)
ctx.synchronize()
t0 = time.perf_counter()
res.rerun()
ctx.synchronize()
dt = time.perf_counter() - t0
out["float_descriptors"] = dict(pairs=len(pairs), ms=1e3 * dt, Mmatches_per_s=1e-6 * len(pairs) * 4096 / dt,
                                path="MSFM_KNN_EXACT=%s" % os.environ.get("MSFM_KNN_EXACT", "0"))
ctx.profile(True); ctx.profile_reset(); res.rerun(); ctx.synchronize(); out["float_kernels"] = ctx.profile_get(); ctx.profile(False)
out["float_stats"] = res.stats()
# ---- track building at config 3 scale: every ordered pair's matches of the 500-camera scene (+ 2 % wrong ones) ----
if "--no-tracks" not in sys.argv:
    from metricsfm_amd import tracks as T
    sc3 = scene.config_scene(3)
    flat = T.flat_matches_from_scene(sc3, wrong=0.02, seed=5)
    ctx.build_tracks(None, None, None, flat=flat)   # (first call: pool allocations)
    t0 = time.perf_counter()
    got = ctx.build_tracks(None, None, None, flat=flat)
    dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    host = capi.build_tracks_flat(*flat)
    hst = time.perf_counter() - t0
    same = all(np.array_equal(g, h) for g, h in zip(got, host))
    out["tracks_config3"] = dict(matches=int(len(flat[3])), pairs=int(len(flat[1])), tracks=int(len(got[0]) - 1), observations=int(len(got[1])),
                                 device_ms=1e3 * dev, host_walk_ms=1e3 * hst, identical=bool(same),
                                 Mmatches_per_s_device=1e-6 * len(flat[3]) / dev, Mmatches_per_s_host=1e-6 * len(flat[3]) / hst,
                                 note="msfm_tracks_build_device incl. upload of the match lists and download of the CSR tracks, "
                                      "vs msfm_tracks_build (one host thread, the reference's walk)")
    del sc3, flat, got, host
# ---- config 5 ----
if "--c5" in sys.argv:
    t0 = time.time()
    sc = scene.config_scene(5)
    gen = time.time() - t0
    arr = A.BaArrays.from_scene(sc, gps_xyz=sc.gps_xyz, gps_weight=float(sc.n_obs // sc.n_cams))
    t0 = time.time()
    ba = ctx.ba(arr)
    setup = time.time() - t0
    r = ba.run(capi.default_options(max_num_iterations=5, function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0))
    out["config5"] = dict(cams=sc.n_cams, points=sc.n_points, obs=sc.n_obs, reduced_order=r["num_reduced_params"], scene_gen_s=gen,
                          create_s=setup, solve_ms=r["solve_ms"], iterations=r["num_iterations"], ms_per_iteration=r["solve_ms"] / 5,
                          cost=[float(c) for c in r["iterations"]["cost"]])
print(json.dumps(out))
