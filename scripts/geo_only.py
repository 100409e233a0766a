"""Verification stage of the resident chain alone (96 images of config 3, all ordered pairs), with the per-class kernel
timers: what msfm_chain_verify spends where.  gpurun -- 'python scripts/geo_only.py [n_images]'"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metricsfm_amd import capi, scene

n_ci = int(sys.argv[1]) if len(sys.argv) > 1 else 96
ctx = capi.Context(0)
sc = scene.config_scene(3)
scene.add_features(sc, 4096, images=list(range(n_ci)))
descs = [sc.desc[i] for i in range(n_ci)]
kps = [np.ascontiguousarray(sc.kp_xy[i], np.float32) for i in range(n_ci)]
cds = ctx.descset(descs, keypoints=kps)
cpairs = scene.all_pairs(n_ci)
cres = cds.match_pairs(cpairs, 0.6, 0.85)
ctx.synchronize()
for rep in range(3):
    prof = rep == 2
    if prof:
        ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter()
    ch = capi.Chain(cres)
    n_m, okc, _ = ch.verify(3.0)
    ctx.synchronize()
    t1 = time.perf_counter()
    print("verify %.3f ms  pairs_ok %d  matches %d" % (1e3 * (t1 - t0), int(okc.sum()), int(n_m.sum())), flush=True)
    if prof:
        st = ctx.profile_get(); ctx.profile(False)
        for k, v in sorted(st.items(), key=lambda kv: -kv[1]["total_ms"]):
            print("  %-28s %4d launches %9.3f ms" % (k, v["launches"], v["total_ms"]))
    ch.close()
