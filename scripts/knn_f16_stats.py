"""Developer tool: epilogue counters of the float matcher (library built with -DMSFM_KNN_F16_STATS, MSFM_LIB pointing at it):
valid queries, queries whose code came from certified intervals, exact evaluations, 64-entry list rounds per wave."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from metricsfm_amd import capi, scene
sc = scene.config_scene(3)
n = 16
scene.add_features(sc, 4096, images=range(n))
descs = [(512.0 * d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32) for d in sc.desc[:n]]
ctx = capi.Context(0)
ds = ctx.descset(descs)
L = capi.lib()
out = (C.c_ulonglong * 4)()
for keep in (False, True):
    L.msfm_dbg_f16_stats(out, 1)
    res = ds.match_pairs(scene.all_pairs(n), 0.6, 0.85, keep_knn=keep)
    ctx.synchronize()
    L.msfm_dbg_f16_stats(out, 0)
    q, c, e, r = [int(x) for x in out]
    print("keep_knn=%d: queries %d, decided from intervals %.2f %%, exact evaluations %.3f per query, list rounds per wave %.2f, slow path %d"
          % (keep, q, 100.0 * c / q, e / q, r / (q / 64.0), res.stats()["slow_path"]))
