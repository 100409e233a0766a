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
out = (C.c_ulonglong * 8)()
for keep in (False, True):
    L.msfm_dbg_f16_stats(out, 1)
    res = ds.match_pairs(scene.all_pairs(n), 0.6, 0.85, keep_knn=keep)
    ctx.synchronize()
    L.msfm_dbg_f16_stats(out, 0)
    q, c, e, r, e2, r2 = [int(x) for x in out][:6]
    print("keep_knn=%d: queries %d, decided from f16 intervals %.2f %%, candidates filed %.3f per query (%.2f rounds per wave); "
          "after the binary32 pass: exact evaluations %.4f per query, %.3f rounds per wave; slow path %d"
          % (keep, q, 100.0 * c / q, e / q, r / (q / 64.0), (e2 if not keep else e) / q, (r2 if not keep else r) / (q / 64.0), res.stats()["slow_path"]))
