import os, sys, ctypes as C
sys.path.insert(0, "/root/repo")
os.environ["MSFM_LIB"] = "/root/repo/metricsfm_amd/libmsfm_hs.so"
import numpy as np
from metricsfm_amd import capi, scene
sc = scene.config_scene(3); n = 32
scene.add_features(sc, 4096, images=range(n))
ctx = capi.Context(0)
ds = ctx.descset([sc.desc[i] for i in range(n)])
L = capi.lib()
buf = (C.c_ulonglong * 64)()
res = ds.match_pairs(scene.all_pairs(n), 0.6, 0.85); ctx.synchronize()
L.msfm_dbg_knn_hits(buf, 1)
res.rerun(); ctx.synchronize()
L.msfm_dbg_knn_hits(buf, 0)
h = np.array(list(buf), dtype=np.float64).reshape(8, 8)[:, :5]
print("groups of four register slots (a + b sets, 64 lanes) by the number of slots with a candidate under the threshold:")
print("train rows      0      1      2      3      4   mean slots hit of 4")
for t in range(8):
    row = h[t] / max(1.0, h[t].sum())
    print("%4d-%4d  " % (t * 512, t * 512 + 511) + " ".join("%6.3f" % v for v in row) + "   %.3f" % (row * np.arange(5)).sum())
tot = h.sum(0) / h.sum()
print("all        " + " ".join("%6.3f" % v for v in tot) + "   %.3f  (of 16 per 32-row step: %.2f)" % ((tot * np.arange(5)).sum(), 4 * (tot * np.arange(5)).sum()))
