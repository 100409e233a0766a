"""Latency of the FLANN-shaped one-shot call msfm_knn2_f32 (descriptors in host memory, one pair per call) against the
batched resident form (msfm_descset_* + msfm_match_pairs) on the same data."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from metricsfm_amd import capi, scene
ctx = capi.Context(0)
rng = np.random.default_rng(1)
d = [scene._sift_like(np.random.default_rng(i), 4096).astype(np.float32) for i in range(8)]
ctx.knn2(d[0], d[1])                      # warm-up (code objects, pool)
t0 = time.perf_counter()
n = 0
for i in range(8):
    for j in range(8):
        if i != j:
            ctx.knn2(d[i], d[j]); n += 1
dt = time.perf_counter() - t0
print("one-shot msfm_knn2_f32: %d pairs, %.3f ms per pair, %.1f Mmatches/s" % (n, 1e3 * dt / n, 1e-6 * n * 4096 / dt))
ds = ctx.descset(d)
pairs = scene.all_pairs(8)
res = ds.match_pairs(pairs); ctx.synchronize()
t0 = time.perf_counter(); res = ds.match_pairs(pairs); ctx.synchronize(); dt2 = time.perf_counter() - t0
print("batched msfm_match_pairs on resident descriptors: %d pairs, %.3f ms per pair, %.1f Mmatches/s" % (len(pairs), 1e3 * dt2 / len(pairs), 1e-6 * len(pairs) * 4096 / dt2))
