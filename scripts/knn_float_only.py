"""Runs only the float-descriptor matching leg a few times (for counter collection): 24 images of 4096 non-integral
descriptors, 552 ordered pairs."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from metricsfm_amd import capi, scene
ctx = capi.Context(0)
rng = np.random.default_rng(0)
d = [(rng.gamma(0.6, 1.0, (4096, 128)) * 40).astype(np.float32) for _ in range(24)]
ds = capi.DescSet(ctx, d)
pairs = scene.all_pairs(24)
res = ds.match_pairs(pairs)
for _ in range(3):
    ctx.synchronize()
    t0 = time.perf_counter()
    res.rerun()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print("pairs %d: %.2f ms, %.1f Mmatches/s, slow path %d" % (len(pairs), dt * 1e3, len(pairs) * 4096 / dt * 1e-6, res.stats()["slow_path"]))
