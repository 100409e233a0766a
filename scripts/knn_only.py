"""Runs only the int8 matching leg of bench.py a few times (for counter collection)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from metricsfm_amd import capi, scene
ctx = capi.Context(0)
sc = scene.config_scene(2)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
scene.add_features(sc, 4096, images=list(range(N)))
ds = capi.DescSet(ctx, [sc.desc[i] for i in range(N)])
pairs = np.array([(i, j) for i in range(N) for j in range(N) if i != j], dtype=np.int32)
for _ in range(3):
    t0 = time.perf_counter()
    res = ds.match_pairs(pairs)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print("pairs %d: %.2f ms, %.1f Mmatches/s" % (len(pairs), dt * 1e3, len(pairs) * 4096 / dt * 1e-6))
