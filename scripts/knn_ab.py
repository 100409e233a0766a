"""A/B timing of differently built libmsfm on the integer matcher (developer tool): python scripts/knn_ab.py v1 v2 ... runs
`MSFM_LIB=metricsfm_amd/libmsfm_<v>.so` in a child process each, on the SAME box: 64 images x 4096 SIFT-like features of
config 3, all 4 032 ordered pairs, five timed passes; prints Mmatches/s and a checksum of the codes."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, zlib
sys.path.insert(0, %r)
import numpy as np
from metricsfm_amd import capi, scene
sc = scene.config_scene(3)
n = 64
scene.add_features(sc, 4096, images=range(n))
descs = [sc.desc[i] for i in range(n)]
if %d:
    descs = [(512.0 * d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32) for d in descs]
pairs = scene.all_pairs(n)
ctx = capi.Context(0)
ds = ctx.descset(descs)
res = ds.match_pairs(pairs, 0.6, 0.85)
ctx.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); res.rerun(); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
if %d:
    ctx.profile(True); ctx.profile_reset(); res.rerun(); ctx.synchronize()
    print("   ", {k: round(v["total_ms"], 3) for k, v in ctx.profile_get().items()}); ctx.profile(False)
na, ng = res.counts()
crc = 0
for p in range(0, len(pairs), 97):
    crc = zlib.crc32(res.fetch(p)[0].tobytes(), crc)
print("%%-10s %%8.1f Mmatches/s  (%%.2f ms)  all %%d good %%d crc %%08x" %% (%r, 1e-6 * len(pairs) * 4096 / best, 1e3 * best, na.sum(), ng.sum(), crc))
'''
flt = 1 if os.environ.get("KNN_AB_FLOAT") else 0
prof = 1 if os.environ.get("KNN_AB_PROFILE") else 0
for v in sys.argv[1:]:
    env = dict(os.environ, PYTHONPATH=ROOT)
    if v != "default":
        env["MSFM_LIB"] = os.path.join(ROOT, "metricsfm_amd", "libmsfm_%s.so" % v)
    else:
        env.pop("MSFM_LIB", None)
    out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, flt, prof, v)], env=env, capture_output=True, text=True, timeout=600)
    print(out.stdout.strip() or out.stderr[-800:])
