"""Staged (lane = observation phases) against thread-per-track triangulation / reprojection on config 3's 200 k tracks:
kernel times from the library's timers, results compared bit for bit.  gpurun -- 'python scripts/tri_ab.py'"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, zlib
sys.path.insert(0, %r)
import numpy as np
from metricsfm_amd import _abi as A, capi, scene
sc = scene.config_scene(3)
R, t, c, fk = scene.cameras_for_tracks(sc)
tr = A.TrackArrays(sc.track_offsets(), sc.obs_cam, sc.obs_xy, R, t, c, fk)
ctx = capi.Context(0)
ctx.triangulate_midpoint(tr, 7.0, np.deg2rad(3.0))
ctx.profile(True); ctx.profile_reset()
for _ in range(5):
    X, m, ok = ctx.triangulate_midpoint(tr, 7.0, np.deg2rad(3.0))
    m2 = ctx.reproject_mse(tr, X)
st = ctx.profile_get()
print({k: round(1e3 * v["total_ms"] / v["launches"], 1) for k, v in st.items() if k.startswith("tri_")}, "us per launch;", int(ok.sum()), "accepted; crc %%08x %%08x %%08x" %% (zlib.crc32(X.tobytes()), zlib.crc32(m.tobytes()), zlib.crc32(m2.tobytes())))
'''
for v in ("1", "0", "1", "0"):
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=dict(os.environ, MSFM_TRI_STAGED=v), capture_output=True, text=True, timeout=600)
    print("MSFM_TRI_STAGED=" + v, out.stdout.strip() or out.stderr[-800:])
