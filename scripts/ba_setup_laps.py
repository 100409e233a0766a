"""Developer tool: the set-up laps (MSFM_VERBOSE) of two small fresh solves - a window of config 2's incremental sequence and a
window of config 5 - the calls that scripts/ba_incremental.py counts."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from metricsfm_amd import capi, scene, window
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ba_incremental import prefix_scene
ctx = capi.Context(0)
sc = scene.config_scene(2)
sub, kept = prefix_scene(sc, 30, sc.cam_pose, sc.point)
scene.perturb_camera(sub, 29)
for rep in range(3):
    arr, info = window.partial_bundle_adjustment_problem(sub, 29, compact=True)
    if rep == 2: os.environ["MSFM_VERBOSE"] = "1"
    r = ctx.ba_solve(arr, capi.default_options(max_num_iterations=3))
    os.environ.pop("MSFM_VERBOSE", None)
print("C2 window: obs %d points %d setup %.3f ms" % (len(arr.obs_cam), len(arr.point), r["setup_ms"]), file=sys.stderr)
arr, kp = window.gather(sub, compact=True)
for rep in range(3):
    arr, kp = window.gather(sub, compact=True)
    if rep == 2: os.environ["MSFM_VERBOSE"] = "1"
    r = ctx.ba_solve(arr, capi.default_options(max_num_iterations=3))
    os.environ.pop("MSFM_VERBOSE", None)
print("C2 full (30 cameras): obs %d points %d setup %.3f ms" % (len(arr.obs_cam), len(arr.point), r["setup_ms"]), file=sys.stderr)
