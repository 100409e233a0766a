import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from metricsfm_amd import _abi as A, capi, scene
ctx = capi.Context(0)
for cfg in (2, 3):
    sc = scene.config_scene(cfg)
    arr = A.BaArrays.from_scene(sc)
    for rep in range(2):
        t0 = time.perf_counter(); ba = ctx.ba(arr); ctx.synchronize(); t1 = time.perf_counter()
        r = ba.run(capi.default_options(max_num_iterations=10, function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0))
        t2 = time.perf_counter()
        ba.upload(arr.cam_pose, arr.cam_model, arr.point)
        ta = time.perf_counter()
        r2 = ba.run(capi.default_options(max_num_iterations=10, function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0))
        tb = time.perf_counter()
        print("   second run of the same object: %.1f ms (solve_ms %.1f)" % (1e3 * (tb - ta), r2["solve_ms"]))
        t2b = time.perf_counter(); ba.close(); t3 = time.perf_counter(); t2 = t2 if True else t2b
        print("config %d: create %.1f ms, 10 iterations %.1f ms (solve_ms %.1f), destroy %.1f ms" % (cfg, 1e3*(t1-t0), 1e3*(t2-t1), r["solve_ms"], 1e3*(t3-t2b)))
    t0 = time.perf_counter(); r = ctx.ba_solve(A.BaArrays.from_scene(sc), capi.default_options(max_num_iterations=10)); t1 = time.perf_counter()
    print("config %d: msfm_ba_solve one-shot %.1f ms for %d iterations" % (cfg, 1e3*(t1-t0), r["num_iterations"]))
