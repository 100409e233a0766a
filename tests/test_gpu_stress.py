"""A short run of the randomised parity sweep (scripts/stress_parity.py): ragged and degenerate batches of every
RANSAC / pose / matching leg compared bit for bit with the oracle, small random bundle adjustments (all-free ring scenes,
and aerial scenes with random frozen cameras / points / intrinsics, several intrinsics blocks, GPS rows) within the gates.
(30000 rounds of the first version and 1600 rounds with the masked bundle adjustments, 200 of those, ran clean on the
MI355X box when they were written; this keeps 150 rounds in the suite.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomised_parity_sweep():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "stress_parity.py"), "150"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert "150 rounds, 0 mismatches" in out.stdout


def test_pool_debug_sequence():
    """The device-block pool cross-checks every free against its own record; with MSFM_POOL_DEBUG=1 a disagreement aborts.
    A bundle adjustment with accepted steps (cam / cam_c swap) followed by descriptor uploads and matching - the sequence
    that once handed a 32 KB block out as 64 KB - must run clean in a fresh process, and so must a windowed solve and a
    re-upload."""
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from metricsfm_amd import _abi as A, capi, scene, window\n"
        "ctx = capi.Context(0)\n"
        "sc = scene.make_aerial_scene(30, 2500, seed=71, n_models=30, gps_sigma=0.5)\n"
        "r = ctx.ba_solve(A.BaArrays.from_scene(sc), capi.default_options(max_num_iterations=8))\n"
        "assert r['num_successful_steps'] >= 3\n"
        "arr, _ = window.partial_bundle_adjustment_problem(sc, 29, gps=True)\n"
        "ctx.ba_solve(arr, capi.default_options(max_num_iterations=5))\n"
        "scene.add_features(sc, 700, images=[0, 1, 2])\n"
        "ds = ctx.descset([sc.desc[i] for i in range(3)])\n"
        "res = ds.match_pairs(scene.all_pairs(3), keep_knn=True)\n"
        "na, ng = res.counts()\n"
        "assert ng.sum() > 0\n"
        "ctx.ba_solve(A.BaArrays.from_scene(sc), capi.default_options(max_num_iterations=4))\n"
        "ds2 = ctx.descset([sc.desc[i] * 0.37 for i in range(3)])\n"
        "res2 = ds2.match_pairs(scene.all_pairs(3))\n"
        "res2.counts(); res.close(); res2.close(); ds.close(); ds2.close(); ctx.close()\n"
        "print('pool ok')\n") % ROOT
    env = dict(os.environ, MSFM_POOL_DEBUG="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "pool ok" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])
    assert "msfm pool:" not in out.stderr
