"""The N > 1 bundle-adjustment path with real kernels: 2 and 3 ranks share the one GPU of the test
box (gloo backend; the RCCL/xGMI case differs only in the backend string of torch.distributed) and
must reproduce the single-rank solve to ~1e-9 (SURVEY.md §8d parity gate for 8-GPU vs 1-GPU)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,which,port", [(2, "c2", 29741), (3, "gps", 29742), (2, "domains", 29743), (2, "c3", 29745), (2, "c5w", 29746), (2, "c2+fold", 29748), (3, "gps+fold", 29749)])
def test_sharded_ba_matches_single_rank(tmp_path, world, which, port):
    out = tmp_path / "mr.npz"
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env["MSFM_DEVICE_SHARE"] = str(world)   # the ranks share the one GPU of the test box: the persistent kernels leave each other room
    if which == "domains":
        env["MSFM_CHOL_DOMAINS"] = "2"
    if which.endswith("+fold"):   # the Schur products formed inside k_point (FoldTables) also on these small problems
        env["MSFM_FOLD_MIN"] = "0"
        which = which[:-5]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "multirank_worker.py"), str(out), which]
    subprocess.check_call(cmd, env=env, cwd=ROOT, timeout=600)
    r = np.load(out)
    assert r["it"] == r["it1"]
    np.testing.assert_array_equal(r["ok"], r["ok1"])
    np.testing.assert_allclose(r["cost"], r["cost1"], rtol=1e-9)
    for a, b in (("point", "point1"), ("cam", "cam1"), ("model", "model1")):
        assert np.abs(r[a] - r[b]).max() <= 1e-8 * np.abs(r[b]).max(), a
    assert r["calls"] >= 3 * r["it"]  # the hook really carried the reduction (3 sums per LM iteration)


def test_hook_runs_on_the_rccl_backend(tmp_path):
    """backend "nccl" (= RCCL) with a single rank: the hook wraps the library's stream and a raw device pointer and hands
    them to torch.distributed.all_reduce - the call sequence of the N-GPU bench, minus the peers."""
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from metricsfm_amd import capi, shard
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29744")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
ctx = capi.Context(0)
hook = shard.TorchAllReduce(dist, 0)
assert hook.backend == "nccl"
buf = torch.arange(1000, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for op in (0, 1, 0):
    assert hook(buf.data_ptr(), buf.numel(), op, ctx.stream) == 0
ctx.synchronize(); torch.cuda.synchronize()
assert (buf.cpu().numpy() == np.arange(1000)).all() and hook.calls == 3 and len(hook._views) == 1
ctx.close(); dist.destroy_process_group()
print("ok")
''' % ROOT
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, timeout=300, capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_native_rccl_collective_world1(tmp_path):
    """msfm_ctx_init_rccl: the library loads librccl itself, creates a communicator (one rank) and reduces device buffers
    with ncclAllReduce on its own stream - the C++ path of the N-GPU bench, minus the peers.  A bundle adjustment on the
    same context afterwards is unaffected."""
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from metricsfm_amd import _abi as A, capi, scene
torch.cuda.set_device(0)
ctx = capi.Context(0)
uid = ctx.rccl_unique_id()
assert len(uid) == 128 and any(uid)
ctx.init_rccl(uid, 0, 1)
buf = torch.arange(5000, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for op in (0, 1, 0):
    ctx.allreduce(buf.data_ptr(), buf.numel(), op)
ctx.synchronize()
assert (buf.cpu().numpy() == np.arange(5000)).all()
sc = scene.make_ring_scene(5, 300, seed=4)
r = ctx.ba_solve(A.BaArrays.from_scene(sc), capi.default_options(max_num_iterations=5))
assert r["num_iterations"] >= 3
try:
    ctx.init_rccl(uid, 0, 1)
    raise SystemExit("a second communicator on one context must be refused")
except capi.MsfmError as e:
    assert e.code == A.MSFM_E_INVAL
ctx.close()
print("ok")
''' % ROOT
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, timeout=300, capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the way the driver starts --gpus 1) spawns its two ranks as child
    processes and prints rank 0's JSON line; here the ranks share the one GPU of the test box, hence gloo."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", "2", "--steps", "4",
                          "--warmup", "1", "--no-matching", "--no-cpu-baseline", "--verify-pairs", "64", "--pose-images", "32"],
                         env=env, cwd=ROOT, timeout=900, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["value"] > 0 and r["roofline"]["frac"] > 0
    assert r["triangulation"]["midpoint"]["tracks"] == 20000


def test_bench_two_ranks_over_rccl():
    """`python bench.py --gpus 2` the way the driver's scaling run starts it (its own torch.distributed.run child, backend
    nccl = RCCL, one rank per GPU, ncclAllReduce inside libmsfm): skipped on a one-GPU box.  The line must name RCCL as the
    collective, report two ranks and a positive rate."""
    import json
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MSFM_SYNC_TIMEOUT_S="60")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                          "--no-extras", "--no-matching"], env=env, cwd=ROOT, timeout=900, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["value"] > 0
    assert "RCCL" in r["config"]["collective"]
