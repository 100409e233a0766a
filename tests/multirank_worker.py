"""Worker of tests/test_gpu_multirank.py: `world` ranks share cuda:0 (gloo backend, host-staged
all-reduce through the C ABI hook); rank 0 also solves the un-sharded problem and stores both."""
import sys

import numpy as np
import torch
import torch.distributed as dist

from metricsfm_amd import _abi as A
from metricsfm_amd import capi, scene, shard


def main():
    out, which = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    if which == "gps":
        sc = scene.make_aerial_scene(20, 2500, seed=21, gps_sigma=0.5, rot_sigma=0.02, trans_sigma=0.3, point_sigma=0.2)
        kw = dict(gps_xyz=sc.gps_xyz, gps_weight=float(sc.n_obs // sc.n_cams))
    elif which == "c3":   # BASELINE config 4's workload: config 3 sharded over ranks
        sc = scene.config_scene(3)
        kw = {}
    elif which == "c5w":  # config 5's shape at a size the test box generates quickly: windowed BA + GPS rows, sharded
        from metricsfm_amd import window
        sc = scene.make_aerial_scene(300, 60000, seed=55, n_models=300, gps_sigma=0.5, rot_sigma=2e-4, trans_sigma=0.01, point_sigma=0.02)
        scene.perturb_camera(sc, 299)
    elif which == "domains":
        # >= 128 cameras: the union camera graph is bisected identically on every rank (MSFM_CHOL_DOMAINS forced by the test)
        sc = scene.make_aerial_scene(150, 5000, seed=8)
        kw = {}
    else:
        sc = scene.config_scene(2)
        kw = {}
    make = (lambda: window.partial_bundle_adjustment_problem(sc, 299, gps=True)[0]) if which == "c5w" else (lambda: A.BaArrays.from_scene(sc, **kw))
    full = make()
    mine = shard.shard_ba_arrays(full, rank, world)
    ctx = capi.Context(0)
    hook = shard.TorchAllReduce(dist, 0)
    ctx.set_allreduce(hook, rank, world)
    opts = capi.default_options(max_num_iterations=12 if which == "c3" else 40)
    res = ctx.ba_solve(mine, opts)
    # gather the points back (host concatenation; cameras are replicated and must agree bitwise)
    pts = [None] * world
    dist.all_gather_object(pts, mine.point)
    cams = [None] * world
    dist.all_gather_object(cams, mine.cam_pose)
    if rank == 0:
        for c in cams[1:]:
            assert (c == cams[0]).all(), "replicated cameras diverged between ranks"
        ctx1 = capi.Context(0)
        ref = make()
        res1 = ctx1.ba_solve(ref, opts)
        np.savez(out, point=np.concatenate(pts), cam=mine.cam_pose, model=mine.cam_model, point1=ref.point, cam1=ref.cam_pose,
                 model1=ref.cam_model, cost=res["iterations"]["cost"], cost1=res1["iterations"]["cost"],
                 ok=res["iterations"]["step_is_successful"], ok1=res1["iterations"]["step_is_successful"],
                 calls=hook.calls, bytes=hook.bytes, it=res["num_iterations"], it1=res1["num_iterations"])
        ctx1.close()
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
