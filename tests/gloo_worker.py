"""Worker of tests/test_shard.py::test_gloo_world2_reduced_system (CPU, gloo)."""
import sys

import numpy as np
import torch
import torch.distributed as dist

from metricsfm_amd import _abi as A
from metricsfm_amd import scene, shard
from oracle import oracle as O


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    sc = scene.make_aerial_scene(10, 600, seed=8, rot_sigma=0.02, trans_sigma=0.2, point_sigma=0.2)
    full = A.BaArrays.from_scene(sc)
    # additivity holds for J^T J and the Schur terms; the LM diagonal is added once after the
    # reduction on the GPU path, so switch it off here (radius -> infinity) along with the
    # column scaling (which needs global column norms)
    opt = O.default_options(jacobi_scaling=0, min_lm_diagonal=1e-300)
    radius = 1e300
    mine = shard.shard_ba_arrays(full, rank, world)
    S, rhs, cost, _ = O.ba_reduced_system(mine, radius, opt)
    n = 6 * sc.n_cams + 3
    assert S.shape == (n, n), "every rank must see every camera block for this test scene"
    buf = torch.from_numpy(np.concatenate([np.triu(S).ravel(), rhs, [cost]]))
    dist.all_reduce(buf)
    if rank == 0:
        Sf, rf, cf, _ = O.ba_reduced_system(full, radius, opt)
        b = buf.numpy()
        np.savez(sys.argv[1], S_sum=b[: n * n].reshape(n, n), rhs_sum=b[n * n: n * n + n], cost_sum=b[-1],
                 S_full=np.triu(Sf), rhs_full=rf, cost_full=cf)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
