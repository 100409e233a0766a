"""Host logic of the N > 1 path on CPU: sharding covers everything exactly once, the Schur
complement is additive over point shards, and a world_size-2 gloo all-reduce reproduces the
single-rank reduced system."""
import os
import subprocess
import sys

import numpy as np

from metricsfm_amd import _abi as A
from metricsfm_amd import scene, shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_point_ranges_cover_and_balance():
    sc = scene.make_aerial_scene(20, 3000, seed=4)
    for world in (1, 2, 3, 8):
        rs = shard.point_ranges(sc.obs_pt, sc.n_points, world)
        assert rs[0][0] == 0 and rs[-1][1] == sc.n_points
        assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
        k = np.bincount(sc.obs_pt, minlength=sc.n_points).astype(float)
        cost = np.array([(k[a:b] ** 2 + 4 * k[a:b]).sum() for a, b in rs])
        assert cost.max() <= 1.1 * cost.mean() + 200
    # degenerate: fewer points than ranks
    rs = shard.point_ranges(np.array([0, 0, 1, 1], np.int32), 2, 4)
    assert rs[0][0] == 0 and rs[-1][1] == 2 and sum(b - a for a, b in rs) == 2


def test_shards_partition_the_problem():
    sc = scene.make_aerial_scene(12, 900, seed=6)
    full = A.BaArrays.from_scene(sc)
    seen_obs, seen_pts = 0, 0
    for r in range(3):
        s = shard.shard_ba_arrays(full, r, 3)
        lo, hi = s.point_range
        assert (s.cam_pose == full.cam_pose).all() and len(s.point) == hi - lo
        sel = (full.obs_pt >= lo) & (full.obs_pt < hi)
        assert (s.obs_cam == full.obs_cam[sel]).all() and (s.obs_xy == full.obs_xy[sel]).all()
        assert (s.obs_pt + lo == full.obs_pt[sel]).all() and (np.diff(s.obs_pt) >= 0).all()
        seen_obs += len(s.obs_cam)
        seen_pts += len(s.point)
    assert seen_obs == sc.n_obs and seen_pts == sc.n_points


def test_pair_shards():
    pairs = scene.all_pairs(9)
    assert len(pairs) == 72 and (pairs[:, 0] != pairs[:, 1]).all()
    got = np.concatenate([shard.shard_pairs(pairs, r, 4) for r in range(4)])
    assert (got == pairs).all()
    counts = np.array([100, 4000, 50, 4000, 4000, 10, 10, 10, 3000])
    parts = [shard.shard_pairs(pairs, r, 3, counts) for r in range(3)]
    assert (np.concatenate(parts) == pairs).all()
    w = [float((counts[p[:, 0]] * counts[p[:, 1]]).sum()) for p in parts]
    assert max(w) < 2.0 * (sum(w) / 3)


def test_track_shards():
    sc = scene.make_aerial_scene(12, 900, seed=6)
    R, t, c, fk = scene.cameras_for_tracks(sc)
    tr = A.TrackArrays(sc.track_offsets(), sc.obs_cam, sc.obs_xy, R, t, c, fk)
    for world in (1, 3, 8):
        parts = [shard.shard_tracks(tr, r, world) for r in range(world)]
        assert parts[0].track_range[0] == 0 and parts[-1].track_range[1] == sc.n_points
        assert all(parts[r].track_range[1] == parts[r + 1].track_range[0] for r in range(world - 1))
        assert (np.concatenate([p.track_cam for p in parts]) == tr.track_cam).all()
        assert (np.concatenate([p.track_xy for p in parts]) == tr.track_xy).all()
        for p in parts:
            lo, hi = p.track_range
            assert p.track_off[0] == 0 and (np.diff(p.track_off) == np.diff(tr.track_off[lo:hi + 1])).all()
            assert p.struct.n_tracks == hi - lo and p.struct.n_cams == sc.n_cams
        n = np.array([len(p.track_cam) for p in parts])
        assert n.max() <= n.mean() + 12          # balanced to within one track
    # degenerate: fewer tracks than ranks, and none at all
    one = A.TrackArrays(np.array([0, 3], np.int32), sc.obs_cam[:3], sc.obs_xy[:3], R, t, c, fk)
    assert sum(p.struct.n_tracks for p in (shard.shard_tracks(one, r, 4) for r in range(4))) == 1
    none = A.TrackArrays(np.array([0], np.int32), sc.obs_cam[:0], sc.obs_xy[:0], R, t, c, fk)
    assert all(shard.shard_tracks(none, r, 2).struct.n_tracks == 0 for r in range(2))


def test_gloo_world2_reduced_system(tmp_path):
    """Two CPU ranks, each assembling the oracle's reduced system on its point shard; the gloo
    all-reduce of [S | rhs] equals the single-rank system (the identity the GPU path relies on)."""
    out = tmp_path / "res.npz"
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(ROOT, "tests", "gloo_worker.py"), str(out)]
    subprocess.check_call(cmd, env=env, cwd=ROOT, timeout=300)
    r = np.load(out)
    scale = np.abs(r["S_full"]).max()
    assert np.abs(r["S_sum"] - r["S_full"]).max() < 1e-10 * scale
    assert np.abs(r["rhs_sum"] - r["rhs_full"]).max() < 1e-10 * np.abs(r["rhs_full"]).max()
    assert abs(r["cost_sum"] - r["cost_full"]) < 1e-12 * r["cost_full"]
