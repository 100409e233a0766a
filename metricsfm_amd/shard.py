"""Multi-GPU sharding of the hot path: one process per GPU.

* Bundle adjustment: points (with all their observations) are split over ranks, cameras and
  intrinsics are replicated.  A point's V, W, g and its Schur contribution depend only on its own
  observations, so each rank builds a partial reduced system; the all-reduce hook of the C ABI
  (msfm_ctx_set_allreduce) sums the per-camera J^T J terms and the partial S / rhs over ranks —
  torch.distributed all_reduce, i.e. RCCL over xGMI with backend "nccl".  The reference has no
  counterpart (no collective anywhere in SfM/src).
* Matching: ordered image pairs are independent (fine_matching_graph.cc:58,87): contiguous,
  cost-balanced slices of the idx1-major pair list, no collective.
* Triangulation / reprojection: tracks are independent (structure.cc:211-355): contiguous ranges balanced by
  observation count, cameras replicated, no collective.
"""
import numpy as np

from . import _abi as A


def point_ranges(obs_pt: np.ndarray, n_points: int, world: int, obs_cam=None, cam_mutable=None, pt_mutable=None):
    """Contiguous point ranges [lo, hi) per rank, balanced by sum k_p^2 (the Schur pair count).  With window masks
    (PartialBundleAdjustment, sfm_incremental.cc:917-945) only the work the masks leave counts: a free point costs the
    square of its free-camera rows plus its rows, a frozen point only its free-camera rows, both frozen nothing."""
    if cam_mutable is None and pt_mutable is None:
        k = np.bincount(obs_pt, minlength=n_points).astype(np.int64)
        cost = np.cumsum(k * k + 4 * k)
    else:
        cm = np.ones(len(obs_pt), bool) if cam_mutable is None else np.asarray(cam_mutable)[obs_cam] != 0
        pm = np.ones(n_points, bool) if pt_mutable is None else np.asarray(pt_mutable) != 0
        kc = np.bincount(obs_pt[cm], minlength=n_points).astype(np.int64)      # rows with a camera block
        k = np.bincount(obs_pt, minlength=n_points).astype(np.int64)
        cost = np.cumsum(np.where(pm, kc * kc + 4 * k, 4 * kc))
    total = cost[-1] if n_points else 0
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(cost, total * r / world, side="left")))
    cuts.append(n_points)
    for r in range(1, world + 1):
        cuts[r] = max(cuts[r], cuts[r - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard_ba_arrays(full: A.BaArrays, rank: int, world: int) -> A.BaArrays:
    """The sub-problem of `rank`: its point range and their observations; every camera."""
    if world == 1:
        return full
    lo, hi = point_ranges(full.obs_pt, len(full.point), world, full.obs_cam, full.cam_mutable, full.pt_mutable)[rank]
    sel = (full.obs_pt >= lo) & (full.obs_pt < hi)
    sub = lambda a: None if a is None else a[lo:hi]
    out = A.BaArrays(full.cam_pose, full.cam_model, full.cam_model_of_cam, full.point[lo:hi], full.obs_cam[sel],
                     full.obs_pt[sel] - lo, full.obs_xy[sel], full.pt_weight[lo:hi], cam_mutable=full.cam_mutable,
                     model_mutable=full.model_mutable, pt_mutable=sub(full.pt_mutable), gps_xyz=full.gps_xyz,
                     gps_weight=full.struct.gps_weight)
    out.point_range = (lo, hi)
    return out


def shard_pairs(pairs: np.ndarray, rank: int, world: int, counts=None) -> np.ndarray:
    """Contiguous slice of the (idx1-major) pair list; balanced by M1*M2 when `counts` is given."""
    pairs = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
    if world == 1 or len(pairs) == 0:
        return pairs
    if counts is None:
        w = np.ones(len(pairs))
    else:
        counts = np.asarray(counts, dtype=np.float64)
        w = counts[pairs[:, 0]] * counts[pairs[:, 1]] + 1.0
    c = np.cumsum(w)
    lo = int(np.searchsorted(c, c[-1] * rank / world, side="left")) if rank else 0
    hi = int(np.searchsorted(c, c[-1] * (rank + 1) / world, side="left")) if rank + 1 < world else len(pairs)
    return pairs[lo:hi]


def track_ranges(track_off: np.ndarray, world: int):
    """Contiguous track ranges [lo, hi) per rank, balanced by observation count (triangulation /
    reprojection are independent per track; the split is by point index range, cameras replicated, no collective)."""
    track_off = np.asarray(track_off, dtype=np.int64)
    n = len(track_off) - 1
    total = int(track_off[-1] - track_off[0]) if n > 0 else 0
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(track_off[1:] - track_off[0], total * r / world, side="left")) if n else 0)
    cuts.append(n)
    for r in range(1, world + 1):
        cuts[r] = min(n, max(cuts[r], cuts[r - 1]))
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard_tracks(tracks: A.TrackArrays, rank: int, world: int) -> A.TrackArrays:
    """The tracks of `rank` for msfm_triangulate_*_batch / msfm_reproject_mse_batch: its contiguous track range with the
    offsets rebased, every camera.  `.track_range` gives the range; the host concatenates X / mse / ok in rank order."""
    if world == 1:
        tracks.track_range = (0, len(tracks.track_off) - 1)
        return tracks
    lo, hi = track_ranges(tracks.track_off, world)[rank]
    o0, o1 = int(tracks.track_off[lo]), int(tracks.track_off[hi])
    out = A.TrackArrays(tracks.track_off[lo:hi + 1] - o0, tracks.track_cam[o0:o1], tracks.track_xy[o0:o1], tracks.cam_R, tracks.cam_t,
                        tracks.cam_c, tracks.cam_fk)
    out.track_range = (lo, hi)
    return out


class _DevPtr:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class TorchAllReduce:
    """The msfm_allreduce_fn hook on torch.distributed: backend "nccl" (= RCCL over xGMI) reduces
    the device buffer in place, stream-ordered on the library's stream; with "gloo" (CPU tests,
    or several ranks sharing one GPU) the buffer is staged through the host."""

    def __init__(self, dist, device_index=0, group=None):
        import torch
        self.torch, self.dist, self.group = torch, dist, group
        self.device = torch.device("cuda", device_index)
        self.backend = dist.get_backend(group)
        self.calls = 0
        self.bytes = 0
        self._streams, self._views = {}, {}

    def __call__(self, buf_ptr, count, op, stream_ptr):
        torch, dist = self.torch, self.dist
        rop = dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM
        self.calls += 1
        self.bytes += 8 * count
        # the library reduces the same few device buffers every LM iteration: wrap each (pointer, count) and the
        # stream once, the hook then costs one dictionary lookup plus the collective's own dispatch
        ext = self._streams.get(stream_ptr)
        if ext is None:
            ext = self._streams[stream_ptr] = torch.cuda.ExternalStream(stream_ptr, device=self.device)
        with torch.cuda.stream(ext):
            t = self._views.get((buf_ptr, count))
            if t is None:
                t = self._views[(buf_ptr, count)] = torch.as_tensor(_DevPtr(buf_ptr, count), device=self.device)
                if len(self._views) > 64:   # buffers of destroyed problems
                    self._views = {(buf_ptr, count): t}
            if self.backend == "nccl":
                dist.all_reduce(t, op=rop, group=self.group)  # ordered after / before work on `ext`
            else:
                h = t.cpu()
                dist.all_reduce(h, op=rop, group=self.group)
                t.copy_(h)
                ext.synchronize()
        return 0


def make_allreduce(dist, device_index, ctx=None, group=None):
    """The collective the library's hook calls (msfm_ctx_set_allreduce): torch.distributed on the library's stream."""
    return TorchAllReduce(dist, device_index, group)


def init_native_rccl(ctx, dist, rank, world, group=None):
    """The library's own RCCL communicator (msfm_ctx_init_rccl): rank 0 creates the id, torch.distributed carries the
    128 bytes to the other ranks, every rank joins.  After this the LM loop reduces with ncclAllReduce from C++ -
    no Python callback, no torch tensor wrapping per call."""
    box = [ctx.rccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    ctx.init_rccl(box[0], rank, world)
