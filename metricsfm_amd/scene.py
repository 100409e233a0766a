"""Seeded synthetic SfM scenes — inputs for tests and bench.py.

The reference has no sample data; scenes follow its conventions:
4000x3000 images (SfM/test/test_sfm/test_sfm.cc:31), f = 1.2*max(w,h) = 4800
(`f_hyp_`, SfM/src/basic_structs.h:56), one shared CameraModel in UAV mode
(test_sfm.cc:53), centred pixel coordinates (SfM/src/database.cc:522-527), pose data =
(angle-axis, t) (SfM/src/camera.cc:89-99), 128-D float32 descriptors
(database.cc:412-418).  The BA start point is ground truth perturbed the way
BundleAdjuster::Perturb does (SfM/src/optimizer.cc:197-232: rotation noise with the
centre held, then translation noise), with explicit seeded noise instead of std::rand.
"""
from __future__ import annotations

import dataclasses
import numpy as np

SEED_BASE = 0x4D53464D  # 'MSFM'

IMG_W, IMG_H = 4000, 3000
FOCAL = 1.2 * max(IMG_W, IMG_H)


def angle_axis_to_R(aa: np.ndarray) -> np.ndarray:
    """Rodrigues with the reference's small-angle branch (basic_funcs.cc:118-158)."""
    aa = np.asarray(aa, dtype=np.float64)
    a = aa.reshape(-1, 3)
    th2 = np.einsum("ni,ni->n", a, a)
    big = th2 > np.finfo(np.float64).eps
    th = np.sqrt(np.where(big, th2, 1.0))
    w = a / th[:, None]
    c, s = np.cos(th), np.sin(th)
    K = np.zeros((a.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -w[:, 2], w[:, 1]
    K[:, 1, 0], K[:, 1, 2] = w[:, 2], -w[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -w[:, 1], w[:, 0]
    eye = np.eye(3)[None]
    Rb = c[:, None, None] * eye + s[:, None, None] * K + (1 - c)[:, None, None] * np.einsum("ni,nj->nij", w, w)
    Ka = np.zeros_like(K)
    Ka[:, 0, 1], Ka[:, 0, 2] = -a[:, 2], a[:, 1]
    Ka[:, 1, 0], Ka[:, 1, 2] = a[:, 2], -a[:, 0]
    Ka[:, 2, 0], Ka[:, 2, 1] = -a[:, 1], a[:, 0]
    out = np.where(big[:, None, None], Rb, eye + Ka)
    return out.reshape(aa.shape[:-1] + (3, 3))


def R_to_angle_axis(R: np.ndarray) -> np.ndarray:
    """Via quaternion, as basic_funcs.cc:25-107."""
    R = np.asarray(R, dtype=np.float64).reshape(-1, 3, 3)
    out = np.empty((R.shape[0], 3))
    for n, r in enumerate(R):
        tr = r[0, 0] + r[1, 1] + r[2, 2]
        q = np.zeros(4)
        if tr >= 0:
            t = np.sqrt(tr + 1.0)
            q[0] = 0.5 * t
            t = 0.5 / t
            q[1] = (r[2, 1] - r[1, 2]) * t
            q[2] = (r[0, 2] - r[2, 0]) * t
            q[3] = (r[1, 0] - r[0, 1]) * t
        else:
            i = 0
            if r[1, 1] > r[0, 0]:
                i = 1
            if r[2, 2] > r[i, i]:
                i = 2
            j, k = (i + 1) % 3, (i + 2) % 3
            t = np.sqrt(r[i, i] - r[j, j] - r[k, k] + 1.0)
            q[i + 1] = 0.5 * t
            t = 0.5 / t
            q[0] = (r[k, j] - r[j, k]) * t
            q[j + 1] = (r[j, i] + r[i, j]) * t
            q[k + 1] = (r[k, i] + r[i, k]) * t
        s2 = q[1] ** 2 + q[2] ** 2 + q[3] ** 2
        if s2 > 0:
            s = np.sqrt(s2)
            two_theta = 2.0 * (np.arctan2(-s, -q[0]) if q[0] < 0 else np.arctan2(s, q[0]))
            out[n] = q[1:] * (two_theta / s)
        else:
            out[n] = q[1:] * 2.0
    return out


@dataclasses.dataclass
class Scene:
    """Flat arrays in the layout of `msfm_ba_problem` / `msfm_tracks` (include/msfm.h)."""

    name: str
    # ground truth
    cam_pose_gt: np.ndarray  # [Nc,6]
    cam_model_gt: np.ndarray  # [Nm,3]
    point_gt: np.ndarray  # [Np,3]
    # BA start point
    cam_pose: np.ndarray
    cam_model: np.ndarray
    point: np.ndarray
    cam_model_of_cam: np.ndarray  # [Nc] i32
    obs_cam: np.ndarray  # [No] i32
    obs_pt: np.ndarray  # [No] i32, non-decreasing
    obs_xy: np.ndarray  # [No,2]
    pt_weight: np.ndarray  # [Np]
    gps_xyz: np.ndarray | None = None
    # features (filled by add_features)
    desc: list | None = None  # per image float32 [M,128]
    kp_xy: list | None = None  # per image float32 [M,2] centred pixels
    feat_point: list | None = None  # per image i32 [M] point id or -1

    @property
    def n_cams(self):
        return self.cam_pose.shape[0]

    @property
    def n_points(self):
        return self.point.shape[0]

    @property
    def n_obs(self):
        return self.obs_cam.shape[0]

    def track_offsets(self) -> np.ndarray:
        off = np.zeros(self.n_points + 1, dtype=np.int32)
        np.add.at(off, self.obs_pt + 1, 1)
        return np.cumsum(off).astype(np.int32)


def project(pose: np.ndarray, model: np.ndarray, X: np.ndarray):
    """pose [..,6], model [..,3], X [..,3] -> (uv [..,2], depth) — the projection of
    reprojection_error_pose_cam_xyz.h:41-63, vectorised through rotation matrices."""
    R = angle_axis_to_R(pose[..., :3])
    return project_Rt(R, pose[..., 3:], model, X)


def project_Rt(R, t, model, X):
    p = np.einsum("...ij,...j->...i", R, X) + t
    xp, yp = p[..., 0] / p[..., 2], p[..., 1] / p[..., 2]
    r2 = xp * xp + yp * yp
    d = 1.0 + r2 * (model[..., 1] + model[..., 2] * r2)
    return np.stack([model[..., 0] * d * xp, model[..., 0] * d * yp], axis=-1), p[..., 2]


def _perturb(rng, pose_gt, point_gt, rot_sigma, trans_sigma, point_sigma):
    """BundleAdjuster::Perturb (optimizer.cc:197-232) with explicit noise arrays."""
    Nc = pose_gt.shape[0]
    point = point_gt + rng.standard_normal(point_gt.shape) * point_sigma
    R_gt = angle_axis_to_R(pose_gt[:, :3])
    c = -np.einsum("nji,nj->ni", R_gt, pose_gt[:, 3:])  # c = -R^T t
    a = pose_gt[:, :3] + rng.standard_normal((Nc, 3)) * rot_sigma  # SetACPose(a+noise, c)
    R = angle_axis_to_R(a)
    t = -np.einsum("nij,nj->ni", R, c)
    t = t + rng.standard_normal((Nc, 3)) * trans_sigma  # SetRTPose(R, t+noise)
    return np.concatenate([a, t], axis=1), point


_TRACK_PATTERN = np.array([2, 3, 4, 5, 6, 6, 7, 8, 9, 10], dtype=np.int32)  # mean 6.0


def make_ring_scene(n_cams=10, n_points=2000, seed=SEED_BASE + 1, noise_px=0.5, rot_sigma=0.1,
                    trans_sigma=0.5, point_sigma=0.5, name="C1") -> Scene:
    """BASELINE config 1: cameras on a ring looking at the origin, every point seen by
    every camera (No = n_cams*n_points)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    radius, height = 150.0, 60.0
    pts = rng.uniform([-30, -30, -10], [30, 30, 10], size=(n_points, 3))
    poses = np.zeros((n_cams, 6))
    for i in range(n_cams):
        ang = 2 * np.pi * i / n_cams
        c = np.array([radius * np.cos(ang), radius * np.sin(ang), height])
        z = -c / np.linalg.norm(c)  # optical axis towards the origin (+z forward)
        x = np.cross(z, [0, 0, 1.0])
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])  # rows = camera axes in world coordinates
        jit = angle_axis_to_R(rng.uniform(-0.03, 0.03, 3))
        R = jit @ R
        poses[i, :3] = R_to_angle_axis(R)[0]
        poses[i, 3:] = -angle_axis_to_R(poses[i, :3]) @ c
    model = np.array([[FOCAL, 0.0, 0.0]])
    obs_pt = np.repeat(np.arange(n_points, dtype=np.int32), n_cams)
    obs_cam = np.tile(np.arange(n_cams, dtype=np.int32), n_points)
    uv, depth = project(poses[obs_cam], model[np.zeros_like(obs_cam)], pts[obs_pt])
    assert (depth > 0).all()
    obs_xy = uv + rng.standard_normal(uv.shape) * noise_px
    pose0, point0 = _perturb(rng, poses, pts, rot_sigma, trans_sigma, point_sigma)
    return Scene(name, poses, model, pts, pose0, model.copy(), point0,
                 np.zeros(n_cams, np.int32), obs_cam, obs_pt, obs_xy, np.ones(n_points))


def make_aerial_scene(n_cams=50, n_points=20000, seed=SEED_BASE + 2, noise_px=0.5, rot_sigma=0.1,
                      trans_sigma=0.5, point_sigma=0.5, n_models=1, gps_sigma=None,
                      name="C2") -> Scene:
    """BASELINE configs 2-5: nadir cameras on a serpentine grid at height 100, +-5 deg
    jitter, 85 % forward / 60 % side overlap so that every point has >= 10 candidate
    views; track length follows the fixed pattern 2..10 (mean 6), views = nearest
    visible cameras; No = 6*Np exactly when n_points % 10 == 0."""
    from scipy.spatial import cKDTree

    rng = np.random.Generator(np.random.PCG64(seed))
    H = 100.0
    foot_x, foot_y = H * IMG_W / FOCAL, H * IMG_H / FOCAL  # ground footprint of one image
    step_along, step_across = 0.15 * foot_y, 0.40 * foot_x
    n_strips = max(1, int(round(np.sqrt(n_cams * step_along / step_across))))
    per_strip = int(np.ceil(n_cams / n_strips))
    centres, yaw = [], []
    for s in range(n_strips):
        order = range(per_strip) if s % 2 == 0 else range(per_strip - 1, -1, -1)
        for k in order:
            if len(centres) == n_cams:
                break
            centres.append([s * step_across, k * step_along, H])
            yaw.append(0.0 if s % 2 == 0 else np.pi)
    centres = np.array(centres) + rng.uniform(-1.0, 1.0, (n_cams, 3))
    poses = np.zeros((n_cams, 6))
    Rs = np.zeros((n_cams, 3, 3))
    base = np.diag([1.0, -1.0, -1.0])  # +z forward looks down
    for i in range(n_cams):
        cz, sz = np.cos(yaw[i]), np.sin(yaw[i])
        Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]])
        jit = angle_axis_to_R(rng.uniform(-np.deg2rad(5), np.deg2rad(5), 3))
        R = jit @ base @ Rz
        poses[i, :3] = R_to_angle_axis(R)[0]
        Rs[i] = angle_axis_to_R(poses[i, :3])
        poses[i, 3:] = -Rs[i] @ centres[i]
    model = np.tile(np.array([[FOCAL, 0.0, 0.0]]), (n_models, 1))
    model_of_cam = (np.arange(n_cams) % n_models).astype(np.int32)
    tree = cKDTree(centres[:, :2])
    lo = centres[:, :2].min(0) - 0.2 * np.array([foot_x, foot_y])
    hi = centres[:, :2].max(0) + 0.2 * np.array([foot_x, foot_y])
    ktarget = _TRACK_PATTERN[np.arange(n_points) % len(_TRACK_PATTERN)]
    pts = np.zeros((n_points, 3))
    views = [None] * n_points
    todo = np.arange(n_points)
    ncand = min(n_cams, 32)
    for _ in range(200):
        if todo.size == 0:
            break
        cand = np.column_stack([rng.uniform(lo[0], hi[0], todo.size), rng.uniform(lo[1], hi[1], todo.size),
                                rng.uniform(-5, 5, todo.size)])
        _, nn = tree.query(cand[:, :2], k=ncand)
        nn = nn.reshape(todo.size, ncand)
        uv, depth = project_Rt(Rs[nn], poses[nn, 3:], model[model_of_cam[nn]], cand[:, None, :])
        vis = (depth > 0) & (np.abs(uv[..., 0]) < IMG_W / 2 - 2) & (np.abs(uv[..., 1]) < IMG_H / 2 - 2)
        still = []
        for r, p in enumerate(todo):
            v = nn[r][vis[r]]
            if v.size >= ktarget[p]:
                pts[p] = cand[r]
                views[p] = np.sort(v[: ktarget[p]])  # nearest-first from the tree, then cam order
            else:
                still.append(p)
        todo = np.array(still, dtype=np.int64)
    if todo.size:
        raise RuntimeError("scene generator could not place %d points" % todo.size)
    obs_pt = np.repeat(np.arange(n_points, dtype=np.int32), ktarget)
    obs_cam = np.concatenate(views).astype(np.int32)
    uv, depth = project_Rt(Rs[obs_cam], poses[obs_cam, 3:], model[model_of_cam[obs_cam]], pts[obs_pt])
    assert (depth > 0).all()
    obs_xy = uv + rng.standard_normal(uv.shape) * noise_px
    pose0, point0 = _perturb(rng, poses, pts, rot_sigma, trans_sigma, point_sigma)
    gps = None
    if gps_sigma is not None:
        # the reference constrains pose[3:6] = t, not the centre (gps_error_pose_absolute.h:36-38)
        gps = poses[:, 3:] + rng.standard_normal((n_cams, 3)) * gps_sigma * np.array([1.0, 1.0, 5.0])
    return Scene(name, poses, model, pts, pose0, model.copy(), point0, model_of_cam, obs_cam, obs_pt,
                 obs_xy, np.ones(n_points), gps)


def config_scene(config: int, **kw) -> Scene:
    """Scenes of BASELINE.json `configs` (1-based numbering as in BASELINE.md §2)."""
    if config == 1:
        return make_ring_scene(10, 2000, seed=SEED_BASE + 1, name="C1", **kw)
    if config == 2:
        return make_aerial_scene(50, 20000, seed=SEED_BASE + 2, name="C2", **kw)
    if config in (3, 4):
        return make_aerial_scene(500, 200000, seed=SEED_BASE + 3, name="C3", **kw)
    if config == 5:
        return make_aerial_scene(2000, 1000000, seed=SEED_BASE + 5, name="C5", gps_sigma=0.5, **kw)
    raise ValueError(config)


def _sift_like(rng, n):
    v = rng.gamma(0.6, 1.0, size=(n, 128))
    v *= 512.0 / np.linalg.norm(v, axis=1, keepdims=True)
    return np.clip(np.rint(v), 0, 255)


def add_features(scene: Scene, feats_per_image: int, seed=None, images=None) -> Scene:
    """Integer-valued SIFT-like descriptors in [0,255] stored as float32: one base vector
    per 3-D point, +-4 integer noise per observation, distractors up to M per image,
    features of an image in a seeded random order."""
    rng = np.random.Generator(np.random.PCG64(SEED_BASE + 0x100 if seed is None else seed))
    Nc = scene.n_cams
    base = _sift_like(rng, scene.n_points).astype(np.int16)
    order = np.argsort(scene.obs_cam, kind="stable")
    start = np.searchsorted(scene.obs_cam[order], np.arange(Nc + 1))
    scene.desc, scene.kp_xy, scene.feat_point = [None] * Nc, [None] * Nc, [None] * Nc
    for c in (range(Nc) if images is None else images):
        sub = np.random.Generator(np.random.PCG64([SEED_BASE + 0x200, c]))
        o = order[start[c]:start[c + 1]]
        n_obs = o.size
        M = max(feats_per_image, n_obs)
        d = np.empty((M, 128), dtype=np.float32)
        xy = np.empty((M, 2), dtype=np.float32)
        pid = np.full(M, -1, dtype=np.int32)
        d[:n_obs] = np.clip(base[scene.obs_pt[o]] + sub.integers(-4, 5, size=(n_obs, 128)), 0, 255)
        xy[:n_obs] = scene.obs_xy[o]
        pid[:n_obs] = scene.obs_pt[o]
        n_dis = M - n_obs
        d[n_obs:] = _sift_like(sub, n_dis)
        xy[n_obs:] = np.column_stack([sub.uniform(-IMG_W / 2, IMG_W / 2, n_dis), sub.uniform(-IMG_H / 2, IMG_H / 2, n_dis)])
        perm = sub.permutation(M)
        scene.desc[c], scene.kp_xy[c], scene.feat_point[c] = d[perm], xy[perm], pid[perm]
    return scene


def all_pairs(n_images: int) -> np.ndarray:
    """matching_type = "all": every ordered pair (i, j != i), idx1-major
    (SfM/src/graph/initial_matching_graph.cc:55-63)."""
    i, j = np.meshgrid(np.arange(n_images), np.arange(n_images), indexing="ij")
    m = i != j
    return np.column_stack([i[m], j[m]]).astype(np.int32)


def cameras_for_tracks(scene: Scene, pose=None, model=None):
    """R, t, c, (f,k1,k2) per camera as Camera::UpdatePoseFromData keeps them
    (SfM/src/camera.cc:113-137)."""
    pose = scene.cam_pose_gt if pose is None else pose
    model = scene.cam_model_gt if model is None else model
    R = angle_axis_to_R(pose[:, :3])
    t = pose[:, 3:].copy()
    c = -np.einsum("nji,nj->ni", R, t)
    return R.reshape(-1, 9).copy(), t, c, model[scene.cam_model_of_cam].copy()


def perturb_camera(scene: Scene, idx: int, rot_sigma=0.05, trans_sigma=0.5, seed=None) -> Scene:
    """A freshly localised camera inside an already adjusted model (the state PartialBundleAdjustment starts from,
    sfm_incremental.cc:917): camera `idx` gets BundleAdjuster::Perturb-style noise on top of its current pose
    (rotation with the centre held, then translation), everything else is left alone.  In place."""
    rng = np.random.Generator(np.random.PCG64([SEED_BASE + 0x300, idx] if seed is None else seed))
    pose, _ = _perturb(rng, scene.cam_pose[idx:idx + 1], np.zeros((0, 3)), rot_sigma, trans_sigma, 0.0)
    scene.cam_pose[idx] = pose[0]
    return scene
