#!/usr/bin/env python3
"""bench.py — BA iterations/s (+ Mmatches/s) on BASELINE.json's headline configuration.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one Levenberg–Marquardt iteration (linearise if the last step was accepted, build the
Schur complement, factor + solve the reduced camera system, back-substitute, evaluate the trial
cost) of the 500-camera / 200k-point / 1.2M-observation synthetic scene (BASELINE config 3), with the
problem already resident in HBM.  Rank 0 prints ONE JSON line.  The matching leg (exhaustive 2-NN +
ratio tests over ordered image pairs of the same scene) is reported inside the same line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from metricsfm_amd import _abi as A  # noqa: E402
from metricsfm_amd import capi, scene, shard  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
BF16_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
I8_PEAK_TOPS = 5000.0          # MI355X_MICROARCH.md: I8 32x32x32 = 2x the bf16 rate per clock
FP64_PEAK_TFLOPS = 78.6        # vendor FP64 vector = matrix figure (not in the micro-arch guide)


def fixed_iteration_options(steps):
    """Exactly `steps` LM iterations: stopping rules off (negative tolerances can never fire)."""
    return capi.default_options(max_num_iterations=steps, function_tolerance=-1.0, gradient_tolerance=-1.0,
                                parameter_tolerance=-1.0, max_num_consecutive_invalid_steps=1 << 30,
                                min_trust_region_radius=0.0)


def ba_algorithmic_bytes(n_obs, n_pts, n_cams, n_models):
    """SURVEY.md §8d per-LM-iteration algorithmic HBM bytes."""
    n = 6 * n_cams + 3 * n_models
    jac = n_obs * (16 + 8) + n_pts * (24 + 24 + 8) + n_cams * 96
    trial = n_obs * 24 + n_pts * 24
    return jac + trial + 2 * n * n * 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, help="BASELINE config number (3 = headline)")
    ap.add_argument("--match-images", type=int, default=48, help="images of the scene used by the matching leg")
    ap.add_argument("--feats", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-matching", action="store_true")
    ap.add_argument("--verify-pairs", type=int, default=2048, help="image pairs in the geometric-verification leg")
    ap.add_argument("--pose-images", type=int, default=1024, help="images / pairs in the pose-initialiser leg")
    ap.add_argument("--backend", default=None, help="torch.distributed backend for N>1 (default nccl = RCCL)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))

    import torch
    # one rank per GPU on the real node; a rehearsal with more ranks than GPUs (gloo, one test GPU) shares devices
    local_rank = local_rank % max(1, torch.cuda.device_count())
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = args.backend or "nccl"
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)

    ctx = capi.Context(local_rank)
    if world > 1:
        ctx.set_allreduce(shard.TorchAllReduce(dist, local_rank), rank, world)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ------------------------------------------------------------------ scene (seeded, identical on every rank)
    t0 = time.time()
    sc = scene.config_scene(args.config)
    gen_s = time.time() - t0
    full = A.BaArrays.from_scene(sc)
    mine = shard.shard_ba_arrays(full, rank, world)  # points (+ their observations) of this rank; cameras replicated
    ba = ctx.ba(mine)
    start = (mine.cam_pose.copy(), mine.cam_model.copy(), mine.point.copy())

    # ------------------------------------------------------------------ BA: warmup, then exactly K iterations
    if args.warmup > 0:
        ba.run(fixed_iteration_options(args.warmup))
    ba.upload(*start)
    barrier()
    t0 = time.perf_counter()
    res = ba.run(fixed_iteration_options(args.steps))
    barrier()
    ba_s = max_over_ranks(time.perf_counter() - t0)
    assert res["num_iterations"] == args.steps, res["termination"]

    # profiled pass (HIP events per kernel class on the ctx stream) for the roofline figures
    ba.upload(*start)
    ctx.profile(True)
    ctx.profile_reset()
    res_p = ba.run(fixed_iteration_options(args.steps))
    ba_stats = ctx.profile_get()
    ctx.profile(False)

    # the same K iterations as ONE msfm_ba_solve call on the host arrays (what replaces ceres::Solve in the reference):
    # index-structure setup + upload + K iterations + download, on the warm GPU; reported beside `value`, never as it
    one_shot = None
    if rank == 0 and world == 1:
        tmp = A.BaArrays.from_scene(sc)
        t0 = time.perf_counter()
        r1 = ctx.ba_solve(tmp, fixed_iteration_options(args.steps))
        os_s = time.perf_counter() - t0
        one_shot = dict(ms=1e3 * os_s, iterations=r1["num_iterations"], setup_ms=r1["setup_ms"], iterations_per_s=r1["num_iterations"] / os_s,
                        note="msfm_ba_create (host index structures, %s host threads) + upload + iterations + download + destroy"
                             % os.environ.get("MSFM_HOST_THREADS", "default"))

    n_red = res["num_reduced_params"]
    it_s = args.steps / ba_s
    alg_bytes = ba_algorithmic_bytes(sc.n_obs, sc.n_points, sc.n_cams, len(sc.cam_model))
    chol_flops = n_red ** 3 / 3.0 + 2.0 * n_red ** 2
    kernels = []
    for name, st in sorted(ba_stats.items(), key=lambda kv: -kv[1]["total_ms"]):
        kernels.append(dict(kernel=name, launches=st["launches"], ms_per_step=st["total_ms"] / args.steps,
                            avg_launch_us=1e3 * st["total_ms"] / max(1, st["launches"])))
    # per-kernel rooflines from the live HIP-event timings (profiled pass); algorithmic work per launch:
    #   chol_panel_mfma : the panel launches carry the whole factorisation, n^3/3 flops per solve (FP64 MFMA)
    #   ba_linearize    : No * (24 B indices + 16 B observation + 26 + 20 doubles written)          (HBM)
    #   ba_point        : No * (20 doubles read + 24 written) + Np * 12 doubles                      (HBM)
    #   ba_schur_pairs  : (pairs) * 2 * 144 B gathered                                               (HBM / L2)
    #   ba_backsub      : No * 26 doubles read + Np * 9 doubles                                      (HBM)
    rooflines = {}
    n_solves = res_p["num_iterations"]

    def add(kname, bound, work_per_launch, note):
        st = ba_stats.get(kname)
        if not st or not st["launches"]:
            return
        t = st["total_ms"] * 1e-3 / st["launches"]
        if bound == "mfma":
            ach, peak, unit = work_per_launch / t / 1e12, FP64_PEAK_TFLOPS, "TFLOP/s"
        else:
            ach, peak, unit = work_per_launch / t / 1e9, HBM_PEAK_GBS, "GB/s"
        rooflines[kname] = dict(bound=bound, achieved=ach, peak=peak, unit=unit, frac=ach / peak, traffic=None,
                                avg_launch_us=t * 1e6, launches=st["launches"], note=note)

    k = np.bincount(sc.obs_pt, minlength=sc.n_points).astype(np.int64)
    n_pairs_cc = int((k * (k + 1) // 2).sum())
    st = ba_stats.get("chol_panel_mfma")
    lay = ba.layout()
    if st:
        dense_flops = n_red ** 3 / 3.0
        # `achieved` follows the contract: ALGORITHMIC flops of the path (the dense factorisation, n^3/3) per launch
        # / launch time.  With camera domains the kernel executes fewer flops for the same result; that figure is given beside it.
        exe_flops = dense_flops
        note = ("n^3/3 = %.2f GFLOP per factorisation (SURVEY 8d) spread over %d panel launches (trailing update + next potrf + trsm fused)"
                % (dense_flops / 1e9, st["launches"] // max(1, n_solves)))
        if lay["n_domains"] > 1:
            sep = lay["separator_cols"] + 1
            exe_flops = sep ** 3 / 3.0 + sum(nk ** 3 / 3.0 + nk * nk * sep + nk * sep * sep for nk in lay["domain_cols"])
            note += ("; %d camera domains of %s columns are factored side by side, then the %d-column separator: %.2f GFLOP actually executed"
                     % (lay["n_domains"], lay["domain_cols"], lay["separator_cols"], exe_flops / 1e9))
        add("chol_panel_mfma", "mfma", dense_flops * n_solves / max(1, st["launches"]), note)
        rooflines["chol_panel_mfma"]["executed_tflops"] = exe_flops * n_solves / (st["total_ms"] * 1e-3) / 1e12
        rooflines["chol_panel_mfma"]["layout"] = lay
    add("ba_linearize", "hbm", sc.n_obs * (24 + 16 + 46 * 8), "bytes read + written per linearisation")
    add("ba_point", "hbm", sc.n_obs * 44 * 8 + sc.n_points * 12 * 8, "SoA Jacobians in, T / T.u records out")
    add("ba_schur_pairs", "hbm", (n_pairs_cc + sc.n_obs) * 288 + sc.n_points * 144, "two 144-byte T records gathered per pair entry")
    add("ba_backsub", "hbm", sc.n_obs * 26 * 8 + sc.n_points * 9 * 8, "SoA Jacobians in, candidate points out")
    add("ba_ftf", "hbm", sc.n_obs * 26 * 8, "camera-major rows in")
    dom = kernels[0]
    step_bw = alg_bytes / (ba_s / args.steps) / 1e9
    whole = dict(bound="hbm", achieved=step_bw, peak=HBM_PEAK_GBS, unit="GB/s", frac=step_bw / HBM_PEAK_GBS, traffic=None,
                 kernel="lm_iteration (all kernels)",
                 note="algorithmic bytes of one LM iteration (SURVEY 8d: %.0f MB) / measured time per iteration; "
                      "FP64 side: %.2f GFLOP Cholesky per iteration" % (alg_bytes / 1e6, chol_flops / 1e9))
    roofline = dict(rooflines[dom["kernel"]], kernel=dom["kernel"]) if dom["kernel"] in rooflines else whole
    # HBM traffic of the dominant kernel from the rocprofv3 PMC passes (profiles/), per launch, if recorded
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path) and args.config == 3 and world == 1:  # the counters were collected on the headline configuration
        try:
            pmc = json.load(open(pmc_path))
            if roofline.get("kernel") in pmc:
                roofline["traffic"] = pmc[roofline["kernel"]]["bytes_per_launch"]
                roofline["traffic_source"] = pmc[roofline["kernel"]].get("source")
        except Exception:
            pass

    out = dict(metric="BA iterations/sec", value=it_s, unit="iterations/s", n_gpus=world, steps=args.steps,
               warmup=args.warmup, ms_per_step=1e3 * ba_s / args.steps, higher_is_better=True, scaling="strong",
               vs_baseline=None, dtype="f64", data="synthetic",
               config=dict(workload="BASELINE config %d: %d cameras / %d points / %d observations, dense-Schur LM, Huber(1)" %
                           (args.config, sc.n_cams, sc.n_points, sc.n_obs), reduced_system_order=n_red,
                           parallelism="points sharded over %d rank(s), camera block all-reduced" % world,
                           successful_steps=res["num_successful_steps"], unsuccessful_steps=res["num_unsuccessful_steps"],
                           setup_ms=res["setup_ms"], scene_gen_s=gen_s),
               roofline=roofline, roofline_whole_step=whole, kernel_rooflines=rooflines, ba_kernels=kernels,
               ba_cost=dict(initial=res["initial_cost"], final=res["final_cost"]), ba_one_shot=one_shot)

    # ------------------------------------------------------------------ matching leg
    if not args.no_matching:
        n_img = min(args.match_images, sc.n_cams)
        scene.add_features(sc, args.feats, images=range(n_img))
        descs = [sc.desc[i] for i in range(n_img)]
        pairs = scene.all_pairs(n_img)
        my_pairs = shard.shard_pairs(pairs, rank, world)
        ds = ctx.descset(descs)
        mres = ds.match_pairs(my_pairs, 0.6, 0.85, keep_knn=False)
        for _ in range(max(0, args.warmup - 1)):
            mres.rerun()
        msteps = max(1, args.steps)
        barrier()
        t0 = time.perf_counter()
        for _ in range(msteps):
            mres.rerun()
        barrier()
        m_s = max_over_ranks(time.perf_counter() - t0)
        ctx.profile(True)
        ctx.profile_reset()
        mres.rerun()
        ctx.synchronize()
        mstats = ctx.profile_get()
        ctx.profile(False)
        counts = np.array([len(d) for d in descs], dtype=np.int64)
        queries = int(counts[pairs[:, 1]].sum())
        flops = float(2 * 128 * (counts[pairs[:, 0]] * counts[pairs[:, 1]]).sum())
        my_flops = float(2 * 128 * (counts[my_pairs[:, 0]] * counts[my_pairs[:, 1]]).sum())
        na, ng = mres.counts()
        kname = next((k for k in ("knn2_i8_mfma", "knn2_bf16_mfma", "knn2_exact_f64") if k in mstats), None)
        kst = mstats.get(kname)
        peak = {"knn2_i8_mfma": I8_PEAK_TOPS, "knn2_bf16_mfma": BF16_PEAK_TFLOPS, "knn2_exact_f64": FP64_PEAK_TFLOPS}.get(kname)
        kern_tf = my_flops / (kst["total_ms"] * 1e-3) / 1e12 if kst else None
        out["mmatches_per_sec"] = 1e-6 * queries * msteps / m_s
        out["matching"] = dict(metric="Mmatches/sec", value=1e-6 * queries * msteps / m_s, unit="Mmatches/s",
                               images=n_img, pairs=int(len(pairs)), feats_per_image=args.feats, steps=msteps,
                               ms_per_step=1e3 * m_s / msteps, tflops=flops * msteps / m_s / 1e12,
                               dtype={"knn2_i8_mfma": "i8", "knn2_bf16_mfma": "bf16"}.get(kname, "f64"),
                               matches_all=int(na.sum()), matches_good=int(ng.sum()),
                               roofline=dict(bound="mfma", achieved=kern_tf, peak=peak, unit="TFLOP/s",
                                             frac=(kern_tf / peak) if kern_tf else None, traffic=None, kernel=kname,
                                             avg_launch_ms=kst["total_ms"] if kst else None,
                                             note="algorithmic 2*128*M1*M2 operations per pair / kernel time; int8 MFMA, exact integer distances"))

    # ------------------------------------------------------------------ geometric-verification leg
    if not args.no_matching and rank == 0:
        rng = np.random.default_rng(0x4D53464D)
        n_vp, n_vm = args.verify_pairs, 256

        def two_view(n):
            X = np.column_stack([rng.uniform(-40, 40, n), rng.uniform(-30, 30, n), rng.uniform(80, 120, n)])
            a = rng.normal(0, 0.05, 3)
            th = np.linalg.norm(a)
            K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]]) / th
            R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
            Xc = X @ R.T + np.array([10.0, 1.0, 0.5])
            x1 = 4800 * X[:, :2] / X[:, 2:3] + rng.normal(0, 0.5, (n, 2))
            x2 = 4800 * Xc[:, :2] / Xc[:, 2:3] + rng.normal(0, 0.5, (n, 2))
            bad = rng.choice(n, int(0.3 * n), replace=False)
            x2[bad] = np.column_stack([rng.uniform(-2000, 2000, len(bad)), rng.uniform(-1500, 1500, len(bad))])
            return x1.astype(np.float32), x2.astype(np.float32)

        tv = [two_view(n_vm) for _ in range(64)]   # 64 distinct geometries, tiled to n_vp pairs
        v1 = np.concatenate([tv[p % 64][0] for p in range(n_vp)])
        v2 = np.concatenate([tv[p % 64][1] for p in range(n_vp)])
        voff = (np.arange(n_vp + 1) * n_vm).astype(np.int32)
        ctx.fundamental_ransac(voff[:9], v1[:8 * n_vm], v2[:8 * n_vm])   # warm-up
        ctx.profile(True)
        ctx.profile_reset()
        t0 = time.perf_counter()
        _, _, vnin, vok = ctx.fundamental_ransac(voff, v1, v2)
        v_s = time.perf_counter() - t0
        vst = ctx.profile_get()
        ctx.profile(False)
        kms = sum(vst[k]["total_ms"] for k in ("geo_fransac_score", "geo_fransac_select") if k in vst)
        out["geo_verification"] = dict(metric="pairs verified/sec", value=n_vp / v_s, unit="pairs/s", pairs=n_vp, matches_per_pair=n_vm,
                                       outlier_fraction=0.3, samples_per_pair="up to 2000 (128 scored first, the rest only for pairs whose adaptive budget is still open)", accepted=int(vok.sum()), mean_inliers=float(vnin.mean()),
                                       kernel_ms=kms, kernel_pairs_per_sec=(n_vp / (kms * 1e-3)) if kms else None, dtype="f64",
                                       note="host arrays in, host arrays out (PCIe inclusive); FM_RANSAC restatement, 7-point solver")
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            t0 = time.perf_counter()
            O.fundamental_ransac(voff[:5], v1[:4 * n_vm], v2[:4 * n_vm])
            out["geo_verification"]["cpu_baseline"] = dict(value=4 / (time.perf_counter() - t0), unit="pairs/s", cores=1, kind="port",
                                                           sample="4 pairs by the sequential CPU oracle (adaptive stop active)")

    # ------------------------------------------------------------------ pose-initialiser leg
    if not args.no_matching and rank == 0:
        rng = np.random.default_rng(0x4D53464D + 3)
        n_img, n_corr = args.pose_images, 256

        def rod(a):
            th = np.linalg.norm(a)
            K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]]) / th
            return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K

        def pnp(n):
            X = np.column_stack([rng.uniform(-40, 40, n), rng.uniform(-30, 30, n), rng.uniform(-5, 5, n)])
            R = rod(np.array([np.pi, 0.0, 0.0]) + rng.normal(0, 0.05, 3))
            t = -R @ np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), 100.0])
            Xc = X @ R.T + t
            return X, 4800 * Xc[:, :2] / Xc[:, 2:3] + rng.normal(0, 0.5, (n, 2))

        def rel(n):
            X = np.column_stack([rng.uniform(-40, 40, n), rng.uniform(-30, 30, n), rng.uniform(80, 120, n)])
            Xc = X @ rod(rng.normal(0, 0.05, 3)).T + np.array([10.0, 1.0, 0.5])
            return (4800 * X[:, :2] / X[:, 2:3] + rng.normal(0, 0.5, (n, 2)), 4800 * Xc[:, :2] / Xc[:, 2:3] + rng.normal(0, 0.5, (n, 2)))

        pn = [pnp(n_corr) for _ in range(64)]
        pX = np.concatenate([pn[p % 64][0] for p in range(n_img)])
        px = np.concatenate([pn[p % 64][1] for p in range(n_img)])
        rl = [rel(n_corr) for _ in range(64)]
        ra = np.concatenate([rl[p % 64][0] for p in range(n_img)])
        rb = np.concatenate([rl[p % 64][1] for p in range(n_img)])
        poff = (np.arange(n_img + 1) * n_corr).astype(np.int32)
        ctx.epnp_ransac(poff[:9], pX[:8 * n_corr], px[:8 * n_corr], 4800.0)      # warm-up
        ctx.relpose_5pt(poff[:9], ra[:8 * n_corr], rb[:8 * n_corr], 4800.0, 4800.0)
        ctx.profile(True)
        ctx.profile_reset()
        t0 = time.perf_counter()
        _, _, _, pavg, _ = ctx.epnp_ransac(poff, pX, px, 4800.0)
        p_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        _, _, _, rok, rnc = ctx.relpose_5pt(poff, ra, rb, 4800.0, 4800.0)
        r_s = time.perf_counter() - t0
        pst = ctx.profile_get()
        ctx.profile(False)
        pk = sum(pst[k]["total_ms"] for k in ("pose_epnp_hyp", "pose_epnp_select") if k in pst)
        rk = sum(pst[k]["total_ms"] for k in ("pose_e5_hyp", "pose_e5_score", "pose_e5_select") if k in pst)
        out["pose_initialisers"] = dict(
            absolute=dict(metric="images localised/sec", value=n_img / p_s, unit="images/s", images=n_img, correspondences_per_image=n_corr,
                          samples_per_image=200, localised=int((pavg < 5.0).sum()), kernel_ms=pk,
                          kernel_images_per_sec=(n_img / (pk * 1e-3)) if pk else None, kernels={k: pst[k] for k in pst if k.startswith("pose_epnp")}),
            relative=dict(metric="pairs oriented/sec", value=n_img / r_s, unit="pairs/s", pairs=n_img, matches_per_pair=n_corr, samples_per_pair=100,
                          oriented=int(rok.sum()), mean_candidates=float(rnc.mean()), kernel_ms=rk,
                          kernel_pairs_per_sec=(n_img / (rk * 1e-3)) if rk else None, kernels={k: pst[k] for k in pst if k.startswith("pose_e5")}),
            dtype="f64", note="host arrays in, host arrays out (PCIe inclusive); EPnP on 4-point samples / Nister five-point, one GPU thread per sample")
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            t0 = time.perf_counter()
            O.epnp_ransac(poff[:33], pX[:32 * n_corr], px[:32 * n_corr], 4800.0)
            out["pose_initialisers"]["absolute"]["cpu_baseline"] = dict(value=32 / (time.perf_counter() - t0), unit="images/s", cores=1, kind="port",
                                                                        sample="32 images by the sequential CPU oracle")
            t0 = time.perf_counter()
            O.relpose_5pt(poff[:33], ra[:32 * n_corr], rb[:32 * n_corr], 4800.0, 4800.0)
            out["pose_initialisers"]["relative"]["cpu_baseline"] = dict(value=32 / (time.perf_counter() - t0), unit="pairs/s", cores=1, kind="port",
                                                                        sample="32 pairs by the sequential CPU oracle")

    # ------------------------------------------------------------------ CPU baseline (rank 0, N = 1 only)
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        cpu_iters = 1
        ref = A.BaArrays.from_scene(sc)
        t0 = time.perf_counter()
        r = O.ba_solve(ref, O.default_options(max_num_iterations=cpu_iters, function_tolerance=-1.0,
                                              gradient_tolerance=-1.0, parameter_tolerance=-1.0))
        cpu_s = time.perf_counter() - t0
        cpu = dict(value=r["num_iterations"] / (r["solve_ms"] * 1e-3), unit="iterations/s", cores=1, kind="port",
                   sample="%d LM iteration(s) of the same config-%d problem by the CPU oracle (restated reference path, "
                          "num_threads = 1 as basic_structs.h:234), %.1f s wall" % (r["num_iterations"], args.config, cpu_s))
        if not args.no_matching:
            t0 = time.perf_counter()
            O.knn2(sc.desc[0], sc.desc[1][:1024], fast=True)
            ks = time.perf_counter() - t0
            cpu["matching"] = dict(value=1e-6 * 1024 / ks, unit="Mmatches/s", cores=1,
                                   sample="1024 queries x %d train descriptors, brute force float32 (FLANN-L2 arithmetic)" % len(sc.desc[0]))
        out["cpu_baseline"] = cpu
        out["speedup_vs_cpu_port"] = it_s / cpu["value"]

    if rank == 0:
        print(json.dumps(out))
    ba.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
