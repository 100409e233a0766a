// Cycle-counter probe of the fused Cholesky panel kernel (developer tool, not part of the library):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/chol_probe.hip metricsfm_amd/csrc/ctx.o -o /tmp/chol_probe
// Probes are taken by thread 0 of workgroup `PROBE_WG` of the launch with j0 == g_probe_j0.
#include <hip/hip_runtime.h>
__device__ long long g_probe[32];
__device__ int g_probe_j0 = -1;
__device__ int g_probe_wg = 0;
__device__ volatile int g_probe_on = 0;
#define MSFM_PROBE_ARM(j0v) do { if (threadIdx.x == 0 && (int)blockIdx.x == g_probe_wg) g_probe_on = ((j0v) == g_probe_j0); } while (0)
#define MSFM_PROBE(i) do { if (threadIdx.x == 0 && (int)blockIdx.x == g_probe_wg && g_probe_on) g_probe[i] = clock64(); } while (0)
#define MSFM_CHAIN_STAMPS 1
#include "../metricsfm_amd/csrc/chol.hip"
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
#include <cstring>
#include <algorithm>

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 3003;
  const int pj0 = argc > 2 ? atoi(argv[2]) : 640;
  const int pwg = argc > 3 ? atoi(argv[3]) : 0;
  const int K = argc > 4 ? atoi(argv[4]) : 1;
  const int npad = (n + 1 + 63) / 64 * 64;
  msfm_ctx* ctx = nullptr;
  if (msfm_ctx_create(0, &ctx) != 0) { printf("no ctx\n"); return 1; }
  std::vector<double> h((size_t)npad * npad, 0.0);
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> U(-1, 1);
  (void)K;
  // argv[3] = "plan": config 3's elimination tree (4 leaves of 5 / 10 / 8 / 10 blocks | 2 separators of 4 | root), n = 2624 + root
  const bool with_plan = argc > 3 && !strcmp(argv[3], "plan");
  msfm_chol_plan plan;
  std::vector<int> node_of(n + 1, 6);   // 0..3 leaves, 4..5 separators, 6 root
  if (with_plan) {
    const int lb[5] = {0, 320, 960, 1472, 2112}, sb[3] = {2112, 2368, 2624};
    if (n <= 2624) { printf("plan needs n > 2624\n"); return 1; }
    plan.n_levels = 2;
    plan.level[0].K = 4; plan.level[0].begin = 0; plan.level[0].b0 = 2112;
    for (int k = 0; k < 4; k++) plan.level[0].node[k] = msfm_chol_node{lb[k], lb[k + 1], k, k};
    plan.level[1].K = 2; plan.level[1].begin = 2112; plan.level[1].b0 = 2624;
    for (int k = 0; k < 2; k++) plan.level[1].node[k] = msfm_chol_node{sb[k], sb[k + 1], 2 * k, 2 * k + 1};
    for (int c = 0; c < n; c++) { int nd = 6; for (int k = 0; k < 4; k++) if (c >= lb[k] && c < lb[k + 1]) nd = k; for (int k = 0; k < 2; k++) if (c >= sb[k] && c < sb[k + 1]) nd = 4 + k; node_of[c] = nd; }
    plan.ldc = 64 * ((n + 1 - 2112 + 63) / 64);
    hipMalloc(&plan.corners, sizeof(double) * 8 * (size_t)plan.ldc * plan.ldc);
  }
  auto coupled = [&](int r, int c) {   // r > c: a leaf couples to itself, its separator and the root; a separator to itself and the root
    if (!with_plan) return true;
    const int a = node_of[r], b = node_of[c];
    if (a == b || a == 6) return true;
    if (a >= 4 && b < 4) return b / 2 == a - 4;
    return false;
  };
  for (int r = 0; r < n; r++) {
    for (int c = 0; c < r; c++) {
      h[(size_t)r * npad + c] = coupled(r, c) ? U(g) : 0.0;
    }
    h[(size_t)r * npad + r] = n + 1.0;
  }
  for (int c = 0; c < n; c++) h[(size_t)n * npad + c] = U(g);
  double *M, *work, *w, *z; int* fail;
  hipMalloc(&M, sizeof(double) * h.size());
  hipMalloc(&work, sizeof(double) * (size_t)npad * 144);
  hipMalloc(&w, sizeof(double) * npad); hipMalloc(&z, sizeof(double) * npad);
  hipMalloc(&fail, 16); hipMemset(fail, 0, 16);
  hipMemcpyToSymbol(HIP_SYMBOL(g_probe_j0), &pj0, sizeof(int));
  hipMemcpyToSymbol(HIP_SYMBOL(g_probe_wg), &pwg, sizeof(int));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  msfm_chol_ws* ws = nullptr;
  if (msfm_chol_ws_create(ctx, npad, &ws) != 0) { printf("no ws\n"); return 1; }
  printf("k_chain resident workgroups: %d\n", ws->capacity);
  std::vector<double> Mref, Mnew((size_t)npad * npad), zref(npad), znew(npad), wk_ref, wk_new((size_t)npad * 144);
  for (int mode = 0; mode < 2; mode++) {   // 0: one launch per panel (round 3), 1: the persistent chain
    for (int rep = 0; rep < 4; rep++) {
      hipMemcpy(M, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
      hipMemset(fail, 0, 16);
      hipDeviceSynchronize();
      hipEventRecord(e0, ctx->stream);
      int rc = msfm_chol_factor_solve(ctx, M, npad, n, work, w, z, fail, with_plan ? &plan : nullptr, nullptr, mode ? ws : nullptr);
      hipEventRecord(e1, ctx->stream);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      int hf; hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
      printf("mode %d rep %d rc %d fail %d total %.3f ms\n", mode, rep, rc, hf, ms);
      fflush(stdout);
      if (hf & (1 << 20)) {
        int dbg[8];
        hipMemcpy(dbg, ws->tickets.p + 8, sizeof dbg, hipMemcpyDeviceToHost);
        printf("SYNC failure: site %d workgroup %d step %d need %d seen %d index %d thread %d - stopping\n", dbg[0], dbg[1], dbg[2], dbg[3], dbg[4], dbg[5], dbg[6]);
        return 2;
      }
    }
    std::vector<double>& Mh = mode ? Mnew : (Mref.resize((size_t)npad * npad), Mref);
    std::vector<double>& zh = mode ? znew : zref;
    std::vector<double>& wk = mode ? wk_new : (wk_ref.resize((size_t)npad * 144), wk_ref);
    hipMemcpy(Mh.data(), M, sizeof(double) * Mh.size(), hipMemcpyDeviceToHost);
    hipMemcpy(zh.data(), z, sizeof(double) * npad, hipMemcpyDeviceToHost);
    hipMemcpy(wk.data(), work, sizeof(double) * wk.size(), hipMemcpyDeviceToHost);
    double rmax = 0, bmax = 0;
    for (int r = 0; r < n; r++) {
      double acc = 0;
      for (int c = 0; c < n; c++) acc += (c <= r ? h[(size_t)r * npad + c] : h[(size_t)c * npad + r]) * zh[c];
      rmax = std::max(rmax, std::fabs(acc - h[(size_t)n * npad + r]));
      bmax = std::max(bmax, std::fabs(h[(size_t)n * npad + r]));
    }
    printf("mode %d residual max |S z - b| = %.3e (|b|max %.3e)\n", mode, rmax, bmax);
  }
  {
    // cycle stamps of the last row owner (alive for every step): wait for the updated tiles, wait for L[t, t-1], operands
    // in LDS, pivot chain, tail (X_3, stores, flag) - and the step period
    const int nrtp = (n + 1 + 15) / 16, ncwp = std::max(1, (nrtp - 4 + 2) / 3);
    int wgp = argc > 2 && atoi(argv[2]) >= 0 ? atoi(argv[2]) : ncwp - 1;
    hipMemcpyToSymbol(HIP_SYMBOL(g_chain_stamp_wg), &wgp, sizeof(int));
    const int sj = argc > 4 ? atoi(argv[4]) : 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_chain_stamp_jobs), &sj, sizeof(int));
    hipMemcpy(M, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    msfm_chol_factor_solve(ctx, M, npad, n, work, w, z, fail, with_plan ? &plan : nullptr, nullptr, ws);
    hipDeviceSynchronize();
    static long long st[256][8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_chain_stamp), sizeof st);
    const int P = with_plan ? 10 : (n + 63) / 64;   // (with the plan the stamps are those of the last launch that reached the stamped workgroup index)
    printf("stamps of row owner %d: step | L wait (pivot wave) | helper at S1 (since step start) | S1 + syrk + S2 | pivots | X3 + hb put (h) | M stores (h) | period\n", wgp);
    for (int l = 0; l < P && l < 256; l++)
      printf("  %3d | %6lld | %6lld | %6lld | %6lld | %6lld | %6lld | %6lld\n", l, st[l][2] - st[l][0], st[l][5] - st[l][0], st[l][3] - st[l][2], st[l][4] - st[l][3],
             st[l][6] - st[l][4], st[l][7] - st[l][6], l ? st[l][0] - st[l - 1][0] : 0);
    if (with_plan && ws->launch.size() > 0) {
      // the bulk side of the stamped launch: per launch step, when its tiles were taken / had their counters / were done,
      // relative to the end of the pivot chain of row step l - 1 (the moment the panel of step l - 1 exists)
      static long long tt[8192][4], rt[256][2];
      hipMemcpyFromSymbol(tt, HIP_SYMBOL(g_chain_task_t), sizeof tt);
      hipMemcpyFromSymbol(rt, HIP_SYMBOL(g_chain_row_t), sizeof rt);
      const ChainLaunch& L = ws->launch[0];
      std::vector<ChainTask> th(L.n_tasks);
      hipMemcpy(th.data(), ws->tasks.p + L.task_off, sizeof(ChainTask) * L.n_tasks, hipMemcpyDeviceToHost);
      printf("level 0: %d tasks, %d row owners, %d bulk workgroups\n", L.n_tasks, L.jobs.n_row_wg, L.jobs.n_bulk_wg);
      printf("launch step | urgent tiles: n, taken / counters / done (us after the pivots of step l-1 ended; mean, max) | other tiles: n, done mean, max\n");
      for (int l = 1; l < 10; l++) {
        double su[3] = {0, 0, 0}, mu[3] = {-1e9, -1e9, -1e9}, so = 0, mo = -1e9; int nu = 0, no = 0;
        const double T = (double)rt[l - 1][1];
        for (int t = 0; t < L.n_tasks && t < 8192; t++) {
          if (th[t].l != l || th[t].k != 2) continue;   // the stamped row owner's job
          if (th[t].J == 1) { nu++; for (int i = 0; i < 3; i++) { const double d = (tt[t][i] - T) * 0.01; su[i] += d; mu[i] = std::max(mu[i], d); } }
          else { no++; const double d = (tt[t][2] - T) * 0.01; so += d; mo = std::max(mo, d); }
        }
        if (nu) printf("  %2d | %3d  %6.1f %6.1f | %6.1f %6.1f | %6.1f %6.1f || %4d %6.1f %6.1f   (step %d lasted %.1f us)\n", l, nu, su[0] / nu, mu[0], su[1] / nu, mu[1], su[2] / nu, mu[2], no,
                       no ? so / no : 0.0, mo, l, (rt[l][1] - rt[l - 1][1]) * 0.01);
      }
    }
  }
  // the factor: blocks below the diagonal blocks in M, the diagonal blocks and the 16 x 16 inverses in the workspace
  size_t nd = 0; double dmax = 0;
  for (int r = 0; r <= n; r++)
    for (int c = 0; c < (r / 64) * 64 && c < n; c++) {
      const double a = Mref[(size_t)r * npad + c], b = Mnew[(size_t)r * npad + c];
      if (memcmp(&a, &b, 8)) { nd++; dmax = std::max(dmax, std::fabs(a - b)); }
    }
  size_t nw = 0;
  for (size_t i = 0; i < (size_t)npad * 16; i++) nw += memcmp(&wk_ref[i], &wk_new[i], 8) != 0;
  for (size_t i = (size_t)npad * 80; i < (size_t)npad * 144; i++) nw += memcmp(&wk_ref[i], &wk_new[i], 8) != 0;
  size_t nz = 0;
  for (int i = 0; i < n; i++) nz += memcmp(&zref[i], &znew[i], 8) != 0;
  printf("factor: %zu entries differ (max %.3e); Dinv / Ldiag: %zu differ; solution: %zu differ\n", nd, dmax, nw, nz);
  return (nd || nw || nz) ? 3 : 0;
}
