/*
 * msfm.h — C ABI of the MI355X-native bundle-adjustment + feature-matching core.
 *
 * Every entry point replaces one inline third-party call (or one short loop) of the
 * reference's SfM/src hot path; the reference file:line each one stands in for is cited
 * on the declaration.  Plain pointers and sizes only: no C++ types, no torch types.
 *
 * Conventions
 *   - All buffers are caller-owned HOST memory unless the name ends in `_dev`.
 *     The library never retains a host pointer past return.
 *   - Return value: MSFM_OK (0) or a negative MSFM_E_* code; msfm_last_error(ctx)
 *     gives the text.  No C++ exception crosses this boundary.
 *   - One msfm_ctx per GPU (one process per GPU).  A ctx is not re-entrant; the
 *     reference's OpenMP-parallel kNN loop (fine_matching_graph.cc:87-100) is hoisted
 *     into the one batched call msfm_match_pairs().
 *   - Bundle adjustment arithmetic is IEEE binary64 throughout; matching takes
 *     binary32 descriptors as the reference does (database.cc:412-418).
 */
#ifndef MSFM_H_
#define MSFM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSFM_VERSION 100 /* 0.1.0 */

/* ---- error codes ------------------------------------------------------------------ */
#define MSFM_OK 0
#define MSFM_E_INVAL (-1)   /* bad argument / inconsistent sizes                         */
#define MSFM_E_NOMEM (-2)   /* host or device allocation failed                          */
#define MSFM_E_DEVICE (-3)  /* HIP runtime error (no GPU, launch failure, ...)           */
#define MSFM_E_NUMERIC (-4) /* unrecoverable numeric failure (Ceres: FAILURE)            */

typedef struct msfm_ctx msfm_ctx;

int msfm_version(void);
/* device < 0: use the current HIP device.  Fails with MSFM_E_DEVICE when no GPU is
 * visible — there is no CPU fallback behind this ABI. */
int msfm_ctx_create(int device, msfm_ctx** out);
/* Objects created from a ctx (descriptor sets, match results, resident BA problems) should be destroyed first; if some
 * are still alive the context is kept until the last of them is destroyed, then released. */
void msfm_ctx_destroy(msfm_ctx* ctx);
const char* msfm_last_error(const msfm_ctx* ctx);
/* The HIP stream (hipStream_t) every kernel of this ctx is launched on. */
void* msfm_ctx_stream(msfm_ctx* ctx);
int msfm_ctx_synchronize(msfm_ctx* ctx);

/* Per-kernel-class device time accumulated with HIP events on the ctx stream since the
 * last reset (used by bench.py for the live roofline figure).  Off by default. */
#define MSFM_MAX_KERNEL_STATS 32
typedef struct msfm_kernel_stat {
  char name[48];
  uint64_t launches;
  double total_ms;
} msfm_kernel_stat;
int msfm_ctx_profile_enable(msfm_ctx* ctx, int enable);
int msfm_ctx_profile_reset(msfm_ctx* ctx);
int msfm_ctx_profile_get(msfm_ctx* ctx, msfm_kernel_stat* stats, int cap, int* n_out);

/* ==================================================================================== *
 *  Matching  
 * ==================================================================================== */

/* Exact 2-nearest-neighbour search, squared L2, of every query descriptor among the
 * train descriptors.  Replaces the FLANN pair
 *     flann_build_index(...)                   SfM/src/graph/fine_matching_graph.cc:72-81
 *     flann_find_nearest_neighbors_index(...)  SfM/src/graph/fine_matching_graph.cc:99
 * (twin call site SfM/src/slam_gps.cc:438-447,463) and fills the same two arrays FLANN
 * fills (`knn_id[j]`, `knn_dis[j]`, fine_matching_graph.cc:96-99):
 *     ids     [n_query][2]  index into train, nearest first
 *     sqdists [n_query][2]  squared L2 distance, ascending
 * Unlike the kd-tree (8 trees, 64 checks — approximate) the result is the exact 2-NN;
 * equal distances are ordered by lower train index.  n_train >= 2 is required (the
 * reference divides dists[0]/dists[1]).  dim must be 128 (SIFT; database.cc:412-418 stores 128 columns): the
 * kernels are specialised for it and every other value is refused with MSFM_E_INVAL.
 */
int msfm_knn2_f32(msfm_ctx* ctx, const float* train, int n_train, const float* query, int n_query,
                  int dim, int* ids, float* sqdists);

/* Device-resident descriptor store: one entry per image, uploaded once and reused by
 * every pair that touches the image (the reference re-reads `<idx2>_feature` from disk
 * per pair, fine_matching_graph.cc:91). */
typedef struct msfm_descset msfm_descset;
int msfm_descset_create(msfm_ctx* ctx, int n_images, int dim /* 128 */, msfm_descset** out);
/* Replaces the image's descriptors.  Waits for work already enqueued on the ctx stream (a running match may still read
 * the old buffers).  Match results created before the upload become stale: msfm_match_pairs_rerun /
 * msfm_match_result_fetch on them return MSFM_E_INVAL; create a new result with msfm_match_pairs. */
int msfm_descset_upload(msfm_descset* set, int image, const float* desc, int count);
int msfm_descset_count(const msfm_descset* set, int image);
/* Ordering rule: a set may be destroyed before its match results.  The results then keep what they own - counts, codes
 * and (with keep_knn) the 2-NN arrays stay readable through msfm_match_result_counts / _fetch / _stats - while
 * msfm_match_pairs_rerun on them returns MSFM_E_INVAL; they are released by msfm_match_result_destroy as usual. */
void msfm_descset_destroy(msfm_descset* set);

/* Batched pair matching with the ratio tests fused in
 * (fine_matching_graph.cc:87-133; the SLAM variant slam_gps.cc:469-503 is msfm_match_pairs_slam below).
 * pairs[p] = {idx1 (train image), idx2 (query image)}.
 * For pair p and query feature m of image idx2 (in feature order, which is the order of
 * the reference loop fine_matching_graph.cc:116-133):
 *     ratio = sqdist0 / sqdist1
 *     good  = ratio < ratio_good,  all = ratio < ratio_all       (two independent tests, :118-130)
 *     code  = -1                                                  if neither
 *           = id0 | (good ? MSFM_MATCH_GOOD : 0) | (all ? 0 : MSFM_MATCH_NOT_ALL)   otherwise
 * With the reference's thresholds (0.6 < 0.85) a good match is always in the "all" set and MSFM_MATCH_NOT_ALL never
 * appears; id0 = code & MSFM_MATCH_ID_MASK.
 * The codes live in device memory inside the result object; fetch copies one pair to
 * the host.  n_all / n_good are the sizes of the reference's matches_all / matches_good.
 */
#define MSFM_MATCH_GOOD 0x40000000
#define MSFM_MATCH_NOT_ALL 0x20000000
#define MSFM_MATCH_ID_MASK 0x1FFFFFFF
typedef struct msfm_match_result msfm_match_result;
int msfm_match_pairs(msfm_descset* set, const int* pairs /*[n_pairs][2]*/, int n_pairs,
                     float ratio_good, float ratio_all, int keep_knn, msfm_match_result** out);
int msfm_match_result_counts(msfm_match_result* res, int* n_all /*[n_pairs]*/,
                             int* n_good /*[n_pairs]*/);
/* code[count(idx2)]; ids/sqdists may be NULL, and are only available with keep_knn. */
int msfm_match_result_fetch(msfm_match_result* res, int pair, int32_t* code, int* ids,
                            float* sqdists);
/* Non-integral descriptors go through an approximate shortlist + exact re-rank; queries whose
 * shortlist could not be certified are redone by exact brute force.  n_slow_path counts them
 * (0 for integer-valued data, which takes the exact int8 kernel). */
int msfm_match_result_stats(msfm_match_result* res, int* n_queries, int* n_slow_path);
void msfm_match_result_destroy(msfm_match_result* res);
/* Re-run the same pair list into an existing result object (steady-state bench loop:
 * no allocation, no host transfer). */
int msfm_match_pairs_rerun(msfm_descset* set, msfm_match_result* res);

/* Keypoint positions of an image's features, [count][2] float as cv::Point2f (`db_.keypoints_[id]->pts[k].pt`), resident
 * beside its descriptors; `count` must equal the image's descriptor count.  Needed by msfm_match_pairs_slam; uploading the
 * descriptors of the image again drops them. */
int msfm_descset_upload_keypoints(msfm_descset* set, int image, const float* xy, int count);

/* The matching loop of SLAMGPS::FeatureMatching step 2 (SfM/src/slam_gps.cc:455-503), batched over image pairs, in place of
 * flann_find_nearest_neighbors_index (:463) and the three checks behind it.  For pair p = {id1 (train), id2 (query)} with
 * its prior F = Fs[i][j] and H = Hs[i][j] (row-major [9], as cv::Mat_<double>(3,3)) and query feature m:
 *     check1 (:469-475)  ratio = sqdist0 / sqdist1;  rejected if ratio > th_first_second_ratio
 *                        (`>`: a ratio equal to the threshold, or 0/0 from duplicate descriptors, passes - unlike the
 *                        `ratio < th` tests of fine_matching_graph.cc:118-130 behind msfm_match_pairs)
 *     check2 (:478-488)  l = F [x1 y1 1]^T, rejected if |l . [x2 y2 1]| / sqrt(l0^2 + l1^2) > th_epipolar
 *     check3 (:490-499)  q = H [x1 y1 1]^T / q2, rejected if |[x2 y2] - q| > 40 * th_distance
 * with (x1, y1) the position of train feature id0 and (x2, y2) of query feature m, binary64 arithmetic as cv::Mat does it.
 *     code = id0 (no flag bits) for a match that passes all three, -1 otherwise   -> matches[j] = (code, m) in m order (:503)
 *     n_good[p] = survivors of check1,  n_all[p] = survivors of all three (= matches[j].size() before :509)
 * The survivors then go to msfm_fundamental_ransac_batch (GeoVerificationFundamental, :509). */
typedef struct msfm_slam_match_options {
  float th_first_second_ratio; /* 0.80               (slam_gps.cc:320) */
  float th_epipolar;           /* 2.0 / resize_ratio (slam_gps.cc:316) */
  float th_distance;           /* 5.0 / resize_ratio (slam_gps.cc:317) */
} msfm_slam_match_options;
void msfm_slam_match_default_options(msfm_slam_match_options* opt);
int msfm_match_pairs_slam(msfm_descset* set, const int* pairs /*[n_pairs][2]*/, int n_pairs,
                          const double* F /*[n_pairs][9]*/, const double* H /*[n_pairs][9]*/,
                          const msfm_slam_match_options* opt, int keep_knn, msfm_match_result** out);

/* ==================================================================================== *
 *  Bundle adjustment  
 * ==================================================================================== */

/* The problem BundleAdjuster::RunOptimizetion assembles (SfM/src/optimizer.cc:59-129)
 * gathered into flat arrays.  Gather order = point index ascending, then std::map key
 * order of the point's observations (optimizer.cc:62,80-82); bad points
 * (is_bad_estimated_, :64) are simply not gathered.
 *
 * Which reference functor an observation becomes follows optimizer.cc:86-125:
 *   pt_mutable & cam_mutable & model_mutable -> ReprojectionErrorPoseCamXYZ (2;6,3,3)
 *   pt_mutable & cam_mutable & !model_mutable-> ReprojectionErrorPoseXYZ    (2;6,3)
 *   pt_mutable & !cam_mutable                -> ReprojectionErrorXYZ        (2;3)
 *   !pt_mutable & cam_mutable & model_mutable-> ReprojectionErrorPoseCam    (2;6,3)
 *   !pt_mutable & cam_mutable & !model_mut.  -> ReprojectionErrorPose       (2;6)
 *   !pt_mutable & !cam_mutable               -> no residual
 */
typedef struct msfm_ba_problem {
  int n_cams, n_models, n_points, n_obs;
  double* cam_pose;            /* [n_cams][6] angle-axis, t (camera.cc:89-99)      in/out */
  double* cam_model;           /* [n_models][3] f, k1, k2 (basic_structs.h:83-90)  in/out */
  const int32_t* cam_model_of_cam; /* [n_cams]                                            */
  double* point;               /* [n_points][3] (structure.h:64)                   in/out */
  const int32_t* obs_cam;      /* [n_obs]                                                 */
  const int32_t* obs_pt;       /* [n_obs] non-decreasing                                  */
  const double* obs_xy;        /* [n_obs][2] centred pixels (database.cc:522-527)         */
  const double* pt_weight;     /* [n_points] residual weight (optimizer.cc:69-78)         */
  const uint8_t* cam_mutable;  /* [n_cams]   NULL = all mutable                           */
  const uint8_t* model_mutable;/* [n_models] NULL = all mutable (basic_structs.h:64)      */
  const uint8_t* pt_mutable;   /* [n_points] NULL = all mutable                           */
  /* Absolute GPS residual per camera (gps_error_pose_absolute.h:31-44, wiring
   * slam_gps.cc:818-830): r = [w|tx-x|, w|ty-y|, (w/5)|tz-z|] on pose[3:6], Huber(1).
   * NULL = none.  */
  const double* gps_xyz;       /* [n_cams][3] */
  double gps_weight;
} msfm_ba_problem;

/* Only the fields the reference sets (optimizer.cc:42-48; slam_gps.cc:681-684) plus the
 * Ceres 1.13 defaults it inherits, spelled out so that tests can vary them.
 * msfm_ba_options_default() fills the Ceres defaults. */
typedef struct msfm_ba_options {
  int max_num_iterations;        /* 200 (basic_structs.h:232); 100 in test_sfm.cc:35-36   */
  int num_threads;               /* passed through; the GPU path ignores it               */
  int progress_to_stdout;        /* minimizer_progress_to_stdout                          */
  double huber_delta;            /* 1.0 (optimizer.cc:84)                                 */
  double function_tolerance;     /* 1e-6                                                  */
  double gradient_tolerance;     /* 1e-10                                                 */
  double parameter_tolerance;    /* 1e-8                                                  */
  double initial_trust_region_radius; /* 1e4                                              */
  double max_trust_region_radius;     /* 1e16                                             */
  double min_trust_region_radius;     /* 1e-32                                            */
  double min_relative_decrease;       /* 1e-3                                             */
  double min_lm_diagonal;             /* 1e-6                                             */
  double max_lm_diagonal;             /* 1e32                                             */
  int max_num_consecutive_invalid_steps; /* 5                                             */
  int jacobi_scaling;                 /* 1                                                */
} msfm_ba_options;
void msfm_ba_options_default(msfm_ba_options* opt);

/* Termination (ceres::TerminationType + which test fired). */
#define MSFM_BA_CONVERGENCE_FUNCTION 1
#define MSFM_BA_CONVERGENCE_GRADIENT 2
#define MSFM_BA_CONVERGENCE_PARAMETER 3
#define MSFM_BA_NO_CONVERGENCE 4 /* iteration cap */
#define MSFM_BA_FAILURE 5        /* too many consecutive invalid steps */
#define MSFM_BA_MIN_RADIUS 6

/* One row of the Ceres progress table per iteration (row 0 = iteration 0). */
typedef struct msfm_ba_iteration {
  double cost;              /* after the iteration                                        */
  double cost_change;
  double gradient_max_norm;
  double step_norm;
  double relative_decrease; /* tr_ratio                                                   */
  double trust_region_radius;
  int32_t step_is_valid;
  int32_t step_is_successful;
} msfm_ba_iteration;

typedef struct msfm_ba_summary {
  int termination;
  int num_iterations;       /* rows written to `iterations` minus 1                       */
  int num_successful_steps;
  int num_unsuccessful_steps;
  double initial_cost, final_cost;
  int num_residuals;        /* scalar residual count                                      */
  int num_reduced_params;   /* order of the reduced camera system                         */
  msfm_ba_iteration* iterations; /* caller-provided, may be NULL                          */
  int iterations_capacity;
  double solve_ms;          /* device time of the LM loop, problem already resident       */
  double setup_ms;          /* upload + symbolic set-up (block-pair lists)                */
} msfm_ba_summary;

/* Replaces `ceres::Solve(options_, &problem_, &summary_)` at SfM/src/optimizer.cc:133 and
 * SfM/src/slam_gps.cc:841: trust-region Levenberg–Marquardt, Huber(1) loss, Jacobi
 * scaling, dense Schur complement on the points, dense Cholesky of the reduced camera
 * system, with the Ceres 1.13 control flow (step acceptance, radius update, stopping
 * rules).  Updates cam_pose / cam_model / point in place, like Ceres does through the
 * `data` blocks. */
int msfm_ba_solve(msfm_ctx* ctx, msfm_ba_problem* problem, const msfm_ba_options* options,
                  msfm_ba_summary* summary);

/* Split form, for callers that keep a problem resident across solves (bench loop,
 * windowed BA re-solves): create = upload + symbolic set-up, run = LM loop on the
 * resident state, download = copy parameters back, reset = re-upload parameters only. */
typedef struct msfm_ba msfm_ba;
int msfm_ba_create(msfm_ctx* ctx, const msfm_ba_problem* problem, msfm_ba** out);
int msfm_ba_run(msfm_ba* ba, const msfm_ba_options* options, msfm_ba_summary* summary);
int msfm_ba_upload_params(msfm_ba* ba, const double* cam_pose, const double* cam_model,
                          const double* point);
int msfm_ba_download_params(msfm_ba* ba, double* cam_pose, double* cam_model, double* point);
void msfm_ba_destroy(msfm_ba* ba);
/* How the reduced camera system of a resident problem is laid out and eliminated (no reference counterpart:
 * Ceres' DENSE_SCHUR factors S in the given camera order).  n_domains <= 1: dense order. */
typedef struct msfm_ba_layout {
  int reduced_order;      /* 6 * camera blocks + 3 * intrinsics blocks */
  int system_order;       /* order of the factored matrix: reduced_order + identity padding of the domains */
  int n_domains;          /* mutually uncoupled camera domains whose panel chains share launches */
  int domain_cols[8];     /* columns of each domain (multiples of 64) */
  int separator_cols;     /* every separator + intrinsics (everything behind the leaf domains) */
  int panel_launches;     /* panel launches per factorisation */
  /* the elimination tree behind it: level 0 = the leaf domains above, level 1.. = separators from the deepest cut to
   * the shallowest (nodes of one level are mutually uncoupled and share launches), then the root chain */
  int n_levels;
  int level_nodes[3];
  int level_begin[3];     /* first column of each level */
  int root_cols;          /* root separator + intrinsics: the final dense chain */
  /* Schur products formed inside the point kernel instead of by the gather kernels (0 everywhere: gather path only):
   * pair-list entries of the camera x camera list in all / folded; (workgroup, block) slots = 288-byte partial sums the
   * point kernel writes and the assembly reads; the same for the intrinsics x camera list (144-byte partials) */
  long long cc_entries, cc_entries_folded;
  int fold_slots, fold_passes;
  long long mc_entries, mc_entries_folded;
  int fold_mc_slots;
  int reserved_;
} msfm_ba_layout;
int msfm_ba_get_layout(const msfm_ba* ba, msfm_ba_layout* out);

/* The elimination order of msfm_ba_create as a host-only function (no device, no context: a diagnostic for tests and for
 * sizing): the nested dissection of a camera graph given as an n x n 0/1 adjacency matrix (cameras that see a common
 * eliminated point; the reduced camera matrix of optimizer.cc:133's Schur complement has a block exactly there).
 * tail_cols = columns that follow the last camera (3 per intrinsics block + 1), force_depth = -1 (choose) or 1..3.
 * label[c] = leaf domain of camera c (0 .. n_leaves - 1), or -(d + 1) for a camera of a separator cut at depth d (-1 = the
 * root separator).  *chain_steps = 64-column panel steps on the critical path (longest leaf + longest separator of every
 * depth + root).  *n_leaves = 0 (and every label 0): the dense order is kept. */
int msfm_camera_graph_dissection(int n_cams, const uint8_t* adjacency, int tail_cols, int force_depth, int32_t* label,
                                 int* n_leaves, int* chain_steps);

/* Multi-GPU: points (with all their observations) are sharded over ranks, cameras and
 * intrinsics are replicated; a few times per LM iteration `count` doubles at `buf_dev` (the per-camera J^T J sums, the
 * partial reduced system [S | rhs], a handful of scalars) must be reduced over ranks in place
 * with `op` (sum or max).
 * The host supplies the collective (RCCL all-reduce through torch.distributed in the
 * Python host, ncclAllReduce in a C++ host); it is called on the host thread that runs
 * the solve, after the producing kernels have been enqueued on `stream`, and must leave
 * the reduced data visible to work enqueued on `stream` afterwards.
 * The reference has no counterpart (no collective anywhere in SfM/src). */
#define MSFM_REDUCE_SUM 0
#define MSFM_REDUCE_MAX 1
typedef int (*msfm_allreduce_fn)(void* user, double* buf_dev, size_t count, int op, void* stream);
int msfm_ctx_set_allreduce(msfm_ctx* ctx, msfm_allreduce_fn fn, void* user, int rank,
                           int world_size);

/* The same collective supplied by the library itself: RCCL all-reduce over xGMI, one process per GPU (the
 * multi-GPU context owns the communicator).  librccl is loaded at run time by these two calls only, so a single-GPU
 * host has no RCCL dependency.  Rank 0 obtains an id, the host hands the 128 bytes to the other ranks by whatever it has
 * (MPI_Bcast, a file, torch.distributed.broadcast_object_list), then EVERY rank calls msfm_ctx_init_rccl - a collective
 * call - which creates the communicator on the context's device and installs ncclAllReduce (ncclDouble, sum / max,
 * in place, on the context's stream) as the reduction hook.  msfm_ctx_destroy releases the communicator. */
#define MSFM_RCCL_ID_BYTES 128
int msfm_rccl_get_unique_id(msfm_ctx* ctx, unsigned char id[MSFM_RCCL_ID_BYTES]);
int msfm_ctx_init_rccl(msfm_ctx* ctx, const unsigned char id[MSFM_RCCL_ID_BYTES], int rank, int world_size);
/* Runs the installed collective (host hook or the native RCCL one) on `count` doubles at `buf_dev`, in place, ordered on
 * the context's stream - what the solver does internally; exposed so that a host can reduce its own per-rank results
 * (the N x N match-count matrix of fine_matching_graph.cc:275-292, timing maxima) through the same communicator. */
int msfm_ctx_allreduce(msfm_ctx* ctx, double* buf_dev, size_t count, int op);

/* ==================================================================================== *
 *  Triangulation / reprojection  
 * ==================================================================================== */

/* Tracks in CSR form; cameras as the reference keeps them (camera.h:60-75):
 *   cam_R [n_cams][9] row-major R, cam_t [n_cams][3], cam_c [n_cams][3] centre,
 *   cam_fk [n_cams][3] = f, k1, k2 of the camera's CameraModel. */
typedef struct msfm_tracks {
  int n_tracks, n_cams;
  const int32_t* track_off; /* [n_tracks+1] */
  const int32_t* track_cam; /* [track_off[n_tracks]] */
  const double* track_xy;   /* [..][2] centred pixels */
  const double* cam_R;
  const double* cam_t;
  const double* cam_c;
  const double* cam_fk;
} msfm_tracks;

/* Point3D::Trianglate2 (SfM/src/structure.cc:211-265): ray-midpoint normal equations
 * solved by 4x4 LLT, then Reprojection() (:267-300) and
 * SufficientTriangulationAngle() (:325-355).
 *   X [n][3]; mse [n] (1e5 on negative depth, :280-284); ok [n] = return value.
 * A failed LLT leaves X untouched and ok = 0 (:248-251); X must therefore be
 * initialised by the caller. th_angle in radians. */
int msfm_triangulate_midpoint_batch(msfm_ctx* ctx, const msfm_tracks* tracks, double th_error,
                                    double th_angle, double* X, double* mse, uint8_t* ok);
/* Point3D::Trianglate (SfM/src/structure.cc:163-209): DLT rows :179-182, last right
 * singular vector (:187), same acceptance test.  Tracks with < 2 views return ok = 0. */
int msfm_triangulate_dlt_batch(msfm_ctx* ctx, const msfm_tracks* tracks, double th_error,
                               double th_angle, double* X, double* mse, uint8_t* ok);
/* Point3D::Reprojection (SfM/src/structure.cc:267-300) for given X; this is what
 * IncrementalSfM::RemovePointOutliers recomputes per point (sfm_incremental.cc:1831-1863). */
int msfm_reproject_mse_batch(msfm_ctx* ctx, const msfm_tracks* tracks, const double* X,
                             double* mse);

/* Closed-form fundamental-matrix filter (SfM/src/utils/geo_verification.cc:60-79):
 * l = F*[x1,y1,1]; l /= hypot(l0,l1); inlier iff |l . [x2,y2,1]| < th (3.0).
 * pt1/pt2 are float pixel pairs as cv::Point2f; inlier[n] gets 0/1. */
int msfm_epipolar_filter(msfm_ctx* ctx, const float* pt1, const float* pt2, int n,
                         const double F[9], double th, uint8_t* inlier);

/* GeoVerification::GeoVerificationFundamental (SfM/src/utils/geo_verification.cc:30-58), batched over image
 * pairs: cv::findFundamentalMat(pt1, pt2, status, cv::FM_RANSAC, 3.0) followed by the >= 30 inliers gate
 * (:34-36, :54-56).  OpenCV 2.4's FM_RANSAC restated: 7-point minimal solver (up to 3 models per sample),
 * error = max of the two squared point-to-epipolar-line distances <= threshold^2, confidence 0.99, at most
 * 2000 samples with the adaptive stop of cvRANSACUpdateNumIters, best model returned without refit.  The
 * sampler is counter based: sample h of pair p (its index in this call) depends only on (seed, p, h).
 * Pair p owns matches [offsets[p], offsets[p+1]); pt1/pt2 are cv::Point2f pairs (centred pixels).
 * Out: F[p][9] row-major (x2^T F x1 = 0, F[8] = 1 when possible; zeros when no model), inlier[total] 0/1
 * against the returned F, n_inliers[p], ok[p] = the bool GeoVerificationFundamental returns. */
typedef struct msfm_fransac_options {
  double threshold;      /* 3.0  (th_epipolar1, geo_verification.cc:44) */
  double confidence;     /* 0.99 (findFundamentalMat default param2)    */
  int max_iterations;    /* 2000 (CvModelEstimator2::runRANSAC default) */
  int min_points;        /* 30   (geo_verification.cc:34)               */
  int min_inliers;       /* 30   (geo_verification.cc:54)               */
  uint64_t seed;
} msfm_fransac_options;
void msfm_fransac_default_options(msfm_fransac_options* opt);
int msfm_fundamental_ransac_batch(msfm_ctx* ctx, int n_pairs, const int* offsets, const float* pt1,
                                  const float* pt2, const msfm_fransac_options* opt, double* F,
                                  uint8_t* inlier, int* n_inliers, uint8_t* ok);
/* The closed-form filter above for many pairs at once (fine_matching_graph.cc:148-150: applied to the
 * "all" match set only when the pair's RANSAC succeeded): pairs with ok[p] == 0 get all-zero masks
 * (ok may be NULL = every pair). */
int msfm_epipolar_filter_batch(msfm_ctx* ctx, int n_pairs, const int* offsets, const float* pt1,
                               const float* pt2, const double* F, const uint8_t* ok, double th,
                               uint8_t* inlier);

/* ======================================================================================
 *  Track building between matching and triangulation  
 * ====================================================================================== */
/* The data association of SLAMGPS::Triangulation (SfM/src/slam_gps.cc:565-635): walk the image pairs in the
 * caller's order (the reference: idx1 ascending, idx2 ascending over match_graph_[idx1][idx2] > 0) and their matches
 * (feature in idx1, feature in idx2): a match whose first feature already belongs to a point adds the second
 * feature to it (:597-606), else one whose second feature belongs to a point adds the first (:607-616), else a new
 * point is created (:617-633).  std::map::insert semantics are kept: a point holds at most one observation per
 * image (the first), a feature stays with the first point it was mapped to, two existing points are never merged.
 * msfm_tracks_build is that walk on the host; msfm_tracks_build_device gives the identical result from the GPU (first
 * appearances by atomicMin, the "joins" forest, a stable sort of the observations by (point, image)) - the form for the
 * match lists of a whole image set (config 3: 6 M matches).  The tracks then go to msfm_triangulate_midpoint_batch (th_tri_angle = 3 degrees, points with fewer
 * than 3 views or a failed triangulation are marked bad, :638-648).
 * Out: CSR tracks, observations of a track in ascending image order (std::map iteration order). */
typedef struct msfm_track_set msfm_track_set;
int msfm_tracks_build(int n_images, const int* n_features, int n_pairs, const int* pair_img /*[n_pairs][2]*/,
                      const int* match_off /*[n_pairs+1]*/, const int* matches /*[match_off[n_pairs]][2]*/,
                      msfm_track_set** out);
int msfm_tracks_build_device(msfm_ctx* ctx, int n_images, const int* n_features, int n_pairs, const int* pair_img,
                             const int* match_off, const int* matches, msfm_track_set** out);
int msfm_track_set_size(const msfm_track_set* set, int* n_tracks, int* n_observations);
int msfm_track_set_fetch(const msfm_track_set* set, int* track_off /*[n_tracks+1]*/, int* obs_image, int* obs_feature);
void msfm_track_set_destroy(msfm_track_set* set);

/* ======================================================================================
 *  The same chain with every intermediate result resident on the GPU
 * ====================================================================================== */
/* match codes -> geometric verification -> track building -> triangulation -> bundle adjustment without a host round trip
 * in between: the hand-over the reference makes through std::vector / std::map objects per pair and per point
 * (fine_matching_graph.cc:116-186 -> slam_gps.cc:557-648 -> optimizer.cc:59-133; incremental form
 * sfm_incremental.cc:755-915).  Each step runs the kernels of its host-array counterpart on device buffers and gives
 * the same result bit for bit; between the descriptor / keypoint upload and msfm_ba_download_params only a few integers
 * per image pair and the cameras cross PCIe.  Image index = camera index.
 *   msfm_chain_create       takes the codes of a msfm_match_pairs result (not the SLAM form) whose descriptor set holds the
 *                           keypoints of every image involved (msfm_descset_upload_keypoints)
 *   msfm_chain_verify       per pair: GeoVerificationFundamental on matches_good (msfm_fundamental_ransac_batch, pair p of the
 *                           list = pair p of the sampler), then, if it succeeded, the closed-form filter with th_filter (3.0)
 *                           on matches_all; a pair keeps the surviving matches_all entries, a failed pair none
 *                           (fine_matching_graph.cc:138-153, :182-186)
 *   msfm_chain_build_tracks msfm_tracks_build_device on those matches, pairs in the order of the match result
 *   msfm_chain_triangulate  msfm_triangulate_midpoint_batch on every track (X starts at 0), observations from the keypoints
 *   msfm_chain_ba_create    msfm_ba_create on the tracks with ok = 1 and >= min_views observations (slam_gps.cc:638-648:
 *                           3), gather order and weights of optimizer.cc:59-129 (2 views -> 1.0, more -> weight_ge3);
 *                           cameras / intrinsics from the caller (host), every block mutable.  The msfm_ba is the caller's.
 * The fetch functions copy a stage's result to the host (tests, file writers); none is needed to go on. */
typedef struct msfm_chain msfm_chain;
int msfm_chain_create(msfm_match_result* res, msfm_chain** out);
int msfm_chain_verify(msfm_chain* chain, msfm_match_result* res, const msfm_fransac_options* opt, double th_filter);
int msfm_chain_matches(msfm_chain* chain, int* n_matches /*[n_pairs]*/, uint8_t* ok /*[n_pairs]*/, double* F /*[n_pairs][9]*/);
int msfm_chain_fetch_matches(msfm_chain* chain, int pair, int* matches /*[n_matches[pair]][2]*/);
int msfm_chain_build_tracks(msfm_chain* chain, int* n_tracks, int* n_observations);
int msfm_chain_fetch_tracks(msfm_chain* chain, int* track_off, int* obs_image, int* obs_feature);
int msfm_chain_triangulate(msfm_chain* chain, int n_cams, const double* cam_R, const double* cam_t, const double* cam_c,
                           const double* cam_fk, double th_error, double th_angle, int* n_accepted);
int msfm_chain_fetch_points(msfm_chain* chain, double* X /*[n_tracks][3]*/, double* mse, uint8_t* ok);
int msfm_chain_ba_create(msfm_chain* chain, int n_cams, int n_models, double* cam_pose, double* cam_model,
                         const int32_t* cam_model_of_cam, int min_views, double weight_ge3, msfm_ba** out, int* n_points,
                         int* n_observations);
/* track index of every point of the bundle adjustment (to put the adjusted points back) */
int msfm_chain_fetch_point_tracks(msfm_chain* chain, int* track_of_point /*[n_points]*/);
void msfm_chain_destroy(msfm_chain* chain);

/* ======================================================================================
 *  Pose initialisers ahead of each bundle adjustment  
 * ====================================================================================== */
/* AbsolutePoseEstimation::AbsolutePoseWithFocalLength (SfM/src/orientation/absolute_pose_estimation.cc:42-58, called
 * when an image is localised against the model, sfm_incremental.cc:646) for a batch of images.  Per image:
 * AbsolutePoseEPNP::EPNPRansac (absolute_pose_via_epnp.cc:103-139) - `max_iter` (reference: 200) samples of 4
 * correspondences, EPnP on each (:142-185; compute_pose :472-519 with OpenCV's Jacobi SVD restated), the sample
 * whose error over its own four points is smallest is kept - then AbsolutePoseEstimation::Error over all
 * correspondences (:67-103).  The reference samples with std::random_shuffle; here sample `it` of image `p` is a
 * pure function of (seed, p, it), so results do not depend on the batch split.
 * In : offsets[n+1] delimits each image's correspondences; pts_w [total][3] world points, pts_2d [total][2] centred
 *      pixels, f[n] focal lengths.
 * Out: R [n][9] row-major and t [n][3] with Xc = R Xw + t (zeros when no sample qualified or fewer than 4 points);
 *      errors [total] = reprojection error, 1000.0 where it is >= 10 px; avg_error [n] = rms of the errors < 10 px
 *      (10000.0 when there is none) - the value compared with th_mse_localization (sfm_incremental.cc:648);
 *      best_iter [n] (may be NULL) = index of the kept sample, -1 if none. */
int msfm_epnp_ransac_batch(msfm_ctx* ctx, int n_problems, const int* offsets, const double* pts_w,
                           const double* pts_2d, const double* f, int max_iter, uint64_t seed, double* R,
                           double* t, double* errors, double* avg_error, int* best_iter);
/* RelativePoseEstimation::RelativePoseWithFocalLength (SfM/src/orientation/relative_pose_estimation.cc:91-120,
 * called for the seed pair, sfm_incremental.cc:309) for a batch of image pairs.  Per pair, on points divided by the
 * focal lengths: EssentialMatrixFivePoints::FivePointEssentialMatrixRANSAC (essential_matrix_five_point.cc:30-92) -
 * `ransac_times` (reference: 100) samples of 5 matches (all matches at once when there are 5..9), Nister's solver
 * through the 10x10 action matrix (:97-178; Eigen's FullPivLU / EigenSolver restated), every real solution scored
 * by the Sampson sum over all matches (:333-349), fewer than 4 solutions in total = failure - then
 * RelativePoseFromEssentialMatrix::ReltivePoseFromEMatrix (relative_pose_from_essential_matrix.cc:33-104): SVD of
 * E, four (R, t) hypotheses, each match votes for the first hypothesis that puts it in front of both cameras.
 * Out: E [n][9] row-major with x_cur^T E x_ref = 0 on (pixel / f, 1); R [n][9], t [n][3] exactly as the reference
 *      returns them in RTPoseRelative; ok [n]; n_candidates [n] (may be NULL) = number of essential matrices scored. */
int msfm_relpose_5pt_batch(msfm_ctx* ctx, int n_pairs, const int* offsets, const double* pts_ref,
                           const double* pts_cur, const double* f_ref, const double* f_cur, int ransac_times,
                           uint64_t seed, double* E, double* R, double* t, uint8_t* ok, int* n_candidates);

/* ==================================================================================== *
 *  Single-process multi-GPU context
 * ==================================================================================== */

/* ---- one process, several GPUs (threading row: "multi-GPU context msfm_ctx_create_multi(n_gpus) owns the
 * communicator") ----
 * The reference's entry point is ONE process (SfM/test/test_sfm/test_sfm.cc:22-70 `main`; BundleAdjuster::RunOptimizetion,
 * optimizer.cc:59-133, is called from a single thread).  A multi context keeps that shape: it owns one msfm_ctx per device,
 * a host thread per device inside the library, and the communicator between them - RCCL's ncclCommInitAll over xGMI when
 * the devices are distinct, an in-process reduction (host barrier + a device-side sum in rank order) when the contexts share
 * one device (`devices` may name the same device more than once: how the path is tested on a one-GPU box).
 *   devices == NULL: devices 0 .. n_gpus-1.
 * The msfm_multi_* calls below are the single-context calls with the split inside: same arguments, same results
 * (bundle adjustment to 1e-9 of the one-context solve - the sums are formed in another order; everything else bit for bit). */
typedef struct msfm_multi msfm_multi;
int msfm_ctx_create_multi(int n_gpus, const int* devices, msfm_multi** out);
void msfm_multi_destroy(msfm_multi* mc);
int msfm_multi_size(const msfm_multi* mc);
msfm_ctx* msfm_multi_ctx(msfm_multi* mc, int rank);          /* the context of rank r (rank 0: where single-context work goes) */
const char* msfm_multi_last_error(const msfm_multi* mc);
/* ceres::Solve (optimizer.cc:133) with the points - and all their observations - split over the contexts in contiguous
 * ranges balanced by the work the mutability masks leave, cameras and intrinsics replicated, three reductions per linear
 * solve over the communicator.  Arguments and results as msfm_ba_solve. */
int msfm_multi_ba_solve(msfm_multi* mc, msfm_ba_problem* problem, const msfm_ba_options* options, msfm_ba_summary* summary);
/* Point3D::Trianglate2 / Trianglate / Reprojection (structure.cc:163-300) with the tracks split by observation count; no
 * collective (tracks are independent). */
int msfm_multi_triangulate_midpoint_batch(msfm_multi* mc, const msfm_tracks* tracks, double th_error, double th_angle, double* X,
                                          double* mse, uint8_t* ok);
int msfm_multi_triangulate_dlt_batch(msfm_multi* mc, const msfm_tracks* tracks, double th_error, double th_angle, double* X, double* mse,
                                     uint8_t* ok);
int msfm_multi_reproject_mse_batch(msfm_multi* mc, const msfm_tracks* tracks, const double* X, double* mse);
/* The matching loop of FineMatchingGraph::BuildMatchGraph (fine_matching_graph.cc:87-133) over a pair list, the idx1-major
 * list cut into contiguous slices balanced by M1 * M2, one per context; no collective (pairs are independent).
 *   desc[i]: [count[i]][dim] float descriptors of image i (host); pairs [n_pairs][2] = (idx1, idx2);
 *   code[p]: count[idx2 of pair p] match codes (MSFM_MATCH_*), n_all / n_good [n_pairs] (either may be NULL). */
int msfm_multi_match_pairs(msfm_multi* mc, int n_images, const float* const* desc, const int* count, int dim, const int* pairs, int n_pairs,
                           float ratio_good, float ratio_all, int32_t* const* code, int* n_all, int* n_good);

#ifdef __cplusplus
}
#endif
#endif /* MSFM_H_ */
