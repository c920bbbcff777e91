/*
 * cfear_hip.h -- C-ABI of libcfear_hip.so: the MI355X (gfx950) implementation of the CFEAR
 * scan-registration hot path of TBV Radar SLAM (dan11003/tbv_slam_public).
 *
 * The reference has no FFI layer: its boundary is the C++ class API of the catkin library
 * `cfear_radarodometry` (cfear_radarodometry/CMakeLists.txt:46-72).  Each entry point below names
 * the reference function(s) it replaces (paths relative to
 * cfear_radarodometry/{include,src}/cfear_radarodometry/).  INTEGRATION.md shows the shim a
 * maintainer adds inside radarDriver / MapPointNormal / n_scan_normal_reg to call them.
 * Three neighbours of the path follow the same rules further down: covariance by cost sampling
 * (odometrykeyframefuser.cpp:261-380), CorAl alignment quality (coral_alignment_quality) and the radar
 * Scan Context arithmetic (place_recognition_radar) -- SURVEY.md 8(f).
 *
 * Conventions
 *   - plain C: pointers + sizes, caller-owned buffers, opaque handles, no C++/torch types;
 *   - every function returns an int status (CFEAR_OK = 0, < 0 = error); nothing exits or throws
 *     (the reference calls exit(0) on several errors, e.g. pointnormal.cpp:72-75);
 *   - data pointers may be HOST or DEVICE memory of the context's GPU (detected with
 *     hipPointerGetAttributes); all buffers of one call must live in the same space;
 *   - one cfear_ctx per host thread / HIP stream; calls on different contexts are independent
 *     (the reference builds a fresh n_scan_normal_reg per loop-closure candidate on the
 *     loop-closure thread, tbv_slam/src/tbv_slam/loopclosure.cpp:56);
 *   - calls with host pointers are synchronous; calls with device pointers are enqueued on the
 *     context's stream (use cfear_ctx_synchronize) unless they return values through host
 *     scalars, in which case they synchronise themselves;
 *   - float parameters that the reference holds as float (radarDriver::Parameters,
 *     radar_driver.h:40-45) are float here and widened to double exactly where the reference
 *     widens them.
 */
#ifndef CFEAR_HIP_H
#define CFEAR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFEAR_ABI_VERSION 1

/* ---- status codes ------------------------------------------------------------------------ */
#define CFEAR_OK 0
#define CFEAR_ERR_INVALID_ARGUMENT (-1)
#define CFEAR_ERR_HIP (-2)              /* a HIP runtime call failed; see cfear_last_error     */
#define CFEAR_ERR_CAPACITY (-3)         /* an output/workspace capacity was exceeded           */
#define CFEAR_ERR_TOO_FEW_RESIDUALS (-4)/* n_scan_normal.cpp:368-369, 444-447                  */
#define CFEAR_ERR_SOLVER (-5)           /* !summary_.IsSolutionUsable() (n_scan_normal.cpp:449)*/
#define CFEAR_ERR_EMPTY_CLOUD (-6)      /* pointnormal.cpp:72-75 ("error, cloud empty")        */
#define CFEAR_ERR_NO_DEVICE (-7)        /* no HIP device / kernels not loadable                */
#define CFEAR_ERR_IO (-8)               /* a file could not be opened / read / written         */
#define CFEAR_ERR_FORMAT (-9)           /* a file is not a simple_graph archive this reader understands */

/* ---- enums (values follow the reference's enums) ------------------------------------------ */
enum cfear_cost_metric { CFEAR_P2P = 0, CFEAR_P2L = 1, CFEAR_P2D = 2 };          /* registration.h:55 */
enum cfear_loss_type { CFEAR_LOSS_NONE = 0, CFEAR_LOSS_HUBER = 1, CFEAR_LOSS_CAUCHY = 2,
                       CFEAR_LOSS_SOFTLONE = 3, CFEAR_LOSS_COMBINED = 4,
                       CFEAR_LOSS_TUKEY = 5 };                                    /* registration.h:60 */
enum cfear_weight_option { CFEAR_W_UNIFORM = 0, CFEAR_W_SIM_N = 1, CFEAR_W_SIM_DIRECTION = 2,
                           CFEAR_W_SIM_SCALE = 3, CFEAR_W_COMBINED = 4 };         /* registration.h:50 */
enum cfear_filter_type { CFEAR_FILTER_KSTRONG = 0, CFEAR_FILTER_CACFAR = 1 };     /* radar_driver.h:25 */

/* ---- context ------------------------------------------------------------------------------ */
typedef struct cfear_ctx cfear_ctx;

int cfear_abi_version(void);
const char* cfear_status_string(int status);
/* hip_stream: a hipStream_t to enqueue on (e.g. torch's current stream; hipStreamLegacy / hipStreamPerThread are
 * accepted too), or NULL to create a private non-blocking stream.                                   */
int cfear_ctx_create(int device, void* hip_stream, cfear_ctx** out);
int cfear_ctx_destroy(cfear_ctx* ctx);
int cfear_ctx_synchronize(cfear_ctx* ctx);
const char* cfear_last_error(const cfear_ctx* ctx);
/* The hipStream_t the context enqueues on (the caller's, or the private one cfear_ctx_create made): lets the host
 * order other work -- a collective, a copy -- behind the library's kernels without a host synchronisation.        */
int cfear_ctx_get_stream(const cfear_ctx* ctx, void** hip_stream);
/* Context options.  None is needed in production: they are TEST / MEASUREMENT hooks that select between routes the
 * library otherwise chooses by itself, so that the parity tests can drive every route with ordinary inputs and an A/B
 * run can compare two routes inside one process.  The library never reads the environment.
 *   CFEAR_OPT_FUSED_DECODE   1 (default): [range bins][azimuths] sweeps are decoded inside the filter kernels
 *                            (radar_driver.cpp:74-90 fused into the sweep); 0: rotation kernel + row sweep.  Read when
 *                            an odometry object is created and per filter call.
 *   CFEAR_OPT_MATCHER_LDS_KB 0 (default): the matcher sizes its LDS by the batch; 8..160: every registration runs in
 *                            that many KB, so that ordinary scans exercise the keyframe groups and the global tail of
 *                            the correspondence arrays (which only unusually large registrations reach otherwise).
 *   CFEAR_OPT_MATCHER_WAVES  0 (default): wavefronts per registration chosen by the batch; 2, 4, 8, 16 force a form.
 *   CFEAR_OPT_HOST_TIMELINE  1: the batched odometry prints where the host spends a frame (every 256 calls).
 * Returns CFEAR_ERR_INVALID_ARGUMENT for an unknown option or a value outside its range.                          */
enum cfear_option { CFEAR_OPT_FUSED_DECODE = 0, CFEAR_OPT_MATCHER_LDS_KB = 1, CFEAR_OPT_MATCHER_WAVES = 2,
                    CFEAR_OPT_HOST_TIMELINE = 3, CFEAR_OPT_COUNT = 4 };
int cfear_ctx_set_option(cfear_ctx* ctx, int32_t option, int64_t value);
int cfear_ctx_get_option(const cfear_ctx* ctx, int32_t option, int64_t* value);
/* Per-kernel-family device time measured with hipEvents on the context's stream.
 * enable=1 brackets every launch with events (adds a sync per read-out, not per launch); enable=2 only the polar
 * filter's row kernels (kstrongest_rows / cacfar_rows: the one HBM-bound launch of the path), which costs a batched
 * pipeline 0.3 % instead of 1.7 %.
 * cfear_ctx_profile_read: names[i] (static strings), total_ms[i], launches[i], up to cap rows;
 * returns the number of rows; reset != 0 clears the accumulators.                             */
int cfear_ctx_profile_enable(cfear_ctx* ctx, int enable);
int cfear_ctx_profile_read(cfear_ctx* ctx, const char** names, double* total_ms, int64_t* launches,
                           int cap, int reset);

/* ---- F: polar filters ----------------------------------------------------------------------
 * Replaces radarDriver::Process (radar_driver.cpp:48-73) =
 *   StructuredKStrongest ctor + FilterKstrongest      radar_filters.cpp:198-237
 *   getPeaksFilteredPointCloud(cloud,false)           radar_filters.cpp:300-337
 *   getPeaksFilteredPointCloud(peaks,true) -> AxialNonMaxSupress   radar_filters.cpp:238-298
 *   AzimuthCACFAR::getFilteredPointCloud               cfar.cpp:35-71
 * Image: row-major uint8, rows = azimuths, cols = range bins, `stride` bytes between rows,
 * `batch_stride` bytes between the `batch` images.                                            */
typedef struct cfear_polar_desc {
  int32_t rows, cols, stride, batch;
  int64_t batch_stride;
} cfear_polar_desc;

/* Image decode of the non-Oxford sensors.  Replaces cv::rotate(image, image, ROTATE_90_COUNTERCLOCKWISE) in
 * radarDriver::Callback (radar_driver.cpp:74-90): those drivers publish the sweep as [range bins][azimuths];
 * the filters want rows = azimuths.  src_desc describes the SOURCE images (rows = range bins, cols = azimuths);
 * dst holds batch images of cols x rows bytes, dst[i][j] = src[j][cols - 1 - i].  src and dst both host or both
 * device; they must not overlap.                                                                             */
int cfear_polar_rotate_ccw(cfear_ctx* ctx, const uint8_t* src, const cfear_polar_desc* src_desc, uint8_t* dst,
                           int32_t dst_stride, int64_t dst_batch_stride);

typedef struct cfear_kstrong_params {   /* radarDriver::Parameters, radar_driver.h:40-45 */
  int32_t k_strongest;                  /* >= 1, <= 1024 */
  float z_min;                          /* converted float -> int -> uchar like radar_driver.cpp:58 */
  float range_res;
  float min_distance;
  int32_t want_peaks;                   /* also run AxialNonMaxSupress */
} cfear_kstrong_params;

/* All output pointers are optional (NULL = not wanted) and per image b of the batch:
 *   sel_range     int32 [batch][rows][k]  range bins kept per azimuth, ascending (intensity,range)
 *                                          order (= dense_filtered_), padded with -1
 *   sel_intensity uint8 [batch][rows][k]  their intensities, padded with 0
 *   sel_count     int32 [batch][rows]
 *   is_peak       uint8 [batch][rows][k]  1 where AxialNonMaxSupress keeps the bin (want_peaks)
 *   xyzi          float [batch][rows*k][4] compacted PointXYZI cloud (x,y,z=0,intensity),
 *                                          rows ascending, only bins > ceil(min_distance/range_res)
 *   n_points      int32 [batch]
 *   xyzi_peaks / n_peaks: the same for the peaks cloud (want_peaks)                           */
typedef struct cfear_kstrong_out {
  int32_t* sel_range;
  uint8_t* sel_intensity;
  int32_t* sel_count;
  uint8_t* is_peak;
  float* xyzi;
  int32_t* n_points;
  float* xyzi_peaks;
  int32_t* n_peaks;
} cfear_kstrong_out;

int cfear_filter_kstrongest(cfear_ctx* ctx, const uint8_t* polar, const cfear_polar_desc* desc,
                            const cfear_kstrong_params* par, const cfear_kstrong_out* out);

/* The filter stage of the batched odometry on its own (device memory only): radarDriver::Callback's decode
 * (radar_driver.cpp:74-90), FilterKstrongest (radar_filters.cpp:209-237) and the selection of
 * getPeaksFilteredPointCloud(cloud, false) (radar_filters.cpp:309-337) in ONE pass over the sweep.  Per azimuth row the kept
 * bins beyond ceil(min_distance / range_res) as packed keys (intensity << 24 | range bin), in the reference's cloud order
 * (ascending (intensity, range)):
 *   row_keys   uint32 [batch][azimuths][k]     (k = k_strongest <= 64; entries beyond the row's count are unspecified)
 *   row_counts int32  [batch][azimuths][2]     {kept bins of the row, 0}
 * flags: CFEAR_ROWKEYS_BINS_MAJOR -- the images are [range bins][azimuths] (desc->rows = bins, desc->cols = azimuths) and
 * row r of the result is source column cols - 1 - r (cv::ROTATE_90_COUNTERCLOCKWISE).  The rotated image is never built:
 * one streaming pass lists, per azimuth, the bins >= uchar(z_min) (the only ones the filter can keep), a second picks the
 * k strongest of each list; azimuths with more than 256 such bins have their 16-column tile transposed in LDS and swept
 * there (CFEAR_ROWKEYS_TILE_SWEEP: every tile takes that route).  Needs 16-byte aligned images, cols % 16 == 0, rows % 4 == 0
 * and <= 4096 bins; any other geometry, a batch of fewer than 128 images (unless a route flag asks otherwise), or
 * CFEAR_ROWKEYS_TWO_PASS, takes cfear_polar_rotate_ccw's kernel into a workspace first -- same result.  The lists cost time
 * per bin >= z_min: beyond ~80 of them per azimuth CFEAR_ROWKEYS_TWO_PASS is the quicker route (the batched odometry measures
 * this and switches by itself).  want_peaks is ignored.                                                                        */
#define CFEAR_ROWKEYS_BINS_MAJOR 1
#define CFEAR_ROWKEYS_TWO_PASS 2
#define CFEAR_ROWKEYS_TILE_SWEEP 4
#define CFEAR_ROWKEYS_ROUTE_LISTS 16   /* candidate lists in global memory (default only when an image's lists do not fit the LDS) */
#define CFEAR_ROWKEYS_ROUTE_IMAGE 32   /* one workgroup per image, lists in LDS (the default route), whatever the batch size */
int cfear_filter_kstrongest_rowkeys(cfear_ctx* ctx, const uint8_t* polar, const cfear_polar_desc* desc,
                                    const cfear_kstrong_params* par, int32_t flags, uint32_t* row_keys, int32_t* row_counts);

/* Legacy filter: replaces k_strongest_filter / InsertStrongestK (radar_filters.cpp:25-78), which CorAl's standalone
 * kstrongRadar scan type still calls (coral_alignment_quality/src/alignment_checker/ScanType.cpp:104-114); TBV itself uses
 * the structured filter above.  Different rule: the first bin >= z_min of a row sets a floor, later bins
 * <= the running minimum are rejected even while fewer than k are held, ties at the cut keep the SMALLER ranges, points
 * sit at bin EDGES (range_res * bin) with a float azimuth, and the near cut is x^2 + y^2 > min_distance^2.  z_min,
 * range_res, min_distance are doubles as in the reference's signature.  xyzi float [batch][cap_points][4] (rows ascending,
 * within a row descending intensity, ascending range on ties -- the order of a stable sort; libstdc++'s std::sort is stable
 * up to 16 elements, i.e. k <= 15), n_points int32 [batch]; all three buffers host or all device.               */
int cfear_filter_kstrongest_legacy(cfear_ctx* ctx, const uint8_t* polar, const cfear_polar_desc* desc, int32_t k_strongest,
                                   double z_min, double range_res, double min_distance, float* xyzi, int32_t* n_points,
                                   int32_t cap_points);

typedef struct cfear_cacfar_params {    /* AzimuthCACFAR ctor, cfar.cpp:28-33; radar_driver.cpp:54 */
  int32_t window_size;                  /* cells per side */
  int32_t nb_guard_cells;
  float false_alarm_rate;
  float range_res;
  float z_min;                          /* static_threshold */
  float min_distance;
  double max_distance;                  /* radar_driver.cpp:54 passes 400.0 */
} cfear_cacfar_params;

/* xyzi float [batch][cap_points][4], n_points int32 [batch]; det_mask (optional) uint8
 * [batch][rows][cols] 0/1 per bin.  Detections are ordered (row, bin) like cfar.cpp:37-70.
 * A batch element with more than cap_points detections yields CFEAR_ERR_CAPACITY.            */
int cfear_filter_cacfar(cfear_ctx* ctx, const uint8_t* polar, const cfear_polar_desc* desc,
                        const cfear_cacfar_params* par, float* xyzi, int32_t* n_points,
                        int32_t cap_points, uint8_t* det_mask);

/* ---- C: motion compensation ----------------------------------------------------------------
 * Replaces Compensate(cloud, mot, ccw)  utils.cpp:96-107 (+ GetRelTimeStamp utils.h:28-32).
 * xyzi float [n][4] modified in place; mot = (x, y, theta) of the previous motion.           */
int cfear_compensate(cfear_ctx* ctx, float* xyzi, int32_t n, const double mot[3], int32_t ccw);

/* ---- N: oriented surface points ------------------------------------------------------------
 * cfear_scan is the device-resident MapPointNormal (pointnormal.h:110-243): the `cell`s plus the
 * float copy of their means that the matcher searches (pointnormal.cpp:151-162).              */
typedef struct cfear_scan cfear_scan;

typedef struct cfear_cell {             /* class cell, pointnormal.h:45-105 */
  double mean[2];                       /* u_ */
  double normal[2];                     /* snormal_ */
  double cov[4];                        /* cov_ row-major */
  double scale;                         /* scale_ (GetPlanarity) */
  double avg_intensity;                 /* avg_intensity_ */
  double lambda_min, lambda_max;
  int32_t nsamples;                     /* Nsamples_ */
  int32_t pad;
} cfear_cell;

typedef struct cfear_feature_params {   /* MapPointNormal ctor arguments, pointnormal.cpp:65 */
  float radius;                         /* par.res */
  double downsample_factor;             /* MapPointNormal::downsample_factor (static, =1) */
  double origin[2];
  int32_t weight_intensity;
  int32_t compensate;                   /* 1: run Compensate(mot, ccw) on the cloud first
                                           (odometrykeyframefuser.cpp:146-150), in place */
  double mot[3];
  int32_t ccw;
  int32_t pad;
} cfear_feature_params;

/* Replaces `new MapPointNormal(cloud, res, origin, weight_intensity, false)`
 * (pointnormal.cpp:65-90 -> ComputeNormals :265-297 -> cell::cell :7-63 ->
 * ComputeSearchTreeFromCells :151-162).  xyzi [n][4] host or device (modified in place only if
 * compensate).  n == 0 -> CFEAR_ERR_EMPTY_CLOUD.                                              */
int cfear_scan_create(cfear_ctx* ctx, float* xyzi, int32_t n, const cfear_feature_params* par,
                      cfear_scan** out);
/* Upload precomputed cells (loop closure consumes the MapPointNormal cached in each graph node,
 * types.h:119-122; also `raw` mode, pointnormal.cpp:76-82).  cells: host array.               */
int cfear_scan_from_cells(cfear_ctx* ctx, const cfear_cell* cells, int32_t n_cells, cfear_scan** out);
int cfear_scan_size(const cfear_scan* scan);                            /* GetSize()  */
int cfear_scan_get_cells(const cfear_scan* scan, cfear_cell* out_host, int32_t cap);  /* GetCells() */
/* MapPointNormal::GetClosestIdx (pointnormal.cpp:238-254) for n_queries points p (x, y doubles): idx[i] = index of
 * the cell whose float mean is nearest to float(p) (FLANN L2_Simple in float, lowest index on ties) if that squared
 * distance is < d * d, else -1 (the reference returns an empty vector).  queries_xy / idx both host or both device. */
int cfear_scan_closest_idx(const cfear_scan* scan, const double* queries_xy, int32_t n_queries, double d, int32_t* idx);
int cfear_scan_destroy(cfear_scan* scan);

/* ---- M: registration -----------------------------------------------------------------------
 * n_scan_normal_reg (n_scan_normal.h:27-85, n_scan_normal.cpp) in mode
 * incremental_last_to_previous: scans[0..n-2] are fixed targets, scans[n-1] is the free source. */
typedef struct cfear_reg_params {
  int32_t cost;                         /* cfear_cost_metric; n_scan_normal_reg ctor */
  int32_t loss;                         /* cfear_loss_type   */
  double loss_limit;                    /* loss_limit_ (0.1) */
  int32_t weight_opt;                   /* cfear_weight_option */
  int32_t max_itr_association;          /* SetParameters(.,) / default 8 */
  int32_t max_itr_solver;               /* options_.max_num_iterations, default 20 */
  int32_t min_itr;                      /* min_itr_ = 3 */
  double radius;                        /* radius_ = 2.0 (registration.h:122) */
  double cov_scale;                     /* SetD2dPar */
  double regularization;                /* SetD2dPar */
  double score_tolerance;               /* 1e-5 (n_scan_normal.h:74) */
  int32_t itr;                          /* GetCost only: the object's leftover itr_ (0 if fresh) */
  int32_t pad;
} cfear_reg_params;

/* Fills the n_scan_normal_reg defaults (P2L, Huber 0.1, Uniform, 8 x 20, radius 2.0). */
void cfear_reg_params_default(cfear_reg_params* p);

typedef struct cfear_reg_result {
  double pose[3];                       /* (x, y, theta) of the source after registration */
  double score;                         /* score_ = final_cost / num_residuals (getScore) */
  double final_cost;                    /* summary_.final_cost */
  int32_t num_residuals;                /* summary_.num_residuals (elements): of the last problem that was solved -- a
                                         * registration that ends with too few residuals in a later pass keeps the pass
                                         * before's, as summary_ does; 0 when none was solved */
  int32_t outer_iters;                  /* itr_ on exit (timing key "itrs") */
  int32_t lm_iters;                     /* total LM iterations */
  int32_t status;                       /* CFEAR_OK / CFEAR_ERR_TOO_FEW_RESIDUALS / CFEAR_ERR_SOLVER */
  double last_relative_decrease;        /* summary_.iterations.back().relative_decrease */
  double reserved;                      /* diagnostic: 1.0 when the registration is one the matcher's regular form (4 wavefronts,
                                         * 40 KB of LDS) is not good at -- dense scans; a caller that streams batches keeps
                                         * the large forms switched on while it sees these; 0.0 otherwise */
} cfear_reg_result;                     /* 72 bytes */

/* Replaces n_scan_normal_reg::Register (n_scan_normal.cpp:82-185).  poses_xyt [n_scans][3]
 * host, in/out: Affine3dToVectorXYeZ(Tsrc[i]) on entry; on return the last row holds the
 * registered source pose (Tsrc.back() = vectorToAffine3d(parameters.back())).  Returns the
 * result's status; reg_cov is the constant diag(0.01,0.01,0,0,0,1e-4) (n_scan_normal.cpp:171). */
int cfear_register(cfear_ctx* ctx, const cfear_scan* const* scans, int32_t n_scans,
                   double* poses_xyt, const cfear_reg_params* par, cfear_reg_result* result);

/* One launch, many independent registrations (loop-closure candidate batches,
 * loopclosure.cpp:35-97 called per candidate from :658-721).  results: host memory (the call
 * returns after the read-back) or DEVICE memory (the records stay on the GPU; the launch is
 * enqueued on the context's stream and not synchronised -- what a collective over the records
 * wants, cfear_register_batch_sharded).
 * The matcher runs in the form the batch calls for (wavefronts per registration x LDS per workgroup: 8 wavefronts up to two
 * workgroups per CU, 4 beyond, 2 for large batches of two-scan candidates); the forms add the fp64 sums in different
 * orders, so the SAME job may differ by a few ulp of cost / pose between a small and a large batch -- e.g. between world 1
 * and a sharded run.  Iteration counts and the accept / reject decisions are the same in every test of this repository
 * (tests/test_gpu_register.py::test_every_form_of_the_matcher_agrees), the poses agree to 1e-11.                       */
typedef struct cfear_reg_job {
  const cfear_scan* const* scans;       /* n_scans handles */
  int32_t n_scans;
  int32_t pad;
  const double* poses_xyt;              /* [n_scans][3] host */
} cfear_reg_job;
int cfear_register_batch(cfear_ctx* ctx, const cfear_reg_job* jobs, int32_t n_jobs,
                         const cfear_reg_params* par, cfear_reg_result* results);

/* The same for a loop-closure thread that registers candidate PAIRS among scans it keeps (every graph node's
 * cloud_normal_, types.h:119-122; loopclosure.cpp:658-721 walks the candidates of a query one by one): the scans' device
 * views are uploaded ONCE as a table, and a batch is then 56 bytes per candidate -- two table indices and the two poses of
 * loopclosure::Register's problem {to = fixed target, from = free source} (loopclosure.cpp:35-97) -- instead of a job
 * record with both scans' views (608 bytes) marshalled on the host per candidate and call; the records the matcher reads
 * are written by a small kernel on the device.  Results as for cfear_register_batch (host, or device: not synchronised).
 * The table keeps no reference on the scans: they must outlive it.                                                  */
typedef struct cfear_scan_table cfear_scan_table;
typedef struct cfear_candidate {
  int32_t target, source;               /* indices into the table */
  double target_xyt[3], source_xyt[3];  /* Affine3dToVectorXYeZ of the two poses; the source pose is the initial guess */
} cfear_candidate;                      /* 56 bytes */
int cfear_scan_table_create(cfear_ctx* ctx, const cfear_scan* const* scans, int32_t n_scans, cfear_scan_table** out);
int cfear_scan_table_size(const cfear_scan_table* table);
int cfear_scan_table_destroy(cfear_scan_table* table);
int cfear_register_candidates(cfear_ctx* ctx, const cfear_scan_table* table, const cfear_candidate* candidates,
                              int32_t n_candidates, const cfear_reg_params* par, cfear_reg_result* results);

/* Replaces n_scan_normal_reg::GetCost (n_scan_normal.cpp:186-211): one association pass at the
 * given poses + robust cost.  residuals (optional, host, cap entries) receives the robustified
 * residual vector; n_residuals its length; score = cost / max(n_residuals, 1).                */
int cfear_get_cost(cfear_ctx* ctx, const cfear_scan* const* scans, int32_t n_scans,
                   const double* poses_xyt, const cfear_reg_params* par, double* cost,
                   double* residuals, int32_t cap, int32_t* n_residuals, double* score);

/* GetCost for many (scan list, pose list) pairs in one launch (the 27 evaluations per registration of
 * approximateCovarianceBySampling, CFEARQuality over candidate batches).  results[j]: final_cost = robust
 * cost, score = cost / n_residuals, num_residuals, status (CFEAR_OK / CFEAR_ERR_TOO_FEW_RESIDUALS /
 * CFEAR_ERR_CAPACITY); pose echoes the evaluated source pose.  par->itr selects the radius as above. */
int cfear_get_cost_batch(cfear_ctx* ctx, const cfear_reg_job* jobs, int32_t n_jobs,
                         const cfear_reg_params* par, cfear_reg_result* results);

/* Covariance of a registration by cost sampling.  Replaces
 * OdometryKeyframeFuser::approximateCovarianceBySampling (odometrykeyframefuser.cpp:261-380) and
 * loopclosure::approximateCovarianceBySampling (tbv_slam/src/tbv_slam/loopclosure.cpp:99-208):
 * samples_per_axis^3 GetCost evaluations on a (yaw, x, y) grid around the registered source pose (one
 * kernel launch for all of them), quadratic least-squares fit, cov = 2 H^-1 * GetCovarianceScaler() *
 * covariance_scaler embedded in a 6x6 (x, y, -, -, -, yaw).                                       */
typedef struct cfear_cov_sampling_params {
  double xy_range;                      /* cov_sampling_xy_range: samples span +- xy_range / 2        */
  double yaw_range;                     /* cov_sampling_yaw_range                                     */
  int32_t samples_per_axis;             /* cov_sampling_samples_per_axis (3)                          */
  int32_t pad;
  double covariance_scaler;             /* cov_sampling_covariance_scaler (4.0)                       */
} cfear_cov_sampling_params;            /* 32 bytes */
void cfear_cov_sampling_params_default(cfear_cov_sampling_params* p);   /* odometrykeyframefuser.h:107-110 */
/* poses_xyt: the poses AFTER Register (T_vek); par: the registration parameters that produced them;
 * reg: that Register's result (final_cost, num_residuals -> GetCovarianceScaler; outer_iters -> the
 * radius GetCost uses).  cov36: row-major 6x6; samples (optional): [n^3][4] = x, y, yaw offset, cost in
 * the reference's loop order (yaw outer, x, y inner).  *success = 1 when the fit is convex and the
 * scaler defined (the reference then replaces reg_cov); otherwise cov36 holds Register's constant
 * diag(0.01, 0.01, 0, 0, 0, 1e-4).                                                                  */
int cfear_covariance_by_sampling(cfear_ctx* ctx, const cfear_scan* const* scans, int32_t n_scans,
                                 const double* poses_xyt, const cfear_reg_params* par,
                                 const cfear_reg_result* reg, const cfear_cov_sampling_params* sp,
                                 double* cov36, double* samples, int32_t* success);
/* Same for a batch (loop-closure candidates): regs [n_jobs], cov36 [n_jobs][36], samples (optional)
 * [n_jobs][n^3][4], success [n_jobs].                                                               */
int cfear_covariance_by_sampling_batch(cfear_ctx* ctx, const cfear_reg_job* jobs, int32_t n_jobs,
                                       const cfear_reg_params* par, const cfear_reg_result* regs,
                                       const cfear_cov_sampling_params* sp, double* cov36, double* samples,
                                       int32_t* success);

/* Ceres-compatible evaluation of one association set (what AddScanPairCost would hand to
 * ceres::Problem, n_scan_normal.cpp:264-318): prepare associates once at `poses_xyt`;
 * evaluate returns the RAW residuals r [n_res] and Jacobian J [n_res][3] (row-major, like
 * ceres::CostFunction::Evaluate), the per-block weights w [n_blocks] of ScaledLoss, and
 * normal_eq the robustified H = J^T J [9], g = J^T r [3], cost = 1/2 sum rho at x.            */
typedef struct cfear_cost cfear_cost;
int cfear_cost_prepare(cfear_ctx* ctx, const cfear_scan* const* scans, int32_t n_scans,
                       const double* poses_xyt, const cfear_reg_params* par, int32_t itr,
                       cfear_cost** out);
int cfear_cost_num_blocks(const cfear_cost* c);
int cfear_cost_num_residuals(const cfear_cost* c);
/* pairs int32 [n_blocks][3] = (target scan, target cell, source cell); weights [n_blocks]. */
int cfear_cost_get_blocks(const cfear_cost* c, int32_t* pairs, double* weights);
int cfear_cost_evaluate(cfear_cost* c, const double x[3], double* residuals, double* jacobian);
int cfear_cost_normal_eq(cfear_cost* c, const double x[3], double H[9], double g[3], double* cost);
int cfear_cost_destroy(cfear_cost* c);

/* ---- caller: CorAl alignment quality (loop-closure verification) ---------------------------------
 * Replaces CorAlRadarQuality (coral_alignment_quality/src/alignment_checker/AlignmentQuality.cpp:8-230)
 * as ScanLearningInterface::getCorAlQualityMeasure builds it (alignmentinterface.cpp:437-456): both
 * scans are kstrongStructuredRadar objects over the stored PEAK clouds, the source is placed at
 * src_pose * Toffset, every point of the merged cloud gets the entropy of its radius-neighbourhood in
 * its own cloud (sep) and in both clouds (joint).  quality_ = {joint, sep, overlap}; valid_ = overlap
 * >= 0.1.  Only ent_cfg = any (ComputeEntropy) is built: TBV never selects the kl variant.          */
typedef struct cfear_coral_params {
  double radius;                        /* AlignmentQuality::parameters::radius; TBV: 1.0              */
  int32_t weight_res_intensity;         /* weight_res_intensity; TBV: false                            */
  int32_t pad;
} cfear_coral_params;
void cfear_coral_params_default(cfear_coral_params* p);

typedef struct cfear_coral_job {
  const float* ref_xyzi;                /* [n_ref][4] x,y,z,intensity in the reference scan's sensor frame; host or device */
  const float* src_xyzi;                /* [n_src][4]                                                   */
  int32_t n_ref, n_src;
  double ref_pose[3];                   /* ref->GetAffine() as (x, y, theta)                           */
  double src_pose[3];                   /* src->GetAffine()                                            */
  double offset[3];                     /* Toffset (perturbation), applied as src_pose * Toffset       */
} cfear_coral_job;

typedef struct cfear_coral_result {
  double joint, sep, overlap;           /* quality_[0..2] (output_overlap = true)                      */
  int32_t valid;                        /* valid_                                                      */
  int32_t count_valid;                  /* points with both covariances and finite entropies           */
  int32_t status;                       /* CFEAR_OK / CFEAR_ERR_EMPTY_CLOUD / CFEAR_ERR_CAPACITY       */
  int32_t pad;
} cfear_coral_result;                   /* 40 bytes */

/* per_point (optional, host): [n_src + n_ref][3] = joint_res_, sep_res_, sep_valid in the reference's
 * index order (source points first); invalid points hold the reference's initial 100.0.             */
int cfear_coral_quality(cfear_ctx* ctx, const cfear_coral_job* job, const cfear_coral_params* par,
                        cfear_coral_result* result, double* per_point);
/* One launch for a batch (the 13 perturbations of a training pair, a loop-closure candidate list);
 * clouds shared between jobs are uploaded once.  per_point (optional): jobs' arrays back to back.     */
int cfear_coral_quality_batch(cfear_ctx* ctx, const cfear_coral_job* jobs, int32_t n_jobs,
                              const cfear_coral_params* par, cfear_coral_result* results, double* per_point);

/* ---- before the path: radar Scan Context (loop-candidate generation) ------------------------------
 * Replaces the arithmetic of RSCManager / SCManager (place_recognition_radar/src/place_recognition_radar/
 * RadarScancontext.cpp:59-131, 156-180; Scancontext.cpp:60-268): the ring x sector descriptor of a
 * local-map cloud (TBV: merged peak clouds of 2 N_aggregate + 1 nodes in the node's frame, loopclosure.cpp:
 * 552-590), its ring / sector keys, and distanceBtnScanContext.  The descriptor database, the odometry-
 * coupled ring-key search and the candidate ranking stay on the host (api.py RSCManager mirrors them). */
typedef struct cfear_sc_params {
  int32_t num_ring, num_sector;         /* PC_NUM_RING 40, PC_NUM_SECTORS 120                          */
  double max_radius;                    /* PC_MAX_RADIUS 80                                            */
  double search_ratio;                  /* SEARCH_RATIO 0.1                                            */
  int32_t desc_function;                /* 0 = "sum", 1 = "max"                                        */
  int32_t pad;
  double desc_divider;                  /* 1000 in TBV's launch defaults                               */
  double no_point;                      /* value of empty bins (reached only when desc_divider == 1)   */
} cfear_sc_params;
void cfear_sc_params_default(cfear_sc_params* p);
typedef struct cfear_sc_cloud {
  const float* xyzi;                    /* [n][4] x,y,z,intensity in the node's frame; host or device   */
  int32_t n, pad;
} cfear_sc_cloud;
/* MakeRadarCloudContext for n_clouds clouds x n_aug lateral shifts (shifts_y[0] is normally 0; TBV augments
 * with {-2, 2, -4, 4}).  desc [n_clouds][n_aug][num_ring * num_sector] row-major (ring, sector); ringkey
 * [..][num_ring] and sectorkey [..][num_sector] optional.  desc may be HOST or DEVICE memory (a descriptor
 * database kept in HBM feeds cfear_sc_distance_batch without crossing PCIe); the keys are host arrays -- the
 * retrieval policy that consumes them runs on the host.                                                */
int cfear_sc_descriptors(cfear_ctx* ctx, const cfear_sc_cloud* clouds, int32_t n_clouds,
                         const cfear_sc_params* par, const double* shifts_y, int32_t n_aug, double* desc,
                         double* ringkey, double* sectorkey);
/* distanceBtnScanContext for pairs[i] = (query index into desc_q, candidate index into desc_c); descriptors
 * host or device; dist / shift host [n_pairs] (shift = argmin column shift of the candidate).          */
int cfear_sc_distance_batch(cfear_ctx* ctx, const double* desc_q, int32_t n_q, const double* desc_c, int32_t n_c,
                            const int32_t* pairs, int32_t n_pairs, const cfear_sc_params* par, double* dist,
                            int32_t* shift);

/* RSCManager as a library object (place_recognition_radar RadarScancontext.cpp:156-345): the descriptor database
 * lives in HBM; makeAndSaveScancontextAndKeysRadarCloud = add, detectLoopClosureID = detect.  Host policy (recent-node
 * exclusion, odometry likelihood, ring-key search, candidate ranking) runs in the library's C++.                    */
typedef struct cfear_sc_manager cfear_sc_manager;
typedef struct cfear_sc_manager_params {
  cfear_sc_params sc;
  int32_t num_candidates_from_tree;     /* NUM_CANDIDATES_FROM_TREE (10)                                 */
  int32_t n_candidates;                 /* N_CANDIDATES kept after ranking (3)                           */
  double odom_sigma_error;              /* 0.05                                                          */
  int32_t odometry_coupled_closure;     /* ring key extended by 10 x odometry similarity (true)          */
  int32_t augment_sc;                   /* lateral augmentations {-2, 2, -4, 4} m of the query (true)    */
  double distance_exclude_recent;       /* DISTANCE_EXCLUDE_RECENT (10 m)                                */
  int64_t pad;
} cfear_sc_manager_params;              /* 88 bytes */
void cfear_sc_manager_params_default(cfear_sc_manager_params* p);
typedef struct cfear_sc_candidate {     /* RSCManager::candidate */
  double min_dist, min_dist_sc, min_dist_odom;
  float yaw_diff_rad;
  int32_t nn_idx;
  int32_t argmin_shift;
  int32_t pad;
  double Taug[3];                       /* augmentation transform of the winning query as (x, y, theta)  */
} cfear_sc_candidate;                   /* 64 bytes */
int cfear_sc_manager_create(cfear_ctx* ctx, const cfear_sc_manager_params* par, cfear_sc_manager** out);
/* cloud: the node's local map [n][4] in the node frame (host or device); Todom: the node's pose (x, y, theta). */
int cfear_sc_manager_add(cfear_sc_manager* m, const float* xyzi, int32_t n_points, const double Todom[3]);
/* candidates for the node added last, closest first; *n_out <= n_candidates.                                   */
int cfear_sc_manager_detect(cfear_sc_manager* m, cfear_sc_candidate* out, int32_t cap, int32_t* n_out);
int cfear_sc_manager_size(const cfear_sc_manager* m);
int cfear_sc_manager_destroy(cfear_sc_manager* m);

/* ---- caller: loop-candidate verification --------------------------------------------------------------
 * What the loop-closure thread does per candidate (tbv_slam/src/tbv_slam/loopclosure.cpp:658-725), for a batch:
 * RegisterLoopCandidate (:320-364, loopclosure::Register :35-97), VerifyLoopCandidate (:365-384) =
 * ScanLearningInterface::PredAlignment (coral_alignment_quality/src/alignment_checker/alignmentinterface.cpp:349-367:
 * CorAl + CFEAR quality -> combined logistic score "alignment_quality") + VerificationModel (:220-238), then
 * ApplyConstratins (:261-274).  One registration launch, one CorAl launch, one GetCost launch for all candidates. */
typedef struct cfear_verify_params {
  double align_intercept;               /* combined_class: file format "intercept,coef..." (alignmentinterface.cpp:224-269) */
  double align_coef[6];                 /* over {CorAl joint, sep, overlap, CFEAR cost, #residuals, mean #cells} (:357-360)  */
  double loop_intercept;                /* verification model over par_.model_features = {odom-bounds, sc-sim,               */
  double loop_coef[3];                  /*   alignment_quality} (loopclosure.h:138); preset loopclosure.cpp:224-232          */
  double model_threshold;               /* par_.model_threshold (0.8)                                                        */
  int32_t all_candidates;               /* par_.all_candidates (true): every candidate above threshold, else only the best   */
  int32_t verification_disabled;        /* probability 0 for everything (loopclosure.cpp:377)                                */
  int32_t use_covariance_sampling;      /* par_.use_covariance_sampling_in_loop_closure (false)                              */
  int32_t pad;
  cfear_coral_params coral;             /* radius 1.0, weight_res_intensity false (alignmentinterface.cpp:444)               */
  cfear_cov_sampling_params sampling;   /* loopclosure.cpp:108-112: +-0.2 m, +-0.0022 rad, 3 per axis, scaler 4              */
} cfear_verify_params;                  /* 160 bytes */
void cfear_verify_params_default(cfear_verify_params* p);

typedef struct cfear_verify_job {
  const cfear_scan* from_scan;          /* (*graph_)[from].cloud_normal_  (the query node: source of the registration)       */
  const cfear_scan* to_scan;            /* (*graph_)[to].cloud_normal_    (the candidate: fixed target)                      */
  const float* from_peaks;              /* (*graph_)[from].cloud_peaks_ [n_from][4], sensor frame; host or device            */
  const float* to_peaks;                /* (*graph_)[to].cloud_peaks_                                                        */
  int32_t n_from, n_to;
  double from_pose[3];                  /* (*graph_)[from].GetPose() as (x, y, theta)                                        */
  double t_be_guess[3];                 /* constraint.t_be on entry (Scan Context yaw / lateral guess): Tto = Tfrom * t_be   */
  double sc_sim;                        /* quality["sc-sim"]                                                                 */
  double odom_bounds;                   /* quality["odom-bounds"], e.g. from cfear_verify_by_odometry                        */
  int32_t group;                        /* candidates of one query node share a group (ApplyConstratins runs per query)      */
  int32_t pad;
} cfear_verify_job;                     /* 112 bytes */

typedef struct cfear_verify_result {
  double t_be[3];                       /* constraint.t_be after registration = Trevised^-1 * Tto; Identity if it failed     */
  double cov[36];                       /* Cov (the reference stores information = Cov.inverse()); Identity if it failed     */
  double coral[3];                      /* X_CorAl = {joint, sep, overlap}                                                   */
  double cfear[3];                      /* X_CFEAR = {cost, #residuals, mean #cells}                                         */
  double alignment_quality;             /* quality["alignment_quality"]                                                      */
  double odom_bounds, sc_sim;           /* echoed features                                                                   */
  double probability;                   /* VerifyLoopCandidate                                                               */
  int32_t reg_ok;                       /* RegisterLoopCandidate's return                                                    */
  int32_t cov_sampled;                  /* the sampled covariance replaced Register's constant one                           */
  int32_t accepted;                     /* ApplyConstratins added the loop constraint                                        */
  int32_t rank;                         /* position in the query's probability-sorted candidate list                         */
  cfear_reg_result reg;                 /* the registration's own record                                                     */
} cfear_verify_result;                  /* 480 bytes */

/* results: host memory -- every field filled, ApplyConstratins applied over the batch; or DEVICE memory (a sharded caller
 * gathers the records there, cfear_rccl_allgather_device): the records are left on the device with accepted = rank = 0, the
 * call still waits for the chain and reports its errors, and the selection is the caller's, over the gathered list
 * (cfear_verify_apply_constraints; not available with use_covariance_sampling).                                       */
int cfear_verify_loop_candidates(cfear_ctx* ctx, const cfear_verify_job* jobs, int32_t n_jobs,
                                 const cfear_verify_params* par, cfear_verify_result* results);
/* ApplyConstratins (loopclosure.cpp:261-274) over host records: groups [n] = the candidates' query ids.  No context. */
int cfear_verify_apply_constraints(const int32_t* groups, int32_t n, const cfear_verify_params* par, cfear_verify_result* results);
/* loopclosure::VerifyByOdometry (loopclosure.cpp:776-808).  rel_xyt [n][3]: the odometry constraints'
 * RelativeMotion(i, i+1), i = to .. from-1.  similarity = 1 - exp(-(max(|T_odom| - 5, 0) / travelled)^2 / 2 sigma^2);
 * 1 when verify_via_odometry is 0.  No context: pure host arithmetic.                                               */
int cfear_verify_by_odometry(const double* rel_xyt, int32_t n, double odom_sigma_error, int32_t verify_via_odometry,
                             double* similarity);

/* ---- caller: candidate batches sharded over the GPUs of one node, for a C++ host ------------------------------
 * Loop-closure candidates are independent (loopclosure.cpp:658-721), so a batch shards by contiguous blocks of
 * ceil(n / world) candidates per rank (one process, context and GPU per rank), with ONE all_gather of the fixed-size
 * result records at the end -- rank order = candidate order.  The host owns the communicator: the collective is a
 * callback gather(user, send, recv, bytes) that must place every rank's `bytes` bytes, rank after rank, into recv on
 * every rank and return 0.  cfear_rccl_allgather is that callback over an ncclComm_t (RCCL over xGMI), user = a
 * cfear_rccl_comm; librccl.so is resolved at run time.  jobs / n_jobs are the FULL candidate list on every rank and
 * results receives all n_jobs records on every rank.  world == 1 needs no callback (NULL); a callback that is given is
 * called at every world size, 1 included.  A rank whose own block fails still takes part in the exchange (its status rides
 * in an 8-byte trailer behind its block) and EVERY rank returns the first failed rank's status.  With cfear_rccl_allgather
 * as the callback cfear_register_batch_sharded keeps the records on the device: the kernel writes into the send buffer,
 * ncclAllGather moves them, one device-to-host copy returns all of them.                                                */
typedef int (*cfear_allgather_fn)(void* user, const void* send, void* recv_all, size_t bytes_per_rank);
typedef struct cfear_rccl_comm { cfear_ctx* ctx; void* nccl_comm; int32_t world, pad; } cfear_rccl_comm;
int cfear_rccl_allgather(void* user /* cfear_rccl_comm* */, const void* send, void* recv_all, size_t bytes_per_rank);
/* the same collective on DEVICE buffers: enqueued on the communicator's context stream, not synchronised */
int cfear_rccl_allgather_device(void* user /* cfear_rccl_comm* */, const void* d_send, void* d_recv_all, size_t bytes_per_rank);
/* A communicator of the library's own making for hosts that have none (librccl.so resolved at run time): rank 0 asks for
 * the 128-byte id and hands it to its peers by whatever means it has (MPI, a socket, torch.distributed); every rank then
 * calls cfear_rccl_comm_init with it.  The communicator's collectives run on streams of `ctx`.                          */
int cfear_rccl_unique_id(char id128[128]);
int cfear_rccl_comm_init(cfear_ctx* ctx, const char id128[128], int32_t world, int32_t rank, cfear_rccl_comm* out);
int cfear_rccl_comm_destroy(cfear_rccl_comm* comm);
int cfear_shard_range(int32_t n, int32_t world, int32_t rank, int32_t* lo, int32_t* hi, int32_t* per_rank);
/* the gather step alone: local = this rank's hi - lo records; all = n_total records in candidate order */
int cfear_gather_records(const void* local, int32_t n_total, int32_t record_bytes, int32_t world, int32_t rank,
                         cfear_allgather_fn gather, void* user, void* all);
int cfear_register_batch_sharded(cfear_ctx* ctx, const cfear_reg_job* jobs, int32_t n_jobs, const cfear_reg_params* par,
                                 int32_t rank, int32_t world, cfear_allgather_fn gather, void* user,
                                 cfear_reg_result* results);
/* Pipelined steps of a sharded candidate batch (loopclosure.cpp:658-721 hands candidates over as the odometry produces
 * nodes): a rank's block of an 8-way sharded batch is ~0.13 ms of kernel, so what sits around the kernel decides the
 * rate.  A pipe keeps up to `depth` steps in flight: submit() stages this rank's block of the FULL candidate list (pinned:
 * the device reads it in place), enqueues the expand kernel on the pipe's preparation stream (beside the previous step's
 * matcher), the matcher on the context's stream and all_gather -> device-to-host copy on the pipe's exchange stream, and
 * returns without waiting; collect() waits on ONE event and returns all n records in candidate
 * order (status rules as cfear_register_batch_sharded: every rank enters the collective, every rank returns the first
 * failed rank's status).  comm = NULL (world 1 only): no collective.  flags & CFEAR_PIPE_GRAPH: a slot's matcher launches
 * are captured into a hipGraph on its first step and replayed while the block's size and geometry, the parameters and the
 * context's scratch stay the same.  Steps are collected in any order, but a slot (ticket % depth) is free again only after its
 * collect.  The table must outlive the pipe.                                                                          */
typedef struct cfear_candidate_pipe cfear_candidate_pipe;
enum { CFEAR_PIPE_GRAPH = 1, CFEAR_PIPE_TIMING = 2 /* hipEvents around the exchange of every step (measurement) */ };
int cfear_candidate_pipe_create(cfear_ctx* ctx, const cfear_scan_table* table, int32_t max_candidates, int32_t rank,
                                int32_t world, const cfear_rccl_comm* comm, int32_t depth, int32_t flags,
                                cfear_candidate_pipe** out);
int cfear_candidate_pipe_submit(cfear_candidate_pipe* pipe, const cfear_candidate* candidates, int32_t n_total,
                                const cfear_reg_params* par, int64_t* ticket);
int cfear_candidate_pipe_collect(cfear_candidate_pipe* pipe, int64_t ticket, cfear_reg_result* results);
int cfear_candidate_pipe_destroy(cfear_candidate_pipe* pipe);
/* measurement: the exchange stream's time (all_gather + read-back, CFEAR_PIPE_TIMING) summed over the collected steps, their
 * number, and how many slots currently replay a captured graph */
int cfear_candidate_pipe_stats(const cfear_candidate_pipe* pipe, double* exchange_ms_sum, int64_t* steps_collected, int32_t* graph_slots);
/* verification: ApplyConstratins (loopclosure.cpp:261-274) is redone over the gathered list, because the candidates
 * of one query may sit on two ranks                                                                             */
int cfear_verify_loop_candidates_sharded(cfear_ctx* ctx, const cfear_verify_job* jobs, int32_t n_jobs,
                                         const cfear_verify_params* par, int32_t rank, int32_t world,
                                         cfear_allgather_fn gather, void* user, cfear_verify_result* results);

/* ---- caller: batched radarDriver + OdometryKeyframeFuser --------------------------------------
 * n_streams independent sequences advance one frame per call: filter (F) -> compensate (C) ->
 * surface points (N) -> Register against the keyframe window (M) -> keyframe policy.  Restates
 * radarDriver::CallbackOffline (radar_driver.cpp:163-176) + OdometryKeyframeFuser::processFrame
 * (odometrykeyframefuser.cpp:143-259) per stream; everything between the polar image and the
 * pose stays on the GPU.                                                                      */
typedef struct cfear_odometry_params {
  int32_t filter_type;                  /* cfear_filter_type */
  cfear_kstrong_params kstrong;
  cfear_cacfar_params cacfar;
  cfear_reg_params reg;
  float res;                            /* par.res */
  int32_t submap_scan_size;
  int32_t weight_intensity, use_guess, compensate, radar_ccw, use_keyframe;
  int32_t rotate_ccw;                   /* 1: the incoming images are [range bins][azimuths] (dataset != oxford), to be
                                           rotated first, radar_driver.cpp:74-90; desc then describes that source layout.
                                           (The k-strongest stage reads such images directly where their geometry allows --
                                           see cfear_filter_kstrongest_rowkeys -- and rotates them otherwise.)          */
  double min_keyframe_dist, min_keyframe_rot_deg, downsample_factor;
  int32_t estimate_cov_by_sampling;     /* par.estimate_cov_by_sampling (false), odometrykeyframefuser.h:104 */
  int32_t keep_nodes;                   /* 1: also build what RadarScan needs (types.h:119-122): the peaks cloud of every
                                           frame, compensated like the cloud (odometrykeyframefuser.cpp:146-150), so that
                                           cfear_odometry_get_scan / _get_cloud / _get_peaks can hand out graph nodes */
  cfear_cov_sampling_params cov_sampling;   /* cov_sampling_* (:107-110) */
} cfear_odometry_params;
void cfear_odometry_params_default(cfear_odometry_params* p);   /* CFEAR-3 preset, Oxford */
/* The reference's shipped configurations (cfear_radarodometry/launch/oxford/eval/params/baseline/oxford_cfear-{1,2,3,
 * 3-s10}:13-26) and sensor setups (tbv_slam/script/{oxford,mulran,kvarntorp,volvo}/run_tbv_simple.sh):
 *   CFEAR-1      P2L, 1 keyframe,  res 3.5, k 12, Huber 0.1, weight option 4, no intensity weights
 *   CFEAR-2      P2L, 3 keyframes, res 3.5, k 12, Huber 0.1
 *   CFEAR-3      P2P, 4 keyframes, res 3,   k 40, Huber 0.1, intensity weights (TBV's default)
 *   CFEAR-3-s10  P2P, 10 keyframes, res 3,  k 40, Cauchy 0.1, regularization 0.1
 *   Oxford       range_res 0.0438,    clockwise sweep,         rows = azimuths
 *   MulRan       range_res 0.0595238, counter-clockwise sweep, [range bins][azimuths] images (rotate_ccw)
 *   Kvarntorp / Volvo  range_res 0.175, counter-clockwise,     [range bins][azimuths] images
 * Everything else as cfear_odometry_params_default.  Returns CFEAR_ERR_INVALID_ARGUMENT for unknown ids.        */
enum cfear_preset { CFEAR_PRESET_CFEAR1 = 1, CFEAR_PRESET_CFEAR2 = 2, CFEAR_PRESET_CFEAR3 = 3, CFEAR_PRESET_CFEAR3_S10 = 4 };
enum cfear_dataset { CFEAR_DATASET_OXFORD = 0, CFEAR_DATASET_MULRAN = 1, CFEAR_DATASET_KVARNTORP = 2, CFEAR_DATASET_VOLVO = 3 };
int cfear_odometry_params_preset(cfear_odometry_params* p, int preset, int dataset);

/* The two per-frame decisions of OdometryKeyframeFuser as pure host functions (no context, no GPU), used by the batched
 * pipeline below and callable on their own:
 *   cfear_keyframe_based_fuse    KeyFrameBasedFuse (odometrykeyframefuser.cpp:62-73): diff = T_keyframe^-1 * Tcurrent as
 *                                (x, y, theta); 1 = add a keyframe (translation norm > min_keyframe_dist or |rotation| >
 *                                min_keyframe_rot_deg, both strict; always 1 without use_keyframe)
 *   cfear_acc_vel_sanity_check   AccelerationVelocitySanityCheck (:76-94): translations of the previous and the current
 *                                motion; 0 = acceleration > 200 m/s^2 or speed > 200 m/s at 4 Hz -> the caller keeps
 *                                its guess (:198-199)                                                              */
int cfear_keyframe_based_fuse(const double diff_xyt[3], int32_t use_keyframe, double min_keyframe_dist,
                              double min_keyframe_rot_deg);
int cfear_acc_vel_sanity_check(const double tmot_prev_xy[2], const double tmot_curr_xy[2]);

typedef struct cfear_odometry cfear_odometry;
typedef struct cfear_frame_info {
  double pose[3];                       /* Tcurrent (x,y,theta) */
  int32_t n_points, n_cells;
  int32_t keyframe_added;
  int32_t reg_status;                   /* cfear_reg_result.status, or 1 for the first frame */
  int32_t outer_iters, lm_iters;
  double score;
} cfear_frame_info;

int cfear_odometry_create(cfear_ctx* ctx, int32_t n_streams, const cfear_polar_desc* desc,
                          const cfear_odometry_params* par, cfear_odometry** out);
/* polar: [n_streams] images laid out per desc (desc.batch must equal n_streams), host or device.
 * info: host array [n_streams].                                                               */
int cfear_odometry_process(cfear_odometry* od, const uint8_t* polar, cfear_frame_info* info);
/* Same, and additionally enqueues the FILTER of the next frame's images (polar_next, may be NULL) behind
 * this frame's kernels, so the GPU sweeps the next polar batch while the host applies this frame's
 * keyframe policy.  The next call must then pass that same pointer as `polar`: the hit is decided by the
 * pointer value only (see cfear_odometry_discard_prefetch).  Return value: per-stream failures (empty sweep,
 * capacity) do not stop the other streams -- every info[b] is filled, info[b].reg_status holds the stream's
 * own status and the call returns the first such status; a failed prefetch is reported the same way after
 * this frame has been completed.                                                                  */
int cfear_odometry_process_prefetch(cfear_odometry* od, const uint8_t* polar, const uint8_t* polar_next,
                                    cfear_frame_info* info);
/* The same step for hosts whose sweeps do not sit at a constant stride (one ring buffer per sequence, sensor drivers
 * with their own allocations): stream b's image starts at base + offsets[b] bytes (device memory, laid out per desc
 * otherwise); offsets_next (may be NULL) names the next frame's images for the prefetch, which is recognised on the
 * next call by the same base and the same offsets.  offsets are HOST arrays [n_streams].  Available for the
 * k-strongest filter with k <= 64, rows = azimuths, keep_nodes = 0 (the fused filter output); otherwise
 * CFEAR_ERR_INVALID_ARGUMENT.                                                                                   */
int cfear_odometry_process_offsets(cfear_odometry* od, const uint8_t* base, const int64_t* offsets,
                                   const int64_t* offsets_next, cfear_frame_info* info);
/* Forget a prefetched filter output.  A prefetch hit is decided by the address (and offsets) of the images alone: a
 * host that REUSES a buffer for different content (drops a frame, rewrites a ring slot) must call this first, or the
 * stale filter output of the earlier content is consumed.                                                        */
int cfear_odometry_discard_prefetch(cfear_odometry* od);
/* The same step for callers whose own driver has filtered the sweep: OdometryKeyframeFuser::pointcloudCallback(cloud,
 * cloud_peaks, Tcurrent, t, cov) (odometrykeyframefuser.cpp:413-426) for every stream.  clouds [n_streams]: the filtered
 * clouds (x, y, z, intensity; host or device, <= rows * k points each); peaks [n_streams] (may be NULL): the peaks
 * clouds, kept (and compensated) only with par.keep_nodes.  Everything after the filter is unchanged.            */
int cfear_odometry_process_clouds(cfear_odometry* od, const cfear_sc_cloud* clouds, const cfear_sc_cloud* peaks,
                                  cfear_frame_info* info);
/* cov_current of every stream after the last processed frame (row-major 6x6, host [n_streams][36]):
 * Identity before the first registration and after a failed one (FormatScans' initial value survives),
 * Register's constant diag(0.01, 0.01, 0, 0, 0, 1e-4) otherwise, or the sampled covariance when
 * estimate_cov_by_sampling is set and the fit succeeded (odometrykeyframefuser.cpp:196, 203-208).
 * sampled (optional, [n_streams]) receives 1 where the sampled covariance was used.                */
int cfear_odometry_get_covariance(cfear_odometry* od, double* cov, int32_t* sampled);
/* The RadarScan of a stream's LAST processed frame (scan_, odometrykeyframefuser.cpp:172, 244) -- call after a frame
 * whose keyframe_added is set to collect a pose-graph node; valid until the next cfear_odometry_process:
 *   get_scan   a copy of cloud_normal_ (MapPointNormal) as a new handle (cfear_scan_destroy it)
 *   get_cloud  cloud_nopeaks_: the filtered cloud after Compensate (what the surface points were built from)
 *   get_peaks  cloud_peaks_: the AxialNonMaxSupress subset after Compensate (needs par.keep_nodes)
 * xyzi: host or device buffer [cap][4], or NULL to query the count in *n_out.                                  */
int cfear_odometry_get_scan(cfear_odometry* od, int32_t stream, cfear_scan** out);
int cfear_odometry_get_cloud(cfear_odometry* od, int32_t stream, float* xyzi, int32_t cap, int32_t* n_out);
int cfear_odometry_get_peaks(cfear_odometry* od, int32_t stream, float* xyzi, int32_t cap, int32_t* n_out);
int cfear_odometry_destroy(cfear_odometry* od);

/* ---- after the path: pose-graph nodes on disk ------------------------------------------------
 * simple_graph.sgh = Boost binary archive of std::vector<std::pair<RadarScan, std::vector<Constraint3d>>>
 * (types.h:46-192, types.cpp:103-130; MapPointNormal::save/load pointnormal.h:206-228; serialization.h): what
 * OdometryKeyframeFuser::SaveGraph writes and the loop-closure tools read back ("advanced usage", README.md:113-168).
 * Pure host code, no context.  The archive layout is restated from Boost 1.71's sources; no reference-produced file
 * exists in the repository to pin it against (csrc/graph.hip says what is assumed).                              */
typedef struct cfear_pose3d { double p[3]; double q[4]; } cfear_pose3d;      /* Pose3d: translation, quaternion (x, y, z, w) */
void cfear_pose3d_from_xyt(const double xyt[3], cfear_pose3d* out);          /* PoseEigToCeres of a planar pose (types.cpp:25-32) */
void cfear_pose3d_to_xyt(const cfear_pose3d* p, double xyt[3]);
typedef struct cfear_graph_cloud {      /* pcl::PointCloud<PointXYZI>::Ptr: n < 0 = null pointer */
  const float* xyzi;                    /* [n][4] x, y, z, intensity */
  int32_t n;
  uint32_t seq;                         /* header.seq */
  uint64_t stamp;                       /* header.stamp */
  const char* frame_id;                 /* header.frame_id (NULL = "") */
} cfear_graph_cloud;
typedef struct cfear_graph_constraint { /* Constraint3d, types.h:152-186 */
  uint64_t id_begin, id_end;
  cfear_pose3d t_be;
  double information[36];               /* row-major 6x6 */
  int32_t type;                         /* ConstraintType: 0 odometry, 1 loop_appearance, 2 mini_loop, 3 candidate */
  int32_t n_quality;
  const char* const* quality_keys;      /* std::map<std::string, double> quality, in key order */
  const double* quality_values;
  const char* info;
} cfear_graph_constraint;
typedef struct cfear_graph_node {       /* RadarScan (types.h:88-142) + the constraints stored with it */
  cfear_pose3d T, Tgt;
  int32_t has_Tgt;
  uint32_t idx;                         /* idx_ */
  uint64_t stamp;                       /* stamp_ */
  double motion[16];                    /* motion_: Affine3d::data(), column-major 4x4 */
  cfear_graph_cloud cloud_peaks, cloud_nopeaks;
  int32_t has_normal;                   /* cloud_normal_ != NULL */
  int32_t input_is_nopeaks;             /* cloud_normal_->input_ is the cloud_nopeaks_ object (how the fuser builds nodes): stored once */
  cfear_graph_cloud normal_input;       /* cloud_normal_->input_ otherwise */
  const cfear_cell* cells;              /* cloud_normal_->cells */
  int32_t n_cells;
  float radius;                         /* radius_ */
  int32_t weight_intensity, pad;
  const cfear_graph_constraint* constraints;
  int32_t n_constraints, pad2;
} cfear_graph_node;
typedef struct cfear_graph cfear_graph;
/* SaveSimpleGraph.  Boost stores an object reached through several shared_ptrs once: here the buffer is the identity -- a
 * cloud whose `xyzi` pointer (and size) equals that of a cloud written earlier, in any slot of any node, is written as a
 * reference to it.  cfear_graph_load resolves such references (its nodes then hold equal copies) and rejects files whose
 * counts exceed what is left of the file (CFEAR_ERR_FORMAT) before allocating anything.                              */
int cfear_graph_save(const char* path, const cfear_graph_node* nodes, int32_t n_nodes);
int cfear_graph_load(const char* path, cfear_graph** out);                                    /* LoadSimpleGraph */
int cfear_graph_size(const cfear_graph* g);
int cfear_graph_node_at(const cfear_graph* g, int32_t i, cfear_graph_node* out);              /* pointers live until destroy */
int cfear_graph_destroy(cfear_graph* g);
/* ---- after the path: pose-graph optimisation -----------------------------------------------------
 * Replaces tbv_slam's CeresLeastSquares::Solve (tbv_slam/src/tbv_slam/ceresoptimizer.cpp:13-113) with its
 * PoseGraph3dErrorTerm (include/tbv_slam/ceresoptimizer.h:55-112): one 6-residual block per constraint,
 * sqrt_information = llt(I_scaled).matrixL() with I_scaled = diag(1/odom_vxx, 1/odom_vyy, 1, 1, 1, 1/odom_vtt) (both
 * types -- the loop_v* values are never read, :79-88) or the constraint's own information, times 1 / loop_scaling for
 * loop constraints; no loss on odometry, CauchyLoss(0.1) on loop_appearance; mini_loop / candidate constraints are not
 * optimised; the first node (smallest id) is constant; quaternions move on ceres::EigenQuaternionParameterization;
 * ceres::Solve with the default trust-region LM and max_num_iterations 200.  Host code (csrc/pgo.hip says why), no
 * context.  poses [n] in/out, ids [n] strictly ascending (the reference's node map order).                        */
typedef struct cfear_pgo_params {        /* tbv_slam::OptimizationParamsConfig as CeresLeastSquares::Parameters sets it */
  double loop_vxx, loop_vyy, loop_vtt, odom_vxx, odom_vyy, odom_vtt, loop_scaling;
  int32_t replace_cov_by_identity;
  int32_t max_num_iterations;            /* 200 (ceresoptimizer.cpp:52) */
  double loop_loss_limit;                /* CauchyLoss(0.1) (:36) */
} cfear_pgo_params;
void cfear_pgo_params_default(cfear_pgo_params* p);
typedef struct cfear_pgo_summary {
  double initial_cost, final_cost;       /* ceres::Solver::Summary */
  int32_t iterations;                    /* summary.iterations.size() - 1 */
  int32_t usable;                        /* IsSolutionUsable() */
  int32_t num_residual_blocks;
  int32_t linear_iterations;             /* conjugate-gradient iterations over all steps */
} cfear_pgo_summary;
int cfear_pgo_solve(cfear_pose3d* poses, const uint64_t* ids, int32_t n, const cfear_graph_constraint* constraints,
                    int32_t m, const cfear_pgo_params* par, cfear_pgo_summary* summary);

/* OdometryKeyframeFuser::AddToGraph (odometrykeyframefuser.cpp:428-445) for `stream`: the odometry constraint from the
 * keyframe added by the LAST processed frame to the keyframe before it -- id_begin / id_end are the stream's keyframe
 * ordinals (RadarScan::counter), t_be = Tfrom^-1 * Tto, type odometry.  The reference stores information = C.inverse()
 * of cov_current with its 3x3 block rotated into the from-frame; Register's constant diag(0.01, 0.01, 0, 0, 0, 1e-4) is
 * singular, so the reference's matrix is not finite -- here information is the inverse on the planar (x, y, yaw)
 * sub-space and zero elsewhere.  CFEAR_ERR_INVALID_ARGUMENT if the last frame added no keyframe or the first one.   */
int cfear_odometry_get_constraint(cfear_odometry* od, int32_t stream, cfear_graph_constraint* out);

#ifdef __cplusplus
}
#endif
#endif /* CFEAR_HIP_H */
