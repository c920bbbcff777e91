/*
 * cfear_oracle.h -- C API of the CPU oracle for the CFEAR scan-registration hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so, and there only
 * as the checker / the timed CPU baseline.  The product (libcfear_hip.so) never links, loads or
 * calls it.
 *
 * PARITY UNPINNED: the reference (dan11003/tbv_slam_public) cannot be compiled here (needs
 * ROS/PCL/FLANN/Eigen/Ceres/OpenCV), ships no tests or golden vectors for this path and no radar
 * data.  This oracle is a restatement of the reference algorithm written from its sources
 * (file:line cited at each function in cfear_oracle.cpp) plus the published semantics of the
 * third-party routines it calls (PCL 1.10 VoxelGrid / FLANN 1.9.1 L2_Simple / Ceres 2.1.0
 * trust-region LM).  It is cross-checked in tests/ against independent brute-force NumPy/SciPy
 * computations, not against the reference itself.
 */
#ifndef CFEAR_ORACLE_H
#define CFEAR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One oriented surface point (reference: class cell, pointnormal.h:45-105). */
typedef struct orc_cell {
  double mean[2];        /* u_                                           */
  double normal[2];      /* snormal_ (flipped toward origin)             */
  double cov[4];         /* cov_ row-major                               */
  double scale;          /* scale_ = log(1 + cond/2) ("planarity")       */
  double avg_intensity;  /* avg_intensity_                               */
  double lambda_min, lambda_max;
  int32_t nsamples;      /* Nsamples_                                    */
  int32_t pad;
} orc_cell;

typedef struct orc_reg_params {
  int32_t cost;          /* 0=P2P 1=P2L 2=P2D   (registration.h:55 costmetric)                */
  int32_t loss;          /* 0=None 1=Huber 2=Cauchy 3=SoftLOne 4=Combined 5=Tukey (:60)       */
  double loss_limit;     /* loss_limit_ (registration.h:119)                                  */
  int32_t weight_opt;    /* 0..4 (registration.h:50)                                          */
  int32_t max_outer;     /* max_itr_association_ (n_scan_normal.h:75)                         */
  int32_t max_inner;     /* options_.max_num_iterations (n_scan_normal.cpp:9)                 */
  int32_t min_outer;     /* min_itr_ = 3 (n_scan_normal.h:75)                                 */
  double radius;         /* radius_ = 2.0 (registration.h:122)                                */
  double cov_scale;      /* cov_scale_ (P2D)                                                  */
  double regularization; /* regularization_ (P2D)                                             */
  double score_tolerance;/* 1e-5 (n_scan_normal.h:74)                                         */
  int32_t first_itr;     /* value of itr_ used by GetCost to pick the radius (0 = fresh obj)  */
  int32_t pad;
} orc_reg_params;

typedef struct orc_reg_result {
  double pose[3];        /* final (x,y,theta) of the last (free) scan                         */
  double score;          /* score_ = final_cost / num_residuals                               */
  double final_cost;     /* summary_.final_cost of the last solve                             */
  int32_t num_residuals; /* residual ELEMENTS of the last problem                             */
  int32_t outer_iters;   /* value of itr_ when the loop was left (timing key "itrs")          */
  int32_t lm_iters;      /* total LM iterations (summary_.iterations.size()-1 summed)         */
  int32_t status;        /* 1 = Register returned true, 0 = false                             */
} orc_reg_result;

/* ---- F: polar filters ------------------------------------------------------------------ */
/* radar_filters.cpp:209-237.  sel_* are [rows][k]; entries past sel_count[row] are -1 / 0. */
int orc_kstrongest(const uint8_t* img, int rows, int cols, int stride, int k, int z_min,
                   int32_t* sel_range, uint8_t* sel_intensity, int32_t* sel_count);
/* radar_filters.cpp:238-298.  is_peak[rows][k] (1 = kept by AxialNonMaxSupress).          */
int orc_peaks(const uint8_t* img, int rows, int cols, int stride, int k,
              const int32_t* sel_range, const int32_t* sel_count, uint8_t* is_peak);
/* radar_filters.cpp:309-337.  mask nullable.  Returns number of points written to xyzi.   */
int orc_kstrongest_cloud(int rows, int k, const int32_t* sel_range, const uint8_t* sel_intensity,
                         const int32_t* sel_count, const uint8_t* mask, float range_res,
                         float min_distance, float* xyzi);
/* Legacy k_strongest_filter / InsertStrongestK (radar_filters.cpp:25-78; CorAl's kstrongRadar).  Returns the number
 * of points written ([n][4]: x, y, 0, intensity; rows ascending, within a row descending intensity), -1 if cap is too small. */
int orc_kstrongest_legacy(const uint8_t* img, int rows, int cols, int stride, int k_strongest, double z_min,
                          double range_res, double min_distance, float* xyzi, int cap);
/* cfar.cpp:12-83 + radar_driver.cpp:52-56.  Returns number of detections; det_rc nullable. */
int orc_cacfar(const uint8_t* img, int rows, int cols, int stride, int window, int guard,
               float false_alarm_rate, float range_res, float z_min, float min_distance,
               double max_distance, float* xyzi, int32_t* det_rc, int cap);

/* ---- C: motion compensation (utils.cpp:96-113, utils.h:28-32) --------------------------- */
void orc_compensate(float* xyzi, int n, const double mot[3], int ccw);

/* ---- N: oriented surface points (pointnormal.cpp:7-63, 65-90, 151-162, 265-297) --------- */
/* Returns number of cells (<= cap) or -1 if cap too small.  centroids (nullable) receives the
 * voxel-grid sample points as [n_voxels][2] floats and n_voxels their count.               */
int orc_surface_points(const float* xyzi, int n, float radius, double downsample_factor,
                       const double origin[2], int weight_intensity, orc_cell* cells, int cap,
                       float* centroids, int* n_voxels);

/* ---- M: registration (n_scan_normal.cpp, registration.cpp, Ceres 2.1 LM) ---------------- */
/* scans[i] = cells of scan i (n_cells[i] of them); poses [n_scans][3] in/out.              */
int orc_register(const orc_cell* const* scans, const int32_t* n_cells, int n_scans,
                 double* poses_xyt, const orc_reg_params* par, orc_reg_result* res);
/* n_scan_normal.cpp:186-211.  residuals cap >= 2*sum(n_cells).  Returns 1 on success.     */
int orc_get_cost(const orc_cell* const* scans, const int32_t* n_cells, int n_scans,
                 const double* poses_xyt, const orc_reg_params* par, double* cost,
                 double* residuals, int32_t* n_res, double* score);
/* One association pass at the given poses (n_scan_normal.cpp:213-261).  pairs[n][3] =
 * (target scan index, target cell, source cell); weights[n]. Returns n.                    */
int orc_associate(const orc_cell* const* scans, const int32_t* n_cells, int n_scans,
                  const double* poses_xyt, const orc_reg_params* par, int itr,
                  int32_t* pairs, double* weights, int cap);
/* Robustified normal equations at x for the association set built at `poses` (used to check
 * the device accumulate kernel): H[9] = J^T J, g[3] = J^T r, cost = 1/2 sum rho.           */
int orc_normal_eq(const orc_cell* const* scans, const int32_t* n_cells, int n_scans,
                  const double* poses_xyt, const orc_reg_params* par, int itr, const double x[3],
                  double H[9], double g[3], double* cost, int32_t* n_res);

/* Test hook: one ceres::Solve on the association set built at `poses` (iteration counter `itr`), from the last pose;
 * trace[i] = {cost, relative_decrease, step_is_successful, trust_region_radius} of summary.iterations[i].  Returns the
 * number of iterations recorded in the summary.                                                                   */
int orc_lm_trace(const orc_cell* const* scans, const int32_t* n_cells, int n_scans, const double* poses_xyt,
                 const orc_reg_params* par, int itr, int max_iter, double x_out[3], double* trace, int cap,
                 double* final_cost, int32_t* usable);

/* Covariance by cost sampling (odometrykeyframefuser.cpp:261-380, loopclosure.cpp:99-208): n^3
 * GetCost samples around the registered pose, quadratic least-squares fit, 2 H^-1 scaled by
 * GetCovarianceScaler.  par->first_itr = leftover itr_; final_cost / num_residuals from the
 * Register summary.  cov36 row-major 6x6; samples_out [n^3][4] optional.  Returns 1 = valid. */
/* MapPointNormal::GetClosestIdx (pointnormal.cpp:238-254) for n_queries points: index of the nearest cell mean or -1 */
void orc_closest_idx(const orc_cell* cells, int n_cells, const double* queries_xy, int n_queries, double d, int32_t* idx);
int orc_cov_by_sampling(const orc_cell* const* scans, const int32_t* n_cells, int n_scans,
                        const double* poses_xyt, const orc_reg_params* par, double final_cost,
                        int32_t num_residuals, double xy_range, double yaw_range,
                        int32_t samples_per_axis, double covariance_scaler, double* cov36,
                        double* samples_out);

/* CorAl alignment quality of two peak clouds (AlignmentQuality.cpp:8-230 as called from
 * alignmentinterface.cpp:437-456): ref/src clouds [n][4] (x,y,z,intensity) in their sensor frames,
 * planar poses, Toffset applied to the source.  quality = {joint, sep, overlap}; returns valid_.
 * per_point optional [n_src+n_ref][3] = joint_res, sep_res, valid (src points first).          */
int orc_coral_quality(const float* ref_xyzi, int n_ref, const float* src_xyzi, int n_src,
                      const double ref_pose[3], const double src_pose[3], const double offset[3],
                      double radius, int weight_res_intensity, double quality[3], double* per_point);

/* Scan Context on radar clouds (RadarScancontext.cpp:59-131, Scancontext.cpp:60-268).  Descriptors are
 * row-major [ring][sector].  desc_function 0 = sum, 1 = max; shift_y = the y offset of an augmentation. */
void orc_sc_descriptor(const float* xyzi, int n, int num_ring, int num_sector, double max_radius,
                       int desc_function, double desc_divider, double no_point, double shift_y, double* desc);
void orc_sc_keys(const double* desc, int num_ring, int num_sector, double* ringkey, double* sectorkey);
double orc_sc_distance(const double* sc1, const double* sc2, int num_ring, int num_sector,
                       double search_ratio, int32_t* argmin_shift);

/* ---- caller: OdometryKeyframeFuser (odometrykeyframefuser.cpp:62-94,143-259,470-494) ----- */
typedef struct orc_fuser_params {
  orc_reg_params reg;
  float  res;                 /* par.res (double in the reference, narrowed to the float radius) */
  int32_t submap_scan_size;
  int32_t weight_intensity, use_guess, compensate, radar_ccw, use_keyframe;
  double min_keyframe_dist, min_keyframe_rot_deg;
  double downsample_factor;
  int32_t estimate_cov_by_sampling;  /* par.estimate_cov_by_sampling (odometrykeyframefuser.h:104)      */
  int32_t cov_samples_per_axis;      /* :109 */
  double cov_xy_range, cov_yaw_range, cov_scaler;   /* :107, :108, :110 */
} orc_fuser_params;
typedef struct orc_fuser orc_fuser;
orc_fuser* orc_fuser_create(const orc_fuser_params* p);
void orc_fuser_destroy(orc_fuser* f);
/* wall seconds spent in {Filtering (orc_fuser_run_sequence only), compensate, build_normals, register} so far, named
 * like the reference's timing keys (radar_driver.cpp:87; odometrykeyframefuser.cpp:253-255), and the frames covered */
void orc_fuser_stage_times(const orc_fuser* f, double seconds[4], int64_t* n_frames);
/* cov_current after the last processed frame (row-major 6x6); *sampled = 1 if it is the sampled one. */
void orc_fuser_last_cov(const orc_fuser* f, double cov36[36], int32_t* sampled);
/* Feeds one filtered cloud (modified in place by compensation).  pose_out = Tcurrent (x,y,th).
 * info[0]=n_cells, info[1]=keyframe added, info[2]=register status, info[3]=outer iters.   */
int orc_fuser_process(orc_fuser* f, float* xyzi, int n, double pose_out[3], int32_t info[4]);
/* k-strongest filter + fuser over n_frames images [n_frames][rows][cols] in one call; poses_out [n_frames][3] */
int orc_fuser_run_sequence(orc_fuser* f, const uint8_t* imgs, int n_frames, int rows, int cols, int k, int z_min,
                           float range_res, float min_distance, double* poses_out);

#ifdef __cplusplus
}
#endif
#endif
