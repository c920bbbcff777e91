// common.hpp -- shared host/device infrastructure of libcfear_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cfear_hip.h"

#define CFEAR_WAVE 64

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct ProfRow {
  const char* name;
  double total_ms = 0.0;
  int64_t launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct cfear_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string last_error;
  int profile = 0;            // 0 off, 1 every kernel family, 2 the polar filter's row kernels only (the HBM-bound launch)
  std::vector<ProfRow> prof;
  std::vector<hipEvent_t> event_pool;
  // grow-only device workspaces (indexed by purpose so stages of one pipeline do not alias)
  struct Ws { void* p = nullptr; size_t bytes = 0; };
  Ws ws[16];
  // pinned host staging for small read-backs
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  hipEvent_t pinned_ev = nullptr;   // recorded behind the last asynchronous copy OUT of `pinned` (cfear_pinned_mark)
  bool pinned_busy = false;
  // free list of scan slabs (capacity -> device pointers) so streaming does not hipMalloc
  struct Slab { void* p; int cap; };
  std::vector<Slab> free_slabs;
  int64_t live_scans = 0;
  int trig_rows = 0;       // rows the cos/sin tables in ws[2] were built for
  int n_cu = 256;          // compute units of the device (cfear_ctx_create)
  int64_t opt[CFEAR_OPT_COUNT] = {1, 0, 0, 0};   // cfear_ctx_set_option (test / measurement hooks; include/cfear_hip.h)
  bool surf_list_dirty = true;   // the surface pipeline's hand-over counter may be non-zero (see cfear_surface_launch)
  std::vector<std::pair<const void*, int>> lds_allowed;   // kernels and the dynamic LDS they were already allowed on this device (cfear_allow_lds)
};

int cfear_set_error(cfear_ctx* ctx, int status, const char* fmt, ...);

#define CFEAR_HIP_CHECK(ctx, expr)                                                              \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return cfear_set_error((ctx), CFEAR_ERR_HIP, "%s failed: %s (%s:%d)", #expr,              \
                             hipGetErrorString(_e), __FILE__, __LINE__);                        \
  } while (0)

// true if p is device (or managed) memory visible to kernels without a copy
bool cfear_is_device_ptr(const void* p);
// grow-only workspace `slot` of at least `bytes`; returns nullptr on allocation failure
void* cfear_workspace(cfear_ctx* ctx, int slot, size_t bytes);
// Pinned staging of at least `bytes`.  A caller that leaves an asynchronous copy from it in flight (results kept on the
// device: no synchronisation before it returns) calls cfear_pinned_mark() behind the copy; cfear_pinned() then waits for
// that copy before it hands the buffer out again.
void* cfear_pinned(cfear_ctx* ctx, size_t bytes);
void cfear_pinned_mark(cfear_ctx* ctx);
// hipFuncAttributeMaxDynamicSharedMemorySize, once per (context, kernel, size): the attribute is per device and contexts are
// per device, so a launch path asks every time and pays a linear look-up over a dozen entries instead of a runtime call
int cfear_allow_lds(cfear_ctx* ctx, const void* kernel, size_t bytes);

// profiling: wrap a kernel launch
int cfear_prof_row(cfear_ctx* ctx, const char* name);
void cfear_prof_begin(cfear_ctx* ctx, int row);
void cfear_prof_end(cfear_ctx* ctx, int row);

struct ProfScope {
  cfear_ctx* ctx;
  int row;
  ProfScope(cfear_ctx* c, const char* name) : ctx(c), row(-1) {
    if (ctx->profile == 1 || (ctx->profile == 2 && (!strcmp(name, "kstrongest_rows") || !strcmp(name, "cacfar_rows") || !strcmp(name, "kstrong_image")))) {
      row = cfear_prof_row(ctx, name);
      cfear_prof_begin(ctx, row);
    }
  }
  ~ProfScope() { if (row >= 0) cfear_prof_end(ctx, row); }
};

// ---------------------------------------------------------------------------------------------
// device-resident MapPointNormal ("scan"): SoA slab of `cap` cells
// ---------------------------------------------------------------------------------------------
struct ScanView {          // plain pointers into one slab; passed to kernels by value / in tables
  float2* mean_f;          // float copy of the means (pointnormal.cpp:151-162), NN search space
  double2* mean;           // u_
  double2* normal;         // snormal_
  double4* cov;            // cov_ row-major (c00,c01,c10,c11)
  double* scale;           // scale_
  double* avg_intensity;
  double2* lambda;         // (lambda_min, lambda_max)
  int32_t* nsamples;
  int32_t* n_cells;        // device counter (written by the surface kernel)
  // The matcher's search structure, prebuilt once per scan (grid_cells_block): the float means bucketed into a uniform
  // kScanGrid x kScanGrid grid over the scan's own extent -- what the reference keeps as a kd-tree per MapPointNormal
  // (pointnormal.cpp:151-162).  ONE block that a registration copies into LDS in 16-byte pieces:
  //   [kScanGridStartPad] u16  first record of every grid cell (row-major), entries past the last cell = n
  //   [np] float2              (x, y) of the records, grouped by grid cell            np = scan_grid_pad(n)
  //   [np] u16                 the records' cell indices (10 bytes per record in all: four keyframes + the LM arrays fit 40 KB)
  unsigned short* grid;
  float4* grid_geo;        // (x0, y0, cells per metre, 1 = tables valid | 0 = more cells than the 16-bit table can address)
  int32_t cap;
  int32_t pad;
};
constexpr int kScanGrid = 32;                                  // grid cells per axis
constexpr int kScanGridCells = kScanGrid * kScanGrid;
constexpr int kScanGridStartPad = kScanGridCells + 8;          // u16 entries per scan: a multiple of 16 bytes
__host__ __device__ inline int scan_grid_pad(int n) { return (n + 7) & ~7; }        // records per scan, padded: both record arrays are runs of 16-byte pieces
__host__ __device__ inline size_t scan_grid_bytes(int n) { return (size_t)kScanGridStartPad * 2 + (size_t)scan_grid_pad(n) * 10; }
constexpr float kScanGridMinEdge = 2.0f;                       // cell edge floor [m]: the matcher's radius (registration.h:131)

struct cfear_scan {
  cfear_ctx* ctx;
  void* slab;
  ScanView view;
  int32_t n_cells_host;    // -1 until read back
  int32_t refs = 1;        // the caller's handle + one per scan table that names it (handles are used from one host thread)
};
void cfear_scan_retain(cfear_scan* scan);   // cfear_scan_destroy drops one reference; the slab is recycled with the last

size_t cfear_scan_slab_bytes(int cap);
ScanView cfear_scan_view(void* slab, int cap);
int cfear_scan_alloc(cfear_ctx* ctx, int cap, cfear_scan** out);
int cfear_scan_clone_view(cfear_ctx* ctx, const ScanView& src, int n_cells, cfear_scan** out);

// ---------------------------------------------------------------------------------------------
// cross-file entry points (device-pointer level; used by the batched odometry pipeline)
// ---------------------------------------------------------------------------------------------
// Optional extras of the device-side k-strongest entry (the batched odometry pipeline):
//   row_keys / row_valid    the kept bins beyond min_distance of every row as packed keys (intensity << 24 | range bin)
//                           at [(b * rows + r) * k + j], j < row_valid[(b * rows + r) * 2], in the reference's cloud
//                           order (k <= 64); the surface-point kernel compacts the rows and converts them to PointXYZI,
//                           so neither sel_* arrays nor a cloud kernel are needed;
//   image_offsets           device [batch] byte offsets of the images from d_polar (streams whose sweeps do not sit at
//                           a constant stride: ring buffers, one buffer per sequence) instead of b * batch_stride.
struct cfear_kstrong_fused {
  uint32_t* row_keys = nullptr;
  int32_t* row_valid = nullptr;
  const int64_t* image_offsets = nullptr;
  uint32_t* cand_stats = nullptr;     // fused decode only: device [64] counters, += the candidates (bins >= z_min) of the batch
};
int cfear_kstrong_device(cfear_ctx* ctx, const uint8_t* d_polar, const cfear_polar_desc* desc,
                         const cfear_kstrong_params* par, const cfear_kstrong_out* o, bool dense_halo = false,
                         const cfear_kstrong_fused* fused = nullptr);
// fused decode + sweep for [range bins][azimuths] sources (sd = the SOURCE images); key output only
bool cfear_kstrong_cols_supported(const uint8_t* d_src, const cfear_polar_desc* sd, const cfear_kstrong_params* par);
bool cfear_kstrong_cols_preferred(const uint8_t* d_src, const cfear_polar_desc* sd, const cfear_kstrong_params* par);   // ... and the batch is large enough to pay
int cfear_kstrong_cols_device(cfear_ctx* ctx, const uint8_t* d_src, const cfear_polar_desc* sd, const cfear_kstrong_params* par,
                              const cfear_kstrong_fused* fused, int route = 0);
int cfear_compensate_batch_device(cfear_ctx* ctx, float* d_xyzi, size_t cloud_stride_points, const int32_t* d_n,
                                  const double* d_mot, int n_clouds, int max_points, int ccw);
int cfear_rotate_ccw_device(cfear_ctx* ctx, const uint8_t* d_src, const cfear_polar_desc* src_desc, uint8_t* d_dst,
                            int dst_stride, int64_t dst_batch_stride);
// CA-CFAR for the batched odometry: per-row key lists (intensity << 24 | range bin, ascending bins; row_cnt[2 row] = the
// row's detections, which may exceed kcap -- the keys beyond are dropped and surface_prep_kernel reports the scan) in the
// layout of cfear_kstrong_fused, so surface_prep_kernel compacts and converts them (rho = range_res * bin, cfar.cpp:43).
struct cfear_cacfar_fused {
  uint32_t* row_keys = nullptr;
  int32_t* row_cnt = nullptr;
  int kcap = 0;
  // desc describes [range bins][azimuths] SOURCE images (rows = bins, cols = azimuths): the decode is fused into the filter
  // (cacfar_cols_kernel); only where cfear_cacfar_cols_supported() says so
  bool bins_major = false;
};
bool cfear_cacfar_cols_supported(const uint8_t* d_src, const cfear_polar_desc* sd, const cfear_cacfar_params* par);
int cfear_cacfar_device(cfear_ctx* ctx, const uint8_t* d_polar, const cfear_polar_desc* desc,
                        const cfear_cacfar_params* par, float* d_xyzi, int32_t* d_n_points,
                        int32_t cap_points, uint8_t* d_det_mask, const cfear_cacfar_fused* fused = nullptr);
size_t cfear_surface_lds_bytes();
size_t cfear_surface_scratch_bytes(int cap_points);   // per scan, for clouds of up to cap_points points
size_t cfear_surface_job_bytes();
int cfear_surface_max_points();
void cfear_surface_fill_job(void* dst, float* d_xyzi, const int32_t* d_n, int32_t n_host, int compensate,
                            const double mot[3], const ScanView& out);
// rows mode: the cloud is handed over as the fused filter output (cfear_kstrong_fused) and compacted by the kernel
// into d_xyzi (capacity cfear_surface_max_points()); d_n_out receives the point count.  rows <= 4096.
void cfear_surface_fill_job_rows(void* dst, float* d_xyzi, int32_t* d_n_out, const uint32_t* d_row_keys, const int32_t* d_row_cnt,
                                 int rows, int k, int compensate, const double mot[3], const ScanView& out);
// rows mode needs the polar -> Cartesian constants: call before cfear_surface_launch (per context)
// rho = range_off + range_res * bin: range_off = range_res / 2 for the k-strongest filter (radar_filters.cpp:324), 0 for CA-CFAR
struct cfear_surface_polar { const double* cos_t = nullptr; const double* sin_t = nullptr; double range_res = 0.0, range_off = 0.0; int rows = 0, k = 0; };
int cfear_trig_tables(cfear_ctx* ctx, int rows, double** d_cos, double** d_sin);
// max_cell_cap: the largest cell capacity (ScanView::cap) among the jobs' output slabs
int cfear_surface_launch(cfear_ctx* ctx, const void* d_jobs, int n_jobs, const cfear_feature_params* par,
                         char* d_scratch, int32_t* d_status, int32_t* d_ncells_out, int max_cell_cap, int cap_points,
                         const cfear_surface_polar* polar = nullptr);
int cfear_register_batch_device(cfear_ctx* ctx, const cfear_reg_job* jobs, int32_t n_jobs, const cfear_reg_params* par,
                                cfear_reg_result* d_out, cfear_reg_result** d_used);
size_t cfear_reg_job_bytes();
int cfear_reg_max_scans();
void cfear_reg_fill_job(void* dst, const ScanView* views, int n_scans, const double* poses_xyt);
size_t cfear_register_scratch_bytes(int pairs_cap);
// Cost-only launches of the matcher (GetCost / cost sampling): n_samples = 0 evaluates each job
// at its own source pose, n_samples = samples_per_axis^3 at the sampling grid around it; results then holds
// max(n_samples, 1) records per job.  blocks_per_job workgroups share a job's samples (scratch: n_jobs x
// blocks_per_job x cfear_register_scratch_bytes).
struct RegCostMode {
  int n_samples = 0, samples_per_axis = 0, blocks_per_job = 1;
  double xy_half = 0.0, yaw_half = 0.0;
  const cfear_reg_result* prior = nullptr;   // device, [n_jobs]: evaluate around prior[j].pose with itr = prior[j].outer_iters
};
// What the caller knows about the batch's registrations (the kernel decides per registration from the device-side sizes):
// cfear_coral_quality_batch in two halves (coral.hip): launch, then read back -- host work of the caller in between
struct CoralPending {
  std::vector<char> host_jobs;        // the uploaded job records (pageable: must outlive the copy)
  void* d_res = nullptr;
  void* d_pp = nullptr;
  int n_jobs = 0, cap = 0;
};
int cfear_coral_enqueue(cfear_ctx* ctx, const cfear_coral_job* jobs, int32_t n_jobs, const cfear_coral_params* par, bool want_per_point,
                        CoralPending& pend);
int cfear_coral_collect(cfear_ctx* ctx, const cfear_coral_job* jobs, const CoralPending& pend, cfear_coral_result* results, double* per_point);

// One registration of the matcher (matcher.hip).  The scan views come LAST so that a batch whose jobs use at most m scans can be
// stored with the shorter stride reg_job_stride(m): a two-scan loop-closure candidate is 0.5 KB instead of 1.9 KB to build and
// upload.  Kernels only ever touch scans[0 .. n_scans).
constexpr int kRegMaxScans = 16;
struct RegJob {
  int32_t n_scans;
  int32_t itr;                                // cost-only launches: this job's leftover itr_ (0: use par.itr)
  double poses[kRegMaxScans][3];
  ScanView scans[kRegMaxScans];
};
inline size_t reg_job_stride(int max_scans) {
  return (offsetof(RegJob, scans) + (size_t)max_scans * sizeof(ScanView) + 15) & ~(size_t)15;
}
// One CorAl job of coral_kernel (coral.hip): two peak clouds on the DEVICE and their poses
struct CoralJob {
  const float4* ref;
  const float4* src;
  const int32_t* n_ref_ptr;           // device-side counts (pipelines), or nullptr -> the host values
  const int32_t* n_src_ptr;
  int32_t n_ref, n_src;
  double ref_pose[3], src_pose[3], offset[3];
};
// coral_kernel over job records that already sit on the device (verify.hip builds them there): results to d_results [n_jobs]
// (device), nothing is synchronised.  cap = the largest n_ref + n_src among the jobs (the per-job scratch's size).
int cfear_coral_launch_device(cfear_ctx* ctx, const CoralJob* d_jobs, int n_jobs, int cap, const cfear_coral_params* par,
                              cfear_coral_result* d_results);
int cfear_coral_max_points();

struct RegLaunchHint {
  bool small_pairs = false;   // every job is a two-scan candidate that fits 20 KB of LDS: the 2-wavefront form, eight per CU
  bool big_pass = false;      // registrations the regular form is not good at may be among them (dense scans): add the large forms
  bool whole_cu = false;      // cost-only launches: a job that does not fit half a CU's LDS is among them
};
int cfear_register_launch(cfear_ctx* ctx, const void* d_jobs, int n_jobs, const cfear_reg_params* par, int pairs_cap,
                          char* d_scratch, cfear_reg_result* d_results, const RegCostMode* mode = nullptr,
                          size_t job_stride = 0,    // 0: full records (cfear_reg_job_bytes)
                          RegLaunchHint hint = RegLaunchHint());
// candidate batches in two halves (matcher.hip; cfear_candidate_pipe in shard.hip runs them on different streams)
struct CandGeometry { int pairs_cap = 1; RegLaunchHint hint; };
struct cfear_scan_table;
int cfear_candidates_expand(cfear_ctx* ctx, hipStream_t stream, const cfear_scan_table* table, const cfear_candidate* cands, int32_t n,
                            const cfear_reg_params* par, cfear_candidate* h_stage, char* d_jobs, int32_t* d_trailer, int trailer_status,
                            CandGeometry* geom);
int cfear_candidates_match(cfear_ctx* ctx, const char* d_jobs, int32_t n, const cfear_reg_params* par, const CandGeometry* geom,
                           cfear_reg_result* d_res);
int cfear_candidates_enqueue(cfear_ctx* ctx, const cfear_scan_table* table, const cfear_candidate* cands, int32_t n,
                             const cfear_reg_params* par, cfear_candidate* h_stage, cfear_reg_result* d_res, int32_t* d_trailer,
                             int trailer_status);
int cfear_check_reg_params(cfear_ctx* ctx, const cfear_reg_params* par);
// launch geometry of a batch of two-scan jobs from its largest target / source (what cfear_register_candidates derives)
void cfear_reg_pair_geometry(const cfear_reg_params* par, int max_tar_cells, int max_src_cells, int* pairs_cap, RegLaunchHint* hint);
size_t cfear_reg_job_stride(int max_scans);         // bytes of a record that holds up to max_scans scan views
void cfear_reg_job_set_itr(void* job, int itr);
// quadratic fit of the cost samples -> 6x6 covariance (covariance.hip, host)
struct CovFit {
  int n = 0, m = 0;
  std::vector<double> offsets;   // [m][3] sample offsets (x, y, yaw) in the reference's loop order
  std::vector<double> pinv;      // [10][m] pseudo-inverse of the design matrix
  void prepare(int samples_per_axis, double xy_half, double yaw_half);
  bool solve(const double* costs /*[m]*/, double score_scale, double covariance_scaler, double cov36[36]) const;
};

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#if defined(__HIPCC__)

// Loads and stores through pointers that are KNOWN to address global memory but whose static type is generic (pointers kept
// in job records, LDS tables, ScanView): a generic pointer compiles to flat_load / flat_store, which count in lgkmcnt as
// well as vmcnt, so every wait for an LDS read behind one also waits for the memory round trip -- a loop that mixes LDS
// reads with such loads runs them one at a time.  T: a scalar or an ext_vector_type (HIP's double2 / float4 structs have
// no address-space-qualified copy constructors).
typedef uint32_t g_u32x4 __attribute__((ext_vector_type(4)));
typedef double g_f64x2 __attribute__((ext_vector_type(2)));
typedef double g_f64x4 __attribute__((ext_vector_type(4)));
typedef float g_f32x4 __attribute__((ext_vector_type(4)));
typedef float g_f32x2 __attribute__((ext_vector_type(2)));
template <class T>
__device__ __forceinline__ T gload(const void* p) { return *(const __attribute__((address_space(1))) T*)p; }
template <class T>
__device__ __forceinline__ void gstore(void* p, T v) { *(__attribute__((address_space(1))) T*)p = v; }
__device__ __forceinline__ double2 gload_d2(const double2* p) { const g_f64x2 v = gload<g_f64x2>(p); return make_double2(v.x, v.y); }
__device__ __forceinline__ double4 gload_d4(const double4* p) { const g_f64x4 v = gload<g_f64x4>(p); return make_double4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 gload_f4(const float4* p) { const g_f32x4 v = gload<g_f32x4>(p); return make_float4(v.x, v.y, v.z, v.w); }

// DPP move: lanes without a valid source (or masked off) receive 0.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND = true>
__device__ __forceinline__ int dpp_mov(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, BANK_MASK, BOUND);
}

// Inclusive prefix sum across the 64 lanes of a wavefront (row_shr Hillis-Steele inside each
// 16-lane DPP row, then row_bcast:15 / row_bcast:31 to carry row totals).  6 DPP adds, no LDS.
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
  v += dpp_mov<0x111>(v);                      // row_shr:1
  v += dpp_mov<0x112>(v);                      // row_shr:2
  v += dpp_mov<0x114>(v);                      // row_shr:4
  v += dpp_mov<0x118>(v);                      // row_shr:8
  v += dpp_mov<0x142, 0xa, 0xf, false>(v);     // row_bcast:15 -> rows 1,3
  v += dpp_mov<0x143, 0xc, 0xf, false>(v);     // row_bcast:31 -> rows 2,3
  return v;
}

// Inclusive prefix maximum of non-negative values (same DPP ladder).
__device__ __forceinline__ int wave_incl_scan_max_i32(int v) {
  v = max(v, dpp_mov<0x111>(v));
  v = max(v, dpp_mov<0x112>(v));
  v = max(v, dpp_mov<0x114>(v));
  v = max(v, dpp_mov<0x118>(v));
  v = max(v, dpp_mov<0x142, 0xa, 0xf, false>(v));
  v = max(v, dpp_mov<0x143, 0xc, 0xf, false>(v));
  return v;
}

// Wave-uniform sum (result in an SGPR).
__device__ __forceinline__ int wave_sum_i32(int v) {
  return __builtin_amdgcn_readlane(wave_incl_scan_i32(v), 63);
}

// Sum of a double over each 16-lane DPP row; every lane of the row receives the row total.
// Order: pairwise butterfly (quad xor 1, xor 2, half-mirror, mirror) -- deterministic.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// DPP move with a bank mask: lanes of disabled banks (lane quads of each row) keep `old`.
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_sel_f64(double old, double src) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xf, BANK, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xf, BANK, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum_f64(double v) {
  v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);   // row_half_mirror
  v += dpp_f64<0x140>(v);   // row_mirror
  return v;
}
// Wave total in lane 63 only: row sums, then (R3 + R2) + (R1 + R0) through row_bcast -- no readlanes.
__device__ __forceinline__ double wave_sum_lane63_f64(double v) {
  v = row16_sum_f64(v);
  {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xa, 0xf, false);
    v += __hiloint2double(hi, lo);
  }
  {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xc, 0xf, false);
    v += __hiloint2double(hi, lo);
  }
  return v;
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// Wave-uniform sum of a double: row sums, then rows 0..3 added in order.
__device__ __forceinline__ double wave_sum_f64(double v) {
  v = row16_sum_f64(v);
  return ((readlane_f64(v, 0) + readlane_f64(v, 16)) + readlane_f64(v, 32)) + readlane_f64(v, 48);
}
#endif  // __HIPCC__
