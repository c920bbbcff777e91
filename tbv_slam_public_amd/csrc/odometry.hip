// odometry.hip -- batched radarDriver + OdometryKeyframeFuser: n_streams independent sequences advance
// one frame per call, polar image in -> SE(2) pose out, everything in between on the GPU.
//
// Restates per stream (cfear_radarodometry/src/cfear_radarodometry/):
//   radarDriver::CallbackOffline / Process            radar_driver.cpp:48-73, 163-176
//   OdometryKeyframeFuser::processFrame               odometrykeyframefuser.cpp:143-259
//     KeyFrameBasedFuse                               :62-73
//     AccelerationVelocitySanityCheck                 :76-94
//     FormatScans / AddToReference                    :470-494
// The frame policy (constant-velocity guess, sanity check, keyframe test, sliding keyframe window)
// is host code exactly as in the reference; the four stages F, C, N, M are three kernel launches
// over the whole batch (kstrongest_rows + kstrong_cloud, surface_points, register) and one small
// read-back (pose, counts) per call.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>

#include "common.hpp"

namespace {

struct Aff2 { double l[4]; double t[2]; };
Aff2 aff_identity() { return Aff2{{1, 0, 0, 1}, {0, 0}}; }
Aff2 aff_from_xyt(double x, double y, double th) {      // registration.cpp:128-135 vectorToAffine3d
  const double c = std::cos(th), s = std::sin(th);
  return Aff2{{c, -s, s, c}, {x, y}};
}
Aff2 aff_mul(const Aff2& a, const Aff2& b) {
  Aff2 r;
  r.l[0] = a.l[0] * b.l[0] + a.l[1] * b.l[2];
  r.l[1] = a.l[0] * b.l[1] + a.l[1] * b.l[3];
  r.l[2] = a.l[2] * b.l[0] + a.l[3] * b.l[2];
  r.l[3] = a.l[2] * b.l[1] + a.l[3] * b.l[3];
  r.t[0] = a.l[0] * b.t[0] + a.l[1] * b.t[1] + a.t[0];
  r.t[1] = a.l[2] * b.t[0] + a.l[3] * b.t[1] + a.t[1];
  return r;
}
Aff2 aff_inv(const Aff2& a) {                            // Eigen Affine inverse
  const double det = a.l[0] * a.l[3] - a.l[2] * a.l[1];
  const double invdet = 1.0 / det;
  Aff2 r;
  r.l[0] = a.l[3] * invdet; r.l[1] = -a.l[1] * invdet; r.l[2] = -a.l[2] * invdet; r.l[3] = a.l[0] * invdet;
  r.t[0] = -(r.l[0] * a.t[0] + r.l[1] * a.t[1]);
  r.t[1] = -(r.l[2] * a.t[0] + r.l[3] * a.t[1]);
  return r;
}
void aff_to_xyt(const Aff2& a, double p[3]) {            // utils.cpp:115-122 Affine3dToVectorXYeZ
  p[0] = a.t[0]; p[1] = a.t[1]; p[2] = std::atan2(a.l[2], a.l[3]);
}

}  // namespace

// ---- frame policy as pure host functions (no GPU, no context): the two decisions OdometryKeyframeFuser takes per frame ----
// KeyFrameBasedFuse (odometrykeyframefuser.cpp:62-73): diff = T_keyframe^-1 * Tcurrent as (x, y, theta).  For a planar
// pose Matrix3d::eulerAngles(0,1,2) is (0, 0, theta), so its norm is |theta|.
extern "C" int cfear_keyframe_based_fuse(const double diff_xyt[3], int32_t use_keyframe, double min_keyframe_dist,
                                         double min_keyframe_rot_deg) {
  if (!use_keyframe) return 1;
  const double tn = std::sqrt(diff_xyt[0] * diff_xyt[0] + diff_xyt[1] * diff_xyt[1]);
  const double rot = std::fabs(diff_xyt[2]);
  return (tn > min_keyframe_dist || rot > (min_keyframe_rot_deg * M_PI / 180.0)) ? 1 : 0;
}
// AccelerationVelocitySanityCheck (odometrykeyframefuser.cpp:76-94): 4 Hz, 200 m/s and 200 m/s^2, strict comparisons,
// acceleration tested first; only the translations of the two motions enter.  1 = sane, 0 = use the guess (:198-199).
extern "C" int cfear_acc_vel_sanity_check(const double tmot_prev_xy[2], const double tmot_curr_xy[2]) {
  const double dt = 0.25, vel_limit = 200, acc_limit = 200;
  const double vel = std::sqrt((tmot_curr_xy[0] / dt) * (tmot_curr_xy[0] / dt) + (tmot_curr_xy[1] / dt) * (tmot_curr_xy[1] / dt));
  const double ax = (tmot_curr_xy[0] - tmot_prev_xy[0]) / (dt * dt), ay = (tmot_curr_xy[1] - tmot_prev_xy[1]) / (dt * dt);
  const double acc = std::sqrt(ax * ax + ay * ay);
  if (acc > acc_limit) return 0;
  else if (vel > vel_limit) return 0;
  return 1;
}

namespace {

struct Keyframe { int slab; Aff2 pose; uint64_t idx = 0; };
struct Stream {
  Aff2 T_prev = aff_identity(), Tmot = aff_identity(), Tcurrent = aff_identity();
  std::vector<Keyframe> keyframes;
  std::vector<int> free_slabs;
  int cur_slab = -1;
  Aff2 Tguess = aff_identity();
  int job = -1;              // index into this call's registration batch, -1 = first frame
  uint64_t n_keyframes = 0;  // RadarScan::counter of this stream
  bool has_constraint = false;          // the last frame added a keyframe behind another one (AddToGraph)
  uint64_t c_from = 0, c_to = 0;
  Aff2 c_Tdiff = aff_identity();
  double c_cov[36] = {0};
};

}  // namespace

// Fused decode of [range bins][azimuths] sweeps: candidates per azimuth beyond which the rotation kernel + row sweep are the
// quicker route (the lists cost ~17 ns of a 2048-sweep pass per candidate and azimuth: equal at ~84, tools/decode_bench.py
// [--dense]), and for how many frames that verdict stands before one frame measures again.
constexpr double kDecodeDense = 80.0;
constexpr int kDecodeHold = 32;

struct cfear_odometry {
  cfear_ctx* ctx = nullptr;
  int n_streams = 0;
  cfear_polar_desc desc{};             // layout the filters see: rows = azimuths
  cfear_polar_desc in_desc{};          // layout of the caller's images (differs when par.rotate_ccw)
  cfear_odometry_params par{};
  int cap_points = 0, cell_cap = 0, slabs_per_stream = 0;
  size_t slab_bytes = 0;
  // device memory (one allocation each)
  uint8_t* d_polar = nullptr;          // staging when the caller passes host images
  uint8_t* d_rot = nullptr;            // rotated images (par.rotate_ccw)
  bool fused_decode = false;           // par.rotate_ccw: the filter stage decodes the source itself where it pays (no d_rot)
  // ... which it does on radar-like sweeps (a few dozen bins >= z_min per azimuth): the candidate lists cost time per
  // candidate, the rotation kernel + row sweep do not.  The decode reports the batch's candidates; beyond kDecodeDense per
  // azimuth the next kDecodeHold frames take the two-kernel route, then one frame probes again.
  uint32_t* d_decode_stats = nullptr;  // [2 buffers][64]
  uint32_t* h_decode_stats = nullptr;  // pinned copy
  bool decode_measured[2] = {false, false};
  int decode_hold = 0;
  char* d_sel = nullptr;               // sel_range | sel_intensity | sel_count
  float* d_xyzi2[2] = {nullptr, nullptr};    // filter outputs are double-buffered: the next frame's filter
  int32_t* d_npts2[2] = {nullptr, nullptr};  //   may run while the host applies this frame's policy
  float* d_xyzi = nullptr;                   // buffer of the frame being processed
  int32_t* d_npts = nullptr;
  // fused filter output (k-strongest without keep_nodes, k <= 64): per-row points + counts written by the polar sweep
  // itself; the surface-point kernel compacts them, so neither sel_* arrays nor a cloud kernel exist in this mode
  bool fused = false;
  // the same hand-over for CA-CFAR (no keep_nodes): cacfar_rows_kernel leaves per-row keys with a row capacity row_k
  bool fused_cfar = false;
  int row_k = 0;                                 // keys per row in d_rowpts2: k (k-strongest) or the CA-CFAR row capacity
  uint32_t* d_rowpts2[2] = {nullptr, nullptr};   // [B][rows][k] packed keys (intensity << 24 | range bin)
  int32_t* d_rowcnt2[2] = {nullptr, nullptr};   // [B][rows][2]
  int64_t* d_offsets2[2] = {nullptr, nullptr};  // [B] image offsets of the sweep in each filter buffer (process_offsets)
  int64_t* h_offsets = nullptr;              // pinned staging, [2][B]
  std::vector<int64_t> prefetched_offsets;   // offsets the prefetched filter output was computed from
  // par.keep_nodes: the peaks cloud of every frame (cloud_peaks_ of RadarScan), double-buffered like the cloud
  float* d_pk2[2] = {nullptr, nullptr};
  int32_t* d_npk2[2] = {nullptr, nullptr};
  float* d_pk = nullptr;
  int32_t* d_npk = nullptr;
  double* d_mot = nullptr;                   // [B][3] TprevMot of this frame (peaks compensation)
  double* h_mot = nullptr;
  int32_t* h_npk = nullptr;
  std::vector<int> last_slab;                // slab that holds each stream's last processed scan
  int cur_buf = 0;
  const uint8_t* prefetched = nullptr;       // polar pointer whose filter output sits in buffer cur_buf ^ 1
  hipEvent_t ev_results = nullptr;
  hipStream_t copy_stream = nullptr;         // uploads the registration jobs while the surface kernel runs
  hipEvent_t ev_jobs = nullptr;
  char* d_slabs = nullptr;
  char* d_surf_jobs = nullptr;
  char* d_reg_jobs = nullptr;
  // what a frame returns to the host lives in ONE device block (and one pinned block): results | status | n_cells |
  // n_points -- one read-back per frame instead of four small copies queued between the matcher and the next sweep
  char* d_out = nullptr;
  char* h_out = nullptr;
  size_t out_bytes = 0;
  int32_t* d_npts_out = nullptr;       // n_points slot of the block (rows mode: written by surface_prep_kernel)
  cfear_reg_result* d_results = nullptr;
  int32_t* d_status = nullptr;
  int32_t* d_ncells = nullptr;         // gathered n_cells of the current scans
  char* d_surf_scratch = nullptr;
  char* d_reg_scratch = nullptr;
  // pinned host mirrors
  char* h_surf_jobs = nullptr;
  char* h_reg_jobs = nullptr;
  cfear_reg_result* h_results = nullptr;
  int32_t* h_status = nullptr;
  int32_t* h_npts = nullptr;
  int32_t* h_ncells = nullptr;
  // covariance by cost sampling (optional): n^3 GetCost records per stream and the cached fit
  cfear_reg_result* d_samples = nullptr;
  cfear_reg_result* h_samples = nullptr;
  CovFit fit;
  std::vector<double> cov;             // [n_streams][36] cov_current
  std::vector<int32_t> cov_sampled;    // [n_streams]
  std::vector<Stream> streams;
  int big_regs = 0;                    // > 0: recent frames held registrations too large for 80 KB of LDS -> keep the second launch on
  std::vector<double> cost_est, cost_tmp;   // per stream: work of its last registration (residuals x iterations): orders the next batch
  std::vector<int> job_slot;                 // per stream: class | rank inside the class << 3 of this frame's registration (-1: none)
  std::vector<ScanView> views;         // [n_streams * slabs_per_stream]
};

namespace {

template <typename T>
bool dalloc(T** p, size_t bytes) {
  if (hipMalloc((void**)p, bytes ? bytes : 256) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return false; }
  return true;
}
template <typename T>
bool halloc(T** p, size_t bytes) {
  if (hipHostMalloc((void**)p, bytes ? bytes : 256, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return false; }
  return true;
}

}  // namespace

extern "C" int cfear_odometry_destroy(cfear_odometry* od) {
  if (!od) return CFEAR_OK;
  (void)hipSetDevice(od->ctx->device);
  (void)hipStreamSynchronize(od->ctx->stream);
  if (od->ev_results) (void)hipEventDestroy(od->ev_results);
  if (od->ev_jobs) (void)hipEventDestroy(od->ev_jobs);
  if (od->copy_stream) (void)hipStreamDestroy(od->copy_stream);
  void* dev2[] = {od->d_rowpts2[0], od->d_rowpts2[1], od->d_rowcnt2[0], od->d_rowcnt2[1], od->d_offsets2[0], od->d_offsets2[1]};
  for (void* p : dev2) if (p) (void)hipFree(p);
  if (od->h_offsets) (void)hipHostFree(od->h_offsets);
  void* dev[] = {od->d_decode_stats, od->d_pk2[0], od->d_pk2[1], od->d_npk2[0], od->d_npk2[1], od->d_mot, od->d_polar, od->d_rot, od->d_sel, od->d_xyzi2[0], od->d_xyzi2[1], od->d_npts2[0], od->d_npts2[1], od->d_slabs, od->d_surf_jobs, od->d_reg_jobs,
                 od->d_out, od->d_surf_scratch, od->d_reg_scratch, od->d_samples};
  for (void* p : dev) if (p) (void)hipFree(p);
  void* host[] = {od->h_mot, od->h_npk, od->h_surf_jobs, od->h_reg_jobs, od->h_out, od->h_samples, od->h_decode_stats};
  for (void* p : host) if (p) (void)hipHostFree(p);
  delete od;
  return CFEAR_OK;
}

extern "C" int cfear_odometry_create(cfear_ctx* ctx, int32_t n_streams, const cfear_polar_desc* desc,
                                     const cfear_odometry_params* par, cfear_odometry** out) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!desc || !par || !out || n_streams < 1) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  if (desc->batch != n_streams) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "desc.batch must equal n_streams");
  if (par->submap_scan_size < 1 || par->submap_scan_size + 1 > cfear_reg_max_scans())
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "submap_scan_size must be in [1,%d]", cfear_reg_max_scans() - 1);
  if (!(par->res > 0.05f)) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "res must be > 0.05");   // odometrykeyframefuser.cpp:25
  if (par->estimate_cov_by_sampling && (par->cov_sampling.samples_per_axis < 1 || par->cov_sampling.samples_per_axis > 15))
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "cov_sampling.samples_per_axis must be in [1,15]");
  {
    // the same argument rules as the standalone filter entry points (cfear_filter_kstrongest / cfear_filter_cacfar)
    const int fcols = par->rotate_ccw ? desc->rows : desc->cols;
    if (desc->rows <= 0 || desc->cols <= 0 || desc->stride < desc->cols ||
        (n_streams > 1 && desc->batch_stride < (int64_t)desc->rows * desc->stride))
      return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "bad polar descriptor");
    if (fcols > 8192) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "more than 8192 range bins unsupported");
    if (par->filter_type == CFEAR_FILTER_CACFAR) {
      if (par->cacfar.window_size < 1 || par->cacfar.nb_guard_cells < 0 || !(par->cacfar.range_res > 0.f))
        return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "bad CFAR parameters");
    } else if (par->filter_type == CFEAR_FILTER_KSTRONG) {
      if (par->kstrong.k_strongest < 1 || par->kstrong.k_strongest > 1024)
        return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "k_strongest must be in [1,1024]");
      if (!(par->kstrong.range_res > 0.f) || !(par->kstrong.min_distance >= 0.f))
        return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "range_res must be > 0 and min_distance >= 0");
    } else {
      return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "unknown filter_type %d", par->filter_type);
    }
  }
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  cfear_odometry* od = new cfear_odometry();
  od->ctx = ctx; od->n_streams = n_streams; od->desc = *desc; od->in_desc = *desc; od->par = *par;
  if (par->rotate_ccw) {               // radarDriver::Callback (radar_driver.cpp:74-90): [bins][azimuths] -> rows = azimuths
    od->desc.rows = desc->cols;
    od->desc.cols = desc->rows;
    od->desc.stride = (desc->rows + 127) / 128 * 128;    // rows on 128-byte lines: the rotation then writes whole lines
    od->desc.batch_stride = (int64_t)od->desc.rows * od->desc.stride;
  }
  const int B = n_streams, rows = od->desc.rows, k = par->kstrong.k_strongest;
  // CA-CFAR puts no bound on the detections of a row (cfar.cpp:35-71): room for 65 536 points per sweep (the surface
  // kernels take clouds beyond 16 384 points through their global-memory path); k-strongest keeps at most rows * k
  od->cap_points = par->filter_type == CFEAR_FILTER_CACFAR ? std::min(rows * od->desc.cols, 65536) : rows * k;
  od->cell_cap = 2048;
  od->slabs_per_stream = par->submap_scan_size + 2;
  od->slab_bytes = cfear_scan_slab_bytes(od->cell_cap);
  const size_t nsel = (size_t)B * rows * std::max(k, 1);
  bool ok = true;
  od->fused = par->filter_type == CFEAR_FILTER_KSTRONG && !par->keep_nodes && k <= 64 && rows <= 4096 &&
              rows * k <= cfear_surface_max_points();
  od->row_k = k;
  // [range bins][azimuths] sources: decode fused into the sweep (CFEAR_OPT_FUSED_DECODE = 0 keeps the two-kernel route, for
  // A/B runs and the tests that compare the two)
  od->fused_decode = od->fused && par->rotate_ccw && ctx->opt[CFEAR_OPT_FUSED_DECODE] != 0;
  if (od->fused_decode) {
    ok = ok && dalloc(&od->d_decode_stats, 2 * 64 * 4);
    ok = ok && halloc(&od->h_decode_stats, 2 * 64 * 4);
  }
  // CA-CFAR puts no bound on a row's detections either: 1024 keys per row (a row beyond that marks its scan
  // CFEAR_ERR_CAPACITY, like a sweep beyond cap_points)
  od->fused_cfar = par->filter_type == CFEAR_FILTER_CACFAR && !par->keep_nodes && rows <= 4096;
  if (od->fused_cfar) {
    od->row_k = std::min((od->desc.cols + 3) / 4 * 4, 1024);
    for (int i = 0; i < 2; i++) {
      ok = ok && dalloc(&od->d_rowpts2[i], (size_t)B * rows * od->row_k * 4);
      ok = ok && dalloc(&od->d_rowcnt2[i], (size_t)B * rows * 8);
    }
  }
  if (od->fused) {
    for (int i = 0; i < 2; i++) {
      ok = ok && dalloc(&od->d_rowpts2[i], nsel * 4);
      ok = ok && dalloc(&od->d_rowcnt2[i], (size_t)B * rows * 8);
      ok = ok && dalloc(&od->d_offsets2[i], (size_t)B * 8);
    }
    ok = ok && halloc(&od->h_offsets, (size_t)B * 8 * 2);
  } else {
    ok = ok && dalloc(&od->d_sel, nsel * 4 + 2 * (nsel + 256) + (size_t)B * rows * 4 + 1024);   // + is_peak (keep_nodes)
  }
  if (par->keep_nodes) {
    for (int i = 0; i < 2; i++) {
      ok = ok && dalloc(&od->d_pk2[i], (size_t)B * od->cap_points * 16);
      ok = ok && dalloc(&od->d_npk2[i], (size_t)B * 4);
    }
    ok = ok && dalloc(&od->d_mot, (size_t)B * 3 * sizeof(double));
    ok = ok && halloc(&od->h_mot, (size_t)B * 3 * sizeof(double));
    ok = ok && halloc(&od->h_npk, (size_t)B * 4);
  }
  od->last_slab.assign(B, -1);
  for (int i = 0; i < 2; i++) {
    ok = ok && dalloc(&od->d_xyzi2[i], (size_t)B * od->cap_points * 16);
    ok = ok && dalloc(&od->d_npts2[i], (size_t)B * 4);
  }
  ok = ok && hipEventCreateWithFlags(&od->ev_results, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&od->ev_jobs, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&od->copy_stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && dalloc(&od->d_slabs, (size_t)B * od->slabs_per_stream * od->slab_bytes);
  ok = ok && dalloc(&od->d_surf_jobs, (size_t)B * cfear_surface_job_bytes());
  ok = ok && dalloc(&od->d_reg_jobs, (size_t)B * cfear_reg_job_bytes());
  od->out_bytes = (size_t)B * (sizeof(cfear_reg_result) + 12);
  ok = ok && dalloc(&od->d_out, od->out_bytes);
  ok = ok && halloc(&od->h_out, od->out_bytes);
  if (ok) {
    auto carve = [&](char* base) {
      struct P { cfear_reg_result* r; int32_t *st, *nc, *np; } q;
      q.r = (cfear_reg_result*)base;
      q.st = (int32_t*)(base + (size_t)B * sizeof(cfear_reg_result));
      q.nc = q.st + B; q.np = q.nc + B;
      return q;
    };
    const auto d = carve(od->d_out), h = carve(od->h_out);
    od->d_results = d.r; od->d_status = d.st; od->d_ncells = d.nc; od->d_npts_out = d.np;
    od->h_results = h.r; od->h_status = h.st; od->h_ncells = h.nc; od->h_npts = h.np;
  }
  ok = ok && dalloc(&od->d_surf_scratch, (size_t)B * cfear_surface_scratch_bytes(od->cap_points));
  ok = ok && dalloc(&od->d_reg_scratch, (size_t)B * cfear_register_scratch_bytes(par->submap_scan_size * od->cell_cap));
  ok = ok && halloc(&od->h_surf_jobs, (size_t)B * cfear_surface_job_bytes());
  ok = ok && halloc(&od->h_reg_jobs, (size_t)B * cfear_reg_job_bytes());
  if (par->estimate_cov_by_sampling) {
    od->fit.prepare(par->cov_sampling.samples_per_axis, par->cov_sampling.xy_range * 0.5, par->cov_sampling.yaw_range * 0.5);
    ok = ok && dalloc(&od->d_samples, (size_t)B * od->fit.m * sizeof(cfear_reg_result));
    ok = ok && halloc(&od->h_samples, (size_t)B * od->fit.m * sizeof(cfear_reg_result));
  }
  od->cov.assign((size_t)B * 36, 0.0);
  for (int b = 0; b < B; b++)
    for (int k = 0; k < 6; k++) od->cov[(size_t)b * 36 + k * 7] = 1.0;       // cov_current = Identity (:36)
  od->cov_sampled.assign(B, 0);
  if (!ok) { cfear_odometry_destroy(od); return cfear_set_error(ctx, CFEAR_ERR_HIP, "odometry buffers: allocation failed"); }
  od->streams.resize(B);
  od->cost_est.assign(B, 0.0);
  od->views.resize((size_t)B * od->slabs_per_stream);
  for (int b = 0; b < B; b++) {
    for (int s = 0; s < od->slabs_per_stream; s++) {
      od->views[(size_t)b * od->slabs_per_stream + s] =
          cfear_scan_view(od->d_slabs + ((size_t)b * od->slabs_per_stream + s) * od->slab_bytes, od->cell_cap);
      od->streams[b].free_slabs.push_back(od->slabs_per_stream - 1 - s);
    }
  }
  *out = od;
  return CFEAR_OK;
}

// ---- F: filter (radar_driver.cpp:48-73) of one batch of polar images into filter buffer `buf` -------
static int run_filter(cfear_odometry* od, const uint8_t* polar, int buf, const int64_t* offsets = nullptr) {
  cfear_ctx* ctx = od->ctx;
  const int B = od->n_streams, rows = od->desc.rows, k = od->par.kstrong.k_strongest;
  const cfear_odometry_params& par = od->par;
  const uint8_t* d_polar = polar;
  cfear_polar_desc dd = od->in_desc;
  const int64_t* d_offsets = nullptr;
  if (offsets) {
    if (!od->fused || par.rotate_ccw || !cfear_is_device_ptr(polar))
      return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "image offsets need device images, the k-strongest filter "
                             "(k <= 64, no keep_nodes) and rows = azimuths");
    int64_t* h = od->h_offsets + (size_t)buf * B;
    memcpy(h, offsets, (size_t)B * 8);
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(od->d_offsets2[buf], h, (size_t)B * 8, hipMemcpyHostToDevice, ctx->stream));
    d_offsets = od->d_offsets2[buf];
  }
  if (!cfear_is_device_ptr(polar)) {
    const size_t img_bytes = (size_t)od->in_desc.rows * od->in_desc.stride;
    if (!od->d_polar && !dalloc(&od->d_polar, img_bytes * B)) return cfear_set_error(ctx, CFEAR_ERR_HIP, "staging allocation failed");
    const int64_t bs = B > 1 ? od->in_desc.batch_stride : (int64_t)img_bytes;
    if (bs == (int64_t)img_bytes) {          // a dense batch crosses PCIe as one copy
      CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(od->d_polar, polar, img_bytes * B, hipMemcpyHostToDevice, ctx->stream));
    } else {
      for (int b = 0; b < B; b++)
        CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(od->d_polar + (size_t)b * img_bytes, polar + (size_t)b * bs, img_bytes,
                                            hipMemcpyHostToDevice, ctx->stream));
    }
    d_polar = od->d_polar;
    dd.batch_stride = (int64_t)img_bytes;
  }
  od->decode_measured[buf] = false;
  if (od->fused_decode && od->decode_hold > 0) od->decode_hold--;
  else if (od->fused_decode && cfear_kstrong_cols_preferred(d_polar, &dd, &par.kstrong)) {
    cfear_kstrong_params kp = par.kstrong;
    kp.want_peaks = 0;
    cfear_kstrong_fused fz;
    fz.row_keys = od->d_rowpts2[buf];
    fz.row_valid = od->d_rowcnt2[buf];
    fz.cand_stats = od->d_decode_stats + 64 * buf;
    const int rc = cfear_kstrong_cols_device(ctx, d_polar, &dd, &kp, &fz);
    if (rc != CFEAR_OK) return rc;
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(od->h_decode_stats + 64 * buf, fz.cand_stats, 64 * 4, hipMemcpyDeviceToHost, ctx->stream));
    od->decode_measured[buf] = true;                         // read once this buffer's frame has been synchronised
    return CFEAR_OK;
  }
  // CA-CFAR on [range bins][azimuths] sweeps: the decode fused into the filter (cacfar_cols_kernel), one pass over the image
  // instead of three (CFEAR_OPT_FUSED_DECODE = 0 keeps rotate + cacfar_rows, for A/B runs and the tests that compare the two)
  if (par.rotate_ccw && od->fused_cfar && ctx->opt[CFEAR_OPT_FUSED_DECODE] != 0 && B >= 16 &&
      cfear_cacfar_cols_supported(d_polar, &dd, &par.cacfar)) {
    cfear_cacfar_params cp = par.cacfar;
    cfear_cacfar_fused fz;
    fz.row_keys = od->d_rowpts2[buf]; fz.row_cnt = od->d_rowcnt2[buf]; fz.kcap = od->row_k; fz.bins_major = true;
    return cfear_cacfar_device(ctx, d_polar, &dd, &cp, nullptr, nullptr, od->cap_points, nullptr, &fz);
  }
  if (par.rotate_ccw) {
    const size_t rot_bytes = (size_t)od->desc.rows * od->desc.stride;
    if (!od->d_rot && !dalloc(&od->d_rot, rot_bytes * B)) return cfear_set_error(ctx, CFEAR_ERR_HIP, "rotation buffer allocation failed");
    const int rc = cfear_rotate_ccw_device(ctx, d_polar, &dd, od->d_rot, od->desc.stride, (int64_t)rot_bytes);
    if (rc != CFEAR_OK) return rc;
    d_polar = od->d_rot;
    dd = od->desc;
  }
  if (par.filter_type == CFEAR_FILTER_CACFAR) {
    cfear_cacfar_params cp = par.cacfar;
    if (par.keep_nodes)    // CA-CFAR produces no peaks cloud (radar_driver.cpp:52-56)
      CFEAR_HIP_CHECK(ctx, hipMemsetAsync(od->d_npk2[buf], 0, (size_t)B * 4, ctx->stream));
    if (od->fused_cfar) {
      cfear_cacfar_fused fz;
      fz.row_keys = od->d_rowpts2[buf]; fz.row_cnt = od->d_rowcnt2[buf]; fz.kcap = od->row_k;
      return cfear_cacfar_device(ctx, d_polar, &dd, &cp, nullptr, nullptr, od->cap_points, nullptr, &fz);
    }
    return cfear_cacfar_device(ctx, d_polar, &dd, &cp, od->d_xyzi2[buf], od->d_npts2[buf], od->cap_points, nullptr);
  }
  const size_t nsel = (size_t)B * rows * k;
  cfear_kstrong_out o{};
  if (od->fused) {
    cfear_kstrong_params kp = par.kstrong;
    kp.want_peaks = 0;
    cfear_kstrong_fused fz;
    fz.row_keys = od->d_rowpts2[buf];
    fz.row_valid = od->d_rowcnt2[buf];
    fz.image_offsets = d_offsets;
    return cfear_kstrong_device(ctx, d_polar, &dd, &kp, &o, par.rotate_ccw != 0, &fz);
  }
  o.sel_range = (int32_t*)od->d_sel;
  o.sel_intensity = (uint8_t*)(od->d_sel + nsel * 4);
  o.sel_count = (int32_t*)(od->d_sel + nsel * 4 + (nsel + 255) / 256 * 256);
  o.xyzi = od->d_xyzi2[buf];
  o.n_points = od->d_npts2[buf];
  cfear_kstrong_params kp = par.kstrong;
  kp.want_peaks = 0;     // the peaks cloud feeds CorAl / Scan Context, not the matcher
  if (par.keep_nodes) {  // ... unless the caller builds graph nodes from this pipeline (RadarScan::cloud_peaks_)
    kp.want_peaks = 1;
    o.is_peak = (uint8_t*)(od->d_sel + nsel * 4 + (nsel + 255) / 256 * 256 + (size_t)B * rows * 4 + 256);
    o.xyzi_peaks = od->d_pk2[buf];
    o.n_peaks = od->d_npk2[buf];
  }
  if (od->cap_points != rows * k)
    return cfear_set_error(ctx, CFEAR_ERR_CAPACITY, "rows*k = %d exceeds %d points per scan", rows * k, od->cap_points);
  // the rotated buffer is a padded copy of what the reference holds as a dense cv::Mat: AxialNonMaxSupress's reads
  // past a row end must land on the next row's first bins (radar_filters.cpp:238-298), not on the padding
  return cfear_kstrong_device(ctx, d_polar, &dd, &kp, &o, par.rotate_ccw != 0);
}

extern "C" int cfear_odometry_get_covariance(cfear_odometry* od, double* cov, int32_t* sampled) {
  if (!od || !cov) return CFEAR_ERR_INVALID_ARGUMENT;
  memcpy(cov, od->cov.data(), od->cov.size() * sizeof(double));
  if (sampled) memcpy(sampled, od->cov_sampled.data(), od->cov_sampled.size() * sizeof(int32_t));
  return CFEAR_OK;
}

extern "C" int cfear_odometry_process(cfear_odometry* od, const uint8_t* polar, cfear_frame_info* info) {
  return cfear_odometry_process_prefetch(od, polar, nullptr, info);
}

static int process_frame(cfear_odometry* od, const uint8_t* polar, const uint8_t* polar_next, const cfear_sc_cloud* clouds,
                         const cfear_sc_cloud* peaks, cfear_frame_info* info, const int64_t* offsets = nullptr,
                         const int64_t* offsets_next = nullptr);

extern "C" int cfear_odometry_process_offsets(cfear_odometry* od, const uint8_t* base, const int64_t* offsets,
                                              const int64_t* offsets_next, cfear_frame_info* info) {
  if (!od || !base || !offsets || !info) return CFEAR_ERR_INVALID_ARGUMENT;
  return process_frame(od, base, offsets_next ? base : nullptr, nullptr, nullptr, info, offsets, offsets_next);
}

extern "C" int cfear_odometry_discard_prefetch(cfear_odometry* od) {
  if (!od) return CFEAR_ERR_INVALID_ARGUMENT;
  od->prefetched = nullptr;
  od->prefetched_offsets.clear();
  return CFEAR_OK;
}

extern "C" int cfear_odometry_process_prefetch(cfear_odometry* od, const uint8_t* polar, const uint8_t* polar_next,
                                               cfear_frame_info* info) {
  if (!od || !polar || !info) return CFEAR_ERR_INVALID_ARGUMENT;
  return process_frame(od, polar, polar_next, nullptr, nullptr, info);
}

// OdometryKeyframeFuser::pointcloudCallback(cloud, cloud_peaks, ...) (odometrykeyframefuser.cpp:413-426): the caller's
// own driver has already filtered the sweep; the clouds enter the pipeline where the filter would have left them.
extern "C" int cfear_odometry_process_clouds(cfear_odometry* od, const cfear_sc_cloud* clouds, const cfear_sc_cloud* peaks,
                                             cfear_frame_info* info) {
  if (!od || !clouds || !info) return CFEAR_ERR_INVALID_ARGUMENT;
  return process_frame(od, nullptr, nullptr, clouds, peaks, info);
}

// uploads one cloud per stream into the frame buffers `buf` (what run_filter would have produced)
static int load_clouds(cfear_odometry* od, const cfear_sc_cloud* clouds, float* d_dst, int32_t* d_n, int32_t* h_tmp) {
  cfear_ctx* ctx = od->ctx;
  const int B = od->n_streams;
  for (int b = 0; b < B; b++) {
    const int n = clouds ? clouds[b].n : 0;
    if (n < 0 || (n > 0 && !clouds[b].xyzi)) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "stream %d: null cloud", b);
    if (n > od->cap_points) return cfear_set_error(ctx, CFEAR_ERR_CAPACITY, "stream %d: %d points > %d", b, n, od->cap_points);
    h_tmp[b] = n;
    if (n > 0) {
      const bool dev = cfear_is_device_ptr(clouds[b].xyzi);
      CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(d_dst + (size_t)b * od->cap_points * 4, clouds[b].xyzi, (size_t)n * 16,
                                          dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    }
  }
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(d_n, h_tmp, (size_t)B * 4, hipMemcpyHostToDevice, ctx->stream));
  // h_tmp is reused by the next load / the read-back of this frame: make sure the copy has left the host buffer
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CFEAR_OK;
}

namespace {
struct HostTimeline {          // CFEAR_OPT_HOST_TIMELINE = 1: where the host spends a frame (printed every 256 calls)
  bool on = false;
  double acc[8] = {0};
  int calls = 0;
  std::chrono::steady_clock::time_point last, exit_t;
  bool have_exit = false;
  void start() {
    if (!on) return;
    last = std::chrono::steady_clock::now();
    if (have_exit) acc[6] += std::chrono::duration<double, std::micro>(last - exit_t).count();
  }
  void mark(int k) {
    if (!on) return;
    const auto t = std::chrono::steady_clock::now();
    acc[k] += std::chrono::duration<double, std::micro>(t - last).count();
    last = t;
  }
  void end() {
    if (!on) return;
    exit_t = std::chrono::steady_clock::now(); have_exit = true;
    if (++calls % 256 == 0) {
      fprintf(stderr, "host us/frame: surf jobs+launch %.0f | reg jobs %.0f | enqueue rest %.0f | wait results %.0f | policy %.0f | outside the call %.0f\n",
              acc[0] / 256, acc[1] / 256, acc[2] / 256, acc[3] / 256, acc[4] / 256, acc[6] / 256);
      for (double& a : acc) a = 0;
    }
  }
};
HostTimeline g_tl;
}  // namespace

static int process_frame(cfear_odometry* od, const uint8_t* polar, const uint8_t* polar_next, const cfear_sc_cloud* clouds,
                         const cfear_sc_cloud* peaks, cfear_frame_info* info, const int64_t* offsets, const int64_t* offsets_next) {
  cfear_ctx* ctx = od->ctx;
  g_tl.on = ctx->opt[CFEAR_OPT_HOST_TIMELINE] != 0;
  g_tl.start();
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int B = od->n_streams;
  const cfear_odometry_params& par = od->par;
  int rc;
  if (clouds) {                                    // filtered clouds from the caller: no filter, no prefetch
    od->prefetched = nullptr;
    rc = load_clouds(od, clouds, od->d_xyzi2[od->cur_buf], od->d_npts2[od->cur_buf], od->h_npts);
    if (rc != CFEAR_OK) return rc;
    if (par.keep_nodes) {
      rc = load_clouds(od, peaks, od->d_pk2[od->cur_buf], od->d_npk2[od->cur_buf], od->h_npk);
      if (rc != CFEAR_OK) return rc;
    }
  } else if (od->prefetched == polar &&
             (offsets ? (od->prefetched_offsets.size() == (size_t)B &&
                         memcmp(od->prefetched_offsets.data(), offsets, (size_t)B * 8) == 0)
                      : od->prefetched_offsets.empty())) {   // this frame's filter already ran (or is running)
    od->cur_buf ^= 1;
  } else {
    rc = run_filter(od, polar, od->cur_buf, offsets);
    if (rc != CFEAR_OK) return rc;
  }
  od->prefetched = nullptr;
  od->prefetched_offsets.clear();
  od->d_xyzi = od->d_xyzi2[od->cur_buf];
  od->d_npts = od->d_npts2[od->cur_buf];
  od->d_pk = od->d_pk2[od->cur_buf];
  od->d_npk = od->d_npk2[od->cur_buf];
  const bool rows_mode = (od->fused || od->fused_cfar) && !clouds;     // the filter left per-row points: the surface kernel compacts them
  // ---- C + N: compensate with the previous motion, surface points (odometrykeyframefuser.cpp:146-161)
  const size_t sjb = cfear_surface_job_bytes();
  for (int b = 0; b < B; b++)
    if (od->streams[b].free_slabs.empty()) return cfear_set_error(ctx, CFEAR_ERR_CAPACITY, "stream %d: no free scan slab", b);
  for (int b = 0; b < B; b++) {
    Stream& st = od->streams[b];
    st.cur_slab = st.free_slabs.back();
    st.free_slabs.pop_back();
    double mot[3];
    aff_to_xyt(st.Tmot, mot);                                     // Compensate(cloud, TprevMot, ccw)
    od->last_slab[b] = st.cur_slab;
    if (par.keep_nodes) { od->h_mot[3 * b] = mot[0]; od->h_mot[3 * b + 1] = mot[1]; od->h_mot[3 * b + 2] = mot[2]; }
    if (rows_mode)
      cfear_surface_fill_job_rows(od->h_surf_jobs + (size_t)b * sjb, od->d_xyzi + (size_t)b * od->cap_points * 4, od->d_npts_out + b,
                                  od->d_rowpts2[od->cur_buf] + (size_t)b * od->desc.rows * od->row_k,
                                  od->d_rowcnt2[od->cur_buf] + (size_t)b * od->desc.rows * 2, od->desc.rows,
                                  od->row_k, par.compensate, mot,
                                  od->views[(size_t)b * od->slabs_per_stream + st.cur_slab]);
    else
      cfear_surface_fill_job(od->h_surf_jobs + (size_t)b * sjb, od->d_xyzi + (size_t)b * od->cap_points * 4,
                             od->d_npts + b, 0, par.compensate, mot, od->views[(size_t)b * od->slabs_per_stream + st.cur_slab]);
  }
  // From here on every stream holds a slab: an error exit must hand the slabs back, or submap_scan_size + 2 failed
  // calls would drain a stream's free list.
  auto fail = [&](int code) {
    for (int b = 0; b < B; b++) {
      Stream& st = od->streams[b];
      if (st.cur_slab >= 0) { st.free_slabs.push_back(st.cur_slab); st.cur_slab = -1; }
    }
    return code;
  };
#define OD_CHECK(expr)                                                                                        \
  do {                                                                                                        \
    hipError_t _e = (expr);                                                                                   \
    if (_e != hipSuccess)                                                                                     \
      return fail(cfear_set_error(ctx, CFEAR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__)); \
  } while (0)
  OD_CHECK(hipMemcpyAsync(od->d_surf_jobs, od->h_surf_jobs, (size_t)B * sjb, hipMemcpyHostToDevice, ctx->stream));
  cfear_feature_params fp{};
  fp.radius = par.res;
  fp.downsample_factor = par.downsample_factor;
  fp.origin[0] = fp.origin[1] = 0.0;                              // Eigen::Vector2d(0,0), :161
  fp.weight_intensity = par.weight_intensity;
  fp.ccw = par.radar_ccw;
  cfear_surface_polar sp;
  if (rows_mode) {
    double *d_cos = nullptr, *d_sin = nullptr;
    rc = cfear_trig_tables(ctx, od->desc.rows, &d_cos, &d_sin);
    if (rc != CFEAR_OK) return fail(rc);
    sp.cos_t = d_cos; sp.sin_t = d_sin;
    sp.rows = od->desc.rows; sp.k = od->row_k;
    if (od->fused_cfar) {                 // cfar.cpp:43: range = range_resolution_ * double(range_bin)
      sp.range_res = (double)par.cacfar.range_res; sp.range_off = 0.0;
    } else {                              // radar_filters.cpp:324: range_res / 2 + range_res * bin
      sp.range_res = (double)par.kstrong.range_res; sp.range_off = sp.range_res / 2.0;
    }
  }
  rc = cfear_surface_launch(ctx, od->d_surf_jobs, B, &fp, od->d_surf_scratch, od->d_status, od->d_ncells, od->cell_cap, od->cap_points, rows_mode ? &sp : nullptr);
  if (rc != CFEAR_OK) return fail(rc);
  g_tl.mark(0);
  // ---- M: the registration jobs depend on host state only: they are built and uploaded (copy stream) while the
  //      surface kernels, already enqueued above, run -- 2.3 KB per stream, ~0.3 ms of host time for 2048 streams
  //      that would otherwise sit between the sweep and the surface kernels (:164-186) ------------------------
  const size_t rjb = cfear_reg_job_stride(par.submap_scan_size + 1);   // records cover the keyframe window + the new scan
  int n_jobs = 0;
  std::vector<ScanView> views(cfear_reg_max_scans());
  std::vector<double> poses(3 * (size_t)cfear_reg_max_scans());
  // Longest first: registrations differ by ~2x in work (cells, outer and LM iterations), and 2048 of them over the 512
  // workgroup slots of the chip end with whatever the last-started ones need.  A stream's previous registration is a
  // good estimate of its next one, so the batch is ordered by it (the job index is only a slot in the batch) -- in
  // four classes by the quartiles of the estimate, streams in index order inside a class: a full sort walked the
  // stream records in random order and cost more host time than the kernel saved.
  std::vector<double>& est = od->cost_est;
  // ... plus a class of its own for the heaviest 3 %: the registrations whose correspondences overflow the LDS arrays of
  // the matcher run ~1.5 x as long as the rest and must not be the last ones to start
  double cut[4] = {0.0, 0.0, 0.0, 0.0};
  {
    std::vector<double>& tmp = od->cost_tmp;
    tmp.assign(est.begin(), est.end());
    if (B >= 8) {
      const size_t ks[4] = {(size_t)B * 97 / 100, (size_t)B * 3 / 4, (size_t)B / 2, (size_t)B / 4};   // descending cuts
      for (int q = 0; q < 4; q++) {
        std::nth_element(tmp.begin(), tmp.begin() + ks[q], tmp.end());
        cut[q] = tmp[ks[q]];
      }
    }
  }
  // one pass for the guess and the class, a prefix over the five classes, one pass that writes every record at its slot
  // (five passes over the stream records -- one per class -- were 0.5 ms of host time per 4096 streams)
  std::vector<int>& slot = od->job_slot;
  slot.assign((size_t)B, -1);
  int cls_n[5] = {0, 0, 0, 0, 0};
  for (int b = 0; b < B; b++) {
    Stream& st = od->streams[b];
    st.Tguess = par.use_guess ? aff_mul(st.T_prev, st.Tmot) : st.T_prev;      // :164-168
    st.job = -1;
    if (st.keyframes.empty()) continue;                                       // :171-177 first frame: no registration
    const double e = est[b];
    const int c = e >= cut[0] ? 0 : e >= cut[1] ? 1 : e >= cut[2] ? 2 : e >= cut[3] ? 3 : 4;
    slot[b] = c | (cls_n[c]++ << 3);                                          // class, rank inside it (stream order)
  }
  int cls_off[5];
  for (int c = 0; c < 5; c++) { cls_off[c] = n_jobs; n_jobs += cls_n[c]; }
  for (int b = 0; b < B; b++) {
    if (slot[b] < 0) continue;
    Stream& st = od->streams[b];
    const int ns = (int)st.keyframes.size() + 1;                              // FormatScans :478-494
    for (int i = 0; i < ns - 1; i++) {
      views[i] = od->views[(size_t)b * od->slabs_per_stream + st.keyframes[i].slab];
      aff_to_xyt(st.keyframes[i].pose, &poses[3 * i]);
    }
    views[ns - 1] = od->views[(size_t)b * od->slabs_per_stream + st.cur_slab];
    aff_to_xyt(st.Tguess, &poses[3 * (ns - 1)]);
    st.job = cls_off[slot[b] & 7] + (slot[b] >> 3);
    cfear_reg_fill_job(od->h_reg_jobs + (size_t)st.job * rjb, views.data(), ns, poses.data());
  }
  if (n_jobs > 0) {
    OD_CHECK(hipMemcpyAsync(od->d_reg_jobs, od->h_reg_jobs, (size_t)n_jobs * rjb, hipMemcpyHostToDevice, od->copy_stream));
    OD_CHECK(hipEventRecord(od->ev_jobs, od->copy_stream));
  }
  g_tl.mark(1);
  // the matcher over this frame's jobs (+ the cost samples around its results); big_pass adds the large forms
  auto launch_register = [&](bool big_pass) -> int {
    RegLaunchHint hint;
    hint.big_pass = big_pass;
    int lrc = cfear_register_launch(ctx, od->d_reg_jobs, n_jobs, &par.reg, par.submap_scan_size * od->cell_cap,
                                    od->d_reg_scratch, od->d_results, nullptr, rjb, hint);
    if (lrc != CFEAR_OK) return lrc;
    if (par.estimate_cov_by_sampling) {
      // approximateCovarianceBySampling (:203-208, 261-316): n^3 GetCost evaluations around the pose the
      // registration just produced; pose and leftover itr_ are read from d_results on the device, so no host
      // round trip separates the two launches.
      RegCostMode mode;
      mode.samples_per_axis = od->fit.n;
      mode.n_samples = od->fit.m;
      mode.xy_half = par.cov_sampling.xy_range * 0.5;
      mode.yaw_half = par.cov_sampling.yaw_range * 0.5;
      mode.blocks_per_job = 1;                                    // one workgroup per stream: its scratch is reused
      mode.prior = od->d_results;
      lrc = cfear_register_launch(ctx, od->d_reg_jobs, n_jobs, &par.reg, par.submap_scan_size * od->cell_cap,
                                  od->d_reg_scratch, od->d_samples, &mode, rjb, hint);
      if (lrc != CFEAR_OK) return lrc;
      if (hipMemcpyAsync(od->h_samples, od->d_samples, (size_t)n_jobs * od->fit.m * sizeof(cfear_reg_result),
                         hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        return cfear_set_error(ctx, CFEAR_ERR_HIP, "sample read-back failed");
    }
    return CFEAR_OK;
  };
  const bool launched_big = od->big_regs > 0;
  if (n_jobs > 0) {
    OD_CHECK(hipStreamWaitEvent(ctx->stream, od->ev_jobs, 0));
    rc = launch_register(launched_big);
    if (rc != CFEAR_OK) return fail(rc);
  }
  if (par.keep_nodes) {
    // Compensate(*cloud_peaks, TprevMot, ccw) (odometrykeyframefuser.cpp:149): off the critical path, behind the matcher
    if (par.compensate) {
      OD_CHECK(hipMemcpyAsync(od->d_mot, od->h_mot, (size_t)B * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      rc = cfear_compensate_batch_device(ctx, od->d_pk, (size_t)od->cap_points, od->d_npk, od->d_mot, B, od->cap_points, par.radar_ccw);
      if (rc != CFEAR_OK) return fail(rc);
    }
    OD_CHECK(hipMemcpyAsync(od->h_npk, od->d_npk, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (!rows_mode)     // the point counts came from the filter / the caller: bring them into the block first
    OD_CHECK(hipMemcpyAsync(od->d_npts_out, od->d_npts, (size_t)B * 4, hipMemcpyDeviceToDevice, ctx->stream));
  OD_CHECK(hipMemcpyAsync(od->h_out, od->d_out, od->out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  OD_CHECK(hipEventRecord(od->ev_results, ctx->stream));
  int prefetch_rc = CFEAR_OK;
  if (polar_next) {
    // (Tried twice: the sweep on its own low-priority stream.  Enqueued ahead of this frame's kernels it simply ran first
    // -- every kernel of the path fills the chip, the sweep holds all vector registers of its SIMDs and the matcher all
    // LDS of its CUs, so nothing co-resides: 2.51 ms per frame batch instead of 2.19.  Gated to start with the matcher,
    // to fill the ~0.2 ms in which the last uneven registrations leave CUs idle, its small workgroups took the LDS the
    // next 80 KB registration needed on every CU that drained: matcher 0.98 -> 1.47 ms, sweep 0.52 -> 1.32 ms, 2.68 ms
    // per frame batch.  Round 4, with 52 KB / 127-VGPR matcher workgroups -- three per CU, so that a finished one
    // leaves room for the sweep beside the other two -- and the sweep enqueued behind the surface kernels on a stream of
    // its own: matcher 1.35 -> 2.40 ms, sweep 1.12 -> 1.89 ms, 4.2 ms per 4096-stream frame batch instead of 3.5.  The
    // sweep's short-lived workgroups keep taking the LDS a registration needs; the two do not share a CU gracefully.)
    // The next frame's filter needs no state of this frame: enqueue it now so the GPU sweeps the next
    // polar batch while the host applies the keyframe policy below.
    // A failed prefetch does not throw this frame away: its kernels already ran and its policy is applied below;
    // the error is reported after the frame is complete and the next call simply filters again.
    prefetch_rc = run_filter(od, polar_next, od->cur_buf ^ 1, offsets_next);
    if (prefetch_rc == CFEAR_OK) {
      od->prefetched = polar_next;
      if (offsets_next) od->prefetched_offsets.assign(offsets_next, offsets_next + B);
    }
  }
  g_tl.mark(2);
  OD_CHECK(hipEventSynchronize(od->ev_results));
  g_tl.mark(3);
  if (n_jobs > 0 && !launched_big) {
    // A registration too large for the first form of a launch WITHOUT the large forms comes back as CFEAR_ERR_CAPACITY with
    // `reserved` set when a larger form holds it (matcher_kernel): the first dense frame of a run.  Run the frame's jobs again
    // with the large forms on (the job records are still on the device; everything else of the frame is unaffected) and keep
    // them on for the frames to come (big_regs below) -- without this the stream ran on its motion guess for ever.
    bool retry = false;
    for (int j = 0; j < n_jobs && !retry; j++) retry = od->h_results[j].status == CFEAR_ERR_CAPACITY && od->h_results[j].reserved != 0.0;
    if (retry) {
      rc = launch_register(true);
      if (rc != CFEAR_OK) return fail(rc);
      OD_CHECK(hipMemcpyAsync(od->h_results, od->d_results, (size_t)n_jobs * sizeof(cfear_reg_result), hipMemcpyDeviceToHost, ctx->stream));
      OD_CHECK(hipStreamSynchronize(ctx->stream));
    }
  }
  if (od->decode_measured[od->cur_buf]) {                  // this frame's sweeps went through the fused decode: how dense were they?
    od->decode_measured[od->cur_buf] = false;
    uint64_t cands = 0;
    for (int k = 0; k < 64; k++) cands += od->h_decode_stats[64 * od->cur_buf + k];
    if ((double)cands > kDecodeDense * (double)B * (double)od->desc.rows) od->decode_hold = kDecodeHold;
  }
  // ---- frame policy (:195-257) ------------------------------------------------------------------
  int first_error = CFEAR_OK;
  bool saw_big = false;
  for (int b = 0; b < B; b++) {
    Stream& st = od->streams[b];
    cfear_frame_info& fi = info[b];
    memset(&fi, 0, sizeof(fi));
    fi.n_points = od->h_npts[b];
    fi.n_cells = od->h_ncells[b];
    if (od->h_status[b] != CFEAR_OK) {
      // empty cloud / capacity: the reference would exit(0) (pointnormal.cpp:72-75); report and keep the state
      if (first_error == CFEAR_OK)
        first_error = cfear_set_error(ctx, od->h_status[b], "stream %d: surface points failed (%s)", b, cfear_status_string(od->h_status[b]));
      fi.reg_status = od->h_status[b];
      st.free_slabs.push_back(st.cur_slab);
      aff_to_xyt(st.Tcurrent, fi.pose);
      continue;
    }
    st.has_constraint = false;
    if (st.job < 0) {                                             // first frame becomes the first keyframe
      st.keyframes.push_back(Keyframe{st.cur_slab, aff_identity(), st.n_keyframes++});
      fi.keyframe_added = 1;
      fi.reg_status = 1;
      aff_to_xyt(st.Tcurrent, fi.pose);
      continue;
    }
    const cfear_reg_result& rr = od->h_results[st.job];
    fi.reg_status = rr.status; fi.outer_iters = rr.outer_iters; fi.lm_iters = rr.lm_iters; fi.score = rr.score;
    od->cost_est[b] = (double)rr.num_residuals * (3.0 * rr.outer_iters + rr.lm_iters);   // association ~ 3 LM iterations
    if (rr.reserved != 0.0) saw_big = true;
    {                                                             // cov_current = cov_vek.back() (:196)
      double* cv = od->cov.data() + (size_t)b * 36;
      for (int k = 0; k < 36; k++) cv[k] = 0.0;
      if (rr.status == CFEAR_OK) { cv[0] = 0.1 * 0.1; cv[7] = 0.1 * 0.1; cv[35] = 0.01 * 0.01; }   // n_scan_normal.cpp:171-175
      else for (int k = 0; k < 6; k++) cv[k * 7] = 1.0;          // FormatScans' Identity66 survives a failed Register
      od->cov_sampled[b] = 0;
      if (par.estimate_cov_by_sampling && rr.num_residuals - 3 != 0) {         // GetCovarianceScaler (:433-439)
        const int m = od->fit.m;
        std::vector<double> costs(m);
        double sample_cost = 0.0;                                 // a failed GetCost keeps the previous value (:282, 307)
        for (int s = 0; s < m; s++) {
          const cfear_reg_result& sr = od->h_samples[(size_t)st.job * m + s];
          if (sr.status == CFEAR_OK) sample_cost = sr.final_cost;
          costs[s] = sample_cost;
        }
        double c36[36];
        if (od->fit.solve(costs.data(), rr.final_cost / (double)(rr.num_residuals - 3), par.cov_sampling.covariance_scaler, c36)) {
          memcpy(cv, c36, sizeof(c36));
          od->cov_sampled[b] = 1;
        }
      }
    }
    // Tcurrent = T_vek.back(): Register rewrites Tsrc via vectorToAffine3d (registration failures are
    // ignored by the reference: `bool success` is shadowed, odometrykeyframefuser.cpp:184-193)
    Aff2 Tcurrent = aff_from_xyt(rr.pose[0], rr.pose[1], rr.pose[2]);
    {                                                             // AccelerationVelocitySanityCheck :76-94
      const Aff2 Tmot_current = aff_mul(aff_inv(st.T_prev), Tcurrent);
      if (!cfear_acc_vel_sanity_check(st.Tmot.t, Tmot_current.t)) Tcurrent = st.Tguess;   // :198-199
    }
    st.Tmot = aff_mul(aff_inv(st.T_prev), Tcurrent);              // :200
    const Aff2 Tkeydiff = aff_mul(aff_inv(st.keyframes.back().pose), Tcurrent);
    double kd[3];
    aff_to_xyt(Tkeydiff, kd);
    const bool fuse = cfear_keyframe_based_fuse(kd, par.use_keyframe, par.min_keyframe_dist, par.min_keyframe_rot_deg) != 0;   // :62-73
    if (fuse) {                                                   // :236-250, AddToReference :470-476
      {                                                           // AddToGraph :428-445: constraint to the latest keyframe
        const Keyframe& to = st.keyframes.back();
        st.has_constraint = true;
        st.c_from = st.n_keyframes; st.c_to = to.idx;
        st.c_Tdiff = aff_mul(aff_inv(Tcurrent), to.pose);         // Tfrom^-1 * Tto
        const double* cv = od->cov.data() + (size_t)b * 36;
        memcpy(st.c_cov, cv, sizeof(st.c_cov));
        // C.block<3,3>(0,0) = Tfrom^-1.rotation() * Cov.block<3,3>(0,0) * Tfrom^-1.rotation()^T
        const Aff2 Ti = aff_inv(Tcurrent);
        const double R[9] = {Ti.l[0], Ti.l[1], 0, Ti.l[2], Ti.l[3], 0, 0, 0, 1};
        double t[9], o[9];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { t[r * 3 + c] = 0; for (int k = 0; k < 3; k++) t[r * 3 + c] += R[r * 3 + k] * cv[k * 6 + c]; }
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { o[r * 3 + c] = 0; for (int k = 0; k < 3; k++) o[r * 3 + c] += t[r * 3 + k] * R[c * 3 + k]; }
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) st.c_cov[r * 6 + c] = o[r * 3 + c];
      }
      st.keyframes.push_back(Keyframe{st.cur_slab, Tcurrent, st.n_keyframes++});
      if ((int)st.keyframes.size() > par.submap_scan_size) {
        st.free_slabs.push_back(st.keyframes.front().slab);
        st.keyframes.erase(st.keyframes.begin());
      }
      fi.keyframe_added = 1;
    } else {
      st.free_slabs.push_back(st.cur_slab);
    }
    st.Tcurrent = Tcurrent;
    st.T_prev = Tcurrent;                                         // :257
    aff_to_xyt(Tcurrent, fi.pose);
  }
  for (int b = 0; b < B; b++) od->streams[b].cur_slab = -1;      // every slab is a keyframe or back on the free list
#undef OD_CHECK
  od->big_regs = saw_big ? 64 : std::max(0, od->big_regs - 1);     // large scans come in runs: keep the second launch for 64 frames
  g_tl.mark(4);
  g_tl.end();
  return first_error != CFEAR_OK ? first_error : prefetch_rc;
}

// ---- graph-node export: the RadarScan of the last processed frame (types.h:119-122; scan_ at
// odometrykeyframefuser.cpp:172, 244) --------------------------------------------------------------------------
extern "C" int cfear_odometry_get_scan(cfear_odometry* od, int32_t stream, cfear_scan** out) {
  if (!od || !out || stream < 0 || stream >= od->n_streams) return CFEAR_ERR_INVALID_ARGUMENT;
  cfear_ctx* ctx = od->ctx;
  *out = nullptr;
  if (od->last_slab[stream] < 0) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "stream %d: no frame processed yet", stream);
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const ScanView& v = od->views[(size_t)stream * od->slabs_per_stream + od->last_slab[stream]];
  return cfear_scan_clone_view(ctx, v, od->h_ncells[stream], out);
}

static int copy_cloud_out(cfear_odometry* od, const float* d_src, int n, float* xyzi, int32_t cap, int32_t* n_out) {
  cfear_ctx* ctx = od->ctx;
  if (n_out) *n_out = n;
  if (!xyzi) return CFEAR_OK;
  if (n > cap) return cfear_set_error(ctx, CFEAR_ERR_CAPACITY, "%d points > cap %d", n, cap);
  if (n == 0) return CFEAR_OK;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const bool dev = cfear_is_device_ptr(xyzi);
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(xyzi, d_src, (size_t)n * 16, dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
  if (!dev) CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CFEAR_OK;
}

extern "C" int cfear_odometry_get_cloud(cfear_odometry* od, int32_t stream, float* xyzi, int32_t cap, int32_t* n_out) {
  if (!od || stream < 0 || stream >= od->n_streams || cap < 0) return CFEAR_ERR_INVALID_ARGUMENT;
  if (od->last_slab[stream] < 0) return cfear_set_error(od->ctx, CFEAR_ERR_INVALID_ARGUMENT, "stream %d: no frame processed yet", stream);
  return copy_cloud_out(od, od->d_xyzi + (size_t)stream * od->cap_points * 4, od->h_npts[stream], xyzi, cap, n_out);
}

extern "C" int cfear_odometry_get_peaks(cfear_odometry* od, int32_t stream, float* xyzi, int32_t cap, int32_t* n_out) {
  if (!od || stream < 0 || stream >= od->n_streams || cap < 0) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!od->par.keep_nodes) return cfear_set_error(od->ctx, CFEAR_ERR_INVALID_ARGUMENT, "peaks clouds are kept only with keep_nodes = 1");
  if (od->last_slab[stream] < 0) return cfear_set_error(od->ctx, CFEAR_ERR_INVALID_ARGUMENT, "stream %d: no frame processed yet", stream);
  return copy_cloud_out(od, od->d_pk + (size_t)stream * od->cap_points * 4, od->h_npk[stream], xyzi, cap, n_out);
}

extern "C" int cfear_odometry_get_constraint(cfear_odometry* od, int32_t stream, cfear_graph_constraint* out) {
  if (!od || !out || stream < 0 || stream >= od->n_streams) return CFEAR_ERR_INVALID_ARGUMENT;
  const Stream& st = od->streams[stream];
  if (!st.has_constraint) return cfear_set_error(od->ctx, CFEAR_ERR_INVALID_ARGUMENT, "stream %d: the last frame added no keyframe behind another one", stream);
  memset(out, 0, sizeof(*out));
  out->id_begin = st.c_from; out->id_end = st.c_to;
  double xyt[3];
  aff_to_xyt(st.c_Tdiff, xyt);
  cfear_pose3d_from_xyt(xyt, &out->t_be);
  out->type = 0;                                                  // ConstraintType::odometry
  // information on the planar sub-space (x, y, yaw) = indices 0, 1, 5; the full 6x6 is singular (see cfear_hip.h)
  const int ix[3] = {0, 1, 5};
  double M[9], I[9];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M[r * 3 + c] = st.c_cov[ix[r] * 6 + ix[c]];
  const double det = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
  if (det != 0.0 && std::isfinite(det)) {
    I[0] = (M[4] * M[8] - M[5] * M[7]) / det; I[1] = (M[2] * M[7] - M[1] * M[8]) / det; I[2] = (M[1] * M[5] - M[2] * M[4]) / det;
    I[3] = (M[5] * M[6] - M[3] * M[8]) / det; I[4] = (M[0] * M[8] - M[2] * M[6]) / det; I[5] = (M[2] * M[3] - M[0] * M[5]) / det;
    I[6] = (M[3] * M[7] - M[4] * M[6]) / det; I[7] = (M[1] * M[6] - M[0] * M[7]) / det; I[8] = (M[0] * M[4] - M[1] * M[3]) / det;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) out->information[ix[r] * 6 + ix[c]] = I[r * 3 + c];
  }
  return CFEAR_OK;
}
