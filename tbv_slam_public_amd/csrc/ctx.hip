// ctx.hip -- context, error reporting, profiling, workspaces and scan slabs of libcfear_hip.so.
#include <cstdarg>

#include "common.hpp"

int cfear_set_error(cfear_ctx* ctx, int status, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->last_error = buf;
  return status;
}

bool cfear_is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();   // clear the sticky error an unregistered host pointer leaves
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

void* cfear_workspace(cfear_ctx* ctx, int slot, size_t bytes) {
  cfear_ctx::Ws& w = ctx->ws[slot];
  if (w.bytes >= bytes && w.p) return w.p;
  if (w.p) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(w.p);
    w.p = nullptr;
    w.bytes = 0;
  }
  size_t want = bytes + bytes / 4 + 4096;
  if (hipMalloc(&w.p, want) != hipSuccess) {
    (void)hipGetLastError();
    w.p = nullptr;
    return nullptr;
  }
  w.bytes = want;
  return w.p;
}

void cfear_pinned_mark(cfear_ctx* ctx) {
  if (!ctx->pinned_ev && hipEventCreateWithFlags(&ctx->pinned_ev, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    ctx->pinned_ev = nullptr;
    (void)hipStreamSynchronize(ctx->stream);               // no event: the copy is simply waited for here
    return;
  }
  ctx->pinned_busy = hipEventRecord(ctx->pinned_ev, ctx->stream) == hipSuccess;
  if (!ctx->pinned_busy) (void)hipStreamSynchronize(ctx->stream);
}

void* cfear_pinned(cfear_ctx* ctx, size_t bytes) {
  if (ctx->pinned_busy) { (void)hipEventSynchronize(ctx->pinned_ev); ctx->pinned_busy = false; }
  if (ctx->pinned_bytes >= bytes && ctx->pinned) return ctx->pinned;
  if (ctx->pinned) { (void)hipStreamSynchronize(ctx->stream); (void)hipHostFree(ctx->pinned); ctx->pinned = nullptr; }
  size_t want = bytes + 4096;
  if (hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    ctx->pinned = nullptr;
    ctx->pinned_bytes = 0;
    return nullptr;
  }
  ctx->pinned_bytes = want;
  return ctx->pinned;
}

int cfear_allow_lds(cfear_ctx* ctx, const void* kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return CFEAR_OK;                     // the default limit: nothing to ask for
  for (auto& e : ctx->lds_allowed)
    if (e.first == kernel) {
      if ((size_t)e.second >= bytes) return CFEAR_OK;
      CFEAR_HIP_CHECK(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      e.second = (int)bytes;
      return CFEAR_OK;
    }
  CFEAR_HIP_CHECK(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  ctx->lds_allowed.emplace_back(kernel, (int)bytes);
  return CFEAR_OK;
}

// ---- profiling --------------------------------------------------------------------------------
int cfear_prof_row(cfear_ctx* ctx, const char* name) {
  for (size_t i = 0; i < ctx->prof.size(); i++)
    if (ctx->prof[i].name == name || strcmp(ctx->prof[i].name, name) == 0) return (int)i;
  ProfRow r;
  r.name = name;
  ctx->prof.push_back(r);
  return (int)ctx->prof.size() - 1;
}

static hipEvent_t prof_event(cfear_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

static void prof_resolve(cfear_ctx* ctx) {
  for (auto& r : ctx->prof) {
    for (auto& p : r.pending) {
      float ms = 0.f;
      if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
        r.total_ms += ms;
        r.launches++;
      }
      ctx->event_pool.push_back(p.first);
      ctx->event_pool.push_back(p.second);
    }
    r.pending.clear();
  }
}

void cfear_prof_begin(cfear_ctx* ctx, int row) {
  hipEvent_t a = prof_event(ctx), b = prof_event(ctx);
  (void)hipEventRecord(a, ctx->stream);
  ctx->prof[row].pending.emplace_back(a, b);
  if (ctx->prof[row].pending.size() > 4096) prof_resolve(ctx);
}

void cfear_prof_end(cfear_ctx* ctx, int row) {
  if (ctx->prof[row].pending.empty()) return;
  (void)hipEventRecord(ctx->prof[row].pending.back().second, ctx->stream);
}

// ---- scan slabs ---------------------------------------------------------------------------------
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

size_t cfear_scan_slab_bytes(int cap) {
  size_t c = (size_t)cap;
  size_t b = 256;                                  // header: n_cells counter
  b += align_up(c * sizeof(float2), 256);
  b += align_up(c * sizeof(double2), 256) * 3;     // mean, normal, lambda
  b += align_up(c * sizeof(double4), 256);
  b += align_up(c * sizeof(double), 256) * 2;      // scale, avg_intensity
  b += align_up(c * sizeof(int32_t), 256);
  b += align_up(scan_grid_bytes(cap), 256);        // prebuilt matcher grid: cell starts, then the records (ScanView::grid)
  return b;
}

ScanView cfear_scan_view(void* slab, int cap) {
  size_t c = (size_t)cap;
  char* p = (char*)slab;
  ScanView v;
  v.n_cells = (int32_t*)p; p += 256;
  v.mean_f = (float2*)p; p += align_up(c * sizeof(float2), 256);
  v.mean = (double2*)p; p += align_up(c * sizeof(double2), 256);
  v.normal = (double2*)p; p += align_up(c * sizeof(double2), 256);
  v.lambda = (double2*)p; p += align_up(c * sizeof(double2), 256);
  v.cov = (double4*)p; p += align_up(c * sizeof(double4), 256);
  v.scale = (double*)p; p += align_up(c * sizeof(double), 256);
  v.avg_intensity = (double*)p; p += align_up(c * sizeof(double), 256);
  v.nsamples = (int32_t*)p; p += align_up(c * sizeof(int32_t), 256);
  v.grid = (unsigned short*)p;                     // ONE block: a registration copies cell starts and records in one sweep
  v.grid_geo = (float4*)((char*)slab + 16);        // inside the 256-byte header, behind the counter
  v.cap = cap;
  v.pad = 0;
  return v;
}

int cfear_scan_alloc(cfear_ctx* ctx, int cap, cfear_scan** out) {
  void* slab = nullptr;
  // reuse a freed slab of sufficient capacity (streaming odometry creates one scan per frame)
  int best = -1;
  for (size_t i = 0; i < ctx->free_slabs.size(); i++)
    if (ctx->free_slabs[i].cap >= cap && (best < 0 || ctx->free_slabs[i].cap < ctx->free_slabs[best].cap)) best = (int)i;
  if (best >= 0) {
    slab = ctx->free_slabs[best].p;
    cap = ctx->free_slabs[best].cap;
    ctx->free_slabs.erase(ctx->free_slabs.begin() + best);
  } else {
    CFEAR_HIP_CHECK(ctx, hipMalloc(&slab, cfear_scan_slab_bytes(cap)));
  }
  cfear_scan* s = new cfear_scan();
  s->ctx = ctx;
  s->slab = slab;
  s->view = cfear_scan_view(slab, cap);
  s->n_cells_host = -1;
  ctx->live_scans++;
  *out = s;
  return CFEAR_OK;
}

// Device-to-device copy of a scan (n cells of every array + the counter) into a fresh handle.
int cfear_scan_clone_view(cfear_ctx* ctx, const ScanView& src, int n, cfear_scan** out) {
  cfear_scan* s = nullptr;
  const int rc = cfear_scan_alloc(ctx, std::max(n, 1), &s);
  if (rc != CFEAR_OK) return rc;
  const ScanView& d = s->view;
  const size_t c = (size_t)n;
  auto cp = [&](void* dst, const void* from, size_t bytes) {
    return bytes == 0 ? hipSuccess : hipMemcpyAsync(dst, from, bytes, hipMemcpyDeviceToDevice, ctx->stream);
  };
  hipError_t e = cp(d.n_cells, src.n_cells, 4);
  if (e == hipSuccess) e = cp(d.mean_f, src.mean_f, c * sizeof(float2));
  if (e == hipSuccess) e = cp(d.mean, src.mean, c * sizeof(double2));
  if (e == hipSuccess) e = cp(d.normal, src.normal, c * sizeof(double2));
  if (e == hipSuccess) e = cp(d.lambda, src.lambda, c * sizeof(double2));
  if (e == hipSuccess) e = cp(d.cov, src.cov, c * sizeof(double4));
  if (e == hipSuccess) e = cp(d.scale, src.scale, c * 8);
  if (e == hipSuccess) e = cp(d.avg_intensity, src.avg_intensity, c * 8);
  if (e == hipSuccess) e = cp(d.nsamples, src.nsamples, c * 4);
  if (e == hipSuccess) e = cp(d.grid, src.grid, scan_grid_bytes(n));
  if (e == hipSuccess) e = cp(d.grid_geo, src.grid_geo, sizeof(float4));
  if (e != hipSuccess) { cfear_scan_destroy(s); return cfear_set_error(ctx, CFEAR_ERR_HIP, "scan copy failed: %s", hipGetErrorString(e)); }
  s->n_cells_host = n;
  *out = s;
  return CFEAR_OK;
}

void cfear_scan_retain(cfear_scan* scan) { if (scan) scan->refs++; }

// ---- C-ABI ---------------------------------------------------------------------------------------
extern "C" {

int cfear_abi_version(void) { return CFEAR_ABI_VERSION; }

const char* cfear_status_string(int status) {
  switch (status) {
    case CFEAR_OK: return "ok";
    case CFEAR_ERR_INVALID_ARGUMENT: return "invalid argument";
    case CFEAR_ERR_HIP: return "HIP runtime error";
    case CFEAR_ERR_CAPACITY: return "capacity exceeded";
    case CFEAR_ERR_TOO_FEW_RESIDUALS: return "too few residuals";
    case CFEAR_ERR_SOLVER: return "solver failure";
    case CFEAR_ERR_EMPTY_CLOUD: return "empty cloud";
    case CFEAR_ERR_NO_DEVICE: return "no HIP device";
    case CFEAR_ERR_IO: return "file i/o failed";
    case CFEAR_ERR_FORMAT: return "not a simple_graph archive";
    default: return "unknown status";
  }
}

int cfear_ctx_create(int device, void* hip_stream, cfear_ctx** out) {
  if (!out) return CFEAR_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return CFEAR_ERR_NO_DEVICE;      // the product path fails loudly without a GPU
  }
  if (device < 0 || device >= ndev) return CFEAR_ERR_INVALID_ARGUMENT;
  if (hipSetDevice(device) != hipSuccess) return CFEAR_ERR_HIP;
  cfear_ctx* ctx = new cfear_ctx();
  ctx->device = device;
  if (hip_stream) {
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
  } else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
      delete ctx;
      return CFEAR_ERR_HIP;
    }
    ctx->own_stream = true;
  }
  (void)hipDeviceGetAttribute(&ctx->n_cu, hipDeviceAttributeMultiprocessorCount, device);
  if (ctx->n_cu < 1) ctx->n_cu = 256;
  *out = ctx;
  return CFEAR_OK;
}

int cfear_ctx_get_stream(const cfear_ctx* ctx, void** hip_stream) {
  if (!ctx || !hip_stream) return CFEAR_ERR_INVALID_ARGUMENT;
  *hip_stream = (void*)ctx->stream;
  return CFEAR_OK;
}

int cfear_ctx_set_option(cfear_ctx* ctx, int32_t option, int64_t value) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  bool ok = false;
  switch (option) {
    case CFEAR_OPT_FUSED_DECODE: case CFEAR_OPT_HOST_TIMELINE: ok = value == 0 || value == 1; break;
    case CFEAR_OPT_MATCHER_LDS_KB: ok = value == 0 || (value >= 8 && value <= 160); break;
    case CFEAR_OPT_MATCHER_WAVES: ok = value == 0 || value == 2 || value == 4 || value == 8 || value == 16; break;
    default: return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "unknown option %d", (int)option);
  }
  if (!ok) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "option %d: value %lld out of range", (int)option, (long long)value);
  ctx->opt[option] = value;
  return CFEAR_OK;
}

int cfear_ctx_get_option(const cfear_ctx* ctx, int32_t option, int64_t* value) {
  if (!ctx || !value || option < 0 || option >= CFEAR_OPT_COUNT) return CFEAR_ERR_INVALID_ARGUMENT;
  *value = ctx->opt[option];
  return CFEAR_OK;
}

int cfear_ctx_destroy(cfear_ctx* ctx) {
  if (!ctx) return CFEAR_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  prof_resolve(ctx);
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  for (auto& w : ctx->ws) if (w.p) (void)hipFree(w.p);
  for (auto& s : ctx->free_slabs) (void)hipFree(s.p);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->pinned_ev) (void)hipEventDestroy(ctx->pinned_ev);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return CFEAR_OK;
}

int cfear_ctx_synchronize(cfear_ctx* ctx) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CFEAR_OK;
}

const char* cfear_last_error(const cfear_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int cfear_ctx_profile_enable(cfear_ctx* ctx, int enable) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  ctx->profile = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
  return CFEAR_OK;
}

int cfear_ctx_profile_read(cfear_ctx* ctx, const char** names, double* total_ms, int64_t* launches,
                           int cap, int reset) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  (void)hipStreamSynchronize(ctx->stream);
  prof_resolve(ctx);
  int n = 0;
  for (auto& r : ctx->prof) {
    if (n < cap) {
      if (names) names[n] = r.name;
      if (total_ms) total_ms[n] = r.total_ms;
      if (launches) launches[n] = r.launches;
    }
    n++;
    if (reset) { r.total_ms = 0.0; r.launches = 0; }
  }
  return n;
}

void cfear_reg_params_default(cfear_reg_params* p) {
  memset(p, 0, sizeof(*p));
  p->cost = CFEAR_P2L;                 // registration.h:117
  p->loss = CFEAR_LOSS_HUBER;          // registration.h:118
  p->loss_limit = 0.1;                 // registration.h:119
  p->weight_opt = CFEAR_W_UNIFORM;     // registration.h:106
  p->max_itr_association = 8;          // n_scan_normal.h:75
  p->max_itr_solver = 20;              // n_scan_normal.cpp:9
  p->min_itr = 3;                      // n_scan_normal.h:75
  p->radius = 2.0;                     // registration.h:122
  p->cov_scale = 1.0;                  // n_scan_normal.h:72
  p->regularization = 0.01;            // n_scan_normal.h:73
  p->score_tolerance = 0.00001;        // n_scan_normal.h:74
  p->itr = 0;
}

void cfear_odometry_params_default(cfear_odometry_params* p) {
  memset(p, 0, sizeof(*p));
  // CFEAR-3 preset (launch/oxford/eval/params/baseline/oxford_cfear-3:13-26)
  p->filter_type = CFEAR_FILTER_KSTRONG;
  p->kstrong.k_strongest = 40;
  p->kstrong.z_min = 60.f;
  p->kstrong.range_res = 0.0438f;
  p->kstrong.min_distance = 2.5f;
  p->kstrong.want_peaks = 0;
  p->cacfar.window_size = 40;
  p->cacfar.nb_guard_cells = 10;
  p->cacfar.false_alarm_rate = 0.01f;
  p->cacfar.range_res = 0.0438f;
  p->cacfar.z_min = 20.f;
  p->cacfar.min_distance = 2.5f;
  p->cacfar.max_distance = 400.0;       // radar_driver.cpp:54
  cfear_reg_params_default(&p->reg);
  p->reg.cost = CFEAR_P2P;
  p->reg.weight_opt = CFEAR_W_COMBINED;
  p->reg.regularization = 0.0;          // OdometryKeyframeFuser::Parameters::regularization_
  p->res = 3.0f;
  p->submap_scan_size = 4;
  p->weight_intensity = 1;
  p->use_guess = 1;
  p->compensate = 1;
  p->radar_ccw = 0;
  p->use_keyframe = 1;
  p->rotate_ccw = 0;                   // Oxford images arrive with rows = azimuths (CallbackOxford)
  p->min_keyframe_dist = 1.5;
  p->min_keyframe_rot_deg = 5.0;
  p->downsample_factor = 1.0;
  p->estimate_cov_by_sampling = 0;
  p->keep_nodes = 0;
  cfear_cov_sampling_params_default(&p->cov_sampling);
}

int cfear_odometry_params_preset(cfear_odometry_params* p, int preset, int dataset) {
  if (!p) return CFEAR_ERR_INVALID_ARGUMENT;
  cfear_odometry_params_default(p);
  switch (preset) {                       // launch/oxford/eval/params/baseline/oxford_cfear-*:13-26
    case CFEAR_PRESET_CFEAR1:
    case CFEAR_PRESET_CFEAR2:
      p->reg.regularization = 1.0;        // EVALUATION_regularization="1" (:23); only P2D reads it (n_scan_normal.cpp:288-292)
      p->reg.cost = CFEAR_P2L;
      p->submap_scan_size = preset == CFEAR_PRESET_CFEAR1 ? 1 : 3;
      p->res = 3.5f;
      p->kstrong.k_strongest = 12;
      p->weight_intensity = 0;
      break;
    case CFEAR_PRESET_CFEAR3:
      p->reg.regularization = 1.0;        // oxford_cfear-3:23; the struct default 0.0 is OdometryKeyframeFuser::Parameters' own
      break;
    case CFEAR_PRESET_CFEAR3_S10:
      p->submap_scan_size = 10;
      p->reg.loss = CFEAR_LOSS_CAUCHY;
      p->reg.regularization = 0.1;
      break;
    default:
      return CFEAR_ERR_INVALID_ARGUMENT;
  }
  float range_res;
  switch (dataset) {                      // tbv_slam/script/*/run_tbv_simple.sh (range_res, radar_ccw, dataset)
    case CFEAR_DATASET_OXFORD: range_res = 0.0438f; p->radar_ccw = 0; p->rotate_ccw = 0; break;
    case CFEAR_DATASET_MULRAN: range_res = 0.0595238f; p->radar_ccw = 1; p->rotate_ccw = 1; break;
    case CFEAR_DATASET_KVARNTORP:
    case CFEAR_DATASET_VOLVO: range_res = 0.175f; p->radar_ccw = 1; p->rotate_ccw = 1; break;
    default:
      return CFEAR_ERR_INVALID_ARGUMENT;
  }
  p->kstrong.range_res = range_res;
  p->cacfar.range_res = range_res;
  return CFEAR_OK;
}

int cfear_scan_size(const cfear_scan* scan) {
  if (!scan) return CFEAR_ERR_INVALID_ARGUMENT;
  cfear_scan* s = const_cast<cfear_scan*>(scan);
  if (s->n_cells_host < 0) {
    int32_t n = 0;
    if (hipMemcpyAsync(&n, s->view.n_cells, sizeof(int32_t), hipMemcpyDeviceToHost, s->ctx->stream) != hipSuccess ||
        hipStreamSynchronize(s->ctx->stream) != hipSuccess)
      return cfear_set_error(s->ctx, CFEAR_ERR_HIP, "reading n_cells failed");
    s->n_cells_host = n;
  }
  return s->n_cells_host;
}

int cfear_scan_destroy(cfear_scan* scan) {
  if (!scan) return CFEAR_OK;
  if (--scan->refs > 0) return CFEAR_OK;                   // a scan table still names it (cfear_scan_retain)
  cfear_ctx* ctx = scan->ctx;
  // the slab may still be read by enqueued kernels of this stream; later users are on the same
  // stream, so recycling it is ordered.
  ctx->free_slabs.push_back(cfear_ctx::Slab{scan->slab, scan->view.cap});
  ctx->live_scans--;
  delete scan;
  return CFEAR_OK;
}

}  // extern "C"
