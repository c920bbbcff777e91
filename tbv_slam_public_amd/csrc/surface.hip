// surface.hip -- stages C + N of the CFEAR hot path on gfx950: motion compensation and oriented
// surface point ("cell") extraction.
//
// Replaces (cfear_radarodometry/src/cfear_radarodometry/):
//   Compensate / GetRelTimeStamp                      utils.cpp:96-107, utils.h:28-32
//   MapPointNormal ctor -> ComputeNormals             pointnormal.cpp:65-90, 265-297
//     pcl::VoxelGrid<PointXYZI>::filter (PCL 1.10)    call site pointnormal.cpp:277-280
//     pcl::search::KdTree::radiusSearchT (FLANN)      call site pointnormal.cpp:291
//   cell::cell / cell::ComputeNormal                  pointnormal.cpp:7-63
//   ComputeSearchTreeFromCells (float means)          pointnormal.cpp:151-162
//
// Kernel design: ONE 1024-thread workgroup per scan, one launch for a batch of scans.  The PCL
// voxel grid + kd-tree radius search are replaced by a sort-based uniform grid that lives in LDS:
//   1. (optional) per-point motion compensation in fp64, stored back as float;
//   2. bounding box -> voxel index per point -> LDS bitonic sort of 64-bit (voxel, point) keys
//      (128 KiB of the CU's 160 KiB LDS): voxels come out ascending (PCL's output order) and the
//      points of a voxel in input order (the canonical in-voxel order of the oracle);
//   3. per-voxel float centroid, summed sequentially in that order (bit-exact with the oracle);
//   4. one 16-lane DPP row per voxel gathers the points of the (2*reach+1)^2 neighbouring voxels
//      (contiguous runs of the sorted array), tests the float squared distance exactly as FLANN's
//      L2_Simple does, and accumulates weighted mean / covariance in fp64 with DPP row reductions;
//      closed-form 2x2 eigen decomposition, validity tests, normal orientation;
//   5. block scan over the validity flags -> cells compacted in voxel order.
// Working set per scan is a few hundred KB and stays in LDS / L2; nothing here is HBM-bound.
#include <cfloat>
#include <cmath>
#include <type_traits>

#include "common.hpp"
#include "fastmath.hpp"
#include "gridsort.hpp"

namespace {

constexpr int kSurfThreads = kGridSortThreads;
constexpr int kMaxPoints = kGridSortMaxPoints;   // LDS sort capacity (64-bit keys)
constexpr int kMaxGridRows = 4096;           // rowbeg table
constexpr int kPerThread = kMaxPoints / kSurfThreads;   // 16 sorted elements per thread
// LDS map: [0, 128K) sort keys, later voxel key/start tables; then rowbeg; then small reductions
constexpr size_t kLdsRowbegOff = (size_t)kMaxPoints * 8 + 16;
constexpr size_t kLdsSmallOff = (kLdsRowbegOff + (size_t)(kMaxGridRows + 1) * 4 + 15) / 16 * 16;
constexpr size_t kLdsTotal = kLdsSmallOff + 256 + 64 + 32;

struct SurfJob {                              // one scan
  float4* xyzi;                               // [n] points (compensated in place if requested)
  const int32_t* n_ptr;                       // device-side point count, or nullptr -> n_host
  int32_t n_host;
  int32_t compensate;
  double mot[3];
  ScanView out;
  // rows mode (batched odometry, fused filter output): the cloud arrives as per-row lists of packed keys (intensity
  // << 24 | range bin) -- row r holds row_cnt[2 r] keys at row_keys[r * k ...] -- and is compacted here, in row order,
  // and converted to PointXYZI (radar_filters.cpp:317-330) into xyzi (n_out receives n)
  const uint32_t* row_pts;
  const int32_t* row_cnt;
  int32_t* n_out;
  int32_t rows, k;
};

struct SurfCommon {
  float radius, leaf, inv_leaf, r2;
  int reach;                                  // neighbouring voxels to visit per axis
  int weight_intensity, ccw;
  double origin[2];
  // global scratch, per scan: sorted points + voxel tables + temporary cells
  char* scratch;
  size_t scratch_stride;
  int32_t* status;                            // [n_jobs] CFEAR_OK / error
  int32_t* ncells_out;                        // [n_jobs] dense copy of the cell counts (nullable)
  const double* cos_t;                        // rows mode: [rows] azimuth tables (host-computed doubles) and range_res
  const double* sin_t;
  double range_res, range_off;                // rho = range_off + range_res * bin
  int32_t* fallback;                          // [1 + n_jobs]: count, then the jobs the fast pipeline handed to the single-kernel path
  int32_t n_jobs;
  int32_t fast_ok;                            // reach == 1: the fast pipeline applies
  int32_t scratch_cap;                        // points the per-scan scratch is sized for (>= kMaxPoints)
  int32_t finish_keys;                        // sort keys surface_finish_kernel's LDS holds
  uint32_t finish_lds;                        // its dynamic LDS bytes
};

struct TmpCell {                              // one candidate cell per voxel (before compaction)
  double mean[2], normal[2], cov[4], scale, avg_intensity, lmin, lmax;
  int32_t nsamples, valid;
};

// Fast pipeline: what surface_sort_kernel leaves per voxel in the TmpCell slot (the first 64 of its 104 bytes) -- the raw
// moments of the neighbourhood; surface_finish_kernel turns them into the cell.
struct CellMom { double s0, s1x, s1y, sxx, sxy, syy; float cx, cy; int32_t cnt, pad; };
static_assert(sizeof(CellMom) <= sizeof(TmpCell), "CellMom lives in the TmpCell slot");
static_assert(sizeof(CellMom) == 64 && offsetof(CellMom, cx) == 48, "surface_finish_kernel reads a CellMom as four 16-byte pieces");

// Per-scan global scratch (one region per job, shared by the fast pipeline and the single-kernel fallback):
//   [0, 256)            SurfHdr
//   + 0       float4[N]    sorted points (x, y, weight, -)
//   + 16 N    u32[N]       fast: voxel keys         | fallback: float2 centroids [N] (8 N bytes)
//   + 24 N    TmpCell[N]   candidate cell per voxel
//   + 128 N   int32[N]     validity flags / compaction offsets
//   + 132 N   u32[N + 4]   fast: voxel starts (V + 1 entries)
constexpr int kFastMaxCells = 1 << 18;        // grid cells of the fast pipeline (one occupancy bit + 1/16 u16 per cell in LDS; the LDS check decides)
constexpr int kFastThreads = 512;
constexpr int kFastMaxPoints = 32768;         // points per scan the fast pipeline takes: up to kMaxPoints in the regular instantiation of
                                              // surface_sort_kernel (32 points per thread in registers), beyond in its second one (64)
constexpr size_t kFastLds = 78 * 1024;        // two workgroups per CU (160 KiB); three (52 KiB, 80 VGPRs) measured no faster: the kernel is issue-bound
constexpr int kRouteFast = 0, kRouteFallback = 1, kRouteDone = 2, kRoutePrepped = 3;
constexpr int kScanCreateMaxPoints = 1 << 20;  // cfear_scan_create: clouds beyond kMaxPoints take the global-memory path

struct SurfHdr {                              // written by surface_sort_kernel, read by the kernels behind it
  int32_t route;                              // kRouteFast: sorted, cells pending | kRouteFallback | kRouteDone (finished / failed)
  int32_t n, V, dbx, dby, pad[3];
};

// cap = the largest point count a scan of this launch may have (>= kMaxPoints); clouds beyond kMaxPoints take the
// big-cloud path, which also keeps its 64-bit sort keys here (2 x next_pow2 entries).
__host__ __device__ inline size_t scratch_bytes_per_scan(int cap) {
  size_t b = 256;
  b += (size_t)cap * 136 + 16;
  b = (b + 255) / 256 * 256;
  if (cap > kMaxPoints) b += (size_t)cap * 2 * 8 + 256;
  return (b + 255) / 256 * 256;
}
struct SurfScratch {
  SurfHdr* hdr; float4* spt; float2* cen; uint32_t* vkey; TmpCell* tmp; int32_t* coff; uint32_t* vs; unsigned short* ord;
  unsigned long long* bigkeys;
};
__host__ __device__ inline SurfScratch scratch_of(char* base, int cap) {
  SurfScratch r;
  r.hdr = (SurfHdr*)base;
  char* p = base + 256;
  r.spt = (float4*)p;
  r.cen = (float2*)(p + (size_t)cap * 16);
  r.vkey = (uint32_t*)(p + (size_t)cap * 16);
  r.tmp = (TmpCell*)(p + (size_t)cap * 24);
  r.coff = (int32_t*)(p + (size_t)cap * 128);
  r.vs = (uint32_t*)(p + (size_t)cap * 132);
  r.ord = (unsigned short*)(p + (size_t)cap * 136 + 16);
  size_t b = 256 + (size_t)cap * 136 + 16;
  b = (b + 255) / 256 * 256;
  r.bigkeys = (unsigned long long*)(base + b);
  return r;
}

__device__ __forceinline__ int row16_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false);
  return v;
}

// all-reduce over aligned groups of G lanes (G = 4: quad, G = 16: DPP row); every lane ends with the same sum
template <int G> __device__ __forceinline__ int group_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);             // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);             // quad_perm [2,3,0,1]
  if (G == 16) {
    v += __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false);          // row_half_mirror
    v += __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false);          // row_mirror
  }
  return v;
}
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
  const long long b = __double_as_longlong(v);
  int lo = (int)b, hi = (int)(b >> 32);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int G> __device__ __forceinline__ double group_sum_f64(double v) {
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  if (G == 16) { v += dpp_f64<0x141>(v); v += dpp_f64<0x140>(v); }
  return v;
}

// utils.h:28-32
__device__ __forceinline__ double get_rel_time_stamp(double x, double y, bool ccw) {
  const double a = atan2(y, x);
  const double d = ((a > 0.00001 ? a : (2 * M_PI + a)) / (2 * M_PI));
  return ccw ? -(d - 0.5) : (d - 0.5);
}

// The same for a point whose azimuth row is known (fused filter output): p = fl32(rho (cos th, sin th)) with th =
// (row + 1) / rows * 2 pi, so atan2(y, x) = th + delta with |delta| ~ 1e-7 from the float rounding of x and y, and
// delta = atan(t) = t to double precision for t = (y c - x s) / (x c + y s).  One division instead of an fp64 atan2;
// the result agrees with libm's atan2 to ~1 ulp, the level at which device and host libm differ anyway.
__device__ __forceinline__ double rel_time_stamp_known_row(double x, double y, double th, double c, double s, bool ccw) {
  // delta = num / den with den = rho (1 + O(1e-7)) > 0: a float reciprocal refined by one Newton step carries ~1e-14
  // relative error on a term that is itself ~1e-7 of th -- far below the last bit of th + delta
  const double den = x * c + y * s;
  double r = (double)__builtin_amdgcn_rcpf((float)den);
  r = r * (2.0 - den * r);
  const double a_full = th + (y * c - x * s) * r;                       // in (0, 2 pi]
  const double a = a_full > M_PI ? a_full - 2 * M_PI : a_full;          // atan2's range
  const double d = (a > 0.00001 ? a : (2 * M_PI + a)) * (1.0 / (2 * M_PI));
  return ccw ? -(d - 0.5) : (d - 0.5);
}

// utils.cpp:96-107
__device__ __forceinline__ float4 compensate_point_d(float4 p, const double d, const double mot[3]) {
  double s_1, c_1;
  sincos(d * mot[2], &s_1, &c_1);
  const double tx = d * mot[0], ty = d * mot[1];
  const double x = (double)p.x, y = (double)p.y;
  p.x = (float)((c_1 * x + (-s_1) * y) + tx);
  p.y = (float)((s_1 * x + c_1 * y) + ty);
  return p;
}

__device__ __forceinline__ float4 compensate_point_small(float4 p, const double d, const double mot[3]) {
  double s_1, c_1;
  sincos_reduced(d * mot[2], &s_1, &c_1);
  const double tx = d * mot[0], ty = d * mot[1];
  const double x = (double)p.x, y = (double)p.y;
  p.x = (float)((c_1 * x + (-s_1) * y) + tx);
  p.y = (float)((s_1 * x + c_1 * y) + ty);
  return p;
}

__device__ __forceinline__ float4 compensate_point(float4 p, const double mot[3], bool ccw) {
  const double d = get_rel_time_stamp((double)p.x, (double)p.y, ccw);
  double s_1, c_1;
  sincos(d * mot[2], &s_1, &c_1);                        // one range reduction for both
  const double tx = d * mot[0], ty = d * mot[1];
  const double x = (double)p.x, y = (double)p.y;
  p.x = (float)((c_1 * x + (-s_1) * y) + tx);
  p.y = (float)((s_1 * x + c_1 * y) + ty);
  return p;
}

__global__ __launch_bounds__(256) void compensate_kernel(float4* xyzi, int n, double m0, double m1, double m2, int ccw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double mot[3] = {m0, m1, m2};
  xyzi[i] = compensate_point(xyzi[i], mot, ccw != 0);
}

// Compensate for a batch of clouds with their own motion and device-side point counts (the peaks clouds of the
// batched odometry, odometrykeyframefuser.cpp:146-150): grid (chunks, clouds).
__global__ __launch_bounds__(256) void compensate_batch_kernel(float4* xyzi, size_t cloud_stride, const int32_t* n_pts,
                                                               const double* mot3, int ccw) {
  const int b = blockIdx.y;
  const int n = n_pts[b];
  const double mot[3] = {mot3[3 * b], mot3[3 * b + 1], mot3[3 * b + 2]};
  float4* c = xyzi + (size_t)b * cloud_stride;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) c[i] = compensate_point(c[i], mot, ccw != 0);
}

// symmetric 2x2 eigen decomposition (one Jacobi rotation), same formula as the oracle's sym2_eig
__device__ __forceinline__ void sym2_eig(double a, double b, double d, double& l0, double& l1, double v0[2]) {
  double c = 1.0, s = 0.0, e0 = a, e1 = d;
  if (b != 0.0) {
    const double theta = (d - a) / (2.0 * b);
    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    c = 1.0 / sqrt(t * t + 1.0);
    s = t * c;
    e0 = a - t * b;
    e1 = d + t * b;
  }
  if (e0 <= e1) { l0 = e0; l1 = e1; v0[0] = c; v0[1] = -s; }
  else { l0 = e1; l1 = e0; v0[0] = s; v0[1] = c; }
}

// ---- shared by the single-kernel path and surface_cells_kernel: one oriented surface point from its neighbourhood ----
struct Moments { int cnt; double s0, s1x, s1y, sxx, sxy, syy; };

// ONE pass over the candidates: weighted raw moments about the voxel centroid (|x'| <= radius, so forming mean /
// covariance from them loses a few ulp at most; the reference's normalise-then-two-pass form, pointnormal.cpp:18-33,
// is algebraically the same).  q = (x, y, weight max(I - 60, 0)).
__device__ __forceinline__ void accum_point(Moments& m, const float2 c, const double cx, const double cy, const float qx,
                                            const float qy, const float qw, const float r2, const bool weight_intensity) {
  const float dx = __fsub_rn(c.x, qx), dy = __fsub_rn(c.y, qy);
  const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));         // FLANN L2_Simple
  if (d2 < r2) {                                                             // RadiusResultSet: strict <
    const double w = weight_intensity ? (double)qw : 1.0;                    // pointnormal.cpp:15
    const double xr = (double)qx - cx, yr = (double)qy - cy;
    const double wx = w * xr, wy = w * yr;
    m.cnt++;
    m.s0 += w; m.s1x += wx; m.s1y += wy;
    m.sxx = fma(wx, xr, m.sxx); m.sxy = fma(wx, yr, m.sxy); m.syy = fma(wy, yr, m.syy);   // own one-pass form: fusing is free
  }
}

// cell::cell + cell::ComputeNormal (pointnormal.cpp:7-63) from the moments; returns valid_.
__device__ __forceinline__ int finish_cell(const Moments& m, const double cx, const double cy, const double ox, const double oy,
                                           TmpCell& tc) {
  if (m.cnt < 6) return 0;                                                   // pointnormal.cpp:291
  const double sum_intensity = m.s0;
  const double mx = m.s1x / sum_intensity, my = m.s1y / sum_intensity;
  const double u0 = cx + mx, u1 = cy + my;
  const double c00 = m.sxx / sum_intensity - mx * mx;
  const double c10 = m.sxy / sum_intensity - mx * my;
  const double c01 = c10;
  const double c11 = m.syy / sum_intensity - my * my;
  double lmin, lmax, vmin[2];
  sym2_eig(c00, c10, c11, lmin, lmax, vmin);                               // pointnormal.cpp:39-45
  const double condition_number = fabs(lmax / lmin);                       // :53
  const double determinant = lmax * lmin;
  const bool cov_reasonable = (condition_number <= 10000) && (determinant > 0.00001) && lmin > 0 && lmax > 0;   // :56
  if (!cov_reasonable) return 0;
  double n0 = vmin[0], n1 = vmin[1];
  if (n0 * (ox - u0) + n1 * (oy - u1) < 0) { n0 = -n0; n1 = -n1; }         // :59-61
  tc.mean[0] = u0; tc.mean[1] = u1;
  tc.normal[0] = n0; tc.normal[1] = n1;
  tc.cov[0] = c00; tc.cov[1] = c01; tc.cov[2] = c10; tc.cov[3] = c11;
  tc.scale = log(1.0 + condition_number / 2);                              // :57
  tc.avg_intensity = sum_intensity / (double)m.cnt;                        // :19
  tc.lmin = lmin; tc.lmax = lmax;
  tc.nsamples = m.cnt;
  tc.valid = 1;
  return 1;
}

__device__ __forceinline__ int lower_bound_u32(const uint32_t* a, int lo, int hi, uint32_t key) {
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}
__device__ __forceinline__ int upper_bound_u32(const uint32_t* a, int lo, int hi, uint32_t key) {
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] <= key) lo = mid + 1; else hi = mid; }
  return lo;
}

// The matcher's search structure of a scan (ScanView::grid): the float means bucketed into a uniform kScanGrid x kScanGrid
// grid over the scan's own extent, built ONCE per scan -- the counterpart of ComputeSearchTreeFromCells (pointnormal.cpp:
// 151-162), which builds the reference's kd-tree once per MapPointNormal.  A registration copies the block into LDS
// (matcher.hip) instead of bucketing every keyframe again.  Counting sort with LDS atomics; the order of the records inside
// a cell is arbitrary (the nearest-neighbour rule breaks ties by the cell index a record carries).
// Block-wide collective; lds: kScanGridLds bytes of scratch.
constexpr size_t kScanGridLds = (size_t)kScanGridCells * 4 + 128;
__device__ void grid_cells_block(const ScanView& v, int n, uint8_t* lds) {
  const int tid = threadIdx.x, nth = blockDim.x;
  unsigned* cnt = (unsigned*)lds;                            // [cells] counts -> scatter cursors
  unsigned* ext = cnt + kScanGridCells;                      // [4] ordered images of min x, max x, min y, max y | [16] wave totals
  auto ordered = [](float f) { unsigned u = __float_as_uint(f); return (u >> 31) ? ~u : (u | 0x80000000u); };
  auto unordered = [](unsigned u) { return __uint_as_float((u >> 31) ? (u & 0x7fffffffu) : ~u); };
  if (n > 65535) {                                           // the tables are 16-bit: the matcher reports such a scan (CFEAR_ERR_CAPACITY)
    if (tid == 0) *v.grid_geo = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const int np = scan_grid_pad(n);
  unsigned short* cstart = v.grid;
  float2* rec_xy = (float2*)(v.grid + kScanGridStartPad);
  unsigned short* rec_idx = (unsigned short*)(rec_xy + np);
  for (int c = tid; c < kScanGridCells; c += nth) cnt[c] = 0;
  if (tid < 4) ext[tid] = (tid & 1) ? 0u : 0xFFFFFFFFu;
  __syncthreads();
  {
    unsigned xl = 0xFFFFFFFFu, xh = 0u, yl = 0xFFFFFFFFu, yh = 0u;
    for (int i = tid; i < n; i += nth) {
      const g_f32x2 m = gload<g_f32x2>(v.mean_f + i);
      const unsigned ux = ordered(m.x), uy = ordered(m.y);
      xl = min(xl, ux); xh = max(xh, ux); yl = min(yl, uy); yh = max(yh, uy);
    }
    for (int o = 32; o > 0; o >>= 1) {
      xl = min(xl, (unsigned)__shfl_xor((int)xl, o)); xh = max(xh, (unsigned)__shfl_xor((int)xh, o));
      yl = min(yl, (unsigned)__shfl_xor((int)yl, o)); yh = max(yh, (unsigned)__shfl_xor((int)yh, o));
    }
    if ((tid & 63) == 0) { atomicMin(&ext[0], xl); atomicMax(&ext[1], xh); atomicMin(&ext[2], yl); atomicMax(&ext[3], yh); }
  }
  __syncthreads();
  float x0 = 0.f, y0 = 0.f, span = 0.f;
  if (n > 0) {
    x0 = unordered(ext[0]); y0 = unordered(ext[2]);
    span = fmaxf(unordered(ext[1]) - x0, unordered(ext[3]) - y0);
  }
  // cell edge: the extent split into kScanGrid cells, never below the matcher's radius (finer cells would only add rows to a query)
  const float inv = 1.0f / fmaxf(span / (float)kScanGrid * 1.0001f, kScanGridMinEdge);
  auto cell_of = [&](float x, float y) {                     // the same expression locates a query's cells in matcher.hip
    const int cx = min(kScanGrid - 1, max(0, (int)floorf((x - x0) * inv)));
    const int cy = min(kScanGrid - 1, max(0, (int)floorf((y - y0) * inv)));
    return cy * kScanGrid + cx;
  };
  for (int i = tid; i < n; i += nth) { const g_f32x2 m = gload<g_f32x2>(v.mean_f + i); atomicAdd(&cnt[cell_of(m.x, m.y)], 1u); }
  __syncthreads();
  {                                                          // exclusive scan over the cells
    const int per = (kScanGridCells + nth - 1) / nth, c0 = tid * per, c1 = min(kScanGridCells, c0 + per);
    int tot = 0;
    for (int c = c0; c < c1; c++) tot += (int)cnt[c];
    const int incl = wave_incl_scan_i32(tot);
    if ((tid & 63) == 63) ext[4 + (tid >> 6)] = (unsigned)incl;
    __syncthreads();
    int run = incl - tot;
    for (int wv = 0; wv < (tid >> 6); wv++) run += (int)ext[4 + wv];
    for (int c = c0; c < c1; c++) {
      const int k = (int)cnt[c];
      gstore<unsigned short>(cstart + c, (unsigned short)run);
      cnt[c] = (unsigned)run;
      run += k;
    }
    if (tid < kScanGridStartPad - kScanGridCells) gstore<unsigned short>(cstart + kScanGridCells + tid, (unsigned short)n);
  }
  __syncthreads();
  for (int i = tid; i < n; i += nth) {
    const g_f32x2 m = gload<g_f32x2>(v.mean_f + i);
    const unsigned pos = atomicAdd(&cnt[cell_of(m.x, m.y)], 1u);
    gstore<g_f32x2>(rec_xy + pos, m);
    gstore<unsigned short>(rec_idx + pos, (unsigned short)i);
  }
  if (tid < np - n) {                                        // the padding records are copied with the rest: defined values
    gstore<g_f32x2>(rec_xy + n + tid, g_f32x2{0.f, 0.f});
    gstore<unsigned short>(rec_idx + n + tid, (unsigned short)0);
  }
  if (tid == 0) gstore<g_f32x4>(v.grid_geo, g_f32x4{x0, y0, inv, 1.f});
}

__global__ __launch_bounds__(kSurfThreads) void scan_grid_kernel(ScanView v) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[kScanGridLds];
  grid_cells_block(v, *v.n_cells, smem);
}

// Clouds with more than kMaxPoints points (CA-CFAR sweeps: cfar.cpp:35-71 puts no bound on the detections per row):
// the tail of the single-kernel path with every array in global scratch -- bitonic sort of 64-bit (voxel, point) keys,
// voxel table by head flags, one lane per voxel with binary searches for the neighbour runs.  Slow (a few ms per scan)
// but unbounded by LDS; compaction and the x-sort are left to surface_finish_kernel.
__device__ void surface_big_tail(const SurfJob& job, const SurfCommon& cm, const int job_id, const int n, const int dbx,
                                 const int dby, const int min_bx, const int min_by, uint8_t* smem) {
  int* red_i = (int*)(smem + kLdsSmallOff + 256);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const SurfScratch scr = scratch_of(cm.scratch + (size_t)job_id * cm.scratch_stride, cm.scratch_cap);
  const float4* pts = job.xyzi;
  unsigned long long* keys = scr.bigkeys;
  int npad = 1024;
  while (npad < n) npad <<= 1;
  for (int i = tid; i < npad; i += kSurfThreads) {
    unsigned long long key = ~0ull;
    if (i < n) {
      const float4 p = pts[i];
      const int ijk0 = (int)(floorf(p.x * cm.inv_leaf) - (float)min_bx);
      const int ijk1 = (int)(floorf(p.y * cm.inv_leaf) - (float)min_by);
      key = ((unsigned long long)(uint32_t)(ijk0 + ijk1 * dbx) << 32) | (unsigned)i;
    }
    keys[i] = key;
  }
  __threadfence_block();
  __syncthreads();
  for (int k = 2; k <= npad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (npad >> 1); t += kSurfThreads) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int hi = lo | j;
        const bool asc = (lo & k) == 0;
        const unsigned long long a = keys[lo], b = keys[hi];
        if ((a > b) == asc) { keys[lo] = b; keys[hi] = a; }
      }
      __threadfence_block();
      __syncthreads();
    }
  }
  // voxel table: contiguous chunks per thread, head flags, block scan
  const int per = (n + kSurfThreads - 1) / kSurfThreads;
  const int e0 = min(n, tid * per), e1 = min(n, e0 + per);
  int heads = 0;
  for (int e = e0; e < e1; e++) heads += (e == 0 || (uint32_t)(keys[e] >> 32) != (uint32_t)(keys[e - 1] >> 32));
  const int incl = wave_incl_scan_i32(heads);
  if (lane == 63) red_i[wave] = incl;
  __syncthreads();
  int ordv = incl - heads, V = 0;
  for (int wv = 0; wv < 16; wv++) { if (wv < wave) ordv += red_i[wv]; V += red_i[wv]; }
  for (int e = e0; e < e1; e++) {
    const uint32_t vx = (uint32_t)(keys[e] >> 32);
    if (e == 0 || vx != (uint32_t)(keys[e - 1] >> 32)) { scr.vkey[ordv] = vx; scr.vs[ordv] = (uint32_t)e; ordv++; }
    const float4 p = pts[(uint32_t)(keys[e] & 0xFFFFFFFFu)];
    scr.spt[e] = make_float4(p.x, p.y, fmaxf(__fsub_rn(p.w, 60.0f), 0.0f), 0.f);
  }
  if (tid == 0) scr.vs[V] = (uint32_t)n;
  __threadfence_block();
  __syncthreads();
  const bool wi = cm.weight_intensity != 0;
  for (int v = tid; v < V; v += kSurfThreads) {
    const uint32_t key = scr.vkey[v];
    const int s = (int)scr.vs[v], e = (int)scr.vs[v + 1];
    const int iy = (int)(key / (uint32_t)dbx), ix = (int)(key - (uint32_t)iy * (uint32_t)dbx);
    const int x0 = max(ix - cm.reach, 0), x1 = min(ix + cm.reach, dbx - 1);
    float ax = 0.f, ay = 0.f;
    for (int p = s; p < e; p++) { const float4 q = scr.spt[p]; ax = __fadd_rn(ax, q.x); ay = __fadd_rn(ay, q.y); }
    const float cnt = (float)(e - s);
    const float2 c = make_float2(__fdiv_rn(ax, cnt), __fdiv_rn(ay, cnt));
    const double cx = (double)c.x, cy = (double)c.y;
    Moments mo{0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int yy = max(iy - cm.reach, 0); yy <= min(iy + cm.reach, dby - 1); yy++) {
      const uint32_t klo = (uint32_t)(yy * dbx + x0), khi = (uint32_t)(yy * dbx + x1);
      const int a = lower_bound_u32(scr.vkey, 0, V, klo), b = upper_bound_u32(scr.vkey, a, V, khi);
      const int p0 = (int)scr.vs[a], p1 = (int)scr.vs[b];
      for (int p = p0; p < p1; p++) { const float4 q = scr.spt[p]; accum_point(mo, c, cx, cy, q.x, q.y, q.z, cm.r2, wi); }
    }
    CellMom* cmo = (CellMom*)&scr.tmp[v];               // surface_finish_kernel turns the moments into the cell
    cmo->s0 = mo.s0; cmo->s1x = mo.s1x; cmo->s1y = mo.s1y; cmo->sxx = mo.sxx; cmo->sxy = mo.sxy; cmo->syy = mo.syy;
    cmo->cx = c.x; cmo->cy = c.y; cmo->cnt = mo.cnt;
    scr.coff[v] = mo.cnt;                               // (the neighbour counts on their own: surface_finish_kernel reads them first)
  }
  if (tid == 0) {
    scr.hdr->route = kRouteFast;                        // moments computed; cells + compaction + x-sort pending
    scr.hdr->n = n; scr.hdr->V = V; scr.hdr->dbx = dbx; scr.hdr->dby = dby;
  }
}

// The single-kernel path: one 1024-thread workgroup per scan, everything in 148 KiB of LDS; any grid size, n <= 16384.
// It serves the scans the fast pipeline (surface_sort / surface_cells / surface_finish below) hands over: voxel grids
// with more than kFastMaxCells cells, downsample factors != 1, or tables that do not fit the fast path's LDS budget.
__device__ void surface_points_job(const SurfJob* __restrict__ jobs, const SurfCommon& cm, const int job_id, uint8_t* smem) {
  float (*red_f)[16] = (float (*)[16])(smem + kLdsSmallOff);          // [4][16]
  int* red_i = (int*)(smem + kLdsSmallOff + 256);                     // [16]
  int* sh_misc = (int*)(smem + kLdsSmallOff + 256 + 64);              // [8]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  SurfJob job = jobs[job_id];
  if (cm.fallback) {                                                  // handed over by surface_sort_kernel
    const SurfHdr h = *scratch_of(cm.scratch + (size_t)job_id * cm.scratch_stride, cm.scratch_cap).hdr;
    if (h.pad[0]) {                                                   // xyzi already holds the compact, compensated cloud
      job.row_pts = nullptr; job.n_ptr = nullptr; job.n_host = h.n; job.compensate = 0;
    }
  }
  int n = job.n_ptr ? *job.n_ptr : job.n_host;
  int32_t* status = cm.status + job_id;
  int32_t* rowoff = (int32_t*)(smem + kLdsRowbegOff);                 // rows mode: [rows + 1] exclusive prefix of the row counts
  if (job.row_pts) {
    int run = 0, over = 0;
    for (int r0 = 0; r0 < job.rows; r0 += kSurfThreads) {
      const int r = r0 + tid;
      const int v = r < job.rows ? job.row_cnt[2 * r] : 0;
      over |= v > job.k;
      const int incl = wave_incl_scan_i32(v);
      if (lane == 63) red_i[wave] = incl;
      __syncthreads();
      int off = run + incl - v;
      for (int wv = 0; wv < wave; wv++) off += red_i[wv];
      if (r < job.rows) rowoff[r] = off;
      for (int wv = 0; wv < 16; wv++) run += red_i[wv];
      __syncthreads();
    }
    if (__syncthreads_or(over)) {
      if (tid == 0) { *job.out.n_cells = 0; *status = CFEAR_ERR_CAPACITY; if (cm.ncells_out) cm.ncells_out[job_id] = 0; }
      return;
    }
    n = run;
    if (tid == 0) { rowoff[job.rows] = n; if (job.n_out) *job.n_out = n; }
    __syncthreads();
  }
  if (n <= 0) {
    if (tid == 0) { *job.out.n_cells = 0; *status = CFEAR_ERR_EMPTY_CLOUD; if (cm.ncells_out) cm.ncells_out[job_id] = 0; }
    return;
  }
  if (n > cm.scratch_cap) {
    if (tid == 0) { *job.out.n_cells = 0; *status = CFEAR_ERR_CAPACITY; if (cm.ncells_out) cm.ncells_out[job_id] = 0; }
    return;
  }
  float4* pts = job.xyzi;
  const SurfScratch scr = scratch_of(cm.scratch + (size_t)job_id * cm.scratch_stride, cm.scratch_cap);
  float4* spt = scr.spt;                      // sorted points (x, y, weight, -)
  float2* cen = scr.cen;
  TmpCell* tmp = scr.tmp;
  int32_t* coff = scr.coff;

  // ---- 1. compensation + bounding box -------------------------------------------------------
  float mnx = FLT_MAX, mny = FLT_MAX, mxx = -FLT_MAX, mxy = -FLT_MAX;
  for (int i0 = tid; i0 < n; i0 += 4 * kSurfThreads) {  // four loads in flight per thread
    float4 p[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + u * kSurfThreads;
      if (i < n) {
        if (job.row_pts) {                              // row of point i: the last r with rowoff[r] <= i
          int lo = 0, hi = job.rows;
          while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (rowoff[mid] <= i) lo = mid; else hi = mid; }
          const uint32_t key = job.row_pts[(size_t)lo * job.k + (i - rowoff[lo])];
          const double rho = cm.range_off + cm.range_res * (double)(int)(key & 0xFFFFFFu);     // radar_filters.cpp:324-330 / cfar.cpp:43
          p[u] = make_float4((float)(rho * cm.cos_t[lo]), (float)(rho * cm.sin_t[lo]), 0.f, (float)(key >> 24));
        } else {
          p[u] = pts[i];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + u * kSurfThreads;
      if (i < n) {
        if (job.compensate) p[u] = compensate_point(p[u], job.mot, cm.ccw != 0);
        if (job.compensate || job.row_pts) pts[i] = p[u];
        mnx = fminf(mnx, p[u].x); mxx = fmaxf(mxx, p[u].x);
        mny = fminf(mny, p[u].y); mxy = fmaxf(mxy, p[u].y);
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    mnx = fminf(mnx, __shfl_xor(mnx, o)); mxx = fmaxf(mxx, __shfl_xor(mxx, o));
    mny = fminf(mny, __shfl_xor(mny, o)); mxy = fmaxf(mxy, __shfl_xor(mxy, o));
  }
  if (lane == 0) { red_f[0][wave] = mnx; red_f[1][wave] = mxx; red_f[2][wave] = mny; red_f[3][wave] = mxy; }
  __syncthreads();
  mnx = red_f[0][0]; mxx = red_f[1][0]; mny = red_f[2][0]; mxy = red_f[3][0];
  for (int wv = 1; wv < 16; wv++) {
    mnx = fminf(mnx, red_f[0][wv]); mxx = fmaxf(mxx, red_f[1][wv]);
    mny = fminf(mny, red_f[2][wv]); mxy = fmaxf(mxy, red_f[3][wv]);
  }
  // pcl::VoxelGrid::applyFilter: min_b = floor(min * inverse_leaf), div_b = max_b - min_b + 1
  const int min_bx = (int)floorf(mnx * cm.inv_leaf), max_bx = (int)floorf(mxx * cm.inv_leaf);
  const int min_by = (int)floorf(mny * cm.inv_leaf), max_by = (int)floorf(mxy * cm.inv_leaf);
  const long long div_bx = (long long)max_bx - min_bx + 1, div_by = (long long)max_by - min_by + 1;
  if (div_bx * div_by > 0x7fffffffLL || div_by > kMaxGridRows) {
    if (tid == 0) { *job.out.n_cells = 0; *status = CFEAR_ERR_CAPACITY; if (cm.ncells_out) cm.ncells_out[job_id] = 0; }
    return;
  }
  const int dbx = (int)div_bx, dby = (int)div_by;
  if (n > kMaxPoints) {                                 // more points than the LDS sort holds (CA-CFAR clouds): global memory
    surface_big_tail(job, cm, job_id, n, dbx, dby, min_bx, min_by, smem);
    return;
  }

  // ---- 2. (voxel, point) keys -> LDS sort: voxels ascending, points of a voxel in input order --
  unsigned long long* keys = (unsigned long long*)smem;
  const int npad = grid_sort_block(smem, n, (long long)dbx * dby, red_i, [&](int i) {
    const float4 p = pts[i];
    const int ijk0 = (int)(floorf(p.x * cm.inv_leaf) - (float)min_bx);
    const int ijk1 = (int)(floorf(p.y * cm.inv_leaf) - (float)min_by);
    return (uint32_t)(ijk0 + ijk1 * dbx);
  });

  // ---- 3. sorted points -> global scratch; voxel table (key, start) -> LDS --------------------
  // each thread owns kPerThread consecutive sorted elements, held in registers across the barrier
  // because the voxel tables overwrite the key region.
  const int per = npad / kSurfThreads;                 // 1..16
  unsigned long long mine[kPerThread];
  const unsigned prev_vox = (tid * per > 0) ? (unsigned)(keys[tid * per - 1] >> 32) : 0xFFFFFFFFu;
  int heads = 0;
#pragma unroll
  for (int q = 0; q < kPerThread; q++) {
    const int e = tid * per + q;
    mine[q] = (q < per && e < n) ? keys[e] : ~0ull;
  }
  {
    unsigned pv = prev_vox;
#pragma unroll
    for (int q = 0; q < kPerThread; q++) {
      const int e = tid * per + q;
      if (q < per && e < n) {
        const unsigned vx = (unsigned)(mine[q] >> 32);
        heads += (e == 0 || vx != pv);
        pv = vx;
      }
    }
  }
  // block exclusive scan of `heads`
  const int incl = wave_incl_scan_i32(heads);
  if (lane == 63) red_i[wave] = incl;
  __syncthreads();                                      // also: every thread has read its keys
  int voff = incl - heads;
  for (int wv = 0; wv < wave; wv++) voff += red_i[wv];
  int V = 0;
  for (int wv = 0; wv < 16; wv++) V += red_i[wv];
  // LDS map from here on (compact, by V and n): voxel keys | voxel starts | [sorted points x,y | intensity]
  const size_t Vp = ((size_t)V + 4) & ~(size_t)3;
  uint32_t* vox_key = (uint32_t*)smem;                  // [V]
  int32_t* vox_start = (int32_t*)(smem + Vp * 4);       // [V + 1]
  const size_t pts_off = (Vp * 8 + 15) & ~(size_t)15;
  const bool lds_pts = pts_off + (size_t)n * 12 + 16 <= kLdsRowbegOff - 16;   // points fit next to the tables
  float2* lxy = (float2*)(smem + pts_off);              // [n] sorted x,y
  float* lin = (float*)(smem + pts_off + (((size_t)n * 8 + 15) & ~(size_t)15));   // [n] sorted intensity
  int32_t* rowbeg = (int32_t*)(smem + kLdsRowbegOff);          // [dby + 1]
  {
    unsigned pv = prev_vox;
    int ord = voff;
    float4 gp[kPerThread];
#pragma unroll
    for (int q = 0; q < kPerThread; q++) {                // all gathers of this thread in flight together
      const int e = tid * per + q;
      if (q < per && e < n) gp[q] = pts[(unsigned)(mine[q] & 0xFFFFFFFFu)];
    }
#pragma unroll
    for (int q = 0; q < kPerThread; q++) {
      const int e = tid * per + q;
      if (q < per && e < n) {
        const unsigned vx = (unsigned)(mine[q] >> 32);
        if (e == 0 || vx != pv) { vox_key[ord] = vx; vox_start[ord] = e; ord++; }
        pv = vx;
        const float4 p = gp[q];
        // the sorted copies carry the point's WEIGHT max(I - 60, 0) (pointnormal.cpp:15), not its intensity:
        // float(I) - 60 is exact for I >= 60 (the difference of two floats is representable whenever it is not
        // larger in magnitude than both), so the fp64 weight of the reference is just its widening
        const float wgt = fmaxf(__fsub_rn(p.w, 60.0f), 0.0f);
        if (lds_pts) { lxy[e] = make_float2(p.x, p.y); lin[e] = wgt; }    // the usual case: no global copy at all
        else spt[e] = make_float4(p.x, p.y, wgt, 0.f);
      }
    }
  }
  if (tid == 0) vox_start[V] = n;
  __syncthreads();
  // rowbeg[y] = first voxel ordinal whose grid row is >= y
  for (int y = tid; y <= dby; y += kSurfThreads)
    rowbeg[y] = lower_bound_u32(vox_key, 0, V, (uint32_t)((long long)y * dbx));
  // ---- 4a. voxel centroids: sequential float sums in sorted (= input) order -------------------
  __threadfence_block();
  __syncthreads();
  for (int v = tid; v < V; v += kSurfThreads) {
    const int s = vox_start[v], e = vox_start[v + 1];
    float ax = 0.f, ay = 0.f;
    if (lds_pts) {
      for (int p = s; p < e; p++) { const float2 q = lxy[p]; ax = __fadd_rn(ax, q.x); ay = __fadd_rn(ay, q.y); }
    } else {
      for (int p = s; p < e; p++) { const float4 q = spt[p]; ax = __fadd_rn(ax, q.x); ay = __fadd_rn(ay, q.y); }
    }
    const float cnt = (float)(e - s);
    cen[v] = make_float2(__fdiv_rn(ax, cnt), __fdiv_rn(ay, cnt));
  }
  __threadfence_block();
  __syncthreads();

  // ---- 4b. one LANE per voxel: radius gather + weighted mean / covariance ----------------------
  // Neighbouring voxels (adjacent lanes) share most of their candidate points, so the per-lane
  // 16-byte loads hit L1; every lane keeps its own fp64 moments (no cross-lane reduction) and
  // voxels that cannot reach 6 neighbours cost nothing.
  for (int v = tid; v < V; v += kSurfThreads) {
    const float2 c = cen[v];
    const uint32_t key = vox_key[v];
    const int iy = (int)(key / (uint32_t)dbx), ix = (int)(key - (uint32_t)iy * (uint32_t)dbx);
    const int x0 = max(ix - cm.reach, 0), x1 = min(ix + cm.reach, dbx - 1);
    auto run_of = [&](int yy, int& p0, int& p1) {       // sorted-point run of grid row yy, columns x0..x1
      p0 = p1 = 0;
      if (yy < 0 || yy >= dby) return;
      const uint32_t klo = (uint32_t)(yy * dbx + x0), khi = (uint32_t)(yy * dbx + x1);
      const int a = lower_bound_u32(vox_key, rowbeg[yy], rowbeg[yy + 1], klo);
      const int b = upper_bound_u32(vox_key, rowbeg[yy], rowbeg[yy + 1], khi);
      p0 = vox_start[a];
      p1 = vox_start[b];
    };
    const double cx = (double)c.x, cy = (double)c.y;
    Moments mo{0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    auto accum = [&](const float4 q) { accum_point(mo, c, cx, cy, q.x, q.y, q.z, cm.r2, cm.weight_intensity != 0); };
    auto scan_run = [&](int p0, int p1) {
      if (lds_pts) {                                      // candidates from LDS (no L1/TA traffic)
        int p = p0;
        for (; p + 3 < p1; p += 4) {                      // four independent LDS reads in flight
          const float2 qa = lxy[p], qb = lxy[p + 1], qc = lxy[p + 2], qd = lxy[p + 3];
          float ia = 0.f, ib = 0.f, ic = 0.f, id = 0.f;
          if (cm.weight_intensity) { ia = lin[p]; ib = lin[p + 1]; ic = lin[p + 2]; id = lin[p + 3]; }
          accum(make_float4(qa.x, qa.y, ia, 0.f)); accum(make_float4(qb.x, qb.y, ib, 0.f));
          accum(make_float4(qc.x, qc.y, ic, 0.f)); accum(make_float4(qd.x, qd.y, id, 0.f));
        }
        for (; p < p1; p++) {
          const float2 q = lxy[p];
          accum(make_float4(q.x, q.y, cm.weight_intensity ? lin[p] : 0.f, 0.f));
        }
        return;
      }
      int p = p0;
      for (; p + 3 < p1; p += 4) {                        // four independent loads in flight
        const float4 qa = spt[p], qb = spt[p + 1], qc = spt[p + 2], qd = spt[p + 3];
        accum(qa); accum(qb); accum(qc); accum(qd);
      }
      for (; p < p1; p++) accum(spt[p]);
    };
    if (cm.reach == 1) {
      int a0, a1, b0, b1, c0, c1;
      run_of(iy - 1, a0, a1);
      {                                                   // centre row: the neighbours are v-1 / v+1 if occupied
        const int va = (v > 0 && ix > 0 && vox_key[v - 1] == key - 1) ? v - 1 : v;
        const int vb = (v + 1 < V && ix < dbx - 1 && vox_key[v + 1] == key + 1) ? v + 2 : v + 1;
        b0 = vox_start[va];
        b1 = vox_start[vb];
      }
      run_of(iy + 1, c0, c1);
      if ((a1 - a0) + (b1 - b0) + (c1 - c0) >= 6) {       // upper bound on the neighbour count
        scan_run(a0, a1);
        scan_run(b0, b1);
        scan_run(c0, c1);
      }
    } else {
      for (int yy = iy - cm.reach; yy <= iy + cm.reach; yy++) {
        int p0, p1;
        run_of(yy, p0, p1);
        scan_run(p0, p1);
      }
    }
    TmpCell tc;
    const int valid = finish_cell(mo, cx, cy, cm.origin[0], cm.origin[1], tc);
    if (valid) tmp[v] = tc;
    coff[v] = valid;
  }
  __threadfence_block();
  __syncthreads();

  // ---- 5. compaction in voxel order -----------------------------------------------------------
  if (tid == 0) sh_misc[0] = 0;
  __syncthreads();
  for (int v0 = 0; v0 < V; v0 += kSurfThreads) {
    const int v = v0 + tid;
    const int f = v < V ? coff[v] : 0;
    const int inc = wave_incl_scan_i32(f);
    if (lane == 63) red_i[wave] = inc;
    __syncthreads();
    int off = sh_misc[0] + inc - f;
    for (int wv = 0; wv < wave; wv++) off += red_i[wv];
    if (f) {
      if (off < job.out.cap) {
        const TmpCell t = tmp[v];
        job.out.mean_f[off] = make_float2((float)t.mean[0], (float)t.mean[1]);  // pointnormal.cpp:154-157
        job.out.mean[off] = make_double2(t.mean[0], t.mean[1]);
        job.out.normal[off] = make_double2(t.normal[0], t.normal[1]);
        job.out.cov[off] = make_double4(t.cov[0], t.cov[1], t.cov[2], t.cov[3]);
        job.out.scale[off] = t.scale;
        job.out.avg_intensity[off] = t.avg_intensity;
        job.out.lambda[off] = make_double2(t.lmin, t.lmax);
        job.out.nsamples[off] = t.nsamples;
      }
    }
    __syncthreads();
    if (tid == 0) { int tot = 0; for (int wv = 0; wv < 16; wv++) tot += red_i[wv]; sh_misc[0] += tot; }
    __syncthreads();
  }
  // ---- 6. the matcher's grid over the float means (its exact 1-NN search structure) -----------------
  __threadfence_block();
  __syncthreads();
  grid_cells_block(job.out, min(sh_misc[0], job.out.cap), smem);
  if (tid == 0) {
    const int total = sh_misc[0];
    *job.out.n_cells = total <= job.out.cap ? total : job.out.cap;
    *status = total <= job.out.cap ? CFEAR_OK : CFEAR_ERR_CAPACITY;
    if (cm.ncells_out) cm.ncells_out[job_id] = total <= job.out.cap ? total : job.out.cap;
  }
}

__global__ __launch_bounds__(kSurfThreads) void surface_points_kernel(const SurfJob* __restrict__ jobs, const SurfCommon cm) {
  // all LDS is carved from the dynamic region so its base stays 16-byte aligned (64-bit keys)
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (!cm.fallback) { surface_points_job(jobs, cm, blockIdx.x, smem); return; }
  const int count = min(cm.fallback[0], cm.n_jobs);   // persistent over the (usually empty) list of handed-over scans
  for (int w = blockIdx.x; w < count; w += gridDim.x) {
    surface_points_job(jobs, cm, cm.fallback[1 + w], smem);
    __syncthreads();
  }
}


// =================================================================================================================
// Fast pipeline (reach == 1, voxel grid <= kFastMaxCells cells, n <= 16384): two launches per batch
//   surface_sort_kernel    one 512-thread workgroup per scan, 78 KiB of LDS -> TWO scans per CU, whose phases overlap:
//                          row compaction / polar -> Cartesian / motion compensation / bounding box; a counting sort of
//                          the points by voxel (LDS histogram over the grid as u16 pairs, exclusive scan -> cell ->
//                          ordinal map and voxel starts, atomic scatter, an in-voxel rank pass that restores input
//                          order: PCL's centroid sums are sequential in input order) instead of a 4-pass radix sort
//                          with per-thread digit counters; then the cells, one lane per voxel, from slabs of sorted
//                          points staged in LDS, with the three neighbour runs found by O(1) look-ups in the cell ->
//                          ordinal map instead of binary searches.
//   surface_finish_kernel  per scan: compaction in voxel order + the x-sorted copy for the matcher (small LDS, 8
//                          wavefronts per SIMD).
// The first version ran all of this in one 1024-thread workgroup per CU (148 KiB of LDS) at ~3 % of the VALU peak: every
// phase waited on barriers and dependent LDS chains with 4 wavefronts per SIMD.  (A flat lane-per-voxel kernel over all
// scans, fed through global scratch, was tried in between: 1.6 - 2.1 ms per 2048 scans -- its dependent global look-ups
// and the staging barriers at 10 wavefronts per CU cost more than the old gather phase.)  Scans the fast path cannot
// take are appended to a work list that the single-kernel path (above) drains.
// =================================================================================================================
template <bool ROWS>
__global__ __launch_bounds__(kFastThreads) void surface_prep_kernel(const SurfJob* __restrict__ jobs, const SurfCommon cm) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int NT = kFastThreads, NW = NT / 64;
  __shared__ float red_f[4][NW];
  __shared__ int red_i[NW];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int job_id = blockIdx.x;
  const SurfJob job = jobs[job_id];
  const SurfScratch scr = scratch_of(cm.scratch + (size_t)job_id * cm.scratch_stride, cm.scratch_cap);
  int32_t* status = cm.status + job_id;
  auto done = [&](int st) {                                            // nothing (more) to do for this scan
    if (tid == 0) {
      scr.hdr->route = kRouteDone;
      *job.out.n_cells = 0; *status = st;
      if (cm.ncells_out) cm.ncells_out[job_id] = 0;
    }
  };
  auto hand_over = [&](int n, int prepared) {                          // to the single-kernel path
    if (tid == 0) {
      scr.hdr->route = kRouteFallback;
      scr.hdr->n = n;
      scr.hdr->pad[0] = prepared;                                      // 1: xyzi already holds the compact, compensated cloud
      const int w = atomicAdd(&cm.fallback[0], 1);
      if (w < cm.n_jobs) cm.fallback[1 + w] = job_id;         // (a stale count can never index past the list)
    }
  };
  if (!cm.fast_ok || (job.row_pts != nullptr) != ROWS) { hand_over(0, 0); return; }
  // rotations beyond the reduced sincos' range (never a real motion; also NaN): the single-kernel path uses libm
  if (job.compensate && !(fabs(job.mot[2]) <= 1e5)) { hand_over(0, 0); return; }
  // ---- (a) point count; rows mode: exclusive prefix of the row counts ------------------------------------------
  int n = job.n_ptr ? *job.n_ptr : job.n_host;
  int32_t* rowoff = (int32_t*)smem;
  if (ROWS) {
    int run = 0, over = 0;
    for (int r0 = 0; r0 < job.rows; r0 += NT) {
      const int r = r0 + tid;
      const int v = r < job.rows ? gload<int32_t>(job.row_cnt + 2 * r) : 0;
      over |= v > job.k;                                   // a CA-CFAR row beyond its key capacity (keys were dropped)
      const int incl = wave_incl_scan_i32(v);
      if (lane == 63) red_i[wave] = incl;
      __syncthreads();
      int off = run + incl - v;
      for (int wv = 0; wv < wave; wv++) off += red_i[wv];
      if (r < job.rows) rowoff[r] = off;
      for (int wv = 0; wv < NW; wv++) run += red_i[wv];
      __syncthreads();
    }
    if (__syncthreads_or(over)) { done(CFEAR_ERR_CAPACITY); return; }
    n = run;
    if (tid == 0) { rowoff[job.rows] = n; if (job.n_out) *job.n_out = n; }
    __syncthreads();
  }
  if (n <= 0) { done(CFEAR_ERR_EMPTY_CLOUD); return; }
  if (n > cm.scratch_cap) { done(CFEAR_ERR_CAPACITY); return; }
  if (n > kFastMaxPoints) { hand_over(n, 0); return; }     // big cloud: global-memory path of the single-kernel workgroup
  float4* pts = job.xyzi;
  // ---- (b) polar -> Cartesian (rows mode), motion compensation, bounding box ------------------------------------
  // rows mode: the azimuth row of every point as a u16 table behind rowoff (one thread per row fills its <= k slots;
  // a per-point binary search over rowoff cost nine dependent LDS reads)
  // and the (cos, sin) table of the azimuths, staged once per scan
  const size_t rid_off = ((size_t)(job.rows + 1) * 4 + 15) & ~(size_t)15;
  unsigned short* rid = (unsigned short*)(smem + rid_off);
  double2* cs = (double2*)(smem + ((rid_off + (size_t)n * 2 + 15) & ~(size_t)15));
  if (ROWS) {
    for (int r = tid; r < job.rows; r += NT) {
      const int o = rowoff[r], e = rowoff[r + 1];
      for (int q = o; q < e; q++) rid[q] = (unsigned short)r;
      cs[r] = make_double2(cm.cos_t[r], cm.sin_t[r]);
    }
    __syncthreads();
  }
  float mnx = FLT_MAX, mny = FLT_MAX, mxx = -FLT_MAX, mxy = -FLT_MAX;
  const double range_res_half = cm.range_off;
  const double th_step = (2. * M_PI) / (double)max(job.rows, 1);
  constexpr int U = 4;
  for (int i0 = tid; i0 < n; i0 += U * NT) {
    float4 p[U];
    int row[U];
    uint32_t key[U];
#pragma unroll
    for (int u = 0; u < U; u++) {                         // all global loads of the batch first
      const int i = i0 + u * NT;
      row[u] = 0; key[u] = 0;
      if (i < n) {
        if (ROWS) {
          row[u] = rid[i];
          key[u] = gload<uint32_t>(job.row_pts + (size_t)row[u] * job.k + (i - rowoff[row[u]]));   // (global_load, not flat_load)
        } else {
          p[u] = gload_f4(pts + i);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = i0 + u * NT;
      if (i < n) {
        if (ROWS) {
          const double rho = range_res_half + cm.range_res * (double)(int)(key[u] & 0xFFFFFFu);   // radar_filters.cpp:324-330
          const double2 t = cs[row[u]];
          p[u] = make_float4((float)(rho * t.x), (float)(rho * t.y), 0.f, (float)(key[u] >> 24));
          if (job.compensate) {
            const double th = (double)(row[u] + 1) * th_step;                                     // radar_filters.cpp:317
            const double d = rel_time_stamp_known_row((double)p[u].x, (double)p[u].y, th, t.x, t.y, cm.ccw != 0);
            p[u] = compensate_point_small(p[u], d, job.mot);
          }
          gstore<g_f32x4>(pts + i, g_f32x4{p[u].x, p[u].y, p[u].z, p[u].w});
        } else if (job.compensate) {
          p[u] = compensate_point_small(p[u], get_rel_time_stamp((double)p[u].x, (double)p[u].y, cm.ccw != 0), job.mot);
          gstore<g_f32x4>(pts + i, g_f32x4{p[u].x, p[u].y, p[u].z, p[u].w});
        }
        mnx = fminf(mnx, p[u].x); mxx = fmaxf(mxx, p[u].x);
        mny = fminf(mny, p[u].y); mxy = fmaxf(mxy, p[u].y);
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    mnx = fminf(mnx, __shfl_xor(mnx, o)); mxx = fmaxf(mxx, __shfl_xor(mxx, o));
    mny = fminf(mny, __shfl_xor(mny, o)); mxy = fmaxf(mxy, __shfl_xor(mxy, o));
  }
  if (lane == 0) { red_f[0][wave] = mnx; red_f[1][wave] = mxx; red_f[2][wave] = mny; red_f[3][wave] = mxy; }
  __threadfence_block();
  __syncthreads();                                                   // also: pts[] written by this workgroup are visible to it
  mnx = red_f[0][0]; mxx = red_f[1][0]; mny = red_f[2][0]; mxy = red_f[3][0];
  for (int wv = 1; wv < NW; wv++) {
    mnx = fminf(mnx, red_f[0][wv]); mxx = fmaxf(mxx, red_f[1][wv]);
    mny = fminf(mny, red_f[2][wv]); mxy = fmaxf(mxy, red_f[3][wv]);
  }
  // pcl::VoxelGrid::applyFilter: min_b = floor(min * inverse_leaf), div_b = max_b - min_b + 1
  const int min_bx = (int)floorf(mnx * cm.inv_leaf), max_bx = (int)floorf(mxx * cm.inv_leaf);
  const int min_by = (int)floorf(mny * cm.inv_leaf), max_by = (int)floorf(mxy * cm.inv_leaf);
  const long long div_bx = (long long)max_bx - min_bx + 1, div_by = (long long)max_by - min_by + 1;
  if (div_bx * div_by > 0x7fffffffLL || div_by > kMaxGridRows) { done(CFEAR_ERR_CAPACITY); return; }   // non-finite / absurd points
  const int dbx = (int)div_bx, dby = (int)div_by;
  const int ncells = dbx * dby;
  if (ncells > kFastMaxCells) { hand_over(n, 1); return; }
  if (tid == 0) {                                                      // hand-off to surface_sort_kernel
    scr.hdr->route = kRoutePrepped;
    scr.hdr->n = n; scr.hdr->dbx = dbx; scr.hdr->dby = dby; scr.hdr->pad[1] = min_bx; scr.hdr->pad[2] = min_by;
  }
}

constexpr int kTierInFlight = 2;                    // candidates per lane in flight in the lane-group tiers of the moments pass (4: no faster)

// KPER = points per thread the kernel can hold in registers: 32 (scans of up to kMaxPoints = 16 384 points: every k-strongest
// cloud) and 64 (CA-CFAR sweeps beyond that -- cfar.cpp:35-71 puts no bound on the detections; at Pfa 0.01 the false alarms of a
// 400 x 2286-bin sweep alone are ~10 k points -- surface_sort_mixed_kernel below picks the instantiation per scan).  Its 64
// positions per thread are packed two per register (as 64 separate u16 they spilled 316 bytes per lane at the 128 VGPRs two
// workgroups per CU allow; one workgroup per CU with 136 VGPRs was slower still: 0.60 / 0.52 / 0.49 ms per 512 sweeps).
// Without it those scans took the single-kernel path -- milliseconds each: 1.5 ms per frame batch of 512 Kvarntorp-preset
// sweeps of which a quarter exceed 16 384 points.
template <int KPER>
__device__ __forceinline__ void surface_sort_job(const SurfJob* __restrict__ jobs, const SurfCommon& cm) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int NT = kFastThreads, NW = NT / 64;
  constexpr int kPer = KPER;                                          // <= KPER points per thread
  int* red_i = (int*)(smem + kFastLds - 512 + 4 * NW * 4);             // [2][NW]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int job_id = blockIdx.x;
  const SurfScratch scr = scratch_of(cm.scratch + (size_t)job_id * cm.scratch_stride, cm.scratch_cap);
  if (scr.hdr->route != kRoutePrepped) return;                         // failed, empty, handed to the single-kernel path -- or done (second launch)
  if (KPER * NT < kFastMaxPoints && scr.hdr->n > KPER * NT) return;    // (a larger scan than this instantiation holds: the mixed kernel's)
  const SurfJob job = jobs[job_id];
  auto hand_over = [&](int n, int prepared) {                          // to the single-kernel path
    if (tid == 0) {
      scr.hdr->route = kRouteFallback;
      scr.hdr->n = n;
      scr.hdr->pad[0] = prepared;                                      // 1: xyzi already holds the compact, compensated cloud
      const int w = atomicAdd(&cm.fallback[0], 1);
      if (w < cm.n_jobs) cm.fallback[1 + w] = job_id;         // (a stale count can never index past the list)
    }
  };
#ifdef CFEAR_SURF_TIMING
  long long* tstamp = (long long*)((char*)scr.hdr + 64);
#define STAMP(i) do { if (tid == 0) tstamp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
  STAMP(0); STAMP(1);
  const int n = scr.hdr->n, dbx = scr.hdr->dbx, dby = scr.hdr->dby, min_bx = scr.hdr->pad[1], min_by = scr.hdr->pad[2];
  const int ncells = dbx * dby;
  const float4* pts = job.xyzi;
  __syncthreads();                                                     // every thread has read the header before tid 0 may rewrite it
  STAMP(2);
  // ---- (c) occupancy of the voxel grid: ONE BIT per cell + the occupied cells before every 32-cell word.  ord(c) = the
  //      number of occupied cells before cell c is then two LDS reads and a popcount, for any cell, at 0.19 bytes of LDS
  //      per cell (dense u16 counters took 2 bytes: MulRan's 134 x 134 and Kvarntorp's 392 x 392 voxels did not fit). ----
  const int nw32 = (ncells >> 5) + 1;                                 // covers c = ncells
  uint32_t* occ = (uint32_t*)smem;
  unsigned short* wpref = (unsigned short*)(smem + (size_t)nw32 * 4);
  const size_t ord_bytes = ((size_t)nw32 * 6 + 15) & ~(size_t)15;
  if (ord_bytes + 8192 > kFastLds - 512) { hand_over(n, 1); return; }
  __syncthreads();                                                   // rowoff is dead
  for (int w = tid; w < nw32; w += NT) occ[w] = 0u;
  __syncthreads();
  STAMP(9);
  const int jn = (n + NT - 1) / NT;                                   // rounds of NT points (<= kPer)
  auto idx_of = [&](int j) { return tid + j * NT; };
  // the cells of this thread's points (registers); later their voxels, then their positions.  The 64-point instantiation packs
  // two per register (64 separate registers spill at the 128 VGPRs two workgroups per CU allow); all indices are compile-time.
  struct Cells {
    typename std::conditional<KPER == 32, unsigned short, uint32_t>::type w[KPER == 32 ? KPER : KPER / 2];
    __device__ __forceinline__ int get(int j) const { return KPER == 32 ? (int)w[j] : (int)((w[j >> 1] >> ((j & 1) * 16)) & 0xffffu); }
    __device__ __forceinline__ void set(int j, int v) {
      if (KPER == 32) w[j] = (unsigned short)v;
      else w[j >> 1] = (j & 1) ? ((w[j >> 1] & 0x0000ffffu) | ((uint32_t)v << 16)) : ((w[j >> 1] & 0xffff0000u) | ((uint32_t)v & 0xffffu));
    }
  } mycell;
  bool w_small = true;                                                // every weight max(I - 60, 0) an integer in [0, 255]?
#pragma unroll
  for (int j0 = 0; j0 < kPer; j0 += 8) {                              // eight loads in flight per thread
    if (j0 < jn) {
      float4 pp[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = idx_of(j0 + u);
        if (i < n) pp[u] = gload_f4(pts + i);           // (global_load, not flat_load: gload's comment in common.hpp)
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int j = j0 + u, i = idx_of(j);
        mycell.set(j, 0);
        if (i < n) {
          const float4 p = pp[u];
          const int ijk0 = (int)(floorf(p.x * cm.inv_leaf) - (float)min_bx);
          const int ijk1 = (int)(floorf(p.y * cm.inv_leaf) - (float)min_by);
          const int c = ijk0 + ijk1 * dbx;
          mycell.set(j, c & 0xffff);                                    // complete for grids up to 65 536 cells
          atomicOr(&occ[c >> 5], 1u << (c & 31));
          const float wgt = fmaxf(__fsub_rn(p.w, 60.0f), 0.0f);
          w_small = w_small && wgt <= 255.0f && wgt == truncf(wgt);
        }
      }
    }
    if (j0 == 0) STAMP(16);
  }
  STAMP(17);
  const bool wbyte = __syncthreads_and(w_small ? 1 : 0) != 0;
  STAMP(3);
  // ---- (d) occupied cells before every word (exclusive scan of the popcounts); the voxel keys in cell order --------
  const int per = (nw32 + NT - 1) / NT;                               // consecutive words per thread
  const int w0 = min(nw32, tid * per), w1 = min(nw32, w0 + per);
  int to = 0;
  for (int w = w0; w < w1; w++) to += __popc(occ[w]);
  int incl = wave_incl_scan_i32(to);
  if (lane == 63) red_i[wave] = incl;
  __syncthreads();
  int excl = incl - to, V = 0;
  for (int wv = 0; wv < NW; wv++) { if (wv < wave) excl += red_i[wv]; V += red_i[wv]; }
  // LDS budget: occupancy | voxel cursors u16[V + 2] | order u16[n] (later: staged points)
  const size_t vs_off = ord_bytes, vs_bytes = (((size_t)V + 2) * 2 + 15) & ~(size_t)15;
  const size_t ord2_off = vs_off + vs_bytes;
  if (ord2_off + (((size_t)n * 2 + 15) & ~(size_t)15) > kFastLds - 512) { hand_over(n, 1); return; }
  unsigned short* vs16 = (unsigned short*)(smem + vs_off);
  uint32_t* vs32 = (uint32_t*)(smem + vs_off);
  unsigned short* order = (unsigned short*)(smem + ord2_off);
  // Staging area (from ord2_off on, reusing the order array's LDS): points (x, y) 8 bytes + weight 1 or 4 bytes, then per
  // voxel of the slab a list entry (2 bytes) and its grid position (iy << 18 | ix, 4 bytes).  A scan whose points and
  // voxels fit at once (the usual 5 000-point scan) is ONE slab, known here -- its grid positions are written right
  // where the keys are produced instead of being read back from global scratch at the head of the cells phase.
  const size_t avail_all = kFastLds - 512 - ord2_off;
  const int cap_all = (n + 3) & ~3;
  const bool single = V <= 4 * NT && (size_t)cap_all * (wbyte ? 9 : 12) + (size_t)V * 6 + 48 <= avail_all;
  auto vlist_off = [&](int cap_pts, int pb) { return ord2_off + (((size_t)cap_pts * pb + 15) & ~(size_t)15); };
  auto vxy_off = [&](int cap_pts, int pb, int nv) { return vlist_off(cap_pts, pb) + (((size_t)nv * 2 + 15) & ~(size_t)15); };
  {
    uint32_t* vxy1 = (uint32_t*)(smem + vxy_off(cap_all, wbyte ? 9 : 12, V));        // (behind the order array: 9 n > 2 n)
    int run_o = excl;
    for (int w = w0; w < w1; w++) {
      wpref[w] = (unsigned short)run_o;
      uint32_t bits = occ[w];
      while (bits) {
        const int t = __ffs(bits) - 1; bits &= bits - 1;
        const uint32_t key = (uint32_t)(32 * w + t);
        if (single) { const uint32_t iy = key / (uint32_t)dbx; vxy1[run_o] = (iy << 18) | (key - iy * (uint32_t)dbx); }
        scr.vkey[run_o++] = key;
      }
    }
  }
  for (int k = tid; k < (V + 3) / 2; k += NT) vs32[k] = 0u;           // per-voxel point counters
  __syncthreads();
  STAMP(10);
  auto ord = [&](int c) { const int w = c >> 5; return (int)wpref[w] + __popc(occ[w] & ((1u << (c & 31)) - 1u)); };
  // points per voxel; mycell[] holds the VOXEL (ordinal) of each point from here on.  Grids beyond 65 536
  // cells (long-range sensors) read the points a second time (L2): their cell needs 18 bits, and 32 more registers
  // spilled the kernel.
  if (ncells <= 65536) {
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      const int i = idx_of(j);
      if (i < n) {
        const int v = ord(mycell.get(j));
        mycell.set(j, v);
        atomicAdd(&vs32[v >> 1], (v & 1) ? 0x10000u : 1u);
      }
    }
  } else {
#pragma unroll
    for (int j0 = 0; j0 < kPer; j0 += 8) {
      if (j0 < jn) {
        float2 pq[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int i = idx_of(j0 + u);
          if (i < n) { const g_f32x2 t = gload<g_f32x2>(pts + i); pq[u] = make_float2(t.x, t.y); }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int j = j0 + u, i = idx_of(j);
          if (i < n) {
            const int ijk0 = (int)(floorf(pq[u].x * cm.inv_leaf) - (float)min_bx);
            const int ijk1 = (int)(floorf(pq[u].y * cm.inv_leaf) - (float)min_by);
            const int v = ord(ijk0 + ijk1 * dbx);
            mycell.set(j, v);
            atomicAdd(&vs32[v >> 1], (v & 1) ? 0x10000u : 1u);
          }
        }
      }
    }
  }
  __syncthreads();
  {                                                                  // counts -> voxel starts (exclusive scan over the voxels)
    const int perv = (V + NT - 1) / NT;
    const int o0 = min(V, tid * perv), o1 = min(V, o0 + perv);
    int tp = 0;
    for (int o = o0; o < o1; o++) tp += (int)vs16[o];
    incl = wave_incl_scan_i32(tp);
    if (lane == 63) red_i[NW + wave] = incl;
    __syncthreads();
    int run_p = incl - tp;
    for (int wv = 0; wv < wave; wv++) run_p += red_i[NW + wv];
    for (int o = o0; o < o1; o++) { const int cnt = (int)vs16[o]; vs16[o] = (unsigned short)run_p; run_p += cnt; }
  }
  __syncthreads();
  STAMP(4);
  // ---- (e) scatter: the atomic cursor of a voxel ends at the start of the next one ------------------------------
#pragma unroll
  for (int j = 0; j < kPer; j++) {
    const int i = idx_of(j);
    if (i < n) {
      const int v = mycell.get(j);
      const uint32_t old = atomicAdd(&vs32[v >> 1], (v & 1) ? 0x10000u : 1u);
      const int pos = (v & 1) ? (int)(old >> 16) : (int)(old & 0xffffu);
      order[pos] = (unsigned short)i;
    }
  }
  __syncthreads();
  STAMP(5);
  // ---- (f) input order inside every voxel: rank of a point among its voxel's points (runs are a few points long) ---
#pragma unroll
  for (int j = 0; j < kPer; j++) {
    const int i = idx_of(j);
    if (i < n) {
      const int v = mycell.get(j);
      const int s0 = v ? (int)vs16[v - 1] : 0, e0 = (int)vs16[v];
      // The loop is a chain of dependent LDS round trips, not arithmetic: sixteen indices per trip (two aligned 16-byte
      // reads in flight) from the 16-byte boundary below the run on; positions outside the run are masked out, so there
      // are no head or tail loops.  (Reads stay inside the order array's padded allocation.)
      int rank = 0;
      auto below = [&](const uint4 o, const int q) {                    // indices < i among positions q .. q + 7 inside [s0, e0)
        uint32_t bits = 0;
        bits |= ((int)(o.x & 0xffffu) < i) ? 1u : 0u;   bits |= ((int)(o.x >> 16) < i) ? 2u : 0u;
        bits |= ((int)(o.y & 0xffffu) < i) ? 4u : 0u;   bits |= ((int)(o.y >> 16) < i) ? 8u : 0u;
        bits |= ((int)(o.z & 0xffffu) < i) ? 16u : 0u;  bits |= ((int)(o.z >> 16) < i) ? 32u : 0u;
        bits |= ((int)(o.w & 0xffffu) < i) ? 64u : 0u;  bits |= ((int)(o.w >> 16) < i) ? 128u : 0u;
        const int lo = min(max(s0 - q, 0), 8), hi = min(max(e0 - q, 0), 8);
        return __popc(bits & ((1u << hi) - 1u) & ~((1u << lo) - 1u));
      };
      for (int q = s0 & ~7; q < e0; q += 16) {
        const uint4 oa = *(const uint4*)(order + q), ob = *(const uint4*)(order + q + 8);
        rank += below(oa, q) + below(ob, q + 8);
      }
      mycell.set(j, s0 + rank);
    }
  }
  __syncthreads();                                                     // the order array is dead: its LDS becomes the staging area
  STAMP(6);
  // ---- (g) sorted points.  mycell[j] is now the POSITION of the thread's j-th point in the sorted array, so every thread
  //      re-reads its own points (coalesced, L2) and puts them in place: a one-slab scan straight into the LDS staging
  //      area, larger scans into global scratch (L2) from where the slabs are staged.  The staged copies carry the
  //      point's WEIGHT max(I - 60, 0) (pointnormal.cpp:15), not its intensity: float(I) - 60 is exact for I >= 60, so
  //      the fp64 weight of the reference is just its widening. --------------------------------------------------------
  {
    float2* sxy = (float2*)(smem + ord2_off);
    uint8_t* sw = (uint8_t*)(smem + ord2_off + (size_t)cap_all * 8);
    float* swf = (float*)sw;
#pragma unroll
    for (int j0 = 0; j0 < kPer; j0 += 8) {                            // eight loads in flight per thread
      if (j0 < jn) {
        float4 pp[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (idx_of(j0 + u) < n) pp[u] = gload_f4(pts + idx_of(j0 + u));
#pragma unroll
        for (int u = 0; u < 8; u++) {
          if (idx_of(j0 + u) < n) {
            const float4 p = pp[u];
            const int pos = mycell.get(j0 + u);
            const float wgt = fmaxf(__fsub_rn(p.w, 60.0f), 0.0f);
            if (single) {
              sxy[pos] = make_float2(p.x, p.y);
              if (wbyte) sw[pos] = (uint8_t)wgt; else swf[pos] = wgt;
            } else {
              scr.spt[pos] = make_float4(p.x, p.y, wgt, 0.f);
            }
          }
        }
      }
    }
    if (!single) __threadfence_block();
  }
  __syncthreads();
  STAMP(7);
  // ---- (h) neighbourhood moments of every voxel, slab by slab.  A slab = the voxels of grid rows [ya, yb) whose
  //      candidate points (rows ya - 1 .. yb, one contiguous range of the sorted array) fit the staging area; a sparse
  //      scan is one slab, a 16 000-point scan five.  Neighbour runs are O(1) look-ups: the points of cells [c0, c1] of
  //      one grid row are the run [points before c0, points before c1 + 1).
  //      Voxel loads are very uneven (walls: a few voxels hold hundreds of points, half of the voxels see < 6
  //      candidates and cannot become cells), and one lane per voxel in grid order left ~85 % of the lanes idle behind
  //      the heaviest voxel of each wavefront.  So per slab: classify the voxels by candidate count C into power-of-two
  //      buckets (LDS counters), list them heaviest first, and give a voxel 16 lanes (C > 64), 4 lanes (16 < C <= 64)
  //      or one lane; group partial sums are combined with DPP all-reduces.  The fp64 sums of a split voxel are then
  //      added in a different order than the oracle's sequential loop -- differences of a few ulp, inside the 1e-9
  //      tolerance the cell tests carry; the float voxel centroid stays a sequential sum (bit-exact).
  auto pbefore = [&](int c) { const int o = ord(c); return o ? (int)vs16[o - 1] : 0; };      // points in cells < c
  float2* lxy = (float2*)(smem + ord2_off);
  int* bucket = red_i;                                                 // [16] counters, then cursors (red_i is free here)
  constexpr int kSlabVoxels = 4 * NT;                                  // voxels per slab: their (bucket, slot) stay in registers
  // Staged point: (x, y) + weight.  Radar intensities are integers, so max(I - 60, 0) normally fits ONE byte (9 bytes
  // per point); clouds with other intensities keep a float weight.
  auto cells_phase = [&](auto wb_tag) -> bool {                        // compiled for both weight formats
  constexpr bool WB = decltype(wb_tag)::value;
  constexpr int PB = WB ? 9 : 12;
  const size_t avail = kFastLds - 512 - ord2_off;
  const bool wi = cm.weight_intensity != 0;
  for (int ya = 0; ya < dby;) {                                       // block-uniform
    const int P0 = single ? 0 : pbefore(max(ya - 1, 0) * dbx);
    const int vbeg = single ? 0 : ord(ya * dbx);
    // a slab fits when its points (PB bytes each, padded) and its voxel list (2 bytes each) fit the staging area
    auto fits = [&](int yb) {
      const int np = pbefore(min(yb + 1, dby) * dbx) - P0, nv = ord(yb * dbx) - vbeg;
      return nv <= kSlabVoxels && (size_t)((np + 3) & ~3) * PB + (size_t)nv * 6 + 48 <= avail;
    };
    int lo = single ? dby : ya + 1, hi = dby;                         // largest yb in [ya + 1, dby] that fits (one-slab scan: known)
    if (!single && !fits(lo)) { hand_over(n, 1); return false; }      // three grid rows exceed the staging area
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (fits(mid)) lo = mid; else hi = mid - 1;
    }
    const int yb = lo;
    const int P1 = single ? n : pbefore(min(yb + 1, dby) * dbx);
    const int vend = single ? V : ord(yb * dbx);
    const int cap_pts = (P1 - P0 + 3) & ~3;
    uint8_t* lw = (uint8_t*)(smem + ord2_off + (size_t)cap_pts * 8);
    float* lwf = (float*)lw;
    unsigned short* vlist = (unsigned short*)(smem + vlist_off(cap_pts, PB));
    // grid position of the slab's voxels, (iy << 18 | ix), next to the list: runs_of() reads it for every voxel in the
    // classification and again in its tier -- from global scratch that was an exposed L2 round trip (and an integer
    // division) at the head of every round of every tier.  (A one-slab scan's were written in phase (d).)
    const size_t vxy_o = vxy_off(cap_pts, PB, vend - vbeg);
    uint32_t* vxy = (uint32_t*)(smem + vxy_o);
    // float centroids of the split voxels, by list position, for as many as still fit behind the grid positions (the
    // rest go through the voxel's scratch slot: a global round trip at the head of a tier's round)
    const size_t cen_o = (vxy_o + (size_t)(vend - vbeg) * 4 + 7) & ~(size_t)7;
    float2* cen = (float2*)(smem + cen_o);
    const int cen_cap = cen_o < kFastLds - 512 ? (int)((kFastLds - 512 - cen_o) / 8) : 0;
    if (!single)
      for (int i = tid; i < vend - vbeg; i += NT) {
        const uint32_t key = gload<uint32_t>(scr.vkey + vbeg + i);
        const uint32_t iy = key / (uint32_t)dbx;
        vxy[i] = (iy << 18) | (key - iy * (uint32_t)dbx);             // dbx * dby <= 2^18 cells, dby <= 4096 rows
      }
    if (!single)
      for (int i = tid; i < P1 - P0; i += NT) {
        const float4 q = gload_f4(scr.spt + P0 + i);
        lxy[i] = make_float2(q.x, q.y);
        if (WB) lw[i] = (uint8_t)q.z; else lwf[i] = q.z;
      }
    if (tid < 16) bucket[tid] = 0;
    __syncthreads();
    STAMP(11);
    // neighbour runs of voxel v (relative to the staging area) and its own run
    auto runs_of = [&](int v, int* r0, int* r1, int& s, int& e) {
      const uint32_t kxy = vxy[v - vbeg];
      s = (v ? (int)vs16[v - 1] : 0) - P0; e = (int)vs16[v] - P0;
      const int iy = (int)(kxy >> 18), ix = (int)(kxy & 0x3ffffu);
      const int x0 = max(ix - 1, 0), x1 = min(ix + 1, dbx - 1);
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const int yy = iy - 1 + d;
        r0[d] = r1[d] = 0;
        if (yy >= 0 && yy < dby) { r0[d] = pbefore(yy * dbx + x0) - P0; r1[d] = pbefore(yy * dbx + x1 + 1) - P0; }
      }
    };
    // -- classify: bucket b = ceil(log2(C)) in [3, 12] for C >= 6 candidates (an upper bound on the neighbour count)
    int kb[4], slot[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int v = vbeg + tid + u * NT;
      kb[u] = 0; slot[u] = 0;
      if (v < vend) {
        int r0[3], r1[3], sdum, edum;
        runs_of(v, r0, r1, sdum, edum);
        const int C = (r1[0] - r0[0]) + (r1[1] - r0[1]) + (r1[2] - r0[2]);
        if (C >= 6) {
          kb[u] = min(12, 32 - __clz(C - 1));
          slot[u] = atomicAdd(&bucket[kb[u]], 1);
        } else {
          gstore<int32_t>(scr.coff + v, 0);                               // cannot reach 6 neighbours (pointnormal.cpp:291)
        }
      }
    }
    __syncthreads();
    int boff[13];                                                       // list offset of bucket b: heaviest first
    {
      int run = 0;
#pragma unroll
      for (int b = 12; b >= 3; b--) { boff[b] = run; run += bucket[b]; }
      boff[2] = run;                                                    // = list length
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (kb[u]) {
        int off = 0;
#pragma unroll
        for (int b = 3; b <= 12; b++) off = kb[u] == b ? boff[b] : off;
        vlist[off + slot[u]] = (unsigned short)(tid + u * NT);
      }
    }
    const int n16 = boff[6], n4 = boff[4], nlist = boff[2];             // C > 64 | 16 < C <= 64 | 6 <= C <= 16
    __syncthreads();
    STAMP(12);
    // -- centroids of the split voxels: sequential float sums in sorted (= input) order, bit-exact with
    //    pcl::CentroidPoint; one lane per voxel, handed to the voxel's lane group through its scratch slot
    for (int idx = tid; idx < n4; idx += NT) {
      const int v = vbeg + vlist[idx];
      const int s = (v ? (int)vs16[v - 1] : 0) - P0, e = (int)vs16[v] - P0;
      float ax = 0.f, ay = 0.f;
      int p = s;
      for (; p + 3 < e; p += 4) {                                       // four LDS reads in flight; the sums stay sequential
        const float2 q0 = lxy[p], q1 = lxy[p + 1], q2 = lxy[p + 2], q3 = lxy[p + 3];
        ax = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(ax, q0.x), q1.x), q2.x), q3.x);
        ay = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(ay, q0.y), q1.y), q2.y), q3.y);
      }
      for (; p < e; p++) { const float2 q = lxy[p]; ax = __fadd_rn(ax, q.x); ay = __fadd_rn(ay, q.y); }
      const float cnt = (float)(e - s);
      CellMom* cmo = (CellMom*)&scr.tmp[v];
      const float2 c = make_float2(__fdiv_rn(ax, cnt), __fdiv_rn(ay, cnt));
      cmo->cx = c.x; cmo->cy = c.y;
      if (idx < cen_cap) cen[idx] = c;
    }
    __threadfence_block();
    __syncthreads();
    STAMP(13);
    // -- moments
    auto tier = [&](auto g_tag, const int lbeg, const int lend) {
      constexpr int G = decltype(g_tag)::value;
      const int sub = tid & (G - 1);
      for (int idx = lbeg + tid / G; idx < lend; idx += NT / G) {
        const int v = vbeg + vlist[idx];
        int r0[3], r1[3], s, e;
        runs_of(v, r0, r1, s, e);
        CellMom* cmo = (CellMom*)&scr.tmp[v];
        float2 c;
        if (G == 1) {
          float ax = 0.f, ay = 0.f;
          for (int p = s; p < e; p++) { const float2 q = lxy[p]; ax = __fadd_rn(ax, q.x); ay = __fadd_rn(ay, q.y); }
          const float cnt = (float)(e - s);
          c = make_float2(__fdiv_rn(ax, cnt), __fdiv_rn(ay, cnt));
        } else {
          c = idx < cen_cap ? cen[idx] : make_float2(gload<float>(&cmo->cx), gload<float>(&cmo->cy));
        }
        const double cx = (double)c.x, cy = (double)c.y;
        Moments mo{0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        // The three neighbour runs are walked as ONE sequence (no idle lanes and no loop tail at the end of every run),
        // several candidates per lane in flight.  One lane per voxel keeps the oracle's order of additions.
        const int n0 = r1[0] - r0[0], n01 = n0 + (r1[1] - r0[1]), C = n01 + (r1[2] - r0[2]);
        auto at = [&](int k) { return k < n0 ? r0[0] + k : (k < n01 ? r0[1] + (k - n0) : r0[2] + (k - n01)); };
        auto wgt_at = [&](int p) { return wi ? (WB ? (float)lw[p] : lwf[p]) : 0.f; };
        constexpr int U = G == 1 ? 4 : kTierInFlight;
        for (int k = sub; k < C; k += U * G) {
          int pp[U]; float2 qq[U]; float ww[U];
#pragma unroll
          for (int u = 0; u < U; u++) pp[u] = at(min(k + u * G, C - 1));
#pragma unroll
          for (int u = 0; u < U; u++) { qq[u] = lxy[pp[u]]; ww[u] = wgt_at(pp[u]); }
#pragma unroll
          for (int u = 0; u < U; u++)
            if (u == 0 || k + u * G < C) accum_point(mo, c, cx, cy, qq[u].x, qq[u].y, ww[u], cm.r2, wi);
        }
        if (G > 1) {
          mo.cnt = group_sum_i32<G>(mo.cnt);
          mo.s0 = group_sum_f64<G>(mo.s0); mo.s1x = group_sum_f64<G>(mo.s1x); mo.s1y = group_sum_f64<G>(mo.s1y);
          mo.sxx = group_sum_f64<G>(mo.sxx); mo.sxy = group_sum_f64<G>(mo.sxy); mo.syy = group_sum_f64<G>(mo.syy);
        }
        if (sub == 0) {
          if (mo.cnt >= 6) {                                            // (fewer: no cell, pointnormal.cpp:291 -- only the count is read)
            cmo->s0 = mo.s0; cmo->s1x = mo.s1x; cmo->s1y = mo.s1y; cmo->sxx = mo.sxx; cmo->sxy = mo.sxy; cmo->syy = mo.syy;
            if (G == 1) { cmo->cx = c.x; cmo->cy = c.y; }
            cmo->cnt = mo.cnt;
          }
          gstore<int32_t>(scr.coff + v, mo.cnt);
        }
      }
    };
    tier(std::integral_constant<int, 16>{}, 0, n16);
    STAMP(14);
    tier(std::integral_constant<int, 4>{}, n16, n4);
    STAMP(15);
    tier(std::integral_constant<int, 1>{}, n4, nlist);
    __syncthreads();                                                   // the staging area is reused by the next slab
    ya = yb;
  }
  return true;
  };
  if (!(wbyte ? cells_phase(std::true_type{}) : cells_phase(std::false_type{}))) return;
  STAMP(8);
  if (tid == 0) {
    scr.hdr->route = kRouteFast;                                       // cells computed; compaction + x-sort pending
    scr.hdr->n = n; scr.hdr->V = V; scr.hdr->dbx = dbx; scr.hdr->dby = dby;
  }
}

// The regular launch: every scan fits 32 points per thread (k-strongest clouds).
__global__ __launch_bounds__(kFastThreads, 4) void surface_sort_kernel(const SurfJob* __restrict__ jobs, const SurfCommon cm) {
  surface_sort_job<32>(jobs, cm);
}
// Batches that may hold scans beyond 16 384 points (CA-CFAR): ONE launch in which a workgroup takes the instantiation its scan
// needs.  (Two launches, the second leaving the first's scans untouched, were two workgroup latencies behind each other for a
// batch of one workgroup per slot: 0.246 + 0.257 ms per 512 Kvarntorp-preset sweeps.)
__global__ __launch_bounds__(kFastThreads, 4) void surface_sort_mixed_kernel(const SurfJob* __restrict__ jobs, const SurfCommon cm) {
  const SurfScratch scr = scratch_of(cm.scratch + (size_t)blockIdx.x * cm.scratch_stride, cm.scratch_cap);
  if (scr.hdr->n > 32 * kFastThreads) surface_sort_job<64>(jobs, cm);  // (block-uniform)
  else surface_sort_job<32>(jobs, cm);
}

constexpr int kFinishThreads = 256;

__global__ __launch_bounds__(kFinishThreads) void surface_finish_kernel(const SurfJob* __restrict__ jobs, const SurfCommon cm) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];       // sort keys: next_pow2(cell capacity) x 8 bytes
  __shared__ int red_i2[2][kFinishThreads / 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int job_id = blockIdx.x;
  const SurfScratch scr = scratch_of(cm.scratch + (size_t)job_id * cm.scratch_stride, cm.scratch_cap);
  if (job_id == 0 && tid == 0) cm.fallback[0] = 0;                     // the hand-over list has been drained: ready for the next launch
  if (scr.hdr->route != kRouteFast) return;
  const SurfJob job = jobs[job_id];
  const int V = scr.hdr->V;
  int run = 0;                                                         // valid cells so far: every thread keeps the total
  // Three of four voxels hold fewer than six neighbours and cannot become cells (pointnormal.cpp:291).  Their 64 bytes of
  // moments were most of what this kernel pulled through the memory system (it is bound by that traffic, not by the eigen
  // decompositions): the neighbour counts sit in an array of their own (4 bytes per voxel), read one round ahead, and a
  // voxel's moments (four 16-byte global loads: gload's comment in common.hpp) are only requested when its count says so.
  int cnt_next = tid < V ? gload<int32_t>(scr.coff + tid) : 0;
  for (int v0 = 0, rnd = 0; v0 < V; v0 += kFinishThreads, rnd ^= 1) { // cells + compaction in voxel order (= PCL's output order)
    const int v = v0 + tid;
    const int cnt = cnt_next;
    cnt_next = v + kFinishThreads < V ? gload<int32_t>(scr.coff + v + kFinishThreads) : 0;
    TmpCell t;
    int f = 0;
    if (v < V && cnt >= 6) {
      const char* src = (const char*)&scr.tmp[v];
      const g_f64x2 a = gload<g_f64x2>(src), b = gload<g_f64x2>(src + 16), c = gload<g_f64x2>(src + 32);
      const g_f32x2 d = gload<g_f32x2>(src + 48);
      const Moments mo{cnt, a.x, a.y, b.x, b.y, c.x, c.y};
      f = finish_cell(mo, (double)d.x, (double)d.y, cm.origin[0], cm.origin[1], t);
    }
    const int inc = wave_incl_scan_i32(f);
    if (lane == 63) red_i2[rnd][wave] = inc;                           // double-buffered: ONE barrier per round
    __syncthreads();
    int off = run + inc - f;
    for (int wv = 0; wv < kFinishThreads / 64; wv++) { if (wv < wave) off += red_i2[rnd][wv]; run += red_i2[rnd][wv]; }
    if (f && off < job.out.cap) {
      gstore<g_f32x2>(job.out.mean_f + off, g_f32x2{(float)t.mean[0], (float)t.mean[1]});  // pointnormal.cpp:154-157
      gstore<g_f64x2>(job.out.mean + off, g_f64x2{t.mean[0], t.mean[1]});
      gstore<g_f64x2>(job.out.normal + off, g_f64x2{t.normal[0], t.normal[1]});
      gstore<g_f64x4>(job.out.cov + off, g_f64x4{t.cov[0], t.cov[1], t.cov[2], t.cov[3]});
      gstore<double>(job.out.scale + off, t.scale);
      gstore<double>(job.out.avg_intensity + off, t.avg_intensity);
      gstore<g_f64x2>(job.out.lambda + off, g_f64x2{t.lmin, t.lmax});
      gstore<int32_t>(job.out.nsamples + off, t.nsamples);
    }
  }
  __threadfence_block();
  __syncthreads();
  const int total = run;
  const int cap = min(job.out.cap, cm.finish_keys);
  grid_cells_block(job.out, min(total, cap), smem);
  if (tid == 0) {
    *job.out.n_cells = total <= cap ? total : cap;
    cm.status[job_id] = total <= cap ? CFEAR_OK : CFEAR_ERR_CAPACITY;
    if (cm.ncells_out) cm.ncells_out[job_id] = total <= cap ? total : cap;
  }
}

// raw cells <-> SoA slab
__global__ void cells_to_slab_kernel(const cfear_cell* cells, int n, ScanView v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *v.n_cells = n;
  if (i >= n) return;
  const cfear_cell c = cells[i];
  v.mean_f[i] = make_float2((float)c.mean[0], (float)c.mean[1]);
  v.mean[i] = make_double2(c.mean[0], c.mean[1]);
  v.normal[i] = make_double2(c.normal[0], c.normal[1]);
  v.cov[i] = make_double4(c.cov[0], c.cov[1], c.cov[2], c.cov[3]);
  v.scale[i] = c.scale;
  v.avg_intensity[i] = c.avg_intensity;
  v.lambda[i] = make_double2(c.lambda_min, c.lambda_max);
  v.nsamples[i] = c.nsamples;
}
__global__ void slab_to_cells_kernel(ScanView v, int n, cfear_cell* cells) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  cfear_cell c;
  c.mean[0] = v.mean[i].x; c.mean[1] = v.mean[i].y;
  c.normal[0] = v.normal[i].x; c.normal[1] = v.normal[i].y;
  c.cov[0] = v.cov[i].x; c.cov[1] = v.cov[i].y; c.cov[2] = v.cov[i].z; c.cov[3] = v.cov[i].w;
  c.scale = v.scale[i];
  c.avg_intensity = v.avg_intensity[i];
  c.lambda_min = v.lambda[i].x; c.lambda_max = v.lambda[i].y;
  c.nsamples = v.nsamples[i];
  c.pad = 0;
  cells[i] = c;
}

}  // namespace

size_t cfear_surface_lds_bytes() { return kLdsTotal; }
size_t cfear_surface_scratch_bytes(int cap_points) { return scratch_bytes_per_scan(std::max(cap_points, kMaxPoints)); }
int cfear_surface_max_points() { return kMaxPoints; }

// Launches the surface-point kernel for n_jobs scans.  d_jobs: device array of SurfJob-compatible
// records built by cfear_surface_fill_job; d_status: device int32 [n_jobs].
struct cfear_surf_job_pod { unsigned char bytes[sizeof(SurfJob)]; };
size_t cfear_surface_job_bytes() { return sizeof(SurfJob); }

void cfear_surface_fill_job_rows(void* dst, float* d_xyzi, int32_t* d_n_out, const uint32_t* d_row_pts, const int32_t* d_row_cnt,
                                 int rows, int k, int compensate, const double mot[3], const ScanView& out) {
  cfear_surface_fill_job(dst, d_xyzi, nullptr, 0, compensate, mot, out);
  SurfJob j;
  memcpy(&j, dst, sizeof(j));
  j.row_pts = d_row_pts; j.row_cnt = d_row_cnt; j.n_out = d_n_out; j.rows = rows; j.k = k;
  memcpy(dst, &j, sizeof(j));
}

void cfear_surface_fill_job(void* dst, float* d_xyzi, const int32_t* d_n, int32_t n_host, int compensate,
                            const double mot[3], const ScanView& out) {
  SurfJob j;
  j.row_pts = nullptr; j.row_cnt = nullptr; j.n_out = nullptr; j.rows = 0; j.k = 0;
  j.xyzi = (float4*)d_xyzi;
  j.n_ptr = d_n;
  j.n_host = n_host;
  j.compensate = compensate;
  j.mot[0] = mot ? mot[0] : 0.0; j.mot[1] = mot ? mot[1] : 0.0; j.mot[2] = mot ? mot[2] : 0.0;
  j.out = out;
  memcpy(dst, &j, sizeof(j));
}

int cfear_surface_launch(cfear_ctx* ctx, const void* d_jobs, int n_jobs, const cfear_feature_params* par,
                         char* d_scratch, int32_t* d_status, int32_t* d_ncells_out, int max_cell_cap, int cap_points,
                         const cfear_surface_polar* polar) {
  if (par->radius <= 0.f || !(par->downsample_factor > 0.0))
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "radius / downsample_factor must be > 0");
  SurfCommon cm;
  cm.radius = par->radius;
  cm.leaf = (float)(par->radius / par->downsample_factor);                    // pointnormal.cpp:279
  cm.inv_leaf = 1.0f / cm.leaf;                                               // Array4f::Ones() / leaf
  cm.r2 = (float)((double)par->radius * (double)par->radius);                 // KdTreeFLANN::radiusSearch
  cm.reach = (int)std::ceil((double)par->radius / (double)cm.leaf);
  if (cm.reach < 1) cm.reach = 1;
  if (2 * cm.reach + 1 > 16)
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "downsample_factor too large (reach %d)", cm.reach);
  cm.weight_intensity = par->weight_intensity;
  cm.ccw = par->ccw;
  cm.origin[0] = par->origin[0]; cm.origin[1] = par->origin[1];
  cm.scratch = d_scratch;
  cm.scratch_cap = std::max(cap_points, kMaxPoints);
  cm.scratch_stride = scratch_bytes_per_scan(cm.scratch_cap);
  cm.status = d_status;
  cm.ncells_out = d_ncells_out;
  cm.cos_t = polar ? polar->cos_t : nullptr;
  cm.sin_t = polar ? polar->sin_t : nullptr;
  cm.range_res = polar ? polar->range_res : 0.0;
  cm.range_off = polar ? polar->range_off : 0.0;
  cm.n_jobs = n_jobs;
  cm.fast_ok = cm.reach == 1 ? 1 : 0;
  // work list of the scans the fast pipeline hands to the single-kernel path: count + job ids
  // (its counter is zeroed by surface_finish_kernel for the next launch; a fresh allocation is zeroed here)
  // (the counter is zeroed by surface_finish_kernel for the next launch; it is zeroed HERE when the workspace was (re)allocated
  // -- hipFree + hipMalloc may hand back the same address -- or when the previous launch sequence did not reach its finish
  // kernel: ctx->surf_list_dirty stays set from before the first launch until after the last one was enqueued without error)
  const size_t ws_bytes_before = ctx->ws[11].bytes;
  cm.fallback = (int32_t*)cfear_workspace(ctx, 11, ((size_t)n_jobs + 16) * 4);
  if (!cm.fallback) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  if (ctx->ws[11].bytes != ws_bytes_before || ctx->surf_list_dirty)
    CFEAR_HIP_CHECK(ctx, hipMemsetAsync(cm.fallback, 0, 4, ctx->stream));
  ctx->surf_list_dirty = true;
  // per launch: the attribute is per device, and contexts on other threads / devices share this code
  { const int rc_lds = cfear_allow_lds(ctx, (const void*)surface_points_kernel, cfear_surface_lds_bytes()); if (rc_lds != CFEAR_OK) return rc_lds; }
  { const int rc_lds = cfear_allow_lds(ctx, (const void*)surface_sort_kernel, kFastLds); if (rc_lds != CFEAR_OK) return rc_lds; }
  const bool big_scans = cap_points > kMaxPoints;            // scans beyond 16 384 points may come (CA-CFAR): the kernel with both instantiations
  if (big_scans) { const int rc_lds = cfear_allow_lds(ctx, (const void*)surface_sort_mixed_kernel, kFastLds); if (rc_lds != CFEAR_OK) return rc_lds; }
  cm.finish_keys = std::min(max_cell_cap, kMaxPoints);     // cells a scan may hold (the matcher's 16-bit tables address 65 535)
  const size_t finish_lds = kScanGridLds;                    // the matcher grid's counters
  cm.finish_lds = (uint32_t)finish_lds;
  {
    // rows mode: rowoff i32[rows + 1] | row of every point u16[n] | (cos, sin) f64[rows]
    size_t prep_lds = 0;
    if (polar) {
      const size_t np = (size_t)std::min<long long>((long long)polar->rows * polar->k, kFastMaxPoints);
      prep_lds = (((size_t)(polar->rows + 1) * 4 + 15) & ~(size_t)15) + ((np * 2 + 15) & ~(size_t)15) + (size_t)polar->rows * 16;
      if (prep_lds > 150 * 1024) return cfear_set_error(ctx, CFEAR_ERR_CAPACITY, "%d azimuths exceed the row tables", polar->rows);
      if (prep_lds > 64 * 1024)
        { const int rc_lds = cfear_allow_lds(ctx, (const void*)surface_prep_kernel<true>, prep_lds); if (rc_lds != CFEAR_OK) return rc_lds; }
    }
    ProfScope ps(ctx, "surface_prep");
    if (polar) hipLaunchKernelGGL(surface_prep_kernel<true>, dim3(n_jobs), dim3(kFastThreads), prep_lds, ctx->stream, (const SurfJob*)d_jobs, cm);
    else hipLaunchKernelGGL(surface_prep_kernel<false>, dim3(n_jobs), dim3(kFastThreads), 0, ctx->stream, (const SurfJob*)d_jobs, cm);
  }
  {
    ProfScope ps(ctx, "surface_sort");
    if (big_scans) hipLaunchKernelGGL(surface_sort_mixed_kernel, dim3(n_jobs), dim3(kFastThreads), kFastLds, ctx->stream, (const SurfJob*)d_jobs, cm);
    else hipLaunchKernelGGL(surface_sort_kernel, dim3(n_jobs), dim3(kFastThreads), kFastLds, ctx->stream, (const SurfJob*)d_jobs, cm);
  }
  {
    ProfScope ps(ctx, "surface_points");       // the single-kernel path drains the hand-over list (usually empty)
    hipLaunchKernelGGL(surface_points_kernel, dim3(std::min(n_jobs, 256)), dim3(kSurfThreads), cfear_surface_lds_bytes(), ctx->stream,
                       (const SurfJob*)d_jobs, cm);
  }
  {
    ProfScope ps(ctx, "surface_finish");
    hipLaunchKernelGGL(surface_finish_kernel, dim3(n_jobs), dim3(kFinishThreads), finish_lds, ctx->stream, (const SurfJob*)d_jobs, cm);
  }
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  ctx->surf_list_dirty = false;                              // the finish kernel is enqueued: it leaves the counter at zero
#ifdef CFEAR_SURF_TIMING
  {                                            // debug build only: phase split of surface_sort_kernel, averaged over the jobs
    static int calls = 0;
    if (++calls % 40 == 0) {
      (void)hipStreamSynchronize(ctx->stream);
      {
        int32_t nfb = -1;
        (void)hipMemcpy(&nfb, cm.fallback, 4, hipMemcpyDeviceToHost);   // (already reset by the finish kernel: read the routes instead)
        int routes[4] = {0, 0, 0, 0}, prepared = 0;
        for (int j = 0; j < n_jobs; j++) {
          SurfHdr h;
          (void)hipMemcpy(&h, d_scratch + (size_t)j * cm.scratch_stride, sizeof(h), hipMemcpyDeviceToHost);
          if (h.route >= 0 && h.route < 4) routes[h.route]++;
          if (h.route == kRouteFallback) prepared += h.pad[0];
        }
        fprintf(stderr, "surface routes of %d jobs: fast %d  fallback %d (prepared %d)  done %d  prepped %d\n", n_jobs, routes[0], routes[1], prepared, routes[2], routes[3]);
      }
      double acc[9] = {0}, sub[5] = {0}, cel[6] = {0};
      const int m = std::min(n_jobs, 256);
      for (int j = 0; j < m; j++) {
        long long t[24];
        (void)hipMemcpy(t, d_scratch + (size_t)(j * (n_jobs / m)) * cm.scratch_stride + 64, sizeof(t), hipMemcpyDeviceToHost);   // spread over the launch
        for (int q = 1; q < 9; q++) acc[q] += (double)(t[q] - t[q - 1]);
        sub[0] += (double)(t[9] - t[2]); sub[1] += (double)(t[10] - t[3]);
        sub[2] += (double)(t[16] - t[9]); sub[3] += (double)(t[17] - t[16]); sub[4] += (double)(t[3] - t[17]);
        cel[0] += (double)(t[11] - t[7]); cel[1] += (double)(t[12] - t[11]); cel[2] += (double)(t[13] - t[12]);
        cel[3] += (double)(t[14] - t[13]); cel[4] += (double)(t[15] - t[14]); cel[5] += (double)(t[8] - t[15]);
      }
      fprintf(stderr, "  cells split (last slab, tid 0): stage %.0f  classify+list %.0f  centroids %.0f  16-lane %.0f  4-lane %.0f  1-lane %.0f\n",
              cel[0] / m, cel[1] / m, cel[2] / m, cel[3] / m, cel[4] / m, cel[5] / m);
      fprintf(stderr, "  of hist: clear the bitmap %.0f, first 8 points per thread %.0f, the rest %.0f, barrier %.0f; of scan: word scan + voxel keys %.0f\n",
              sub[0] / m, sub[2] / m, sub[3] / m, sub[4] / m, sub[1] / m);
      fprintf(stderr, "surface_sort phases (cycles, mean of %d jobs): count %.0f  comp+bbox %.0f  hist %.0f  scan %.0f  scatter %.0f  order %.0f  spt %.0f  cells %.0f\n",
              m, acc[1] / m, acc[2] / m, acc[3] / m, acc[4] / m, acc[5] / m, acc[6] / m, acc[7] / m, acc[8] / m);
    }
  }
#endif
  return CFEAR_OK;
}

// d_xyzi [n_clouds][cloud_stride_points][4], d_n [n_clouds], d_mot [n_clouds][3] all device
int cfear_compensate_batch_device(cfear_ctx* ctx, float* d_xyzi, size_t cloud_stride_points, const int32_t* d_n, const double* d_mot,
                                  int n_clouds, int max_points, int ccw) {
  if (n_clouds <= 0) return CFEAR_OK;
  ProfScope ps(ctx, "compensate_peaks");
  const int chunks = std::max(1, std::min(16, (max_points + 255) / 256));
  hipLaunchKernelGGL(compensate_batch_kernel, dim3(chunks, n_clouds), dim3(256), 0, ctx->stream, (float4*)d_xyzi, cloud_stride_points,
                     d_n, d_mot, ccw);
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}

extern "C" int cfear_compensate(cfear_ctx* ctx, float* xyzi, int32_t n, const double mot[3], int32_t ccw) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!xyzi || !mot || n < 0) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  if (n == 0) return CFEAR_OK;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const bool dev = cfear_is_device_ptr(xyzi);
  float* d = xyzi;
  if (!dev) {
    d = (float*)cfear_workspace(ctx, 4, (size_t)n * 16);
    if (!d) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(d, xyzi, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
  }
  {
    ProfScope ps(ctx, "compensate");
    hipLaunchKernelGGL(compensate_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (float4*)d, n, mot[0], mot[1], mot[2], ccw);
  }
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  if (!dev) {
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(xyzi, d, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
    CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return CFEAR_OK;
}

extern "C" int cfear_scan_create(cfear_ctx* ctx, float* xyzi, int32_t n, const cfear_feature_params* par, cfear_scan** out) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!par || !out || (!xyzi && n > 0)) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  if (n <= 0) return cfear_set_error(ctx, CFEAR_ERR_EMPTY_CLOUD, "error, cloud empty");   // pointnormal.cpp:72-75
  if (n > kScanCreateMaxPoints) return cfear_set_error(ctx, CFEAR_ERR_CAPACITY, "n = %d > %d points", n, kScanCreateMaxPoints);
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const bool dev = cfear_is_device_ptr(xyzi);
  float* d = xyzi;
  if (!dev) {
    d = (float*)cfear_workspace(ctx, 4, (size_t)n * 16);
    if (!d) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(d, xyzi, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
  }
  cfear_scan* s = nullptr;
  int rc = cfear_scan_alloc(ctx, n, &s);          // at most one cell per point
  if (rc != CFEAR_OK) return rc;
  char* ws = (char*)cfear_workspace(ctx, 5, cfear_surface_scratch_bytes(n) + 1024);
  if (!ws) { cfear_scan_destroy(s); return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed"); }
  char* d_job = ws;                                // job record + status in front of the scratch
  int32_t* d_status = (int32_t*)(ws + 512);
  char* d_scratch = ws + 1024;
  unsigned char hjob[sizeof(SurfJob)];
  cfear_surface_fill_job(hjob, d, nullptr, n, par->compensate, par->mot, s->view);
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(d_job, hjob, sizeof(SurfJob), hipMemcpyHostToDevice, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));     // hjob is on the stack
  rc = cfear_surface_launch(ctx, d_job, 1, par, d_scratch, d_status, nullptr, n, n);
  if (rc != CFEAR_OK) { cfear_scan_destroy(s); return rc; }
  int32_t hst[2] = {0, 0};
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(&hst[0], d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(&hst[1], s->view.n_cells, 4, hipMemcpyDeviceToHost, ctx->stream));
  if (!dev && par->compensate)
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(xyzi, d, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (hst[0] != CFEAR_OK) {
    cfear_scan_destroy(s);
    return cfear_set_error(ctx, hst[0], "surface point extraction failed: %s", cfear_status_string(hst[0]));
  }
  s->n_cells_host = hst[1];
  *out = s;
  return CFEAR_OK;
}

extern "C" int cfear_scan_from_cells(cfear_ctx* ctx, const cfear_cell* cells, int32_t n_cells, cfear_scan** out) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!out || n_cells < 0 || (!cells && n_cells > 0)) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  cfear_scan* s = nullptr;
  int rc = cfear_scan_alloc(ctx, n_cells > 0 ? n_cells : 1, &s);
  if (rc != CFEAR_OK) return rc;
  cfear_cell* d = (cfear_cell*)cfear_workspace(ctx, 4, (size_t)(n_cells > 0 ? n_cells : 1) * sizeof(cfear_cell));
  if (!d) { cfear_scan_destroy(s); return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed"); }
  if (n_cells > 0)
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(d, cells, (size_t)n_cells * sizeof(cfear_cell), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(cells_to_slab_kernel, dim3((n_cells + 255) / 256 + 1), dim3(256), 0, ctx->stream, d, n_cells, s->view);
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  if (n_cells > kMaxPoints) { cfear_scan_destroy(s); return cfear_set_error(ctx, CFEAR_ERR_CAPACITY, "more than %d cells", kMaxPoints); }
  hipLaunchKernelGGL(scan_grid_kernel, dim3(1), dim3(kSurfThreads), 0, ctx->stream, s->view);
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));     // caller's host array may go away
  s->n_cells_host = n_cells;
  *out = s;
  return CFEAR_OK;
}

extern "C" int cfear_scan_get_cells(const cfear_scan* scan, cfear_cell* out_host, int32_t cap) {
  if (!scan || !out_host) return CFEAR_ERR_INVALID_ARGUMENT;
  cfear_ctx* ctx = scan->ctx;
  const int n = cfear_scan_size(scan);
  if (n < 0) return n;
  if (n > cap) return cfear_set_error(ctx, CFEAR_ERR_CAPACITY, "%d cells > cap %d", n, cap);
  if (n == 0) return 0;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  cfear_cell* d = (cfear_cell*)cfear_workspace(ctx, 4, (size_t)n * sizeof(cfear_cell));
  if (!d) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  hipLaunchKernelGGL(slab_to_cells_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, scan->view, n, d);
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(out_host, d, (size_t)n * sizeof(cfear_cell), hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return n;
}

// ---- MapPointNormal::GetClosestIdx for a batch of queries --------------------------------------------------
// pointnormal.cpp:238-254: exact 1-NN of float(p) among the float cell means (KdTreeFLANN<PointXY>::nearestKSearch,
// L2_Simple in float, lowest index on ties -- the rule the matcher uses), returned iff its squared
// float distance is < d * d (compared in double).  One thread per query; the means are swept from LDS in tiles.
namespace {
__global__ __launch_bounds__(256) void closest_idx_kernel(ScanView v, int n_cells, const double2* __restrict__ q, int nq,
                                                          double d2, int32_t* __restrict__ idx) {
  __shared__ float2 tile[1024];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float qx = 0.f, qy = 0.f;
  if (i < nq) { const double2 p = q[i]; qx = (float)p.x; qy = (float)p.y; }       // pnt.x = p(0): double -> float
  int best = -1;
  float bestd = 0.f;
  for (int t0 = 0; t0 < n_cells; t0 += 1024) {
    const int m = min(1024, n_cells - t0);
    __syncthreads();
    for (int j = threadIdx.x; j < m; j += 256) tile[j] = v.mean_f[t0 + j];
    __syncthreads();
    for (int j = 0; j < m; j++) {
      const float2 c = tile[j];
      const float dx = __fsub_rn(qx, c.x), dy = __fsub_rn(qy, c.y);
      const float dd = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
      if (best < 0 || dd < bestd) { best = t0 + j; bestd = dd; }
    }
  }
  if (i < nq) idx[i] = (best >= 0 && (double)bestd < d2) ? best : -1;
}
}  // namespace

extern "C" int cfear_scan_closest_idx(const cfear_scan* scan, const double* queries_xy, int32_t n_queries, double d,
                                      int32_t* idx) {
  if (!scan) return CFEAR_ERR_INVALID_ARGUMENT;
  cfear_ctx* ctx = scan->ctx;
  if (!queries_xy || !idx || n_queries < 0) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  if (n_queries == 0) return CFEAR_OK;
  const int n = cfear_scan_size(scan);
  if (n < 0) return n;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const bool dev = cfear_is_device_ptr(queries_xy);
  if (dev != cfear_is_device_ptr(idx))
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "queries and idx must both be host or both be device memory");
  const double2* dq = (const double2*)queries_xy;
  int32_t* di = idx;
  if (!dev) {
    char* ws = (char*)cfear_workspace(ctx, 4, (size_t)n_queries * 20 + 256);
    if (!ws) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(ws, queries_xy, (size_t)n_queries * 16, hipMemcpyHostToDevice, ctx->stream));
    dq = (const double2*)ws;
    di = (int32_t*)(ws + (((size_t)n_queries * 16 + 255) & ~(size_t)255));
  }
  hipLaunchKernelGGL(closest_idx_kernel, dim3((n_queries + 255) / 256), dim3(256), 0, ctx->stream, scan->view, n, dq, n_queries,
                     d * d, di);
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  if (!dev) {
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(idx, di, (size_t)n_queries * 4, hipMemcpyDeviceToHost, ctx->stream));
    CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return CFEAR_OK;
}

