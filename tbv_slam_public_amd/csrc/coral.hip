// coral.hip -- CorAl alignment quality of two radar peak clouds on gfx950.
//
// Replaces CorAlRadarQuality (coral_alignment_quality/src/alignment_checker/AlignmentQuality.cpp:8-230)
// as TBV calls it for loop-closure verification and classifier training
// (alignmentinterface.cpp:296-347, 437-456: kstrongStructuredRadar scans from the stored peak clouds,
// radius 1.0, ent_cfg = any -> ComputeEntropy, output_overlap = true):
//   PoseScan::GetCloudCopy + pcl3dto2d            ScanType.cpp:211-215, Utils.cpp:200-212
//   GetNearby (two pcl::KdTreeFLANN radius searches)  AlignmentQuality.cpp:8-29
//   Covariance / ComputeEntropy                   :30-53, :80-98
//   per-point loops and aggregation               :132-204
// The ent_cfg = kl branch (ComputeKLDiv, :54-78) is never selected by TBV and is not built.
//
// Kernel design: ONE 1024-thread workgroup per (ref, src, Toffset) job, one launch per batch.  The two
// kd-trees are replaced by ONE sort-based uniform grid over the merged cloud (cell >= radius, so every
// neighbour of a query lies in the 3 x 3 cells around it; gridsort.hpp): each merged point is a query
// once and accumulates, in a single pass over the contiguous runs of the sorted array, the fp64 moments
// of its source-cloud and reference-cloud neighbours separately -- the joint neighbourhood is their sum,
// so the reference's two radius searches + three matrix copies per point become one sweep.  The float
// squared distance test is FLANN's L2_Simple (`dist < r^2`, strict).  Per-point entropies are written by
// original index and reduced in index order (fixed tree), so results do not depend on scheduling.
// Working set per job: a few hundred KB in LDS + L2; nothing here is HBM-bound.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <map>
#include <vector>

#include "common.hpp"
#include "gridsort.hpp"

namespace {

constexpr int kCoralThreads = kGridSortThreads;
constexpr int kCoralMaxPoints = kGridSortMaxPoints;      // merged points per job
constexpr int kCoralMaxGridRows = 4096;
constexpr int kCoralPerThread = kCoralMaxPoints / kCoralThreads;
constexpr size_t kCoralRowbegOff = (size_t)kCoralMaxPoints * 8 + 16;
constexpr size_t kCoralSmallOff = (kCoralRowbegOff + (size_t)(kCoralMaxGridRows + 1) * 4 + 15) / 16 * 16;
constexpr size_t kCoralLdsTotal = kCoralSmallOff + 1024;

// (struct CoralJob: common.hpp -- verify.hip's kernels write job records too)

struct CoralCommon {
  double radius;
  float r2, inv_cell;
  int32_t weight_res_intensity;
  int32_t cap;                        // merged-point capacity of the per-job scratch
  char* scratch;
  size_t scratch_stride;
  cfear_coral_result* results;
  double* per_point;                  // nullable: [n_jobs][cap][3] joint_res, sep_res, valid
};

__host__ __device__ inline size_t coral_scratch_bytes(int cap) {
  // sorted points float4 | joint_res f64 | sep_res f64 | weight f64 (0 = invalid) | valid i32 | work list i32
  return ((size_t)cap * (16 + 8 + 8 + 8 + 4 + 4) + 255) / 256 * 256;
}

struct Aff2d { double l0, l1, l2, l3, t0, t1; };
__device__ __forceinline__ Aff2d aff_xyt(const double* p) {
  double s, c;
  sincos(p[2], &s, &c);
  return Aff2d{c, -s, s, c, p[0], p[1]};
}
__device__ __forceinline__ Aff2d aff_compose(const Aff2d& a, const Aff2d& b) {     // Eigen Transform * Transform
  Aff2d r;
  r.l0 = a.l0 * b.l0 + a.l1 * b.l2; r.l1 = a.l0 * b.l1 + a.l1 * b.l3;
  r.l2 = a.l2 * b.l0 + a.l3 * b.l2; r.l3 = a.l2 * b.l1 + a.l3 * b.l3;
  r.t0 = a.l0 * b.t0 + a.l1 * b.t1 + a.t0;
  r.t1 = a.l2 * b.t0 + a.l3 * b.t1 + a.t1;
  return r;
}
// pcl::transformPointCloud<PointXYZI, double> (PCL 1.10 common/impl/transforms.hpp): float(((t0 x + t1 y) + t2 z) + t3)
__device__ __forceinline__ float2 tf_point(const float4 p, const Aff2d& T) {
  const double x = (double)p.x, y = (double)p.y, z = (double)p.z;
  return make_float2((float)(((T.l0 * x + T.l1 * y) + 0.0 * z) + T.t0), (float)(((T.l2 * x + T.l3 * y) + 0.0 * z) + T.t1));
}

__device__ __forceinline__ int lower_bound_u32(const uint32_t* a, int lo, int hi, uint32_t key) {
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}
__device__ __forceinline__ int upper_bound_u32(const uint32_t* a, int lo, int hi, uint32_t key) {
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] <= key) lo = mid + 1; else hi = mid; }
  return lo;
}

struct Moments { int n; double sx, sy, sxx, sxy, syy; };
// all-reduce over aligned groups of G lanes (G = 4: quad, G = 16: DPP row); every lane ends with the same sum
template <int G> __device__ __forceinline__ int group_sum_i32(int v) {
  if (G >= 2) v += __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
  if (G >= 4) v += __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  if (G == 16) {
    v += __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false);          // row_half_mirror
    v += __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false);          // row_mirror
  }
  return v;
}
template <int G> __device__ __forceinline__ double group_sum_f64(double v) {
  if (G >= 2) v += dpp_f64<0xB1>(v);
  if (G >= 4) v += dpp_f64<0x4E>(v);
  if (G == 16) { v += dpp_f64<0x141>(v); v += dpp_f64<0x140>(v); }
  return v;
}
typedef float v4f __attribute__((ext_vector_type(4)));
#define CFEAR_LDS __attribute__((address_space(3)))

// CorAlRadarQuality::Covariance (:30-53) from the moments of d = p - q about the query q: the mean shift
// cancels in the covariance, |d| < radius keeps the one-pass form accurate to a few ulp.
__device__ __forceinline__ bool cov_from_moments(const Moments& m, double& c00, double& c01, double& c11) {
  if (m.n <= 2) return false;                                                  // x.rows() <= 2
  const double n = (double)m.n;
  const double mx = m.sx / n, my = m.sy / n;
  const double den = (double)(float)m.n - 1.0;                                 // `float n = x.rows()`; cov = covSum*1.0/(n-1.0)
  c00 = (m.sxx - n * mx * mx) * 1.0 / den;
  c01 = (m.sxy - n * mx * my) * 1.0 / den;
  c11 = (m.syy - n * my * my) * 1.0 / den;
  return true;
}

__device__ __forceinline__ void fail_job(const CoralCommon& cm, int status) {
  if (threadIdx.x == 0) {
    cfear_coral_result& r = cm.results[blockIdx.x];
    r.joint = r.sep = r.overlap = 0.0; r.valid = 0; r.count_valid = 0; r.status = status; r.pad = 0;
  }
}

constexpr int kCoralSweepLanes = 1;                           // lanes per point of the work list in pass B (2: equal, 4: slower)

__global__ __launch_bounds__(kCoralThreads) void coral_kernel(const CoralJob* __restrict__ jobs, const CoralCommon cm) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  float (*red_f)[16] = (float (*)[16])(smem + kCoralSmallOff);          // [4][16]
  int* red_i = (int*)(smem + kCoralSmallOff + 256);                     // [16]
  double* red_d = (double*)(smem + kCoralSmallOff + 384);               // [16][3] + counts
  int* red_c = (int*)(smem + kCoralSmallOff + 384 + 16 * 3 * 8);        // [16]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const CoralJob job = jobs[blockIdx.x];
  const int n_src = job.n_src_ptr ? *job.n_src_ptr : job.n_src;
  const int n_ref = job.n_ref_ptr ? *job.n_ref_ptr : job.n_ref;
  const int n = n_src + n_ref;
  if (n_src <= 0 || n_ref <= 0) { fail_job(cm, CFEAR_ERR_EMPTY_CLOUD); return; }         // assert(size() > 0) (:118)
  if (n > cm.cap || n > kCoralMaxPoints) { fail_job(cm, CFEAR_ERR_CAPACITY); return; }
  char* scr = cm.scratch + (size_t)blockIdx.x * cm.scratch_stride;
  float4* spt = (float4*)scr;                                  // sorted merged points (x, y, intensity, original index)
  double* jres = (double*)(spt + cm.cap);
  double* sres = jres + cm.cap;
  double* wres = sres + cm.cap;
  int32_t* vres = (int32_t*)(wres + cm.cap);
  int32_t* work = vres + cm.cap;                               // sorted positions of the points that overlap the other cloud

#ifdef CFEAR_CORAL_TIMING
  long long tq[8]; tq[0] = __builtin_readcyclecounter();
#define CORAL_T(k) tq[k] = __builtin_readcyclecounter()
#else
#define CORAL_T(k)
#endif
  const Aff2d Tref = aff_xyt(job.ref_pose);
  const Aff2d Tsrc = aff_compose(aff_xyt(job.src_pose), aff_xyt(job.offset));            // src->GetAffine() * Toffset (:101)
  auto point = [&](int i) -> float2 {                          // merged index: source points first (:132, :155)
    return i < n_src ? tf_point(gload_f4(job.src + i), Tsrc) : tf_point(gload_f4(job.ref + (i - n_src)), Tref);   // (global_load: gload's comment in common.hpp)
  };
  // ---- 1. bounding box of the merged cloud ------------------------------------------------------
  float mnx = FLT_MAX, mny = FLT_MAX, mxx = -FLT_MAX, mxy = -FLT_MAX;
  for (int i = tid; i < n; i += kCoralThreads) {
    const float2 p = point(i);
    mnx = fminf(mnx, p.x); mxx = fmaxf(mxx, p.x);
    mny = fminf(mny, p.y); mxy = fmaxf(mxy, p.y);
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
  const int min_bx = (int)floorf(mnx * cm.inv_cell), max_bx = (int)floorf(mxx * cm.inv_cell);
  const int min_by = (int)floorf(mny * cm.inv_cell), max_by = (int)floorf(mxy * cm.inv_cell);
  const long long div_bx = (long long)max_bx - min_bx + 1, div_by = (long long)max_by - min_by + 1;
  if (!(mnx == mnx) || !(mny == mny) || div_bx * div_by > 0x7fffffffLL || div_by > kCoralMaxGridRows) {
    fail_job(cm, CFEAR_ERR_CAPACITY);
    return;
  }
  const int dbx = (int)div_bx, dby = (int)div_by;
  auto cell_xy = [&](float2 p, int& ix, int& iy) {
    ix = (int)(floorf(p.x * cm.inv_cell) - (float)min_bx);
    iy = (int)(floorf(p.y * cm.inv_cell) - (float)min_by);
  };
  CORAL_T(1);
  // ---- 2. sort by (cell, index) ------------------------------------------------------------------
  unsigned long long* keys = (unsigned long long*)smem;
  int npad = grid_sort_rows_block(smem, n, dbx, dby, (uint32_t*)(smem + kCoralRowbegOff), red_i, red_c, 512,
                                  [&](int i, int& ix, int& iy) { cell_xy(point(i), ix, iy); });
  if (npad == 0)                                       // crowded grid row or a large cloud: generic block sort
    npad = grid_sort_block(smem, n, (long long)dbx * dby, red_i, [&](int i) {
      int ix, iy;
      cell_xy(point(i), ix, iy);
      return (uint32_t)(ix + iy * dbx);
    });
  CORAL_T(2);
  // ---- 3. sorted points -> scratch; cell table (key, start) -> LDS ---------------------------------
  const int per = npad / kCoralThreads;                 // 1..16 consecutive sorted elements per thread
  unsigned long long mine[kCoralPerThread];
  const unsigned prev_cell = (tid * per > 0) ? (unsigned)(keys[tid * per - 1] >> 32) : 0xFFFFFFFFu;
  int heads = 0;
#pragma unroll
  for (int q = 0; q < kCoralPerThread; q++) {
    const int e = tid * per + q;
    mine[q] = (q < per && e < n) ? keys[e] : ~0ull;
  }
  {
    unsigned pv = prev_cell;
#pragma unroll
    for (int q = 0; q < kCoralPerThread; q++) {
      const int e = tid * per + q;
      if (q < per && e < n) {
        const unsigned vx = (unsigned)(mine[q] >> 32);
        heads += (e == 0 || vx != pv);
        pv = vx;
      }
    }
  }
  const int incl = wave_incl_scan_i32(heads);
  if (lane == 63) red_i[wave] = incl;
  __syncthreads();                                      // also: every thread has read its keys
  int voff = incl - heads;
  for (int wv = 0; wv < wave; wv++) voff += red_i[wv];
  int V = 0;
  for (int wv = 0; wv < 16; wv++) V += red_i[wv];
  const size_t Vp = ((size_t)V + 4) & ~(size_t)3;
  uint32_t* cell_key = (uint32_t*)smem;                 // [V]
  int32_t* cell_start = (int32_t*)(smem + Vp * 4);      // [V + 1]
  int32_t* rowbeg = (int32_t*)(smem + kCoralRowbegOff); // [dby + 1]
  // The sorted points follow the cell table in LDS when they fit (the usual case: a few thousand peaks), so the
  // neighbour sweep of step 4 reads them at LDS latency; larger clouds keep them in the per-job global scratch.
  const size_t spt_off = (Vp * 4 + ((size_t)V + 1) * 4 + 15) & ~(size_t)15;
  const bool spt_in_lds = spt_off + (size_t)n * 16 <= kCoralRowbegOff;
  if (spt_in_lds) spt = (float4*)(smem + spt_off);
  {
    unsigned pv = prev_cell;
    int ord = voff;
#pragma unroll
    for (int q = 0; q < kCoralPerThread; q++) {
      const int e = tid * per + q;
      if (q < per && e < n) {
        const unsigned vx = (unsigned)(mine[q] >> 32);
        const int idx = (int)(unsigned)(mine[q] & 0xFFFFFFFFu);
        if (e == 0 || vx != pv) { cell_key[ord] = vx; cell_start[ord] = e; ord++; }
        pv = vx;
        const float2 p = point(idx);
        const float inten = idx < n_src ? gload<float>(&job.src[idx].w) : gload<float>(&job.ref[idx - n_src].w);
        spt[e] = make_float4(p.x, p.y, inten, __int_as_float(idx));
      }
    }
  }
  if (tid == 0) cell_start[V] = n;
  __threadfence_block();
  __syncthreads();
  // ---- 3b. O(1) cell look-ups: ONE BIT per grid cell + the occupied cells before every 32-cell word (the map
  //      surface_sort_kernel uses): the points before cell c = cell_start[wpref[c / 32] + popcount(occ[c / 32] below c)] --
  //      two LDS reads and one dependent read instead of two binary searches over the row's cells (ten dependent reads) per
  //      grid row and point.  Kept behind the sorted points when it fits the LDS (grids up to ~3 x 10^5 cells for the usual
  //      peak clouds); otherwise the binary searches below. ---------------------------------------------------------
  const long long ncells_ll = (long long)dbx * dby;
  const size_t occ_off = spt_in_lds ? ((spt_off + (size_t)n * 16 + 15) & ~(size_t)15) : spt_off;
  const long long nw32_ll = (ncells_ll >> 5) + 1;
  const bool bitmap = occ_off + (size_t)nw32_ll * 6 + 16 <= kCoralRowbegOff;
  uint32_t* occ = (uint32_t*)(smem + occ_off);
  const int nw32 = bitmap ? (int)nw32_ll : 0;
  unsigned short* wpref = (unsigned short*)(occ + nw32);
  if (!bitmap) {                                        // first cell of every grid row, for the binary searches
    for (int y = tid; y <= dby; y += kCoralThreads)
      rowbeg[y] = lower_bound_u32(cell_key, 0, V, (uint32_t)((long long)y * dbx));
    __syncthreads();
  }
  if (bitmap) {
    for (int w = tid; w < nw32; w += kCoralThreads) occ[w] = 0u;
    __syncthreads();
    for (int o = tid; o < V; o += kCoralThreads) { const uint32_t c = cell_key[o]; atomicOr(&occ[c >> 5], 1u << (c & 31)); }
    __syncthreads();
    const int perw = (nw32 + kCoralThreads - 1) / kCoralThreads;
    const int w0 = min(nw32, tid * perw), w1 = min(nw32, w0 + perw);
    int to = 0;
    for (int w = w0; w < w1; w++) to += __popc(occ[w]);
    const int inclw = wave_incl_scan_i32(to);
    if (lane == 63) red_c[wave] = inclw;
    __syncthreads();
    int runw = inclw - to;
    for (int wv = 0; wv < wave; wv++) runw += red_c[wv];
    for (int w = w0; w < w1; w++) { wpref[w] = (unsigned short)runw; runw += __popc(occ[w]); }
    __syncthreads();
  }
  auto pbefore = [&](int c) -> int {                            // points in cells < c, 0 <= c <= ncells
    const int w = c >> 5;
    return cell_start[(int)wpref[w] + __popc(occ[w] & ((1u << (c & 31)) - 1u))];
  };
  // the three candidate runs of a query in cell (ix, iy): rows iy - 1 .. iy + 1, columns ix - 1 .. ix + 1 (empty outside the grid)
  auto runs_of = [&](int ix, int iy, int* r0, int* r1) {
    const int x0 = max(ix - 1, 0), x1 = min(ix + 1, dbx - 1);
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const int yy = iy - 1 + d;
      const bool in = yy >= 0 && yy < dby;
      const int cy = in ? yy : 0;
      const int a = pbefore(cy * dbx + x0), b = pbefore(cy * dbx + x1 + 1);
      r0[d] = in ? a : 0; r1[d] = in ? b : 0;
    }
  };
  CORAL_T(3);
  int* n_work = red_i;                                          // LDS counter (red_i is free after the sort)
  if (tid == 0) *n_work = 0;
  __syncthreads();
  // Pass A (cheap, every point): does the point have ANY neighbour of the other cloud within the radius (overlap_req_ = 1,
  // :138, :160)?  Float distance tests only, first hit ends the search.  Points without one are final (100, 100,
  // invalid); the others go to a work list.  Pass B (expensive, work list only): fp64 moments and entropies.
  auto find_overlap_bm = [&](auto* SP) {
    for (int e0 = 0; e0 < n; e0 += kCoralThreads) {
      const int e = e0 + tid;
      bool hit = false;
      if (e < n) {
        const v4f q = SP[e];
        const int idx = __float_as_int(q.w);
        const bool q_is_src = idx < n_src;
        int ix, iy, r0[3], r1[3];
        cell_xy(make_float2(q.x, q.y), ix, iy);
        runs_of(ix, iy, r0, r1);
        auto test = [&](const v4f c) {
          const float dxf = __fsub_rn(q.x, c.x), dyf = __fsub_rn(q.y, c.y);
          const float d2 = __fadd_rn(__fmul_rn(dxf, dxf), __fmul_rn(dyf, dyf));
          return (d2 < cm.r2) && ((__float_as_int(c.w) < n_src) != q_is_src);
        };
        // the three runs as ONE sequence, the point's own row first (the nearest returns of the other cloud usually share
        // it); four independent loads per exit test
        const int n0 = r1[1] - r0[1], n01 = n0 + (r1[0] - r0[0]), C = n01 + (r1[2] - r0[2]);
        auto at = [&](int j) { return j < n0 ? r0[1] + j : (j < n01 ? r0[0] + (j - n0) : r0[2] + (j - n01)); };
        for (int j = 0; j < C && !hit; j += 4) {
          const v4f c0 = SP[at(j)], c1 = SP[at(min(j + 1, C - 1))], c2 = SP[at(min(j + 2, C - 1))], c3 = SP[at(min(j + 3, C - 1))];
          hit = ((int)test(c0) | (int)test(c1) | (int)test(c2) | (int)test(c3)) != 0;
        }
        if (!hit) { gstore<double>(jres + idx, 100.0); gstore<double>(sres + idx, 100.0); gstore<double>(wres + idx, 0.0); gstore<int32_t>(vres + idx, 0); }
      }
      const unsigned long long m = __ballot(hit);                // wave-aggregated append
      int base = 0;
      if (lane == 0 && m) base = atomicAdd(n_work, __popcll(m));
      base = __shfl(base, 0);
      if (hit) gstore<int32_t>(work + base + __popcll(m & ((1ull << lane) - 1ull)), e);
    }
  };
  // Pass B: one lane per point of the work list.  (Lane groups for the heavy neighbourhoods -- sixteen lanes per point with
  // more than 48 candidates, four above 12, DPP all-reduces of the partial moments -- were measured in round 4: the jobs
  // that take 2.5 x the sweep time of the rest are not held up by a few heavy lanes, they simply visit more candidates; the
  // classification pass cost more than the balance gained: 55 k cycles against 50 k per job.)
  auto sweep_bm = [&](auto* SP, const int W) {
    const int* work2 = work;
    const int n16 = 0, n4 = 0;
    auto tier = [&](auto g_tag, const int lbeg, const int lend) {
      constexpr int G = decltype(g_tag)::value;
      const int sub = tid & (G - 1);
      for (int k0 = lbeg; k0 < lend; k0 += kCoralThreads / G) {
        const int k = k0 + tid / G;
        const bool act = k < lend;                               // (whole groups stay in the loop for the DPP sums)
        const int e = act ? gload<int32_t>(work2 + k) : 0;
        const v4f q = SP[e];
        const int idx = __float_as_int(q.w);
        const bool q_is_src = idx < n_src;
        int ix, iy, r0[3], r1[3];
        cell_xy(make_float2(q.x, q.y), ix, iy);
        runs_of(ix, iy, r0, r1);
        Moments ms{0, 0, 0, 0, 0, 0}, mr{0, 0, 0, 0, 0, 0};
        const double qx = (double)q.x, qy = (double)q.y;
        auto visit = [&](const v4f c) {
          const float dxf = __fsub_rn(q.x, c.x), dyf = __fsub_rn(q.y, c.y);
          const float d2 = __fadd_rn(__fmul_rn(dxf, dxf), __fmul_rn(dyf, dyf));  // FLANN L2_Simple
          if (d2 < cm.r2) {                                                       // RadiusResultSet: strict <
            const double dx = (double)c.x - qx, dy = (double)c.y - qy;
            if (__float_as_int(c.w) < n_src) {
              ms.n++; ms.sx += dx; ms.sy += dy; ms.sxx += dx * dx; ms.sxy += dx * dy; ms.syy += dy * dy;
            } else {
              mr.n++; mr.sx += dx; mr.sy += dy; mr.sxx += dx * dx; mr.sxy += dx * dy; mr.syy += dy * dy;
            }
          }
        };
        if (act) {                                               // the three runs as ONE sequence, two loads in flight
          const int n0 = r1[0] - r0[0], n01 = n0 + (r1[1] - r0[1]), C = n01 + (r1[2] - r0[2]);
          auto at = [&](int j) { return j < n0 ? r0[0] + j : (j < n01 ? r0[1] + (j - n0) : r0[2] + (j - n01)); };
          for (int j = sub; j < C; j += 2 * G) {
            const bool two = j + G < C;
            const v4f c0 = SP[at(j)], c1 = SP[at(two ? j + G : j)];
            visit(c0);
            if (two) visit(c1);
          }
        }
        if (G > 1) {
          ms.n = group_sum_i32<G>(ms.n); mr.n = group_sum_i32<G>(mr.n);
          ms.sx = group_sum_f64<G>(ms.sx); ms.sy = group_sum_f64<G>(ms.sy); ms.sxx = group_sum_f64<G>(ms.sxx);
          ms.sxy = group_sum_f64<G>(ms.sxy); ms.syy = group_sum_f64<G>(ms.syy);
          mr.sx = group_sum_f64<G>(mr.sx); mr.sy = group_sum_f64<G>(mr.sy); mr.sxx = group_sum_f64<G>(mr.sxx);
          mr.sxy = group_sum_f64<G>(mr.sxy); mr.syy = group_sum_f64<G>(mr.syy);
        }
        if (act && sub == 0) {
          double jr = 100.0, sr = 100.0, w = 0.0;
          int valid = 0;
          const Moments& own = q_is_src ? ms : mr;
          const Moments& other = q_is_src ? mr : ms;
          if (other.n >= 1) {                                                         // overlap_req_ = 1 (:138, :160)
            const Moments mj{ms.n + mr.n, ms.sx + mr.sx, ms.sy + mr.sy, ms.sxx + mr.sxx, ms.sxy + mr.sxy, ms.syy + mr.syy};
            double s00, s01, s11, j00, j01, j11;
            if (cov_from_moments(own, s00, s01, s11) && cov_from_moments(mj, j00, j01, j11)) {
              const double det_j = j00 * j11 - j01 * j01;                             // ComputeEntropy (:80-98)
              const double det_s = s00 * s11 - s01 * s01;
              if (!(isnan(det_s) || isnan(det_j))) {
                const double sep_entropy = 1.0 / 2.0 * log(2.0 * M_PI * exp(1.0) * det_s + 0.00000001);
                const double joint_entropy = 1.0 / 2.0 * log(2.0 * M_PI * exp(1.0) * det_j + 0.00000001);
                if (!(isnan(sep_entropy) || isnan(joint_entropy))) {
                  w = cm.weight_res_intensity ? (double)q.z : 1.0;                    // :180
                  jr = w * joint_entropy; sr = w * sep_entropy; valid = 1;
                }
              }
            }
          }
          gstore<double>(jres + idx, jr); gstore<double>(sres + idx, sr); gstore<double>(wres + idx, valid ? w : 0.0); gstore<int32_t>(vres + idx, valid);
        }
      }
    };
    (void)n16;
    tier(std::integral_constant<int, kCoralSweepLanes>{}, n4, W);
  };
  if (bitmap) {
    if (spt_in_lds) find_overlap_bm((CFEAR_LDS const v4f*)spt);
    else find_overlap_bm((const v4f*)spt);
    __threadfence_block();
    __syncthreads();
    const int Wb = *n_work;
    CORAL_T(4);
    __syncthreads();                                            // (red_c / red_i are reused by the sweep's counters)
    if (spt_in_lds) sweep_bm((CFEAR_LDS const v4f*)spt, Wb);
    else sweep_bm((const v4f*)spt, Wb);
  } else {
  // ---- 4. moments of the source / reference neighbours of every point -> entropies -----------------------------
  // Instantiated per address space of the sorted points (ds_read_b128 when they sit in LDS); candidates are fetched
  // two at a time so the second load is in flight while the first is tested.
  // Pass A (cheap, every point): does the point have ANY neighbour of the other cloud within the radius
  // (overlap_req_ = 1, :138, :160)?  Float distance tests only, first hit ends the search.  Points without one are
  // final (100, 100, invalid); the others go to a work list.  Pass B (expensive, work list only): fp64 moments and
  // entropies -- typically a quarter to a half of the points, spread evenly over the workgroup.
  auto find_overlap = [&](auto* SP) {
    for (int e0 = 0; e0 < n; e0 += kCoralThreads) {
      const int e = e0 + tid;
      bool hit = false;
      if (e < n) {
        const v4f q = SP[e];
        const int idx = __float_as_int(q.w);
        const bool q_is_src = idx < n_src;
        int ix, iy;
        cell_xy(make_float2(q.x, q.y), ix, iy);
        const int x0 = max(ix - 1, 0), x1 = min(ix + 1, dbx - 1);
        for (int yy = max(iy - 1, 0); yy <= min(iy + 1, dby - 1) && !hit; yy++) {
          const uint32_t klo = (uint32_t)(yy * dbx + x0), khi = (uint32_t)(yy * dbx + x1);
          const int a = lower_bound_u32(cell_key, rowbeg[yy], rowbeg[yy + 1], klo);
          const int b = upper_bound_u32(cell_key, rowbeg[yy], rowbeg[yy + 1], khi);
          const int p1 = cell_start[b];
          auto test = [&](const v4f c) {
            const float dxf = __fsub_rn(q.x, c.x), dyf = __fsub_rn(q.y, c.y);
            const float d2 = __fadd_rn(__fmul_rn(dxf, dxf), __fmul_rn(dyf, dyf));
            return (d2 < cm.r2) && ((__float_as_int(c.w) < n_src) != q_is_src);
          };
          int p = cell_start[a];
          for (; p + 3 < p1 && !hit; p += 4) {                  // four independent loads per exit test
            const v4f c0 = SP[p], c1 = SP[p + 1], c2 = SP[p + 2], c3 = SP[p + 3];
            hit = ((int)test(c0) | (int)test(c1) | (int)test(c2) | (int)test(c3)) != 0;
          }
          for (; p < p1 && !hit; p++) hit = test(SP[p]);
        }
        if (!hit) { jres[idx] = 100.0; sres[idx] = 100.0; wres[idx] = 0.0; vres[idx] = 0; }
      }
      const unsigned long long m = __ballot(hit);                // wave-aggregated append
      int base = 0;
      if (lane == 0 && m) base = atomicAdd(n_work, __popcll(m));
      base = __shfl(base, 0);
      if (hit) work[base + __popcll(m & ((1ull << lane) - 1ull))] = e;
    }
  };
  if (spt_in_lds) find_overlap((CFEAR_LDS const v4f*)spt);
  else find_overlap((const v4f*)spt);
  __threadfence_block();
  __syncthreads();
  const int W = *n_work;
  CORAL_T(4);
  auto sweep = [&](auto* SP) {
    for (int k = tid; k < W; k += kCoralThreads) {
      const int e = work[k];
      const v4f q = SP[e];
      const int idx = __float_as_int(q.w);
      const bool q_is_src = idx < n_src;
      int ix, iy;
      cell_xy(make_float2(q.x, q.y), ix, iy);
      const int x0 = max(ix - 1, 0), x1 = min(ix + 1, dbx - 1);
      Moments ms{0, 0, 0, 0, 0, 0}, mr{0, 0, 0, 0, 0, 0};
      const double qx = (double)q.x, qy = (double)q.y;
      auto visit = [&](const v4f c) {
        const float dxf = __fsub_rn(q.x, c.x), dyf = __fsub_rn(q.y, c.y);
        const float d2 = __fadd_rn(__fmul_rn(dxf, dxf), __fmul_rn(dyf, dyf));  // FLANN L2_Simple
        if (d2 < cm.r2) {                                                       // RadiusResultSet: strict <
          const double dx = (double)c.x - qx, dy = (double)c.y - qy;
          if (__float_as_int(c.w) < n_src) {
            ms.n++; ms.sx += dx; ms.sy += dy; ms.sxx += dx * dx; ms.sxy += dx * dy; ms.syy += dy * dy;
          } else {
            mr.n++; mr.sx += dx; mr.sy += dy; mr.sxx += dx * dx; mr.sxy += dx * dy; mr.syy += dy * dy;
          }
        }
      };
      for (int yy = max(iy - 1, 0); yy <= min(iy + 1, dby - 1); yy++) {
        const uint32_t klo = (uint32_t)(yy * dbx + x0), khi = (uint32_t)(yy * dbx + x1);
        const int a = lower_bound_u32(cell_key, rowbeg[yy], rowbeg[yy + 1], klo);
        const int b = upper_bound_u32(cell_key, rowbeg[yy], rowbeg[yy + 1], khi);
        const int p1 = cell_start[b];
        int p = cell_start[a];
        for (; p + 1 < p1; p += 2) {
          const v4f c0 = SP[p], c1 = SP[p + 1];
          visit(c0);
          visit(c1);
        }
        if (p < p1) visit(SP[p]);
      }
      double jr = 100.0, sr = 100.0, w = 0.0;
      int valid = 0;
      const Moments& own = q_is_src ? ms : mr;
      const Moments& other = q_is_src ? mr : ms;
      if (other.n >= 1) {                                                         // overlap_req_ = 1 (:138, :160)
        const Moments mj{ms.n + mr.n, ms.sx + mr.sx, ms.sy + mr.sy, ms.sxx + mr.sxx, ms.sxy + mr.sxy, ms.syy + mr.syy};
        double s00, s01, s11, j00, j01, j11;
        if (cov_from_moments(own, s00, s01, s11) && cov_from_moments(mj, j00, j01, j11)) {
          const double det_j = j00 * j11 - j01 * j01;                             // ComputeEntropy (:80-98)
          const double det_s = s00 * s11 - s01 * s01;
          if (!(isnan(det_s) || isnan(det_j))) {
            const double sep_entropy = 1.0 / 2.0 * log(2.0 * M_PI * exp(1.0) * det_s + 0.00000001);
            const double joint_entropy = 1.0 / 2.0 * log(2.0 * M_PI * exp(1.0) * det_j + 0.00000001);
            if (!(isnan(sep_entropy) || isnan(joint_entropy))) {
              w = cm.weight_res_intensity ? (double)q.z : 1.0;                    // :180
              jr = w * joint_entropy; sr = w * sep_entropy; valid = 1;
            }
          }
        }
      }
      jres[idx] = jr; sres[idx] = sr; wres[idx] = valid ? w : 0.0; vres[idx] = valid;
    }
  };
  if (spt_in_lds) sweep((CFEAR_LDS const v4f*)spt);
  else sweep((const v4f*)spt);
  }
  __threadfence_block();
  __syncthreads();
  CORAL_T(5);
  // ---- 5. aggregation in index order (:178-204): contiguous chunk per thread, then a fixed tree ------
  {
    const int chunk = (n + kCoralThreads - 1) / kCoralThreads;
    double sj = 0.0, ss = 0.0, sw = 0.0;
    int cnt = 0;
    for (int i = tid * chunk; i < min(n, (tid + 1) * chunk); i++)
      if (gload<int32_t>(vres + i)) { sw += gload<double>(wres + i); sj += gload<double>(jres + i); ss += gload<double>(sres + i); cnt++; }
    sj = wave_sum_lane63_f64(sj); ss = wave_sum_lane63_f64(ss); sw = wave_sum_lane63_f64(sw);
    cnt = wave_sum_i32(cnt);
    if (lane == 63) { red_d[wave * 3] = sj; red_d[wave * 3 + 1] = ss; red_d[wave * 3 + 2] = sw; red_c[wave] = cnt; }
    __syncthreads();
    if (tid == 0) {
      double joint = 0.0, sep = 0.0, w_sum = 0.0;
      int count_valid = 0;
      for (int wv = 0; wv < 16; wv++) { joint += red_d[wv * 3]; sep += red_d[wv * 3 + 1]; w_sum += red_d[wv * 3 + 2]; count_valid += red_c[wv]; }
      if (count_valid > 0) { sep /= w_sum; joint /= w_sum; }
      const double overlap = count_valid / ((double)n);
      cfear_coral_result& r = cm.results[blockIdx.x];
      r.joint = joint; r.sep = sep; r.overlap = overlap;                        // quality_ = {joint_, sep_, overlap_}
      r.valid = overlap < 0.1 ? 0 : 1;                                          // :197-204
      r.count_valid = count_valid; r.status = CFEAR_OK; r.pad = 0;
    }
  }
#ifdef CFEAR_CORAL_TIMING
  CORAL_T(6);
  if (tid == 0 && (blockIdx.x % 997) == 0)
    printf("coral job %d n %d W %d: bbox %lld | sort %lld | table %lld | overlap %lld | sweep %lld | reduce %lld | total %lld\n", blockIdx.x, n, *n_work,
           tq[1] - tq[0], tq[2] - tq[1], tq[3] - tq[2], tq[4] - tq[3], tq[5] - tq[4], tq[6] - tq[5], tq[6] - tq[0]);
#endif
  if (cm.per_point) {
    double* pp = cm.per_point + (size_t)blockIdx.x * cm.cap * 3;
    for (int i = tid; i < n; i += kCoralThreads) { pp[3 * i] = jres[i]; pp[3 * i + 1] = sres[i]; pp[3 * i + 2] = (double)vres[i]; }
  }
}

}  // namespace

int cfear_coral_max_points() { return kCoralMaxPoints; }

int cfear_coral_launch_device(cfear_ctx* ctx, const CoralJob* d_jobs, int n_jobs, int cap, const cfear_coral_params* par,
                              cfear_coral_result* d_results) {
  if (!(par->radius > 0.0)) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "radius must be > 0");
  if (n_jobs <= 0) return CFEAR_OK;
  if (cap > kCoralMaxPoints) return cfear_set_error(ctx, CFEAR_ERR_CAPACITY, "a job's %d points exceed %d", cap, kCoralMaxPoints);
  CoralCommon cm;
  cm.radius = par->radius;
  cm.r2 = (float)(par->radius * par->radius);            // radiusSearch passes float(radius * radius) to FLANN
  cm.inv_cell = (float)(1.0 / (par->radius * 1.0001));
  cm.weight_res_intensity = par->weight_res_intensity;
  cm.cap = std::max(cap, 1);
  cm.scratch_stride = coral_scratch_bytes(cm.cap);
  cm.per_point = nullptr;
  const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_jobs, ((size_t)1 << 30) / cm.scratch_stride));
  char* scr = (char*)cfear_workspace(ctx, 10, cm.scratch_stride * (size_t)chunk);
  if (!scr) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  cm.scratch = scr;
  { const int rc_lds = cfear_allow_lds(ctx, (const void*)coral_kernel, 160 * 1024); if (rc_lds != CFEAR_OK) return rc_lds; }
  ProfScope ps(ctx, "coral_quality");
  for (int j0 = 0; j0 < n_jobs; j0 += chunk) {
    const int nj = std::min(chunk, n_jobs - j0);
    cm.results = d_results + j0;
    hipLaunchKernelGGL(coral_kernel, dim3(nj), dim3(kCoralThreads), kCoralLdsTotal, ctx->stream, d_jobs + j0, cm);
  }
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}

extern "C" void cfear_coral_params_default(cfear_coral_params* p) {
  if (!p) return;
  p->radius = 1.0;                    // alignmentinterface.cpp:444
  p->weight_res_intensity = 0;
  p->pad = 0;
}

// The batch in two halves, so that a caller with host work of its own (verify.hip) can do it while the kernel runs:
// cfear_coral_enqueue stages the clouds, uploads the jobs and launches; cfear_coral_collect reads the results back and
// synchronises.  `pend` carries what must outlive the launch.
static int coral_enqueue_impl(cfear_ctx* ctx, const cfear_coral_job* jobs, int32_t n_jobs, const cfear_coral_params* par, bool want_per_point,
                              CoralPending& pend);

// The enqueue half may leave asynchronous copies FROM pageable host memory in flight (pend.host_jobs, the caller's peak
// clouds): when it fails behind the first of them the stream is drained before the error returns, so that whoever destroys
// `pend` or the clouds next does not pull them from under a copy (collect() is the only other place that waits).
int cfear_coral_enqueue(cfear_ctx* ctx, const cfear_coral_job* jobs, int32_t n_jobs, const cfear_coral_params* par, bool want_per_point,
                        CoralPending& pend) {
  const int rc = coral_enqueue_impl(ctx, jobs, n_jobs, par, want_per_point, pend);
  if (rc != CFEAR_OK) (void)hipStreamSynchronize(ctx->stream);
  return rc;
}

static int coral_enqueue_impl(cfear_ctx* ctx, const cfear_coral_job* jobs, int32_t n_jobs, const cfear_coral_params* par, bool want_per_point,
                              CoralPending& pend) {
  pend.n_jobs = 0;
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!jobs || !par || n_jobs < 0) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  if (!(par->radius > 0.0)) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "radius must be > 0");
  if (n_jobs == 0) return CFEAR_OK;
  const bool per_point = want_per_point;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // stage host clouds once each (perturbation sets and candidate lists share clouds)
  std::map<const float*, size_t> staged;                 // host pointer -> offset (floats) in the staging buffer
  std::map<const float*, bool> on_device;                // one hipPointerGetAttributes per distinct cloud, not per job
  size_t stage_floats = 0;
  int cap = 1;
  for (int j = 0; j < n_jobs; j++) {
    const cfear_coral_job& jb = jobs[j];
    if (jb.n_ref < 0 || jb.n_src < 0 || (jb.n_ref > 0 && !jb.ref_xyzi) || (jb.n_src > 0 && !jb.src_xyzi))
      return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "job %d: null cloud", j);   // empty clouds are a per-job status
    if ((long long)jb.n_ref + jb.n_src > kCoralMaxPoints)
      return cfear_set_error(ctx, CFEAR_ERR_CAPACITY, "job %d: %d + %d points exceed %d", j, jb.n_ref, jb.n_src, kCoralMaxPoints);
    cap = std::max(cap, jb.n_ref + jb.n_src);
    const float* ptrs[2] = {jb.ref_xyzi, jb.src_xyzi};
    const int ns[2] = {jb.n_ref, jb.n_src};
    for (int c = 0; c < 2; c++) {
      if (ns[c] == 0 || !ptrs[c]) continue;
      auto it = on_device.find(ptrs[c]);
      if (it != on_device.end()) continue;
      const bool dev = cfear_is_device_ptr(ptrs[c]);
      on_device[ptrs[c]] = dev;
      if (!dev) {
        staged[ptrs[c]] = stage_floats;
        stage_floats += ((size_t)ns[c] * 4 + 3) & ~(size_t)3;
      }
    }
  }
  float* d_stage = nullptr;
  if (stage_floats) {
    d_stage = (float*)cfear_workspace(ctx, 8, stage_floats * 4);
    if (!d_stage) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
    std::map<const float*, int> len;
    for (int j = 0; j < n_jobs; j++) {
      len[jobs[j].ref_xyzi] = std::max(len[jobs[j].ref_xyzi], jobs[j].n_ref);
      len[jobs[j].src_xyzi] = std::max(len[jobs[j].src_xyzi], jobs[j].n_src);
    }
    for (auto& kv : staged)
      if (len[kv.first] > 0)
        CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(d_stage + kv.second, kv.first, (size_t)len[kv.first] * 16, hipMemcpyHostToDevice, ctx->stream));
  }
  pend.host_jobs.resize((size_t)n_jobs * sizeof(CoralJob));
  CoralJob* hj = (CoralJob*)pend.host_jobs.data();
  for (int j = 0; j < n_jobs; j++) {
    const cfear_coral_job& jb = jobs[j];
    CoralJob& o = hj[j];
    o.ref = (const float4*)(staged.count(jb.ref_xyzi) ? d_stage + staged[jb.ref_xyzi] : jb.ref_xyzi);
    o.src = (const float4*)(staged.count(jb.src_xyzi) ? d_stage + staged[jb.src_xyzi] : jb.src_xyzi);
    o.n_ref_ptr = o.n_src_ptr = nullptr;
    o.n_ref = jb.n_ref; o.n_src = jb.n_src;
    for (int k = 0; k < 3; k++) { o.ref_pose[k] = jb.ref_pose[k]; o.src_pose[k] = jb.src_pose[k]; o.offset[k] = jb.offset[k]; }
  }
  CoralCommon cm;
  cm.radius = par->radius;
  cm.r2 = (float)(par->radius * par->radius);            // radiusSearch passes float(radius * radius) to FLANN
  cm.inv_cell = (float)(1.0 / (par->radius * 1.0001));   // cell a hair wider than the radius: float rounding of
                                                         // x * inv_cell can never put a neighbour two cells away
  cm.weight_res_intensity = par->weight_res_intensity;
  cm.cap = cap;
  cm.scratch_stride = coral_scratch_bytes(cap);
  // scratch bounded to 1 GiB per launch
  const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_jobs, ((size_t)1 << 30) / cm.scratch_stride));
  const size_t jb_bytes = (size_t)n_jobs * sizeof(CoralJob), rb = (size_t)n_jobs * sizeof(cfear_coral_result);
  const size_t pp_bytes = per_point ? (size_t)n_jobs * cap * 3 * sizeof(double) : 0;
  char* ws = (char*)cfear_workspace(ctx, 9, (jb_bytes + 255) / 256 * 256 + (rb + 255) / 256 * 256 + pp_bytes + 512);
  char* scr = (char*)cfear_workspace(ctx, 10, cm.scratch_stride * (size_t)chunk);
  if (!ws || !scr) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  CoralJob* d_jobs = (CoralJob*)ws;
  cfear_coral_result* d_res = (cfear_coral_result*)(ws + (jb_bytes + 255) / 256 * 256);
  double* d_pp = per_point ? (double*)((char*)d_res + (rb + 255) / 256 * 256) : nullptr;
  cm.scratch = scr;
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(d_jobs, hj, jb_bytes, hipMemcpyHostToDevice, ctx->stream));
  { const int rc_lds = cfear_allow_lds(ctx, (const void*)coral_kernel, 160 * 1024); if (rc_lds != CFEAR_OK) return rc_lds; }
  {
    ProfScope ps(ctx, "coral_quality");
    for (int j0 = 0; j0 < n_jobs; j0 += chunk) {
      const int nj = std::min(chunk, n_jobs - j0);
      cm.results = d_res + j0;
      cm.per_point = d_pp ? d_pp + (size_t)j0 * cap * 3 : nullptr;
      hipLaunchKernelGGL(coral_kernel, dim3(nj), dim3(kCoralThreads), kCoralLdsTotal, ctx->stream, d_jobs + j0, cm);
    }
  }
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  pend.n_jobs = n_jobs; pend.cap = cap; pend.d_res = d_res; pend.d_pp = d_pp;
  return CFEAR_OK;
}

int cfear_coral_collect(cfear_ctx* ctx, const cfear_coral_job* jobs, const CoralPending& pend, cfear_coral_result* results, double* per_point) {
  const int n_jobs = pend.n_jobs, cap = pend.cap;
  if (n_jobs == 0) return CFEAR_OK;
  const size_t rb = (size_t)n_jobs * sizeof(cfear_coral_result);
  const size_t pp_bytes = per_point ? (size_t)n_jobs * cap * 3 * sizeof(double) : 0;
  const cfear_coral_result* d_res = (const cfear_coral_result*)pend.d_res;
  const double* d_pp = (const double*)pend.d_pp;
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(results, d_res, rb, hipMemcpyDeviceToHost, ctx->stream));
  std::vector<double> hpp;
  if (per_point) {
    hpp.resize((size_t)n_jobs * cap * 3);
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(hpp.data(), d_pp, pp_bytes, hipMemcpyDeviceToHost, ctx->stream));
  }
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (per_point) {                                       // compact [job][n_src + n_ref][3]
    size_t o = 0;
    for (int j = 0; j < n_jobs; j++) {
      const size_t nn = (size_t)jobs[j].n_ref + jobs[j].n_src;
      std::copy(hpp.begin() + (size_t)j * cap * 3, hpp.begin() + (size_t)j * cap * 3 + nn * 3, per_point + o);
      o += nn * 3;
    }
  }
  for (int j = 0; j < n_jobs; j++)
    if (results[j].status != CFEAR_OK && results[j].status != CFEAR_ERR_EMPTY_CLOUD)
      return cfear_set_error(ctx, results[j].status, "job %d: %s", j, cfear_status_string(results[j].status));
  return CFEAR_OK;
}

extern "C" int cfear_coral_quality_batch(cfear_ctx* ctx, const cfear_coral_job* jobs, int32_t n_jobs,
                                         const cfear_coral_params* par, cfear_coral_result* results, double* per_point) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!results) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  CoralPending pend;
  const int rc = cfear_coral_enqueue(ctx, jobs, n_jobs, par, per_point != nullptr, pend);
  if (rc != CFEAR_OK) return rc;
  return cfear_coral_collect(ctx, jobs, pend, results, per_point);
}

extern "C" int cfear_coral_quality(cfear_ctx* ctx, const cfear_coral_job* job, const cfear_coral_params* par,
                                   cfear_coral_result* result, double* per_point) {
  return cfear_coral_quality_batch(ctx, job, 1, par, result, per_point);
}
