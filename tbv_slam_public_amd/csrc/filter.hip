// filter.hip -- stage F of the CFEAR hot path on gfx950: k-strongest / peaks / CA-CFAR filtering of
// the polar radar image and polar->Cartesian conversion.
//
// Replaces (cfear_radarodometry/src/cfear_radarodometry/):
//   StructuredKStrongest::FilterKstrongest          radar_filters.cpp:209-237
//   StructuredKStrongest::AxialNonMaxSupress        radar_filters.cpp:238-298
//   StructuredKStrongest::getPeaksFilteredPointCloud radar_filters.cpp:300-337
//   AzimuthCACFAR::getFilteredPointCloud / getMean  cfar.cpp:35-83
//
// Kernel design (HBM-bound integer/byte work, no MFMA):
//   kstrongest_rows_kernel  one 64-lane wavefront per azimuth row.  The row (cols bytes) is read
//     ONCE from HBM with 16-byte-per-lane non-temporal loads into registers (cols <= 8192 ->
//     <= 8 x dwordx4 per lane) and never re-read.  The k largest (intensity,range) pairs are found
//     without sorting the row: a SWAR byte-compare + popcount counts "intensity >= t" for a wave-
//     uniform threshold t (3 VALU per 4 bins), an 8-step bisection finds the cut intensity T, ties
//     at T are resolved toward the larger range through a packed DPP prefix scan, and only the
//     <= k survivors are ranked (LDS broadcast reads) to produce the reference's ascending
//     (intensity,range) order.  Results are bit-exact integers.
//   kstrong_cloud_kernel    one workgroup per image: per-row survivor counts -> scan -> compacted
//     PointXYZI cloud, fp64 polar->Cartesian with host-computed cos/sin tables (bit-exact floats).
//   cacfar_rows_kernel      one wavefront per row: exact uint32 prefix sums of squares in LDS,
//     windowed means and thresholds in fp64 exactly as the reference evaluates them.
#include <cmath>

#include "common.hpp"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kRowsPerBlock = 4;   // 4 wavefronts (rows) per 256-thread workgroup
constexpr int kMaxCols = 8192;
constexpr int kMaxK = 1024;

struct KStrongArgs {
  const uint8_t* polar;
  int rows, cols, stride, batch;
  long long batch_stride;
  int k, u_zmin, want_peaks, batch0;
  int32_t* sel_range;
  uint8_t* sel_intensity;
  int32_t* sel_count;
  uint8_t* is_peak;
  int32_t* row_valid;      // [batch][rows][2]: kept bins beyond min_range_bin (all, peaks) -> cloud offsets
  int min_range_bin;
  int dense_halo;          // 1: the image is a padded copy of a DENSE cv::Mat (the pipeline's rotated buffer): bins read
                           // past a row end are the first bins of the next row, not the padding
  // Fused cloud output (the batched odometry pipeline, k <= 64): the kept bins beyond min_range_bin of row r as packed
  // keys (intensity << 24 | range bin) at row_keys[(b * rows + r) * k + j], j < row_valid[..][0], in the reference's
  // order (ascending (intensity, range), radar_filters.cpp:309-337): 4 bytes per point instead of a PointXYZI; the
  // surface-point kernel compacts the rows and converts to Cartesian (surface.hip).
  uint32_t* row_keys;
  const long long* image_offsets;   // optional [batch]: byte offset of image b from `polar` instead of b * batch_stride
};

// bit 7 of every byte of the result is set iff that byte of x is >= t (0 <= t <= 255).
// Per byte (0x80 + low7) - (t & 0x7f) stays in [1, 0xff]: no borrow crosses a byte boundary.
__device__ __forceinline__ uint32_t swar_ge(uint32_t x, uint32_t tl4, bool thi) {
  const uint32_t g = ((x & 0x7f7f7f7fu) | 0x80808080u) - tl4;
  const uint32_t xh = x & 0x80808080u;
  return thi ? (xh & g) : ((g & 0x80808080u) | xh);
}

// Reads the row into registers: lane L of chunk c owns bytes [(c*64+L)*16, +16).  Lanes whose chunk
// lies entirely past the row end hold zeros and issue no loads.
// tail_safe: the 16 bytes of the row's LAST, partial piece may be read whole (they end inside the image: every row but an
// image's last one) -- the bytes beyond the row are then cleared in registers.  Without it the piece is gathered byte by byte:
// sixteen dependent predicated loads on one lane that the whole wavefront waits for (Oxford's native 3768 bins end 8 bytes into
// a piece: 0.220 instead of 0.159 ms per 512 sweeps until round 6).
template <int NCHUNK, bool VEC>
__device__ __forceinline__ void load_row(const uint8_t* rowp, int cols, int lane, uint32_t (&w)[NCHUNK * 4], const bool tail_safe = false) {
#pragma unroll
  for (int c = 0; c < NCHUNK; c++) {
    const int pos = (c * 64 + lane) * 16;
    w[c * 4 + 0] = w[c * 4 + 1] = w[c * 4 + 2] = w[c * 4 + 3] = 0u;
    if (VEC && (pos + 16 <= cols || (tail_safe && pos < cols))) {
      const u32x4 v = __builtin_nontemporal_load((const u32x4*)(rowp + pos));
      w[c * 4 + 0] = v.x; w[c * 4 + 1] = v.y; w[c * 4 + 2] = v.z; w[c * 4 + 3] = v.w;
      if (pos + 16 > cols) {                       // the ragged tail: keep the row's own bytes only
#pragma unroll
        for (int d = 0; d < 4; d++) {
          const int rem = cols - (pos + 4 * d);
          w[c * 4 + d] &= rem >= 4 ? 0xffffffffu : (rem <= 0 ? 0u : ((1u << (8 * rem)) - 1u));
        }
      }
    } else if (pos < cols) {                       // unaligned image or the partial tail chunk of an image's last row
#pragma unroll
      for (int d = 0; d < 4; d++) {
        uint32_t word = 0;
#pragma unroll
        for (int by = 0; by < 4; by++) {
          const int p = pos + d * 4 + by;
          if (p < cols) word |= (uint32_t)rowp[p] << (8 * by);
        }
        w[c * 4 + d] = word;
      }
    }
  }
}

// Candidate bitmaps.  For each 16-byte chunk c of this lane, bit (8*by + 4 + d) of bm[c] is set iff
// byte `by` of word d is >= t.  The SWAR compare leaves its verdict in bit 7 of every byte; a
// v_bfi per word shifts the running bitmap down one bit and inserts the new verdicts, so the
// bitmap costs nothing over the masks themselves (5 VALU per 4 bins).
template <int NCHUNK, bool MASK, bool THI>
__device__ __forceinline__ void candidate_bitmaps(const uint32_t (&w)[NCHUNK * 4], uint32_t tl4, int cols, int lane,
                                                  uint32_t (&bm)[NCHUNK]) {
  constexpr uint32_t M = 0x80808080u;
#pragma unroll
  for (int c = 0; c < NCHUNK; c++) {
    uint32_t acc = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const uint32_t x = w[c * 4 + d];
      const uint32_t g = ((x & 0x7f7f7f7fu) | M) - tl4;   // per byte in [1, 0xff]: no borrow crosses bytes
      uint32_t raw = THI ? (g & x) : (g | x);              // bit 7 of each byte: byte >= t
      if (MASK) {                                          // byte validity (row tail / z_min == 0)
        const int rem = cols - ((c * 64 + lane) * 16 + d * 4);
        raw &= rem >= 4 ? 0xffffffffu : (rem <= 0 ? 0u : ((1u << (8 * rem)) - 1u));
      }
      acc = (raw & M) | ((acc >> 1) & ~M);
    }
    bm[c] = acc;
  }
}

// Calls f(pos) for every candidate bin of this lane.  Two chunks share one loop (their bitmaps
// occupy disjoint nibbles), so a row of <= 2048 bins costs one divergent loop, 4096 bins two.
template <int NCHUNK, typename F>
__device__ __forceinline__ void for_each_candidate(const uint32_t (&bm)[NCHUNK], int lane, F&& f) {
#pragma unroll
  for (int c = 0; c < NCHUNK; c += 2) {
    uint32_t mm = (bm[c] >> 4) | (c + 1 < NCHUNK ? bm[c + 1] : 0u);
    const int base = c * 1024 + lane * 16;
    while (mm) {
      const int t = __ffs(mm) - 1;
      mm &= mm - 1;
      f(base + ((t & 4) << 8) + ((t & 3) << 2) + (t >> 3));
    }
  }
}

// One azimuth row on one wavefront.  STAGED = false: the row is read from `rowp` (global memory) and staged in `rowbuf`;
// STAGED = true: the caller has already placed the row's bytes in `rowbuf` (LDS, 16-byte aligned, readable up to the next
// multiple of 16 bins; kstrongest_cols_kernel transposes them there) -- no peaks in that mode (no halo bytes).
// hist: [kScratch / 4] dwords of scratch, list: [kpad] packed keys; nothing below crosses a workgroup barrier.
template <int NCHUNK, bool VEC, bool MASK, bool STAGED>
__device__ __forceinline__ void kstrong_row(const KStrongArgs& a, const int r, const int b, const uint8_t* img, uint8_t* rowbuf,
                                            uint32_t* hist, uint32_t* list, const int lane) {
  const long long row_lin = (long long)r * a.stride;
  const uint8_t* rowp = img + row_lin;
  const int k = a.k;
  constexpr int NP = (NCHUNK + 1) / 2;             // bitmap words per lane (two chunks per word)

  // Peaks only: the six bytes before and after the row (AxialNonMaxSupress reads them through unchecked cv::Mat::at,
  // radar_filters.cpp:238-298).  Their loads are issued here, together with the row's, so that they cost no extra
  // memory round trip later.
  const bool do_peaks = !STAGED && a.want_peaks && a.is_peak;
  uint8_t halo = 0;
  int halo_pos = 0;                                // rowbuf offset this lane's halo byte belongs to (0 = none)
  if (do_peaks && (lane < 6 || (lane >= 8 && lane < 14))) {
    const long long total = (long long)a.rows * a.stride;
    long long lin;
    if (lane < 6) {
      halo_pos = -6 + lane;
      lin = row_lin - 6 + lane;
      if (a.dense_halo) lin = r > 0 ? row_lin - a.stride + a.cols - 6 + lane : -1;       // last bins of the previous row
    } else {
      halo_pos = a.cols + (lane - 8);
      lin = row_lin + halo_pos;
      if (a.dense_halo) lin = r + 1 < a.rows ? row_lin + a.stride + (lane - 8) : total;   // first bins of the next row
    }
    if (lin >= 0 && lin < total) halo = img[lin];
  }
  const int cols16 = (a.cols + 15) & ~15;          // STAGED: the bytes of rowbuf that may be read
  uint32_t w[NCHUNK * 4];
  if (STAGED) {
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if ((c * 64 + lane) * 16 < cols16) v = *(const uint4*)(rowbuf + (c * 64 + lane) * 16);
      w[c * 4] = v.x; w[c * 4 + 1] = v.y; w[c * 4 + 2] = v.z; w[c * 4 + 3] = v.w;
    }
  } else {
    load_row<NCHUNK, VEC>(rowp, a.cols, lane, w, r + 1 < a.rows);
#pragma unroll
    for (int c = 0; c < NCHUNK; c++)                 // stage the row: candidate bytes are fetched by position
      *(uint4*)(rowbuf + (c * 64 + lane) * 16) = make_uint4(w[c * 4], w[c * 4 + 1], w[c * 4 + 2], w[c * 4 + 3]);
  }

  // The row lives on in LDS: the (rare) later passes over it re-read it from there instead of keeping 4 NCHUNK
  // registers alive across the whole kernel (occupancy: 8 wavefronts per SIMD need <= 64 VGPRs).
  auto reload_row = [&](uint32_t (&x)[NCHUNK * 4]) {
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (!STAGED || (c * 64 + lane) * 16 < cols16) v = *(const uint4*)(rowbuf + (c * 64 + lane) * 16);
      x[c * 4] = v.x; x[c * 4 + 1] = v.y; x[c * 4 + 2] = v.z; x[c * 4 + 3] = v.w;
    }
  };
  // ---- candidates: bins with intensity >= uchar(z_min) (radar_filters.cpp:217) ---------------------
  uint32_t bm[NCHUNK];
  {
    const uint32_t tz4 = (uint32_t)(a.u_zmin & 0x7f) * 0x01010101u;
    if (a.u_zmin & 0x80) candidate_bitmaps<NCHUNK, MASK, true>(w, tz4, a.cols, lane, bm);
    else candidate_bitmaps<NCHUNK, MASK, false>(w, tz4, a.cols, lane, bm);
  }
  int c_lane = 0;
#pragma unroll
  for (int c = 0; c < NCHUNK; c++) c_lane += __popc(bm[c]);
  const int c_incl = wave_incl_scan_i32(c_lane);
  const int n_ge = __builtin_amdgcn_readlane(c_incl, 63);
  int n_sel;                                       // survivors = min(candidates, k)
  int n_all;                                       // keys placed in list[] (== n_sel unless the cut is made by rank)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // One candidate per lane and round, however the candidates cluster (a wall return fills adjacent bins of ONE
  // lane).  Candidates are numbered lane-major ("slots"); owner lanes mark the first slot of their run, a max-scan
  // spreads the owner id over the run, and in round rd lane j takes slot 64 rd + j: it selects bit (slot - first
  // slot) of its owner's bitmap by popcount bisection.  n <= 256 candidates -> at most 4 rounds, no divergent loop.
  uint8_t* marker = (uint8_t*)hist;                // [256] owner lane of the slot that starts a run, else 0
  uint32_t* sexcl = hist + 64;                     // [64] first slot of each lane's run
  uint32_t* spw = hist + 128;                      // [NP][64] bitmaps
  auto scatter_prepare = [&](const uint32_t (&bmx)[NCHUNK], int cnt_lane, int cnt_incl) {
    uint32_t pw[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) pw[p] = (bmx[2 * p] >> 4) | (2 * p + 1 < NCHUNK ? bmx[2 * p + 1] : 0u);
    const int excl = cnt_incl - cnt_lane;
    ((uint32_t*)marker)[lane] = 0;
    sexcl[lane] = excl;
#pragma unroll
    for (int p = 0; p < NP; p++) spw[p * 64 + lane] = pw[p];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (cnt_lane && excl < 256) marker[excl] = (uint8_t)lane;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  int scatter_carry = 0;                           // owner of the last slot of the previous round
  auto scatter_round = [&](int rd, int n, bool& valid) -> uint32_t {
    const int slot = rd * 64 + lane;
    const int own = max(scatter_carry, wave_incl_scan_max_i32((int)marker[slot]));
    scatter_carry = __builtin_amdgcn_readlane(own, 63);
    valid = slot < n;
    uint32_t key = 0;
    if (valid) {
      int q = slot - (int)sexcl[own];
      uint32_t word = spw[own];
      int p = 0;
#pragma unroll
      for (int pp = 1; pp < NP; pp++) {
        const int c = __popc(word);
        const uint32_t nxt = spw[pp * 64 + own];
        if (p == pp - 1 && q >= c) { q -= c; word = nxt; p = pp; }
      }
      int t = 0;
      { const int c = __popc(word & 0xFFFFu); if (q >= c) { q -= c; t = 16; word >>= 16; } }
      { const int c = __popc(word & 0xFFu);   if (q >= c) { q -= c; t += 8; word >>= 8; } }
      { const int c = __popc(word & 0xFu);    if (q >= c) { q -= c; t += 4; word >>= 4; } }
      { const int c = __popc(word & 0x3u);    if (q >= c) { q -= c; t += 2; word >>= 2; } }
      if (q >= (int)(word & 1u)) t += 1;
      const int pos = p * 2048 + own * 16 + ((t & 4) << 8) + ((t & 3) << 2) + (t >> 3);
      key = ((uint32_t)rowbuf[pos] << 24) | (uint32_t)pos;
    }
    return key;
  };
  auto scatter_to_lanes = [&](const uint32_t (&bmx)[NCHUNK], int cnt_lane, int cnt_incl, int n) {   // n <= 64
    scatter_prepare(bmx, cnt_lane, cnt_incl);
    scatter_carry = 0;
    bool valid;
    const uint32_t key = scatter_round(0, n, valid);
    if (valid) list[lane] = key;
  };
  // The cut intensity T of the k largest among the histogrammed keys: lane L owns intensities 4 (63 - L) + {0..3},
  // an inclusive scan over lanes counts from 255 downward.
  auto cut_from_hist = [&](int& T, int& n_gt, int& n_eq) {
    const uint4 h = *(const uint4*)(hist + (63 - lane) * 4);
    const int s_lane = (int)(h.x + h.y + h.z + h.w);
    const int s_incl = wave_incl_scan_i32(s_lane);
    const unsigned long long reach = __ballot(s_incl >= k);
    const int lc = __ffsll((long long)reach) - 1;                  // first lane whose cumulative count reaches k
    int Tl = 0, gl = 0, el = 0;
    {
      int cum = s_incl - s_lane;
      const int hv[4] = {(int)h.w, (int)h.z, (int)h.y, (int)h.x};  // descending intensity
      bool found = false;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (!found && cum + hv[j] >= k) { Tl = 4 * (63 - lane) + 3 - j; gl = cum; el = hv[j]; found = true; }
        cum += hv[j];
      }
    }
    T = __builtin_amdgcn_readlane(Tl, lc);
    n_gt = __builtin_amdgcn_readlane(gl, lc);
    n_eq = __builtin_amdgcn_readlane(el, lc);
  };

  if (__builtin_expect(n_ge <= k || n_ge <= 64, 1)) {
    // ---- at most k candidates, or at most 64: all of them become keys (any order); the ranking below
    //      restores the reference's ascending (intensity, range) order and, when there are more than k,
    //      keeps the k largest keys -- the lexicographic (intensity, range) cut of the reference, ties at
    //      the cut intensity resolved toward the larger range, without building the histogram --------------
    n_sel = min(n_ge, k);
    n_all = n_ge;
    if (__builtin_expect(n_ge <= 64, 1)) {
      scatter_to_lanes(bm, c_lane, c_incl, n_ge);
    } else {
      int slot = c_incl - c_lane;
      for_each_candidate<NCHUNK>(bm, lane, [&](int pos) { list[slot++] = ((uint32_t)rowbuf[pos] << 24) | (uint32_t)pos; });
    }
  } else {
    // ---- more than k and more than 64 candidates: the cut intensity T -------------------------------------------
    n_sel = k;
    n_all = k;
    int T = 0, n_gt = 0, n_eq = 0;
    bool have_list = false;                        // list[] already holds every bin >= T (n_all of them, <= 64)
    // (a) very dense rows (> 256 candidates): raise the candidate threshold until between k and 256 bins pass it --
    //     each trial is one SWAR pass over the register-resident row (5 VALU per 4 bins); the first trial assumes
    //     a flat intensity distribution above the threshold, later ones bisect.  If two neighbouring thresholds
    //     bracket k the cut is known exactly (a plateau) and goes to the tie scan below.
    uint32_t bt[NCHUNK];
    int t_lane = c_lane, t_incl = c_incl, n_c = n_ge;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) bt[c] = bm[c];
    bool exact = false;
    if (n_ge > 256) {
      int lo = a.u_zmin, c_lo = n_ge, hi = 256, c_hi = 0;
      bool first = true;
      while (hi - lo > 1) {
        int mid = first ? 256 - max(1, ((256 - lo) * 128) / c_lo) : (lo + hi) >> 1;
        mid = min(max(mid, lo + 1), hi - 1);
        first = false;
        const uint32_t tm4 = (uint32_t)(mid & 0x7f) * 0x01010101u;
        uint32_t bx[NCHUNK];
        {
          uint32_t wx[NCHUNK * 4];
          reload_row(wx);
          if (mid & 0x80) candidate_bitmaps<NCHUNK, MASK, true>(wx, tm4, a.cols, lane, bx);
          else candidate_bitmaps<NCHUNK, MASK, false>(wx, tm4, a.cols, lane, bx);
        }
        int x_lane = 0;
#pragma unroll
        for (int c = 0; c < NCHUNK; c++) x_lane += __popc(bx[c]);
        const int x_incl = wave_incl_scan_i32(x_lane);
        const int cnt = __builtin_amdgcn_readlane(x_incl, 63);
        if (cnt > 256) { lo = mid; c_lo = cnt; }
        else if (cnt < k) { hi = mid; c_hi = cnt; }
        else {
#pragma unroll
          for (int c = 0; c < NCHUNK; c++) bt[c] = bx[c];
          t_lane = x_lane; t_incl = x_incl; n_c = cnt;
          break;
        }
      }
      if (n_c > 256) { exact = true; T = lo; n_gt = c_hi; n_eq = c_lo - c_hi; }
    }
    if (!exact) {
      // (b) 64 < n_c <= 256 candidates: one key per lane and round (registers), an LDS histogram of the keys'
      //     intensities -> T; the keys >= T (the survivors plus the ties at the cut, <= 64 unless a plateau is
      //     wider) are packed into list[] by ballot and cut by rank below -- the reference's tie rule.
      scatter_prepare(bt, t_lane, t_incl);
      scatter_carry = 0;
      uint32_t kr[4];
      bool kv[4];
      const int rounds = (n_c + 63) >> 6;
#pragma unroll
      for (int rd = 0; rd < 4; rd++) { kr[rd] = 0; kv[rd] = false; if (rd < rounds) kr[rd] = scatter_round(rd, n_c, kv[rd]); }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      *(uint4*)(hist + lane * 4) = make_uint4(0, 0, 0, 0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int rd = 0; rd < 4; rd++) if (kv[rd]) atomicAdd(&hist[kr[rd] >> 24], 1u);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      cut_from_hist(T, n_gt, n_eq);
      if (n_gt + n_eq <= 64) {
        int base = 0;
#pragma unroll
        for (int rd = 0; rd < 4; rd++) {
          const bool sel = kv[rd] && (int)(kr[rd] >> 24) >= T;
          const unsigned long long bal = __ballot(sel);
          if (sel) list[base + __popcll(bal & ((1ull << lane) - 1ull))] = kr[rd];
          base += __popcll(bal);
        }
        n_all = n_gt + n_eq;
        have_list = true;
      }
    }
    if (!have_list) {
    uint32_t wt[NCHUNK * 4];
    reload_row(wt);
    const int skip_eq = n_eq - (k - n_gt);         // drop the lowest-range ties: lexicographic (intensity, range)
    // ---- ordered compaction: all (> T) plus the (== T) bins of rank >= skip_eq in position order -----
    // (T >= z_min, so ">= T" implies candidacy; only the MASK variant needs the validity bits)
    const int thr_gt = T + 1;
    const uint32_t tg4 = (uint32_t)(thr_gt & 0x7f) * 0x01010101u;
    const bool tghi = (thr_gt & 0x80) != 0;
    const uint32_t te4 = (uint32_t)(T & 0x7f) * 0x01010101u;
    const bool tehi = (T & 0x80) != 0;
    int g_base = 0, e_base = 0;                    // survivors > T / bins == T before the current chunk
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      uint32_t mg[4], me[4];
      int cg = 0, ce = 0;
#pragma unroll
      for (int d = 0; d < 4; d++) {
        const int i = c * 4 + d;
        const uint32_t valid = MASK ? ((bm[c] << (3 - d)) & 0x80808080u) : 0x80808080u;
        mg[d] = thr_gt > 255 ? 0u : (swar_ge(wt[i], tg4, tghi) & valid);
        me[d] = swar_ge(wt[i], te4, tehi) & valid & ~mg[d];
        cg += __popc(mg[d]);
        ce += __popc(me[d]);
      }
      const int packed = (ce << 16) | cg;          // one scan carries both prefixes (totals < 65536)
      const int incl = wave_incl_scan_i32(packed);
      const int tot = __builtin_amdgcn_readlane(incl, 63);
      const int excl = incl - packed;
      int g_run = g_base + (excl & 0xFFFF);
      int e_run = e_base + (excl >> 16);
      if (cg + ce) {
#pragma unroll
        for (int d = 0; d < 4; d++) {
          uint32_t mm = mg[d] | me[d];
          while (mm) {
            const int bit = __ffs(mm) - 1;         // bit 7 of byte (bit >> 3)
            mm &= mm - 1;
            const int by = bit >> 3;
            const bool is_eq = (me[d] >> bit) & 1u;
            const bool selected = !is_eq || e_run >= skip_eq;
            const int e_sel_before = e_run > skip_eq ? e_run - skip_eq : 0;
            if (selected) {
              const int pos = (c * 64 + lane) * 16 + d * 4 + by;
              list[g_run + e_sel_before] = (((wt[c * 4 + d] >> (8 * by)) & 0xffu) << 24) | (uint32_t)pos;
            }
            if (is_eq) e_run++; else g_run++;
          }
        }
      }
      g_base += tot & 0xFFFF;
      e_base += tot >> 16;
    }
    }
  }
  const int nq = (n_all + 3) & ~3;
  for (int j = n_all + lane; j < nq; j += 64) list[j] = 0xFFFFFFFFu;      // pad for the b128 reads
  if (a.row_keys && lane < 2) hist[2 + lane] = 0;                       // ranks of the kept bins beyond min_range_bin
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int drop = n_all - n_sel;                  // keys below the cut (only when n_all <= 64)

  // ---- peaks (AxialNonMaxSupress, radar_filters.cpp:238-298): score[r] = sum of raw[r-3..r+3]
  //      exists for r within 3 bins of a kept bin m with 3 <= m < cols - 3; a kept bin is a peak iff its
  //      score is not exceeded by the three scores either side (missing scores are 0).  The reference reads
  //      raw[] through unchecked cv::Mat::at, i.e. up to 6 bytes before / after the row in image memory:
  //      those halo bytes are staged next to the row so that every tap is one LDS read. --------------------
  auto note_kept = [&](int m) {                    // kept VALID bins among the first / last 16 bins of the row
    if (m >= 3 && m < a.cols - 3) {
      if (m < 16) atomicOr(&hist[0], 1u << m);
      if (m >= a.cols - 16) atomicOr(&hist[1], 1u << (m - (a.cols - 16)));
    }
  };
  if (__builtin_expect(do_peaks, 0)) {
    if (halo_pos != 0) rowbuf[halo_pos] = halo;    // after the row staging: the tail chunk's zero padding lies there
    if (lane < 2) hist[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (drop == 0) {                               // every key survives: the kept set is known before ranking
      for (int j = lane; j < n_all; j += 64) note_kept((int)(list[j] & 0xFFFFFFu));
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  // ---- rank the keys: ascending (intensity, range) == ascending packed key; survivors have rank >= drop ----
  const long long obase = ((long long)b * a.rows + r) * k;
  int nvalid = 0, nvalid_pk = 0;                   // wave-uniform (ballot popcounts)
  for (int j0 = 0; j0 < n_all; j0 += 64) {         // one pass unless k > 64
    const int j = j0 + lane;
    bool beyond = false, beyond_pk = false;
    uint32_t key = 0;
    int rank = -1;
    if (j < n_all) {
      key = list[j];
      rank = 0;
      for (int i = 0; i < nq; i += 4) {
        const uint4 q = *(const uint4*)(list + i);  // same address in every lane: LDS broadcast
        rank += (q.x < key) + (q.y < key) + (q.z < key) + (q.w < key);
      }
      rank -= drop;                                 // < 0: below the cut
    }
    const int range = (int)(key & 0xFFFFFFu);
    if (do_peaks && drop != 0) {                    // n_all <= 64: single pass, the kept set follows from the ranks
      if (rank >= 0) note_kept(range);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (rank >= 0) {
      if (a.sel_range) a.sel_range[obase + rank] = range;
      if (a.sel_intensity) a.sel_intensity[obase + rank] = (uint8_t)(key >> 24);
      beyond = range > a.min_range_bin;                             // radar_filters.cpp:327
      if (do_peaks) {
        int v[13];                                  // raw[range - 6 .. range + 6]
#pragma unroll
        for (int i = 0; i < 13; i++) v[i] = rowbuf[range - 6 + i];
        int sc[7];                                  // score[range - 3 .. range + 3]
        sc[0] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + v[6]);
#pragma unroll
        for (int i = 1; i < 7; i++) sc[i] = sc[i - 1] - v[i - 1] + v[i + 6];
        if (!(range >= 3 && range < a.cols - 3)) {
          // border bin: it created no scores itself; score[r] exists only if another kept valid bin lies
          // within 3 bins of r (all such bins are among the first / last 16 of the row)
          const uint32_t klo = hist[0], khi = hist[1];
          const int base = a.cols - 16;
#pragma unroll
          for (int i = 0; i < 7; i++) {
            const int rr = range - 3 + i;
            bool covered = false;
            {                                       // kept valid bins in [rr - 3, rr + 3] among bins 0..15
              const int lo = max(rr - 3, 0), hi = min(rr + 3, 15);
              if (hi >= lo) covered = ((klo >> lo) & ((2u << (hi - lo)) - 1u)) != 0u;
            }
            {                                       // ... among bins cols - 16 .. cols - 1
              const int lo = max(rr - 3 - base, 0), hi = min(rr + 3 - base, 15);
              if (hi >= lo) covered = covered || ((khi >> lo) & ((2u << (hi - lo)) - 1u)) != 0u;
            }
            if (!covered) sc[i] = 0;
          }
        }
        bool pk = true;
#pragma unroll
        for (int i = 1; i <= 3; i++)
          if (sc[3 - i] > sc[3] || sc[3] < sc[3 + i]) pk = false;
        a.is_peak[obase + rank] = pk ? 1 : 0;
        beyond_pk = beyond && pk;
      }
    }
    if (__builtin_expect(a.row_keys != nullptr, 1)) {
      // fused getPeaksFilteredPointCloud(cloud, false) (radar_filters.cpp:309-337): the row's kept bins in rank order;
      // a bin's slot = number of kept bins beyond min_range_bin with a lower rank (k <= 64: one pass)
      if (beyond) atomicOr(&hist[2 + (rank >> 5)], 1u << (rank & 31));
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (beyond) {
        const uint32_t m0 = hist[2], m1 = hist[3];
        const int idx = rank < 32 ? __popc(m0 & ((1u << rank) - 1u)) : __popc(m0) + __popc(m1 & ((1u << (rank - 32)) - 1u));
        a.row_keys[obase + idx] = key;
      }
    }
    nvalid += __popcll(__ballot(beyond));
    nvalid_pk += __popcll(__ballot(beyond_pk));
  }
  for (int j = n_sel + lane; j < k; j += 64) {     // unused slots
    if (a.sel_range) a.sel_range[obase + j] = -1;
    if (a.sel_intensity) a.sel_intensity[obase + j] = 0;
    if (do_peaks) a.is_peak[obase + j] = 0;
  }
  if (lane == 0) {
    if (a.row_valid) {
      a.row_valid[((long long)b * a.rows + r) * 2] = nvalid;
      a.row_valid[((long long)b * a.rows + r) * 2 + 1] = nvalid_pk;
    }
    if (a.sel_count) a.sel_count[(long long)b * a.rows + r] = n_sel;
  }
}

// per-wavefront LDS of a row: [raw row + 16-byte halos] (not STAGED) | scratch | list
__host__ __device__ constexpr int kstrong_scratch_bytes(int nchunk) { return ((nchunk + 1) / 2 + 2) * 256 > 1024 ? ((nchunk + 1) / 2 + 2) * 256 : 1024; }

template <int NCHUNK, bool VEC, bool MASK>
__global__ __launch_bounds__(256, 7) void kstrongest_rows_kernel(const KStrongArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int r = blockIdx.x * kRowsPerBlock + wave;          // grid = (row quads, images): no division
  if (r >= a.rows) return;                                  // no workgroup barrier below
  const int b = a.batch0 + blockIdx.y;
  const uint8_t* img = a.polar + (a.image_offsets ? a.image_offsets[b] : (long long)b * a.batch_stride);
  const int kpad = max((a.k + 3) & ~3, 64);        // list capacity: k survivors, or up to 64 candidates to rank
  constexpr int kScratch = kstrong_scratch_bytes(NCHUNK);   // marker u8[256] | sexcl[64] | spw[NP][64], or hist[256]
  const int per_wave = NCHUNK * 1024 + 32 + kScratch + kpad * 4;
  uint8_t* rowbuf = smem + wave * per_wave + 16;                                  // the raw row, 16-byte halo either side
  uint32_t* hist = (uint32_t*)(smem + wave * per_wave + NCHUNK * 1024 + 32);      // [256] histogram / scatter scratch
  uint32_t* list = (uint32_t*)(smem + wave * per_wave + NCHUNK * 1024 + 32 + kScratch);   // [kpad] survivors (packed keys)
  kstrong_row<NCHUNK, VEC, MASK, false>(a, r, b, img, rowbuf, hist, list, lane);
}

// ---- [range bins][azimuths] sources: the driver's decode fused into the sweep ---------------------------------------
// radarDriver::Callback (radar_driver.cpp:74-90) rotates such a sweep 90 degrees counter-clockwise before Process();
// rotate_ccw_rows_kernel + kstrongest_rows_kernel move the image through HBM three times (read, write, read).  The fused
// route reads it ONCE, along its own rows, and never builds the rotated image:
//   1. extraction   the source is streamed in 16-byte pieces (16 azimuths of one bin); a SWAR compare finds the bytes >=
//      uchar(z_min) -- the only bins FilterKstrongest can keep (radar_filters.cpp:217) -- and each is appended as a key
//      (intensity << 24 | bin) to the list of ITS azimuth (a radar sweep holds a few dozen per azimuth).
//   2. selection    one wavefront per azimuth (two when both lists are short): the k largest keys of the list ARE the
//      reference's selection (lexicographic (intensity, range) cut, ties toward the larger range), ranked into its order.
//      Batches that fill the chip run 1 + 2 in ONE kernel, a workgroup per image with the lists in LDS
//      (kstrong_image_kernel); smaller ones spread an image over many workgroups and keep the lists in global memory
//      (kstrong_extract_kernel: one returning atomic per candidate; kstrong_select_kernel).
//   3. kstrongest_cols_kernel   azimuths with more than kCandCap candidates (dense returns; z_min = 0) need the raw row:
//      pass 2 puts their 16-column tile on a work list, and this kernel transposes those tiles into LDS (v_perm_b32 on
//      4 x 4 byte blocks) and runs the complete row algorithm on them.  Without a list it takes every tile.
// a.rows / a.cols are the ROTATED image's (azimuths, bins); a.stride / a.batch_stride are the SOURCE's (bytes per bin
// row, per image).  Output row r holds source column a.rows - 1 - r (cv::ROTATE_90_COUNTERCLOCKWISE).
constexpr int kColsTile = 16;
constexpr int kColsWaves = 8;
constexpr int kXcds = 8;
constexpr int kCandCap = 256;                 // candidate keys kept per azimuth (1 KiB)
constexpr int kExtractPieces = 8;             // 16-byte pieces in flight per thread and step

// The candidates among kExtractPieces 16-byte pieces held in registers (w[u] = piece p0 + step * u of the image, pieces
// numbered along the source rows: piece p = bin p / segs, source columns 16 (p % segs) ..): emit(r, key) once per byte
// >= uchar(z_min), r = the azimuth row of the ROTATED image.  One candidate per lane and turn, whichever piece it sits in
// (a wall fills adjacent azimuths of ONE piece): a wavefront takes as many turns as its busiest lane holds candidates.
template <typename Emit>
__device__ __forceinline__ void extract_candidates(const KStrongArgs& a, const uint32_t (&w)[kExtractPieces][4], const uint32_t p0,
                                                   const uint32_t step, const uint32_t n_pieces, const uint32_t seg_magic,
                                                   const int segs, Emit&& emit) {
  const uint32_t tz4 = (uint32_t)(a.u_zmin & 0x7f) * 0x01010101u;
  const bool thi = (a.u_zmin & 0x80) != 0;
  uint32_t cm[kExtractPieces / 2];                           // bit 16 (u & 1) + 4 d + byte of word u / 2
#pragma unroll
  for (int h = 0; h < kExtractPieces / 2; h++) cm[h] = 0;
#pragma unroll
  for (int u = 0; u < kExtractPieces; u++) {
    // the SWAR verdicts sit in bit 7 of every byte; v_dot4_u32_u8 with the weights 1, 2, 4, .., 128 sums them into
    // 128 * (one bit per byte) -- two dwords per accumulator
    const uint32_t lo = __builtin_amdgcn_udot4(swar_ge(w[u][1], tz4, thi), 0x80402010u,
                                               __builtin_amdgcn_udot4(swar_ge(w[u][0], tz4, thi), 0x08040201u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(swar_ge(w[u][3], tz4, thi), 0x80402010u,
                                               __builtin_amdgcn_udot4(swar_ge(w[u][2], tz4, thi), 0x08040201u, 0u, false), false);
    uint32_t pm = (lo >> 7) | ((hi >> 7) << 8);
    if (p0 + step * u >= n_pieces) pm = 0;
    cm[u >> 1] |= pm << (16 * (u & 1));
  }
  for (;;) {
    int t = -1;
#pragma unroll
    for (int h = kExtractPieces / 2 - 1; h >= 0; h--)
      if (cm[h]) t = 32 * h + __ffs(cm[h]) - 1;
    if (t < 0) break;
#pragma unroll
    for (int h = 0; h < kExtractPieces / 2; h++)
      if ((t >> 5) == h) cm[h] &= cm[h] - 1;
    const int u = t >> 4, e = t & 15;
    // w[u][e >> 2]: registers cannot be indexed by a lane -- the piece by a chain of selects, then the dword
    uint32_t x0 = 0, x1 = 0, x2 = 0, x3 = 0;
#pragma unroll
    for (int uu = 0; uu < kExtractPieces; uu++) {
      const bool is = u == uu;
      x0 = is ? w[uu][0] : x0; x1 = is ? w[uu][1] : x1; x2 = is ? w[uu][2] : x2; x3 = is ? w[uu][3] : x3;
    }
    const uint32_t word = (e & 8) ? ((e & 4) ? x3 : x2) : ((e & 4) ? x1 : x0);
    const uint32_t p = p0 + step * (uint32_t)u;
    const uint32_t j = segs == 1 ? p : __umulhi(p, seg_magic), sg = p - j * (uint32_t)segs;
    emit(a.rows - 1 - (int)(16u * sg + e), (((word >> (8 * (e & 3))) & 0xffu) << 24) | j);
  }
}

__device__ __forceinline__ void load_pieces(const KStrongArgs& a, const uint8_t* img, const uint32_t p0, const uint32_t step,
                                            const uint32_t n_pieces, const uint32_t seg_magic, const int segs,
                                            uint32_t (&w)[kExtractPieces][4]) {
  if (a.stride == 16 * segs) {                               // no row pitch: piece p is bytes 16 p .. of the image
#pragma unroll
    for (int u = 0; u < kExtractPieces; u++) {
      const uint32_t p = min(p0 + step * u, n_pieces - 1u);
      const u32x4 v = __builtin_nontemporal_load((const u32x4*)(img + (size_t)p * 16u));
      w[u][0] = v.x; w[u][1] = v.y; w[u][2] = v.z; w[u][3] = v.w;
    }
    return;
  }
#pragma unroll
  for (int u = 0; u < kExtractPieces; u++) {
    const uint32_t p = min(p0 + step * u, n_pieces - 1u);
    const uint32_t j = segs == 1 ? p : __umulhi(p, seg_magic), sg = p - j * (uint32_t)segs;
    const u32x4 v = __builtin_nontemporal_load((const u32x4*)(img + (size_t)j * a.stride + 16u * sg));
    w[u][0] = v.x; w[u][1] = v.y; w[u][2] = v.z; w[u][3] = v.w;
  }
}

// The k strongest of one azimuth's n <= kCandCap candidate keys (fetch(j), j < n, any order) -> row_keys / row_valid.
// list: [kCandCap + 8] dwords of this wavefront's LDS.  Survivors = the min(n, k) largest keys -- the lexicographic
// (intensity, range) cut of FilterKstrongest (radar_filters.cpp:214-229), ties toward the larger range; the slot of a
// survivor = its rank among the survivors beyond min_range_bin (getPeaksFilteredPointCloud(cloud, false), :309-337).
template <typename Fetch>
__device__ __forceinline__ void select_row(const KStrongArgs& a, const long long row, const int n, uint32_t* list, const int lane,
                                           Fetch&& fetch) {
  uint32_t* bits = list + kCandCap + 4;
  constexpr int NR = kCandCap / 64;
  uint32_t key[NR];
#pragma unroll
  for (int i = 0; i < NR; i++) {
    const int j = i * 64 + lane;
    key[i] = 0xFFFFFFFFu;                                    // the padding ranks above every key
    if (i * 64 < n) {
      if (j < n) key[i] = fetch(j);
      list[j] = key[i];
    }
  }
  if (lane < 2) bits[lane] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int k = a.k, drop = n - min(n, k), nq = (n + 3) & ~3;
  int rank[NR];
  bool beyond[NR];
#pragma unroll
  for (int i = 0; i < NR; i++) {
    rank[i] = -1;
    beyond[i] = false;
    if (i * 64 < n) {                                        // wave-uniform
      if (i * 64 + lane < n) {
        int c = 0;
        for (int q = 0; q < nq; q += 4) {
          const uint4 x = *(const uint4*)(list + q);         // same address in every lane: LDS broadcast
          c += (x.x < key[i]) + (x.y < key[i]) + (x.z < key[i]) + (x.w < key[i]);
        }
        rank[i] = c - drop;
      }
      beyond[i] = rank[i] >= 0 && (int)(key[i] & 0xFFFFFFu) > a.min_range_bin;
      if (beyond[i]) atomicOr(&bits[rank[i] >> 5], 1u << (rank[i] & 31));      // (the survivors' ranks are < k <= 64)
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const uint32_t m0 = bits[0], m1 = bits[1];
#pragma unroll
  for (int i = 0; i < NR; i++) {
    if (beyond[i]) {
      const int rk = rank[i];
      const int idx = rk < 32 ? __popc(m0 & ((1u << rk) - 1u)) : __popc(m0) + __popc(m1 & ((1u << (rk - 32)) - 1u));
      a.row_keys[row * k + idx] = key[i];
    }
  }
  if (lane == 0) {
    a.row_valid[row * 2] = __popc(m0) + __popc(m1);
    a.row_valid[row * 2 + 1] = 0;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the list is reused by the wavefront's next row
  __builtin_amdgcn_wave_barrier();
}

// Two azimuths with at most 32 candidates each on one wavefront: lanes 0-31 take row[0], lanes 32-63 row[1] (the typical
// radar azimuth holds two or three dozen bins >= z_min, so select_row leaves half of its lanes idle).  fetch(h, j) = key j
// of the half's row; list: [64] keys + [2] bitmap words.
template <typename Fetch>
__device__ __forceinline__ void select_pair(const KStrongArgs& a, const long long row0, const long long row1, const int n0,
                                            const int n1, uint32_t* list, const int lane, Fetch&& fetch) {
  const int half = lane >> 5, h = lane & 31;
  const int n = half ? n1 : n0;
  const long long row = half ? row1 : row0;
  uint32_t* bits = list + 64;
  const uint32_t key = h < n ? fetch(half, h) : 0xFFFFFFFFu;
  list[lane] = key;
  if (h == 0) bits[half] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int k = a.k, drop = n - min(n, k);
  const int nq = (max(n0, n1) + 3) & ~3;                     // wave-uniform
  const uint32_t* mine = list + 32 * half;
  int c = 0;
  for (int q = 0; q < nq; q += 4) {
    const uint4 x = *(const uint4*)(mine + q);               // two addresses per instruction
    c += (x.x < key) + (x.y < key) + (x.z < key) + (x.w < key);
  }
  const int rank = h < n ? c - drop : -1;
  const bool beyond = rank >= 0 && (int)(key & 0xFFFFFFu) > a.min_range_bin;
  if (beyond) atomicOr(&bits[half], 1u << rank);             // (survivors' ranks are < min(n, k) <= 32)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const uint32_t m = bits[half];
  if (beyond) a.row_keys[row * k + __popc(m & ((1u << rank) - 1u))] = key;
  if (h == 0) {
    a.row_valid[row * 2] = __popc(m);
    a.row_valid[row * 2 + 1] = 0;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// a row whose list overflowed: its 16-column tile goes to kstrongest_cols_kernel (once)
__device__ __forceinline__ void flag_tile(const KStrongArgs& a, const int b, const int r, const int tiles, uint32_t* tile_flag,
                                          int32_t* work_n, uint32_t* work) {
  const uint32_t id = (uint32_t)b * (uint32_t)tiles + (uint32_t)((a.rows - 1 - r) / kColsTile);
  if (atomicExch(&tile_flag[id], 1u) == 0u) work[atomicAdd(work_n, 1)] = id;
}

// ---- small batches: many workgroups per image, the lists in global memory (one atomic per candidate) ---------------
__global__ __launch_bounds__(256) void kstrong_extract_kernel(const KStrongArgs a, const uint32_t seg_magic, const int segs,
                                                              int32_t* __restrict__ cand_cnt, uint32_t* __restrict__ cand) {
  const int b = a.batch0 + blockIdx.y;
  const uint8_t* img = a.polar + (long long)b * a.batch_stride;
  const uint32_t n_pieces = (uint32_t)a.cols * (uint32_t)segs;
  const uint32_t p0 = blockIdx.x * (256u * kExtractPieces) + threadIdx.x;
  uint32_t w[kExtractPieces][4];
  load_pieces(a, img, p0, 256u, n_pieces, seg_magic, segs, w);
  extract_candidates(a, w, p0, 256u, n_pieces, seg_magic, segs, [&](const int r, const uint32_t key) {
    const long long row = (long long)b * a.rows + r;
    const int slot = atomicAdd(&cand_cnt[row], 1);
    if (slot < kCandCap) cand[row * kCandCap + slot] = key;
  });
}

__global__ __launch_bounds__(256) void kstrong_select_kernel(const KStrongArgs a, const int tiles, const int32_t* __restrict__ cand_cnt,
                                                             const uint32_t* __restrict__ cand, uint32_t* tile_flag,
                                                             int32_t* work_n, uint32_t* work, uint32_t* stats) {
  __shared__ __attribute__((aligned(16))) uint32_t lists[kRowsPerBlock][kCandCap + 8];   // keys | 2 bitmap words
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int r = blockIdx.x * kRowsPerBlock + wave;
  if (r >= a.rows) return;                                   // no workgroup barrier below
  const int b = a.batch0 + blockIdx.y;
  const long long row = (long long)b * a.rows + r;
  const int n = __builtin_amdgcn_readfirstlane(cand_cnt[row]);
  // how dense the batch is: every 16th azimuth reports, for 16 (an atomic per azimuth on 64 counters took longer than the
  // selection itself: 0.48 ms for 255 sweeps)
  if (stats && lane == 0 && (r & 15) == 0) atomicAdd(&stats[(blockIdx.x + blockIdx.y) & 63], 16u * (uint32_t)n);
  if (n > kCandCap) {                                        // the list is incomplete: the row needs its raw bytes
    if (lane == 0) flag_tile(a, b, r, tiles, tile_flag, work_n, work);
    return;
  }
  select_row(a, row, n, lists[wave], lane, [&](const int j) { return cand[row * kCandCap + j]; });
}

// ---- large batches: ONE workgroup streams a whole image; the lists of all its azimuths sit in LDS (the first lds_cap
// keys of each; the rest, up to kCandCap, in global memory), so a candidate costs an LDS atomic, and the same workgroup
// then picks the k strongest of every list: one kernel, nothing but the keys written.  Two workgroups per CU: one streams
// while the other selects.
constexpr int kImgWaves = 8;

__global__ __launch_bounds__(64 * kImgWaves, 4) void kstrong_image_kernel(const KStrongArgs a, const uint32_t seg_magic, const int segs,
                                                                           const int tiles, const int lds_cap,
                                                                           uint32_t* __restrict__ cand, uint32_t* tile_flag,
                                                                           int32_t* work_n, uint32_t* work, uint32_t* stats) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int rows4 = (a.rows + 3) & ~3;
  uint32_t* cnt = (uint32_t*)smem;                           // [rows]
  uint32_t* lists = cnt + rows4;                             // [rows][lds_cap]
  uint32_t* mine = lists + (size_t)a.rows * lds_cap + wave * (kCandCap + 8);
  const uint32_t n_pieces = (uint32_t)a.cols * (uint32_t)segs;
  constexpr uint32_t kThreads = 64 * kImgWaves;
  for (int b = blockIdx.x; b < a.batch; b += gridDim.x) {
    const uint8_t* img = a.polar + (long long)b * a.batch_stride;
    for (int i = threadIdx.x; i < a.rows; i += kThreads) cnt[i] = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_pieces; base += kThreads * kExtractPieces) {
      uint32_t w[kExtractPieces][4];
      load_pieces(a, img, base + threadIdx.x, kThreads, n_pieces, seg_magic, segs, w);
      extract_candidates(a, w, base + threadIdx.x, kThreads, n_pieces, seg_magic, segs, [&](const int r, const uint32_t key) {
        const int slot = (int)atomicAdd(&cnt[r], 1u);
        if (slot < lds_cap) lists[r * lds_cap + slot] = key;
        else if (slot < kCandCap) cand[((long long)b * a.rows + r) * kCandCap + slot] = key;
      });
    }
    __syncthreads();                                         // (also orders the overflow stores before the loads below)
    if (stats) {                                             // how dense the batch is: the image's candidates, 64 counters (no hot address)
      int local = 0;
      for (int i = threadIdx.x; i < a.rows; i += kThreads) local += (int)cnt[i];
      const int incl = wave_incl_scan_i32(local);
      if (lane == 63) atomicAdd(&stats[(b * kImgWaves + wave) & 63], (uint32_t)incl);
    }
    const int pair_cap = min(32, lds_cap);                   // (a pair's keys all come from LDS)
    for (int r = wave; r < a.rows; r += 2 * kImgWaves) {
      const int r1 = r + kImgWaves;
      const int n0 = (int)cnt[r], n1 = r1 < a.rows ? (int)cnt[r1] : 0;
      if (r1 < a.rows && n0 <= pair_cap && n1 <= pair_cap) {
        select_pair(a, (long long)b * a.rows + r, (long long)b * a.rows + r1, n0, n1, mine, lane,
                    [&](const int half, const int j) { return lists[(half ? r1 : r) * lds_cap + j]; });
        continue;
      }
      for (int i = 0; i < 2; i++) {
        const int rr = i ? r1 : r, n = i ? n1 : n0;
        if (rr >= a.rows) break;
        const long long row = (long long)b * a.rows + rr;
        if (n > kCandCap) {
          if (lane == 0) flag_tile(a, b, rr, tiles, tile_flag, work_n, work);
          continue;
        }
        select_row(a, row, n, mine, lane, [&](const int j) { return j < lds_cap ? lists[rr * lds_cap + j] : cand[row * kCandCap + j]; });
      }
    }
    __syncthreads();                                         // the counters are cleared for the next image
  }
}

// Persistent workgroups, two per CU.  With a work list: tile ids (image * tiles + tile), taken round robin.  Without:
// every tile of every image -- workgroup n lives on XCD n % 8 and is that XCD's slot n / 8; XCD x takes the images
// b = x (mod 8), tile after tile, its slots striding through that sequence together, so the (up to) eight tiles that
// share a 128-byte line are read by neighbouring slots of one XCD at about the same time.  While a workgroup runs the
// row algorithm on the tile in LDS, the 16-byte pieces of its NEXT tile are already in flight into registers.
template <int NCHUNK, bool MASK>
__global__ __launch_bounds__(64 * kColsWaves, 4) void kstrongest_cols_kernel(const KStrongArgs a, const int tiles,
                                                                              const uint32_t* __restrict__ work,
                                                                              const int32_t* __restrict__ work_n) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int xcd = blockIdx.x % kXcds, slot = blockIdx.x / kXcds, slots = gridDim.x / kXcds;
  const int nq = work ? *work_n : ((a.batch - xcd + kXcds - 1) / kXcds) * tiles;       // items of this workgroup's sequence
  const int q0 = work ? (int)blockIdx.x : slot, dq = work ? (int)gridDim.x : slots;
  auto locate = [&](const int q, int& b, int& tile) {
    if (work) { const uint32_t id = work[q]; b = (int)(id / (uint32_t)tiles); tile = (int)(id - (uint32_t)b * (uint32_t)tiles); }
    else { const int im = q / tiles; b = im * kXcds + xcd; tile = q - im * tiles; }
  };
  const int bins = a.cols, cols16 = (bins + 15) & ~15;
  const int kpad = max((a.k + 3) & ~3, 64);
  constexpr int kScratch = kstrong_scratch_bytes(NCHUNK);
  uint8_t* tbase = smem;                                     // [kColsTile][cols16]: row lr = source column c0 + 15 - lr
  uint32_t* hist = (uint32_t*)(smem + kColsTile * cols16 + wave * (kScratch + kpad * 4));
  uint32_t* list = hist + kScratch / 4;
  constexpr int GP = (NCHUNK * 256 + 64 * kColsWaves - 1) / (64 * kColsWaves);   // groups of 4 bins per thread
  uint32_t rw[GP][4][4];
  // one 32-bit byte offset per group and thread (groups past the last bin re-read the last one; their tile bytes are
  // zeroed below), the bin row i and the tile folded into the wave-uniform base: SGPR base + VGPR offset addressing
  uint32_t goff[GP];
#pragma unroll
  for (int p = 0; p < GP; p++)
    goff[p] = (uint32_t)min((int)threadIdx.x + p * 64 * kColsWaves, (bins >> 2) - 1) * 4u * (uint32_t)a.stride;
  auto issue = [&](const int q) {                            // the pieces of item q: bins 4 g .. 4 g + 3, 16 source columns
    int b, tile;
    locate(q, b, tile);
    const uint8_t* src = a.polar + (long long)b * a.batch_stride + tile * kColsTile;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint8_t* bi = src + (long long)i * a.stride;
#pragma unroll
      for (int p = 0; p < GP; p++) {
        const u32x4 v = *(const u32x4*)(bi + (size_t)goff[p]);
        rw[p][i][0] = v.x; rw[p][i][1] = v.y; rw[p][i][2] = v.z; rw[p][i][3] = v.w;
      }
    }
  };
  int q = q0;
  if (q < nq) issue(q);
  while (q < nq) {
#pragma unroll
    for (int p = 0; p < GP; p++) {
      const int g = threadIdx.x + p * 64 * kColsWaves;
      if (4 * g < cols16) {
#pragma unroll
        for (int d = 0; d < 4; d++) {                        // source columns c0 + 4 d .. + 3 of bins 4 g .. 4 g + 3
          const uint32_t w0 = rw[p][0][d], w1 = rw[p][1][d], w2 = rw[p][2][d], w3 = rw[p][3][d];
          const uint32_t t0 = __builtin_amdgcn_perm(w1, w0, 0x05010400u), t1 = __builtin_amdgcn_perm(w1, w0, 0x07030602u);
          const uint32_t t2 = __builtin_amdgcn_perm(w3, w2, 0x05010400u), t3 = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
          uint32_t colw[4];                                  // colw[e] = column c0 + 4 d + e as {bin 4g, +1, +2, +3}
          colw[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u); colw[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
          colw[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u); colw[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
#pragma unroll
          for (int e = 0; e < 4; e++)
            *(uint32_t*)(tbase + (kColsTile - 1 - (4 * d + e)) * cols16 + 4 * g) = 4 * g < bins ? colw[e] : 0u;
        }
      }
    }
    __syncthreads();
    int b, tile;
    locate(q, b, tile);
    const int qn = q + dq;
    if (qn < nq) issue(qn);
    const int r0 = a.rows - kColsTile - tile * kColsTile;    // output row of tile row 0
    for (int lr = wave; lr < kColsTile; lr += kColsWaves)
      kstrong_row<NCHUNK, true, MASK, true>(a, r0 + lr, b, a.polar, tbase + lr * cols16, hist, list, lane);
    __syncthreads();                                         // every row of the tile has been consumed
    q = qn;
  }
}

// ---------------------------------------------------------------------------------------------
// cloud compaction: radar_filters.cpp:309-337
// ---------------------------------------------------------------------------------------------
struct CloudArgs {
  const int32_t* sel_range;
  const uint8_t* sel_intensity;
  const int32_t* sel_count;
  const uint8_t* is_peak;
  const int32_t* row_valid;  // [batch][rows][2] written by kstrongest_rows_kernel
  const double* cos_t;      // [rows]
  const double* sin_t;
  int rows, k, min_range_bin;
  double range_res;
  float* xyzi;              // [batch][rows*k][4]
  int32_t* n_points;        // [batch]
  float* xyzi_peaks;
  int32_t* n_peaks;
};

constexpr int kCloudSplit = 8;   // workgroups per image: each scans all row counts, writes its row slice
constexpr int kCfarCloudSplit = 8;    // CA-CFAR clouds: 25 slices (16 rows per workgroup) measured slower, 0.146 vs 0.137 ms per 512 sweeps: every workgroup re-scans the row counts

__global__ __launch_bounds__(256) void kstrong_cloud_kernel(const CloudArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  int32_t* row_off = (int32_t*)smem;                 // [rows + 1]
  __shared__ int32_t wave_tot[8];
  const int b = blockIdx.x;
  const bool peaks = blockIdx.y == 1;
  float* out = peaks ? a.xyzi_peaks : a.xyzi;
  int32_t* nout = peaks ? a.n_peaks : a.n_points;
  if (!out && !nout) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long ibase = (long long)b * a.rows * a.k;
  // This workgroup writes the rows [rbeg, rend) of the image: it needs the number of points in the rows before
  // its slice (one block reduction over all row counts) and a prefix inside the slice (one wavefront).
  const int rows_per = (a.rows + kCloudSplit - 1) / kCloudSplit;
  const int rbeg = blockIdx.z * rows_per, rend = min(a.rows, rbeg + rows_per);
  const int which = peaks ? 1 : 0;
  int before = 0, total = 0;
  for (int r = threadIdx.x; r < a.rows; r += 256) {
    const int v = a.row_valid[((long long)b * a.rows + r) * 2 + which];
    total += v;
    before += r < rbeg ? v : 0;
  }
  before = wave_sum_i32(before);
  total = wave_sum_i32(total);
  if (lane == 0) { wave_tot[wave] = before; wave_tot[4 + wave] = total; }
  __syncthreads();
  before = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  total = wave_tot[4] + wave_tot[5] + wave_tot[6] + wave_tot[7];
  if (threadIdx.x == 0 && nout && blockIdx.z == 0) nout[b] = total;
  if (!out) return;
  if (wave == 0) {
    int run = before;
    for (int r0 = rbeg; r0 < rend; r0 += 64) {
      const int r = r0 + lane;
      const int v = r < rend ? a.row_valid[((long long)b * a.rows + r) * 2 + which] : 0;
      const int incl = wave_incl_scan_i32(v);
      if (r < rend) row_off[r] = run + incl - v;
      run += __builtin_amdgcn_readlane(incl, 63);
    }
  }
  __syncthreads();
  // one wavefront per row of this workgroup's slice writes its points in (intensity,range) order
  const double range_res_half = a.range_res / 2.0;
  // Every load of a row's first 64 slots is issued before any of them is used (the slots past the row's count
  // exist and are simply ignored), so a row costs one memory round trip instead of three dependent ones.
  for (int r = rbeg + wave; r < rend; r += 4) {
    const long long rb = ibase + (long long)r * a.k;
    const int cnt = a.sel_count[(long long)b * a.rows + r];
    const double cos_t = a.cos_t[r], sin_t = a.sin_t[r];
    int range0 = 0;
    uint8_t inten0 = 0, pk0 = 1;
    if (lane < a.k) {
      range0 = a.sel_range[rb + lane];
      inten0 = a.sel_intensity[rb + lane];
      if (peaks) pk0 = a.is_peak[rb + lane];
    }
    int base = row_off[r];
    for (int j0 = 0; j0 < cnt; j0 += 64) {
      const int j = j0 + lane;
      bool ok = false;
      int range = 0;
      uint8_t inten = 0;
      if (j < cnt) {
        if (j0 == 0) { range = range0; inten = inten0; ok = range > a.min_range_bin && pk0; }
        else {
          range = a.sel_range[rb + j];
          inten = a.sel_intensity[rb + j];
          ok = range > a.min_range_bin && (!peaks || a.is_peak[rb + j]);
        }
      }
      const unsigned long long bal = __ballot(ok);
      if (ok) {
        const int idx = base + __popcll(bal & ((1ull << lane) - 1ull));
        const double rho = range_res_half + a.range_res * (double)range;
        float4 p;
        p.x = (float)(rho * cos_t);
        p.y = (float)(rho * sin_t);
        p.z = 0.f;
        p.w = (float)inten;
        ((float4*)out)[(long long)b * a.rows * a.k + idx] = p;
      }
      base += __popcll(bal);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// CA-CFAR: cfar.cpp:35-83
// ---------------------------------------------------------------------------------------------
struct CfarArgs {
  const uint8_t* polar;
  int rows, cols, stride, batch;
  long long batch_stride;
  long long total_rows;
  int window, guard;
  double scaling, range_res, static_threshold, min_distance, max_distance;
  // bitmap output (standalone filter; cacfar_cloud_kernel compacts it)
  unsigned long long* det_bits;   // [batch][rows][words]  (words = ceil(cols/64))
  int32_t* det_count;             // [batch][rows]
  int words;
  // key output (batched odometry): row r of image b leaves its detections as (intensity << 24 | bin), ascending bins, at
  // row_keys[(b rows + r) kcap ...] and their number in row_cnt[2 (b rows + r)] -- the layout surface_prep_kernel
  // already takes from the k-strongest sweep, so no cloud kernel runs in between
  uint32_t* row_keys;
  int32_t* row_cnt;
  int kcap;
  int list_cap;                   // entries of a wavefront's candidate list (cacfar_cols_kernel: shorter than a chunk)
  int thr_i, bin_lo, bin_hi;      // candidate pre-test in integers: intensity >= thr_i, bin_lo <= bin < bin_hi
  int need_cols;                  // bins a row's arithmetic can touch: min(cols, bin_hi - 1 + guard + window), in 16s
  int colsp;                      // need_cols in whole 1024-bin chunks: the row's length in LDS
  int lut_ok;                     // lut[] decides a candidate with two full windows in integers
  int pre_on;                     // lower-bound pre-filter on (pa*, pb*, kappa_lb valid)
  int pa0, pa1, pb0, pb1;         // quads [2H + pa0, 2H + pa1) / [2H + pb0, 2H + pb1) lie in the trailing / forwarding
                                  // window of EVERY bin of the 8-bin block H
  int pad_lo, pad_hi;             // guard entries of the prefix table below bin 0 / beyond the row
  float kappa_lb;                 // scaling / (2 window), rounded down a little
  uint32_t lut[256];              // see cfar_build_lut
};

// One wavefront per azimuth row, persistent: a wavefront walks rows g, g + W, g + 2 W, ... and requests the NEXT row's
// 16-byte pieces before it works on the current one, so the HBM round trip hides behind its own arithmetic (the version
// before ran one row per wavefront and 12 wavefronts per CU: a third of a row's residence was the wait for its loads).
//
// Per row:
//  A. the row's bytes and the exact uint32 prefix sums of their squares go to LDS: 16 bins are summed inside a lane
//     (v_dot4), ONE wave scan per 1024 bins places the lanes, the table keeps every fourth prefix (P(x) = P4[x / 4] + the
//     squares of up to three bytes of one LDS word).  Only the bins the arithmetic can reach are read at all: with
//     radar_driver.cpp:54's 400 m cap a Kvarntorp row ends at bin 2286 + guard + window.
//  B. candidates.  cfar.cpp:45 lets a bin through when intensity > static_threshold inside the range window; here a bin
//     must ALSO beat a lower bound of its own CFAR threshold: the aligned 4-bin groups that lie inside the trailing /
//     forwarding window of every bin of an 8-bin block (56 of 80 bins for guard 10, window 40) give S_lb <= S_t + S_f from
//     four prefix reads per block, and I^2 > scaling (S_t / n_t + S_f / n_f) / 2 >= scaling S_lb / (2 w) is necessary for
//     a detection (n_t, n_f <= w).  So each block compares its bytes (SWAR) against max(thr_i, floor(sqrt(kappa S_lb)))
//     instead of thr_i alone: a quarter of the candidates survive on the synthetic Kvarntorp rows, and nothing that can
//     fire is lost.  Survivors are listed in LDS in bin order (a wave scan of the popcounts places the lanes).
//  C. the list is evaluated one candidate per lane, 64 at a time, carrying the remainder from chunk to chunk so that
//     every round but the last is full.  With both windows full the decision is an integer compare: with S = S_t + S_f
//     the reference computes  I^2 > scaling ((S_t / w + S_f / w) / 2)  in fp64, which differs from the exact
//     I^2 > S scaling / (2 w) by at most 5 roundings of 2^-53 -- so with B = I^2 2 w / scaling it fires for S <= B - 1e-3
//     and does not for S >= B + 1e-3, and lut[I] = (T << 1 | amb) with T = ceil(B - 1e-3), amb = an integer lies within
//     1e-3 of B:  fires <=> 2 S + 1 < lut[I];  2 S + 1 == lut[I] (practically never) and bins whose windows the row's ends
//     cut take cfar.cpp:45-60 literally in fp64.
//  D. detections leave in list (= bin) order: as keys for surface_prep_kernel (batched odometry) or as a bit per bin.
constexpr int kCfarListSlack = 64 + 64;     // list entries = one chunk's bins + the carried remainder (a 576-entry list with windowed
                                           // appends admits a fifth workgroup per CU and measured 4 % SLOWER: the kernel is
                                           // bound by VALU issue, not by latency)
__host__ __device__ inline size_t cfar_wave_lds(int colsp, int pad_lo, int pad_hi, bool keys, int chunk_bins) {
  // P4 u32[pad_lo + colsp / 4 + 1 + pad_hi] | raw u8[colsp + 16] | det u32[colsp / 32] (bitmap output) | list u16[chunk + slack]
  size_t b = ((size_t)(pad_lo + colsp / 4 + 1 + pad_hi) * 4 + 15) & ~(size_t)15;
  b += (size_t)colsp + 16;
  if (!keys) b += (size_t)colsp / 8;
  b += (size_t)(chunk_bins + kCfarListSlack) * 2;
  return (b + 15) & ~(size_t)15;
}
// D = dwords a lane owns per chunk (4, 6 or 8: a chunk is 256 D bins), NCH = chunks the registers hold.  The per-chunk
// overhead (two wave scans, the list loop, the round bookkeeping) is paid per CHUNK, so the host picks the D that covers
// the reachable bins with the fewest chunks: 2336 bins of a Kvarntorp row are 2 chunks of 1536 (D = 6) instead of 3 of
// 1024, of which the third held 18 busy lanes and cost 18 % of the kernel.
// DL = dwords per lane of the LAST chunk (DL <= D; DL < D only with exactly NCH chunks): 2336 reachable bins are a chunk
// of 1536 (D = 6) and one of 1024 (DL = 4) -- ten dwords per lane and row instead of twelve.
// One row of CA-CFAR on one wavefront (steps A .. D of cacfar_rows_kernel's comment): the row's bytes are in `cur` (lane l
// owns the dwords [j CB + l 4 D_j ..) of chunk j) and -- STAGED -- already in LDS at `raw` (the reachable a.need_cols bytes + 16 zeros).
// grow = the row's index in the OUTPUT (image * rows + azimuth); key_base = grow * kcap.
template <int D, int NCH, int DL, bool KEYS, bool PRE, bool STAGED, typename AfterSwar>
__device__ __forceinline__ void cfar_row(const CfarArgs& a, const int lane, const uint32_t* lut, uint32_t* P4, uint8_t* raw, uint32_t* det32,
                                         unsigned short* list, uint32_t (&cur)[NCH][D], const int nch, const long long grow,
                                         const long long key_base, AfterSwar&& after_swar) {
  constexpr int CB = 256 * D;
  auto DJ = [](int j) { return (DL != D && j == NCH - 1) ? DL : D; };
  const int colsp = a.colsp;
  auto lds_put = [&](void* at, const uint32_t (&v)[D], int dj) {               // the first dj dwords of v
    if (dj % 4 == 0) {
#pragma unroll
      for (int k = 0; k < D / 4; k++) if (4 * k < dj) ((uint4*)at)[k] = make_uint4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    } else {
#pragma unroll
      for (int k = 0; k < D / 2; k++) if (2 * k < dj) ((uint2*)at)[k] = make_uint2(v[2 * k], v[2 * k + 1]);
    }
  };
  // LDS hand-over between the lanes of this wavefront.  The fences name the LDS address space only: a plain wavefront
  // fence makes the compiler wait for vmcnt(0) too, i.e. for the NEXT row's loads that were just requested.
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };
  auto P = [&](int x) -> uint32_t {                                         // sum_{q < x} I_q^2, 0 <= x <= colsp
    const int q = x >> 2, rr = x & 3;
    // the first rr bytes of the word (none for rr = 0; the word of x = colsp is the 16 bytes of padding): no branch
    const uint32_t wd = *(const uint32_t*)(raw + 4 * q) & ((1u << (8 * rr)) - 1u);
    return __builtin_amdgcn_udot4(wd, wd, P4[q], false);
  };
#ifdef CFEAR_CFAR_TIMING
  long long tq[6] = {0, 0, 0, 0, 0, 0}; int ctot = 0, nrounds = 0;
#define CFAR_T0() long long t_ = __builtin_readcyclecounter()
#define CFAR_T(k) { const long long n_ = __builtin_readcyclecounter(); tq[k] += n_ - t_; t_ = n_; }
#else
#define CFAR_T0()
#define CFAR_T(k)
#endif
  CFAR_T0();
    // ---- A: bytes + prefix sums of squares -> LDS ---------------------------------------------------------------
    uint32_t run = 0;
#pragma unroll
    for (int j = 0; j < NCH; j++) {
      if (j >= nch) break;
      const int dj = DJ(j);
      const int pos = j * CB + lane * 4 * dj;
      if (!STAGED) lds_put(raw + pos, cur[j], dj);                            // (a staged row already sits there)
      uint32_t pre[D];                                                      // sums of squares before each of the lane's quads
      uint32_t acc = 0;
#pragma unroll
      for (int d = 0; d < D; d++) if (d < dj) { pre[d] = acc; acc += __builtin_amdgcn_udot4(cur[j][d], cur[j][d], 0u, false); }
      const int incl = wave_incl_scan_i32((int)acc);
      const uint32_t base = run + (uint32_t)incl - acc;                     // sum before this lane's first bin
#pragma unroll
      for (int d = 0; d < D; d++) if (d < dj) pre[d] += base;
      lds_put(P4 + (pos >> 2), pre, dj);
      run += (uint32_t)__builtin_amdgcn_readlane(incl, 63);
    }
    if (lane == 0) P4[colsp >> 2] = run;                                    // P(colsp)
    for (int i = lane; i < a.pad_hi; i += 64) P4[(colsp >> 2) + 1 + i] = run;
    if (!KEYS) for (int i = lane; i < colsp / 32; i += 64) det32[i] = 0u;
    wave_sync();
    CFAR_T(0);
    // ---- B + C -------------------------------------------------------------------------------------------------
    int C = 0, ndet = 0;
    // W candidates per lane and round (W = 1, 2): candidate k0 + 64 w + lane, w < W.  The decision of one candidate is a chain
    // of dependent LDS reads (list -> byte and four prefixes -> table); with two per lane the two chains interleave, and a
    // row's ~100 survivors take ONE round of 128 instead of a full and a partial round of 64 (the rounds were 38 % of a
    // row's cycles).  Detections still leave in list (= bin) order: the first 64 candidates' keys, then the second 64's.
    auto rounds = [&](auto w_tag, int k0, int cnt) {
      constexpr int W = decltype(w_tag)::value;
      bool act[W], valid[W], full[W], det[W], slow[W], edge[W];
      int bin[W], nt[W], nf[W];
      uint32_t v[W], st[W], sf[W];
#pragma unroll
      for (int w = 0; w < W; w++) {
        act[w] = 64 * w + lane < cnt;
        bin[w] = 0; v[w] = 0;
        if (act[w]) bin[w] = list[k0 + 64 * w + lane];
      }
#pragma unroll
      for (int w = 0; w < W; w++) if (act[w]) v[w] = raw[bin[w]];
#pragma unroll
      for (int w = 0; w < W; w++) {
        const int t1 = bin[w] - a.guard, t0 = t1 - a.window, f0 = bin[w] + a.guard, f1 = f0 + a.window;   // cfar.cpp:48-53
        const int lt0 = max(t0, 0), lf1 = min(f1, a.cols);                   // the windows as the row's ends cut them
        nt[w] = t1 - lt0; nf[w] = lf1 - f0;
        // getMean over an empty window is 0 / 0 = NaN: no detection (a window "ending" before bin 0 compares a size_t index
        // with a negative end in the reference -- undefined there, no detection here)
        valid[w] = act[w] && nt[w] > 0 && nf[w] > 0;
        st[w] = P(max(t1, 0)) - P(lt0); sf[w] = P(lf1) - P(min(f0, colsp));
        full[w] = nt[w] == a.window && nf[w] == a.window;
        det[w] = false; slow[w] = false; edge[w] = false;
      }
      if (PRE) {
        bool any_edge = false;
#pragma unroll
        for (int w = 0; w < W; w++) {
          // both windows full (every bin but the row's ends): the integer decision of step C
          const uint32_t X = 2u * (st[w] + sf[w]) + 1u, L = lut[v[w]];
          det[w] = valid[w] && full[w] && X < L;
          slow[w] = valid[w] && full[w] && X == L;
          edge[w] = valid[w] && !full[w];
          any_edge = any_edge || edge[w];
        }
        if (__ballot(any_edge)) {
#pragma unroll
          for (int w = 0; w < W; w++) {
            // a cut window: I^2 > scaling (S_t / n_t + S_f / n_f) / 2  <=>  I^2 2 n_t n_f > scaling (S_t n_f + S_f n_t) up to
            // six roundings of 2^-53; the integers on both sides are exact in fp64, so unless the two sides agree to 1e-12
            // the comparison is decided without the reference's divisions
            const double lhs = (double)(v[w] * v[w]) * (double)(2 * nt[w] * nf[w]);
            const double rhs = a.scaling * ((double)st[w] * (double)nf[w] + (double)sf[w] * (double)nt[w]);
            const double d = lhs - rhs;
            const bool sure = fabs(d) > fabs(rhs) * 1e-12;
            if (edge[w]) { det[w] = sure && d > 0.0; slow[w] = !sure; }
          }
        }
      } else {
#pragma unroll
        for (int w = 0; w < W; w++) slow[w] = valid[w];
      }
      bool any_slow = false;
#pragma unroll
      for (int w = 0; w < W; w++) any_slow = any_slow || slow[w];
      if (__ballot(any_slow)) {
#pragma unroll
        for (int w = 0; w < W; w++)
          if (slow[w]) {                                                      // cfar.cpp:55-60, literally
            const double trailing_mean = (double)st[w] / (double)nt[w];
            const double forwarding_mean = (double)sf[w] / (double)nf[w];
            const double mean = (trailing_mean + forwarding_mean) / 2.0;      // :56
            const double threshold = a.scaling * mean;                        // :58
            det[w] = (double)(v[w] * v[w]) > threshold;                       // :59-60
          }
      }
#pragma unroll
      for (int w = 0; w < W; w++) {
        if (KEYS) {
          const unsigned long long dm = __ballot(det[w]);
          const int at = ndet + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u));
          if (det[w] && at < a.kcap) a.row_keys[key_base + at] = (v[w] << 24) | (uint32_t)bin[w];
          ndet += __popcll(dm);
        } else {
          if (det[w]) atomicOr(&det32[bin[w] >> 5], 1u << (bin[w] & 31));
        }
      }
#ifdef CFEAR_CFAR_TIMING
      nrounds++;
#endif
    };
    auto round = [&](int k0, int cnt) { rounds(std::integral_constant<int, 1>{}, k0, cnt); };
    auto round2 = [&](int k0, int cnt) { rounds(std::integral_constant<int, 2>{}, k0, cnt); };
    // candidate test "byte >= t" for four bytes at once: with tl = t & 127 and y = ((x & 0x7f..) | 0x80..) - tl * 0x0101..,
    // bit 7 of a byte of y says (x & 127) >= tl; the verdict is y & x for t >= 128 and y | x below.
    const int cj_lo = a.bin_lo / CB, cj_hi = (a.bin_hi + CB - 1) / CB;      // chunks that hold bins of the range window
    uint32_t cm[NCH];                                                       // candidate bits of the lane, chunk by chunk
#pragma unroll
    for (int j = 0; j < NCH; j++) {
      cm[j] = 0u;
      if (j >= cj_hi) break;
      if (j < cj_lo) continue;
      const int dj = DJ(j), LB = 4 * dj;                                    // bytes (bins) of the lane in this chunk
      const int pos = j * CB + lane * LB;
      uint32_t cmask = 0;                                                   // bit per bin of the lane (LB <= 32)
#pragma unroll
      for (int h = 0; h < D / 2; h++) {
        if (2 * h >= dj) break;
        int t = a.thr_i;
        if (PRE) {
          const uint32_t* pq = P4 + (pos >> 2) + 2 * h;                     // quad 2 H of this 8-bin block
          const uint32_t slb = (pq[a.pa1] - pq[a.pa0]) + (pq[a.pb1] - pq[a.pb0]);
          const float f = __builtin_amdgcn_sqrtf((float)slb * a.kappa_lb);  // <= sqrt(kappa S_lb): kappa_lb carries the slack
          t = max(t, (int)f);
        }
        t = min(t, 255);                                                     // (a threshold above 255 lets 255 through: harmless, the
                                                                             //  list is decided exactly; thr_i = 256 empties the window on the host)
        const uint32_t tl = (uint32_t)(t & 0x7f);
        const uint32_t lo4 = __builtin_amdgcn_perm(tl, tl, 0u);               // the byte in all four places
        const uint32_t nhi = (t & 0x80) ? 0u : 0xffffffffu;
        const uint32_t x0 = cur[j][2 * h], x1 = cur[j][2 * h + 1];
        const uint32_t y0 = (x0 | 0x80808080u) - lo4, y1 = (x1 | 0x80808080u) - lo4;
        const uint32_t ge0 = (y0 & x0) | ((y0 | x0) & nhi);                  // bit 7 of every byte: byte >= t
        const uint32_t ge1 = (y1 & x1) | ((y1 | x1) & nhi);
        // gather the eight verdict bits: byte k of z holds dword 0's verdict at bit 0 and dword 1's at bit 4, and the dot
        // product of z's bytes with (1, 2, 4, 8) is the mask (one full-rate v_dot4 instead of a 32-bit multiply)
        const uint32_t z = ((ge0 >> 7) & 0x01010101u) | ((ge1 >> 3) & 0x10101010u);
        const uint32_t m8 = __builtin_amdgcn_udot4(z, 0x08040201u, 0u, false);
        cmask |= m8 << (8 * h);
      }
      if (a.bin_lo > j * CB || a.bin_hi < j * CB + 64 * LB) {               // range window: bins [bin_lo, bin_hi) of this lane's LB
        const int lo = min(LB, max(0, a.bin_lo - pos)), hi = min(LB, max(0, a.bin_hi - pos));   // (chunks inside the window skip this)
        const uint32_t win = hi > lo ? ((hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u)) : 0u;
        cmask &= win;
      }
      cm[j] = cmask;
    }
    CFAR_T(1);
    after_swar();                                                           // (the rows kernel moves the NEXT row's pieces into `cur` here)
#pragma unroll
    for (int j = 0; j < NCH; j++) {
      if (j >= cj_hi) break;
      if (j < cj_lo) continue;
      const int pos = j * CB + lane * 4 * DJ(j);
      // the chunk's candidates -> list, behind the carried remainder, in bin order (lane, bit).  STAGED (cacfar_cols_kernel): the
      // list is shorter than a chunk + the carry, so that eight wavefronts share a CU's LDS with two 16-row tiles; a chunk whose
      // candidates would not fit goes in two halves of 32 lanes (<= 1024 bins + 127 carried <= kCfarColsList; the pre-filter
      // leaves ~100 candidates per row, so this is the exception)
      uint32_t m = cm[j];
      const int pc = __popc(m);
      const int incl = wave_incl_scan_i32(pc);
      const int tot = __builtin_amdgcn_readlane(incl, 63);
      const int low = STAGED ? __builtin_amdgcn_readlane(incl, 31) : 0;     // candidates of lanes 0 .. 31
      const int halves = (STAGED && C + tot > a.list_cap) ? 2 : 1;
      for (int hh = 0; hh < halves; hh++) {
        {
          const bool mine = halves == 1 || (lane >> 5) == hh;
          int off = C + incl - pc - (hh == 1 ? low : 0);
          if (mine)
            while (m) {
              list[off++] = (unsigned short)(pos + __ffs((int)m) - 1);
              m &= m - 1u;
            }
          C += halves == 1 ? tot : (hh == 0 ? low : tot - low);
        }
        wave_sync();
        CFAR_T(2);
#ifdef CFEAR_CFAR_TIMING
        ctot += C;
#endif
        int k0 = 0;
        for (; k0 + 128 <= C; k0 += 128) round2(k0, 128);
        if (k0 > 0) {                                                       // carry the remainder (< 128) to the front
          const int rem = C - k0;
          wave_sync();
          const unsigned short tmp0 = lane < rem ? list[k0 + lane] : (unsigned short)0;
          const unsigned short tmp1 = lane + 64 < rem ? list[k0 + 64 + lane] : (unsigned short)0;
          wave_sync();
          if (lane < rem) list[lane] = tmp0;
          if (lane + 64 < rem) list[64 + lane] = tmp1;
          C = rem;
          wave_sync();
        }
#ifdef CFEAR_CFAR_TIMING
        ctot -= C;
#endif
        CFAR_T(3);
      }
    }
    if (C > 64) round2(0, C);
    else if (C > 0) round(0, C);
    CFAR_T(3);
    // ---- D ------------------------------------------------------------------------------------------------------
    if (KEYS) {
      if (lane == 0) { a.row_cnt[2 * grow] = ndet; a.row_cnt[2 * grow + 1] = 0; }
    } else {
      wave_sync();
      int total = 0;
      for (int w0 = 0; w0 < a.words; w0 += 64) {
        const int wd = w0 + lane;
        if (wd < a.words) {
          unsigned long long bits = 0ull;
          if (2 * wd < colsp / 32) bits = (unsigned long long)det32[2 * wd];
          if (2 * wd + 1 < colsp / 32) bits |= (unsigned long long)det32[2 * wd + 1] << 32;
          a.det_bits[grow * a.words + wd] = bits;
          total += __popcll(bits);
        }
      }
      total = wave_sum_i32(total);
      if (lane == 0) a.det_count[grow] = total;
    }
    wave_sync();                                                            // the next row's writes stay behind this row's reads
    CFAR_T(4);
#ifdef CFEAR_CFAR_TIMING
    if (lane == 0 && (grow % 20011) == 0)
      printf("cfar row %lld: prefix %lld | thresholds+swar %lld | list %lld | rounds %lld (%d rounds, %d candidates) | write %lld\n", grow,
             tq[0], tq[1], tq[2], tq[3], nrounds, ctot, tq[4]);
#endif
}

template <int D, int NCH, int DL, bool KEYS, bool PRE>
__global__ __launch_bounds__(256) void cacfar_rows_kernel(const CfarArgs a) {
  constexpr int CB = 256 * D;                                                // bins per chunk (all but a shorter last one)
  auto DJ = [](int j) { return (DL != D && j == NCH - 1) ? DL : D; };        // dwords per lane of chunk j (folds after unrolling)
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(4)));
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t* lut = (uint32_t*)smem;
  lut[threadIdx.x] = a.lut[threadIdx.x];
  __syncthreads();
  const int colsp = a.colsp;
  uint8_t* wbase = smem + 1024 + (size_t)wave * cfar_wave_lds(colsp, a.pad_lo, a.pad_hi, KEYS, CB);   // (the list holds a full chunk)
  uint32_t* P4 = (uint32_t*)wbase + a.pad_lo;                                // P4[i] = sum_{q < 4 i} I_q^2, i in [-pad_lo, colsp / 4 + pad_hi]
  uint8_t* raw = wbase + (((size_t)(a.pad_lo + colsp / 4 + 1 + a.pad_hi) * 4 + 15) & ~(size_t)15);   // the row itself
  uint32_t* det32 = (uint32_t*)(raw + colsp + 16);                           // detections, bit per bin (bitmap output)
  unsigned short* list = (unsigned short*)(raw + colsp + 16 + (KEYS ? 0 : colsp / 8));
  for (int i = lane; i < a.pad_lo; i += 64) P4[-1 - i] = 0u;
  const int nch = DL != D ? NCH : colsp / CB;
  const long long step = (long long)gridDim.x * kRowsPerBlock;
  long long grow = (long long)blockIdx.x * kRowsPerBlock + wave;
  // (image, row) of the current and of the next row walk along with grow: no 64-bit division per row
  const int step_b = (int)(step / a.rows), step_r = (int)(step - (long long)step_b * a.rows);
  int cb = (int)(grow / a.rows), cr = (int)(grow - (long long)cb * a.rows);
  auto row_ptr = [&](int b, int r) -> const uint8_t* { return a.polar + (long long)b * a.batch_stride + (long long)r * a.stride; };
  // Rows are read in 16-byte pieces wherever they start on a 4-byte boundary (global_load_dwordx4 asks for no more).  A row whose
  // length is not a multiple of 16 (Oxford's native 3768 bins) ends inside its last piece: that piece is read whole -- it ends
  // inside the image for every row but the image's last -- and the bytes beyond the row are cleared in registers (mask_tail), as
  // the byte-wise copy below leaves them.  Only rows on odd addresses and the last row of a ragged image are copied into LDS byte
  // by byte, zero-padded, and take their pieces from there (no prefetch).  (Until round 6 every ragged or 8-byte-aligned row took
  // the byte copy: 1.42 instead of 0.43 ms per 512 sweeps of 3768 bins.)
  const bool ragged = (a.cols & 15) != 0;
  auto is_direct = [&](const uint8_t* p, const int row) -> bool {
    return (((uintptr_t)p) & 3) == 0 && (a.stride & 3) == 0 && !(ragged && row == a.rows - 1);
  };
  // a lane's LB bytes of a chunk: 16-byte pieces where LB is a multiple of 16 (D = 4, 8), 8-byte pieces otherwise (D = 6:
  // 24 lane is only 8-byte aligned, and ds_write_b128 wants 16)
  auto issue = [&](const uint8_t* p, uint32_t (&dst)[NCH][D]) {
#pragma unroll
    for (int j = 0; j < NCH; j++) {
      const int dj = DJ(j);
      const int pos = j * CB + lane * 4 * dj;
      // (no zeroing here: the registers of the lanes beyond need_cols are zeroed ONCE before the row loop; the masked loads
      //  never write them, and the copies below move zeros)
      if (dj % 4 == 0) {
#pragma unroll
        for (int k = 0; k < D / 4; k++)
          if (4 * k < dj && pos + 16 * k < a.need_cols) {
            const u32x4 v = __builtin_nontemporal_load((const u32x4*)(p + pos + 16 * k));
            dst[j][4 * k] = v.x; dst[j][4 * k + 1] = v.y; dst[j][4 * k + 2] = v.z; dst[j][4 * k + 3] = v.w;
          }
      } else {
#pragma unroll
        for (int k = 0; k < D / 2; k++)
          if (2 * k < dj && pos + 8 * k < a.need_cols) {
            const u32x2 v = __builtin_nontemporal_load((const u32x2*)(p + pos + 8 * k));
            dst[j][2 * k] = v.x; dst[j][2 * k + 1] = v.y;
          }
      }
    }
  };
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };
  uint32_t cur[NCH][D], nxt[NCH][D];
#pragma unroll
  for (int j = 0; j < NCH; j++)
#pragma unroll
    for (int d = 0; d < D; d++) { cur[j][d] = 0u; nxt[j][d] = 0u; }
  auto mask_tail = [&](uint32_t (&x)[NCH][D]) {             // ragged rows: nothing but zeros beyond bin cols - 1
#pragma unroll
    for (int j = 0; j < NCH; j++) {
      const int dj = DJ(j);
      const int pos = j * CB + lane * 4 * dj;
#pragma unroll
      for (int d = 0; d < D; d++)
        if (d < dj) {
          const int rem = a.cols - (pos + 4 * d);
          x[j][d] &= rem >= 4 ? 0xffffffffu : (rem <= 0 ? 0u : ((1u << (8 * rem)) - 1u));
        }
    }
  };
  bool direct_cur = grow < a.total_rows && is_direct(row_ptr(cb, cr), cr);
  if (direct_cur) issue(row_ptr(cb, cr), cur);
  // the first row's pieces are waited for HERE, so that inside the loop `cur` only ever comes from register copies: the
  // compiler cannot count conditional loads and would otherwise wait for vmcnt(0) -- the NEXT row's requests -- at the
  // first use of `cur` in every iteration
  __builtin_amdgcn_s_waitcnt(0);
  if (ragged && direct_cur) mask_tail(cur);
  const uint8_t* rowp = row_ptr(cb, cr);
  long long key_base = grow * (long long)a.kcap;
  const long long key_step = step * (long long)a.kcap;
  for (; grow < a.total_rows; grow += step, key_base += key_step) {
    cb += step_b; cr += step_r;
    if (cr >= a.rows) { cr -= a.rows; cb++; }
    if (!direct_cur) {
      for (int pos = lane * 16; pos < colsp; pos += 1024) {
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        for (int q = pos; q < min(pos + 16, a.cols); q++) w[(q - pos) >> 2] |= (uint32_t)rowp[q] << (8 * (q & 3));
        *(uint4*)(raw + pos) = make_uint4(w[0], w[1], w[2], w[3]);
      }
      wave_sync();
#pragma unroll
      for (int j = 0; j < NCH; j++) {
        const int dj = DJ(j);
        const int pos = j * CB + lane * 4 * dj;
#pragma unroll
        for (int d = 0; d < D; d++) cur[j][d] = (d < dj && pos < colsp) ? *(const uint32_t*)(raw + pos + 4 * d) : 0u;
      }
    }
    const bool have_next = grow + step < a.total_rows;
    const uint8_t* nextp = have_next ? row_ptr(cb, cr) : rowp;
    const bool next_direct = have_next && is_direct(nextp, cr);
    if (next_direct) issue(nextp, nxt);
    rowp = nextp;                                                           // (this row is in registers / LDS from here on)
    cfar_row<D, NCH, DL, KEYS, PRE, false>(a, lane, lut, P4, raw, det32, list, cur, nch, grow, key_base, [&]() {
      // The row's bytes are dead from here on (the rounds work from LDS): the NEXT row's pieces move into `cur` now, so that
      // the wait for them does not sit behind this row's key stores (vmcnt counts stores too: at the end of the row the copy
      // waited for the stores of the last round every time).
      if (next_direct) {
#pragma unroll
        for (int j = 0; j < NCH; j++)
#pragma unroll
          for (int d = 0; d < D; d++) cur[j][d] = nxt[j][d];
        if (ragged) mask_tail(cur);
      }
    });
    direct_cur = next_direct;
  }
}

// CA-CFAR on [range bins][azimuths] sweeps (the layout the non-Oxford drivers deliver, radar_driver.cpp:74-90): the decode
// (cv::rotate 90 deg counter-clockwise) fused into the filter -- ONE pass over the image instead of rotate (read + write)
// + cacfar_rows (read).  A workgroup takes a tile of 16 azimuths: the 16-byte pieces of the bins the arithmetic can reach
// (a.need_cols source rows) are transposed into LDS with v_perm_b32 on 4 x 4 byte blocks, the next tile's pieces are
// requested, and each wavefront runs cfar_row on four of the tile's rows straight from LDS.  Output row r holds source
// column a.rows - 1 - r.  a.rows / a.cols are the ROTATED image's (azimuths, bins); a.stride / a.batch_stride the SOURCE's.
// Key output only (the batched odometry).  Tiles of one image run on one XCD (blockIdx % 8): the eight tiles that share a
// 128-byte line of a source row meet in that XCD's L2.
constexpr int kCfarTile = 16;
constexpr int kCfarColsWaves = 8;             // 2 workgroups x 8 wavefronts per CU: two rows of a tile per wavefront, four wavefronts per
                                              // SIMD like cacfar_rows_kernel (4 x 4 rows: 0.364 ms per 512 sweeps, 6 x 3|2: 0.338)
constexpr int kCfarColsList = 1152;           // list entries per wavefront (cfar_row's STAGED append)
__host__ __device__ inline int cfar_cols_list_cap(int chunk_bins) { return chunk_bins + kCfarListSlack < kCfarColsList ? chunk_bins + kCfarListSlack : kCfarColsList; }
__host__ __device__ inline size_t cfar_cols_wave_lds(int colsp, int pad_lo, int pad_hi, int chunk_bins) {
  // P4 + list: the row lives in the tile
  return (((size_t)(pad_lo + colsp / 4 + 1 + pad_hi) * 4 + 15) & ~(size_t)15) + (((size_t)cfar_cols_list_cap(chunk_bins) * 2 + 15) & ~(size_t)15);
}
template <int D, int NCH, int DL, bool PRE>
__global__ __launch_bounds__(64 * kCfarColsWaves, 4) void cacfar_cols_kernel(const CfarArgs a, const int tiles) {
  constexpr int CB = 256 * D;
  auto DJ = [](int j) { return (DL != D && j == NCH - 1) ? DL : D; };
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  uint32_t* lut = (uint32_t*)smem;
  if (threadIdx.x < 256) lut[threadIdx.x] = a.lut[threadIdx.x];   // (256 entries: the workgroup has more threads than that)
  const int colsp = a.colsp, tstride = a.need_cols + 16;      // (the arithmetic never reads a row beyond need_cols: cfar_derive)
  uint8_t* tbase = smem + 1024;                               // [kCfarTile][tstride]: row lr = source column c0 + 15 - lr
  uint8_t* wbase = tbase + (((size_t)kCfarTile * tstride + 15) & ~(size_t)15) + (size_t)wave * cfar_cols_wave_lds(colsp, a.pad_lo, a.pad_hi, CB);
  uint32_t* P4 = (uint32_t*)wbase + a.pad_lo;
  unsigned short* list = (unsigned short*)(wbase + (((size_t)(a.pad_lo + colsp / 4 + 1 + a.pad_hi) * 4 + 15) & ~(size_t)15));
  for (int i = lane; i < a.pad_lo; i += 64) P4[-1 - i] = 0u;
  for (int i = threadIdx.x; i < kCfarTile * 4; i += 64 * kCfarColsWaves)     // the 16 bytes of padding behind every tile row
    *(uint32_t*)(tbase + (size_t)(i >> 2) * tstride + a.need_cols + 4 * (i & 3)) = 0u;
  const int nch = DL != D ? NCH : colsp / CB;
  const int xcd = blockIdx.x % kXcds, slot = blockIdx.x / kXcds, slots = gridDim.x / kXcds;
  const int nq = ((a.batch - xcd + kXcds - 1) / kXcds) * tiles;               // (image, tile) items of this workgroup's XCD
  auto locate = [&](const int q, int& b, int& tile) { const int im = q / tiles; b = im * kXcds + xcd; tile = q - im * tiles; };
  constexpr int GP = (NCH * CB / 4 + 64 * kCfarColsWaves - 1) / (64 * kCfarColsWaves);   // groups of 4 bins per thread (upper bound)
  const int need_groups = a.need_cols >> 2;
  uint32_t rw[GP][4][4];
  auto issue = [&](const int q) {                             // the pieces of item q: bins 4 g .. 4 g + 3, 16 source columns
    int b, tile;
    locate(q, b, tile);
    const uint8_t* src = a.polar + (long long)b * a.batch_stride + tile * kCfarTile;
#pragma unroll
    for (int p = 0; p < GP; p++) {
      const int g = (int)threadIdx.x + p * 64 * kCfarColsWaves;
      if (g < need_groups) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const u32x4 v = *(const u32x4*)(src + (size_t)(4 * g + i) * a.stride);
          rw[p][i][0] = v.x; rw[p][i][1] = v.y; rw[p][i][2] = v.z; rw[p][i][3] = v.w;
        }
      }
    }
  };
#pragma unroll
  for (int p = 0; p < GP; p++)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int d = 0; d < 4; d++) rw[p][i][d] = 0u;            // groups beyond need_cols stay zero
  int q = slot;
  if (q < nq) issue(q);
  while (q < nq) {
#pragma unroll
    for (int p = 0; p < GP; p++) {
      const int g = (int)threadIdx.x + p * 64 * kCfarColsWaves;
      if (g < need_groups) {
#pragma unroll
        for (int d = 0; d < 4; d++) {                         // source columns c0 + 4 d .. + 3 of bins 4 g .. 4 g + 3
          const uint32_t w0 = rw[p][0][d], w1 = rw[p][1][d], w2 = rw[p][2][d], w3 = rw[p][3][d];
          const uint32_t t0 = __builtin_amdgcn_perm(w1, w0, 0x05010400u), t1 = __builtin_amdgcn_perm(w1, w0, 0x07030602u);
          const uint32_t t2 = __builtin_amdgcn_perm(w3, w2, 0x05010400u), t3 = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
          uint32_t colw[4];                                   // colw[e] = column c0 + 4 d + e as {bin 4g, +1, +2, +3}
          colw[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u); colw[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
          colw[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u); colw[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
#pragma unroll
          for (int e = 0; e < 4; e++) *(uint32_t*)(tbase + (size_t)(kCfarTile - 1 - (4 * d + e)) * tstride + 4 * g) = colw[e];
        }
      }
    }
    __syncthreads();
    int b, tile;
    locate(q, b, tile);
    const int qn = q + slots;
    if (qn < nq) issue(qn);
    const int r0 = a.rows - kCfarTile - tile * kCfarTile;     // output row of tile row 0
    for (int lr = wave; lr < kCfarTile; lr += kCfarColsWaves) {
      uint8_t* raw = tbase + (size_t)lr * tstride;
      uint32_t cur[NCH][D];
#pragma unroll
      for (int j = 0; j < NCH; j++) {
        const int dj = DJ(j);
        const int pos = j * CB + lane * 4 * dj;
#pragma unroll
        for (int d = 0; d < D; d++) cur[j][d] = (d < dj && j < nch && pos + 4 * d < a.need_cols) ? *(const uint32_t*)(raw + pos + 4 * d) : 0u;
      }
      const long long grow = (long long)b * a.rows + (r0 + lr);
      cfar_row<D, NCH, DL, true, PRE, true>(a, lane, lut, P4, raw, nullptr, list, cur, nch, grow, grow * (long long)a.kcap, []() {});
    }
    __syncthreads();                                          // every row of the tile has been consumed
    q = qn;
  }
}

// lut[I] for cacfar_rows_kernel (step C of its comment): 0 when I does not pass the static threshold.
static bool cfar_build_lut(const CfarArgs& a, uint32_t* lut) {
  for (int i = 0; i < 256; i++) lut[i] = 0u;
  if (!(a.scaling > 0.0) || !std::isfinite(a.scaling) || a.window > 8192) return false;
  const double c = 2.0 * (double)a.window / a.scaling;
  for (int i = 0; i < 256; i++) {
    if (!((double)i > a.static_threshold)) continue;                        // cfar.cpp:45
    const double B = (double)(i * i) * c;
    if (!(B < 1.0e9)) return false;                                         // tiny scalings: the table would not fit 31 bits
    const double T = std::ceil(B - 1e-3), Tn = std::ceil(B + 1e-3);
    const uint32_t t = T < 0.0 ? 0u : (uint32_t)T;
    lut[i] = (t << 1) | (Tn > T ? 1u : 0u);
  }
  return true;
}

struct CfarCloudArgs {
  const uint8_t* polar;
  int rows, cols, stride;
  long long batch_stride;
  const unsigned long long* det_bits;
  const int32_t* det_count;
  int words;
  const double* cos_t;
  const double* sin_t;
  double range_res;
  float* xyzi;
  int32_t* n_points;
  int cap_points;
  uint8_t* det_mask;
};

// Compaction in (row, bin) order.  grid = (image, kCloudSplit row slices): every workgroup scans all row counts (cheap)
// and emits the rows of its slice, one wavefront per row: lane w takes word w of the row's detection bitmap, a wave
// scan of the popcounts places the words, and each lane walks the set bits of its own word.  (The first version gave one
// workgroup a whole image and walked every 64-bin word of every row in turn: 1.5 ms per batch whatever its size.)
__global__ __launch_bounds__(256) void cacfar_cloud_kernel(const CfarCloudArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  int32_t* row_off = (int32_t*)smem;
  __shared__ int32_t wave_tot[4];
  __shared__ int32_t run_base;
  const int b = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) run_base = 0;
  __syncthreads();
  for (int r0 = 0; r0 < a.rows; r0 += 256) {
    const int r = r0 + threadIdx.x;
    const int v = r < a.rows ? a.det_count[(long long)b * a.rows + r] : 0;
    const int incl = wave_incl_scan_i32(v);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int off = run_base;
    for (int wv = 0; wv < wave; wv++) off += wave_tot[wv];
    if (r < a.rows) row_off[r] = off + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) run_base += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
  }
  if (threadIdx.x == 0 && blockIdx.y == 0) a.n_points[b] = run_base;
  const uint8_t* img = a.polar + (long long)b * a.batch_stride;
  const int rows_per = (a.rows + (int)gridDim.y - 1) / (int)gridDim.y;
  const int rbeg = blockIdx.y * rows_per, rend = min(a.rows, rbeg + rows_per);
  // One wavefront per row.  Lane w takes word w of the row's detection bitmap and writes the bins of its set bits into a
  // list in LDS (a wave scan of the popcounts places them); the list is then turned into points one detection per lane,
  // so the intensity gathers of a row are ONE memory round trip however the detections cluster (the form before walked
  // the set bits of its word with a dependent gather per step).  The next row's bitmap words are loaded meanwhile.
  unsigned short* dlist = (unsigned short*)(row_off + a.rows + 1) + (size_t)wave * 64 * 64;   // [<= 64 words x 64 bits]
  auto load_bits = [&](int rr, int w0) -> unsigned long long {
    const int wd = w0 + lane;
    return (rr < rend && wd < a.words) ? a.det_bits[((long long)b * a.rows + rr) * a.words + wd] : 0ull;
  };
  unsigned long long nxt = load_bits(rbeg + wave, 0);
  for (int r = rbeg + wave; r < rend; r += 4) {
    const double cos_t = a.cos_t[r], sin_t = a.sin_t[r];
    int base = row_off[r];
    for (int w0 = 0; w0 < a.words; w0 += 64) {                              // cols <= 8192: at most two rounds
      const int wd = w0 + lane;
      unsigned long long bits = nxt;
      nxt = w0 + 64 < a.words ? load_bits(r, w0 + 64) : load_bits(r + 4, 0);
      if (a.det_mask && wd < a.words)
        for (int j = 0; j < 64 && wd * 64 + j < a.cols; j++)
          a.det_mask[((long long)b * a.rows + r) * a.cols + wd * 64 + j] = (uint8_t)((bits >> j) & 1ull);
      const int pc = __popcll(bits);
      const int incl = wave_incl_scan_i32(pc);
      const int n = __builtin_amdgcn_readlane(incl, 63);
      int off = incl - pc;
      while (bits) {
        dlist[off++] = (unsigned short)(wd * 64 + __ffsll((long long)bits) - 1);
        bits &= bits - 1;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane, idx = base + j;
        if (j < n && idx < a.cap_points) {
          const int bin = dlist[j];
          const double range = a.range_res * (double)bin;
          float4 p;
          p.x = (float)(range * cos_t);                                       // cfar.cpp:63-65
          p.y = (float)(range * sin_t);
          p.z = 0.f;
          p.w = (float)img[(long long)r * a.stride + bin];
          ((float4*)a.xyzi)[(long long)b * a.cap_points + idx] = p;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      base += n;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Legacy k_strongest_filter / InsertStrongestK (radar_filters.cpp:25-78; CorAl's standalone kstrongRadar,
// coral_alignment_quality/src/alignment_checker/ScanType.cpp:104-114).  Different rule than StructuredKStrongest
// (SURVEY App. C): the first bin f with intensity >= z_min sets a floor m0 -- later bins <= the list's minimum are
// rejected even while the list is not full -- and ties at the cut keep the SMALLER ranges.  In closed form: with D = the
// bins after f with intensity > m0, the row keeps the k largest of D under (intensity, -range), plus f iff |D| < k,
// in descending intensity / ascending range order.  "k largest under (intensity, -range)" is StructuredKStrongest on the
// REVERSED row, so the tuned sweep does the selection: a prepare pass writes the row reversed with everything outside D
// zeroed, the sweep runs with z_min = 1, and a post pass undoes the reversal, appends f and converts to PointXYZI.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void legacy_prepare_kernel(const uint8_t* __restrict__ polar, int rows, int cols, int stride,
                                                             long long batch_stride, int u_z, uint8_t* __restrict__ rev,
                                                             int rev_stride, int32_t* __restrict__ first /*[batch][rows][2]*/) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long grow = (long long)blockIdx.x * 4 + wave;
  const int b = blockIdx.y;
  if (grow >= rows) return;
  const int r = (int)grow;
  const uint8_t* rowp = polar + (long long)b * batch_stride + (long long)r * stride;
  int fpos = 0x7fffffff;                                   // first bin with intensity >= z_min
  for (int i0 = 0; i0 < cols && fpos == 0x7fffffff; i0 += 64) {
    const int i = i0 + lane;
    const bool c = i < cols && (int)rowp[i] >= u_z;
    const unsigned long long bal = __ballot(c);
    if (bal) fpos = i0 + __ffsll((long long)bal) - 1;
  }
  const int m0 = fpos < cols ? (int)rowp[fpos] : 255;
  uint8_t* out = rev + ((long long)b * rows + r) * rev_stride;
  for (int i = lane; i < cols; i += 64) {
    const int v = rowp[i];
    out[cols - 1 - i] = (i > fpos && v > m0) ? (uint8_t)v : (uint8_t)0;
  }
  if (lane == 0) { first[((long long)b * rows + r) * 2] = fpos < cols ? fpos : -1; first[((long long)b * rows + r) * 2 + 1] = m0; }
}

// one workgroup per image: per-row output counts -> offsets -> points
__global__ __launch_bounds__(256) void legacy_cloud_kernel(const int32_t* __restrict__ sel_range, const uint8_t* __restrict__ sel_int,
                                                           const int32_t* __restrict__ sel_count, const int32_t* __restrict__ first,
                                                           const float* __restrict__ cosf_t, const float* __restrict__ sinf_t,
                                                           int rows, int cols, int k, double range_res, double min_d2,
                                                           float* __restrict__ xyzi, int32_t* __restrict__ n_points, int cap) {
  extern __shared__ int32_t row_off[];                    // [rows + 1]
  __shared__ int32_t wave_tot[4];
  __shared__ int32_t run_base;
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  auto point = [&](int r, int j, int cnt, float4& p) -> bool {    // j-th entry of row r in the reference's list order
    int bin, inten;
    if (j < cnt) {                                         // descending: the sweep's list reversed; undo the row reversal
      const long long e = ((long long)b * rows + r) * k + (cnt - 1 - j);
      bin = cols - 1 - sel_range[e];
      inten = sel_int[e];
    } else {                                               // f, the floor-setting first bin (only while |D| < k)
      bin = first[((long long)b * rows + r) * 2];
      inten = first[((long long)b * rows + r) * 2 + 1];
    }
    p.x = (float)(range_res * bin * cosf_t[r]);            // :62-63
    p.y = (float)(range_res * bin * sinf_t[r]);
    p.z = 0.f;
    p.w = (float)inten;
    return (double)(p.x * p.x + p.y * p.y) > min_d2;       // :71
  };
  auto row_entries = [&](int r) {
    const int cnt = sel_count[(long long)b * rows + r];
    const bool has_f = first[((long long)b * rows + r) * 2] >= 0 && cnt < k;
    return cnt + (has_f ? 1 : 0);
  };
  if (threadIdx.x == 0) run_base = 0;
  __syncthreads();
  for (int r0 = 0; r0 < rows; r0 += 256) {
    const int r = r0 + threadIdx.x;
    int v = 0;
    if (r < rows) {
      const int cnt = sel_count[(long long)b * rows + r], ne = row_entries(r);
      float4 p;
      for (int j = 0; j < ne; j++) v += point(r, j, cnt, p) ? 1 : 0;
    }
    const int incl = wave_incl_scan_i32(v);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int off = run_base;
    for (int wv = 0; wv < wave; wv++) off += wave_tot[wv];
    if (r < rows) row_off[r] = off + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) run_base += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) n_points[b] = run_base;
  for (int r = threadIdx.x; r < rows; r += 256) {
    const int cnt = sel_count[(long long)b * rows + r], ne = row_entries(r);
    int o = row_off[r];
    for (int j = 0; j < ne; j++) {
      float4 p;
      if (point(r, j, cnt, p)) {
        if (o < cap) ((float4*)xyzi)[(long long)b * cap + o] = p;
        o++;
      }
    }
  }
}

// ---- host helpers --------------------------------------------------------------------------------
}  // namespace
// cos / sin of the azimuths (radar_filters.cpp:317), computed on the host in double so that the device's float
// coordinates are bit-exact with the reference; cached per context.
int cfear_trig_tables(cfear_ctx* ctx, int rows, double** d_cos, double** d_sin) {
  if (ctx->trig_rows == rows && ctx->ws[2].p) {      // tables are cached per context
    *d_cos = (double*)ctx->ws[2].p;
    *d_sin = (double*)ctx->ws[2].p + rows;
    return CFEAR_OK;
  }
  std::vector<double> h(2 * (size_t)rows);
  for (int bearing = 0; bearing < rows; bearing++) {
    const double theta = (double(bearing + 1) / rows) * 2. * M_PI;           // radar_filters.cpp:317
    h[bearing] = std::cos(theta);
    h[rows + bearing] = std::sin(theta);
  }
  double* d = (double*)cfear_workspace(ctx, 2, h.size() * sizeof(double));
  if (!d) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));   // h goes out of scope
  ctx->trig_rows = rows;
  *d_cos = d;
  *d_sin = d + rows;
  return CFEAR_OK;
}
namespace {
int upload_trig(cfear_ctx* ctx, int rows, double** d_cos, double** d_sin) { return cfear_trig_tables(ctx, rows, d_cos, d_sin); }

int check_desc(cfear_ctx* ctx, const cfear_polar_desc* d) {
  if (!d || d->rows <= 0 || d->cols <= 0 || d->stride < d->cols || d->batch <= 0 ||
      (d->batch > 1 && d->batch_stride < (int64_t)d->rows * d->stride))
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "bad polar descriptor");
  if (d->cols > kMaxCols)
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "cols > %d unsupported", kMaxCols);
  return CFEAR_OK;
}

template <int NCHUNK>
void launch_kstrong(cfear_ctx* ctx, const KStrongArgs& a, bool vec, bool mask, dim3 grid, size_t lds) {
  if (vec) {
    if (mask) hipLaunchKernelGGL((kstrongest_rows_kernel<NCHUNK, true, true>), grid, dim3(256), lds, ctx->stream, a);
    else hipLaunchKernelGGL((kstrongest_rows_kernel<NCHUNK, true, false>), grid, dim3(256), lds, ctx->stream, a);
  } else {
    if (mask) hipLaunchKernelGGL((kstrongest_rows_kernel<NCHUNK, false, true>), grid, dim3(256), lds, ctx->stream, a);
    else hipLaunchKernelGGL((kstrongest_rows_kernel<NCHUNK, false, false>), grid, dim3(256), lds, ctx->stream, a);
  }
}

}  // namespace

// Device-side entry used by cfear_filter_kstrongest and by the odometry pipeline: everything is
// already in device memory; outputs that are nullptr are skipped.
int cfear_kstrong_device(cfear_ctx* ctx, const uint8_t* d_polar, const cfear_polar_desc* desc,
                         const cfear_kstrong_params* par, const cfear_kstrong_out* o, bool dense_halo,
                         const cfear_kstrong_fused* fused) {
  const int z_min_i = (int)par->z_min;                         // radar_driver.cpp:58 float -> int
  KStrongArgs a;
  a.row_keys = nullptr;
  a.image_offsets = fused ? (const long long*)fused->image_offsets : nullptr;
  if (fused && fused->row_keys) {
    if (par->k_strongest > 64) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "fused row keys need k <= 64");
    a.row_keys = fused->row_keys;
  }
  a.polar = d_polar;
  a.rows = desc->rows; a.cols = desc->cols; a.stride = desc->stride; a.batch = desc->batch;
  a.batch_stride = desc->batch > 1 ? desc->batch_stride : (int64_t)desc->rows * desc->stride;
  a.k = par->k_strongest;
  a.u_zmin = (int)(uint8_t)z_min_i;                            // radar_filters.cpp:212 uchar(z_min_)
  a.want_peaks = par->want_peaks && (o->is_peak != nullptr);
  a.sel_range = o->sel_range; a.sel_intensity = o->sel_intensity; a.sel_count = o->sel_count;
  a.is_peak = o->is_peak;
  const bool want_cloud = o->xyzi || o->n_points || o->xyzi_peaks || o->n_peaks;
  a.row_valid = nullptr;
  a.dense_halo = (dense_halo && desc->stride != desc->cols) ? 1 : 0;
  {
    const double range_res_ = (double)par->range_res, min_distance_ = (double)par->min_distance;
    a.min_range_bin = (int)std::ceil(min_distance_ / range_res_);            // radar_filters.cpp:315
  }
  if (fused && fused->row_valid) {
    a.row_valid = fused->row_valid;
  } else if (want_cloud) {
    a.row_valid = (int32_t*)cfear_workspace(ctx, 3, (size_t)desc->batch * desc->rows * 8);
    if (!a.row_valid) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  }
  const bool vec = (((uintptr_t)d_polar) % 4 == 0) && (a.stride % 4 == 0) && (a.batch_stride % 4 == 0);
  // Byte-validity masks only where a zero byte could pass a test, i.e. z_min == 0: load_row() zero-fills what lies beyond the row
  // (ragged widths: Oxford's native 3768 bins end 8 bytes into a 16-byte piece), every threshold the kernel compares with is
  // >= uchar(z_min), and the peaks' halo bytes are written behind the last re-read of the staged row.  (Until round 6 every
  // width that is not a multiple of 16 took the masked instantiation: 1.84 instead of 1.25 ms per 4096 Oxford-native sweeps.)
  const bool mask = a.u_zmin == 0;
  const int nchunk = (a.cols + 1023) / 1024;
  const int kpad = std::max((a.k + 3) & ~3, 64);
  {
    ProfScope ps(ctx, "kstrongest_rows");
    for (int b0 = 0; b0 < a.batch; b0 += 65535) {             // gridDim.y limit
      a.batch0 = b0;
      dim3 grid((unsigned)((a.rows + kRowsPerBlock - 1) / kRowsPerBlock), (unsigned)std::min(65535, a.batch - b0));
      auto lds = [&](int nchunk_t) {
        const int np = (nchunk_t + 1) / 2;
        return (size_t)kRowsPerBlock * (nchunk_t * 1024 + 32 + std::max(1024, (np + 2) * 256) + kpad * 4);
      };
      if (nchunk <= 1) launch_kstrong<1>(ctx, a, vec, mask, grid, lds(1));
      else if (nchunk <= 2) launch_kstrong<2>(ctx, a, vec, mask, grid, lds(2));
      else if (nchunk <= 4) launch_kstrong<4>(ctx, a, vec, mask, grid, lds(4));
      else launch_kstrong<8>(ctx, a, vec, mask, grid, lds(8));
    }
  }
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  if (want_cloud) {
    double *d_cos = nullptr, *d_sin = nullptr;
    int rc = upload_trig(ctx, a.rows, &d_cos, &d_sin);
    if (rc != CFEAR_OK) return rc;
    CloudArgs c;
    c.sel_range = o->sel_range; c.sel_intensity = o->sel_intensity; c.sel_count = o->sel_count;
    c.is_peak = o->is_peak;
    c.row_valid = a.row_valid;
    c.cos_t = d_cos; c.sin_t = d_sin;
    c.rows = a.rows; c.k = a.k;
    c.min_range_bin = a.min_range_bin;
    c.range_res = (double)par->range_res;
    c.xyzi = o->xyzi; c.n_points = o->n_points;
    const bool pk = a.want_peaks && (o->xyzi_peaks || o->n_peaks);
    c.xyzi_peaks = pk ? o->xyzi_peaks : nullptr;
    c.n_peaks = pk ? o->n_peaks : nullptr;
    ProfScope ps(ctx, "kstrong_cloud");
    hipLaunchKernelGGL(kstrong_cloud_kernel, dim3(a.batch, pk ? 2 : 1, kCloudSplit), dim3(256),
                       (size_t)(a.rows + 1) * 4, ctx->stream, c);
    CFEAR_HIP_CHECK(ctx, hipGetLastError());
  }
  return CFEAR_OK;
}

// [range bins][azimuths] sources, fused decode + sweep (kstrongest_cols_kernel): `sd` describes the SOURCE images (rows =
// range bins, cols = azimuths).  Only the batched odometry's key output; whatever this path does not take (peaks, k > 64,
// unaligned or ragged images, more than 4096 bins) goes through cfear_rotate_ccw_device + cfear_kstrong_device.
bool cfear_kstrong_cols_supported(const uint8_t* d_src, const cfear_polar_desc* sd, const cfear_kstrong_params* par) {
  // azimuths (sd->cols) <= 4096: the image route's fixed LDS part and the exact range of the umulhi division by the number of
  // 16-azimuth segments (p * segs < 2^32 with p < bins * segs) both depend on it
  return par->k_strongest <= 64 && sd->rows <= 4096 && sd->cols <= 4096 && sd->rows % 4 == 0 && sd->cols % kColsTile == 0 && sd->stride % 16 == 0 &&
         (uintptr_t)d_src % 16 == 0 && (sd->batch <= 1 || sd->batch_stride % 16 == 0) &&
         (int64_t)sd->rows * sd->stride < ((int64_t)1 << 31);
}

// route: 0 = by batch size, 1 = lists in global memory (small batches), 2 = one workgroup per image, 3 = every tile
// through the LDS transposition
// ... and pays: a handful of sweeps (one radar alone) are done sooner by the rotation kernel + the row sweep (20 us against
// 37 us for one image through the global lists; a workgroup alone streams its image in ~100 us whatever the batch).  From
// 128 images on the image-per-workgroup kernel wins (128: 0.11 ms against ~0.10 for the two kernels in the pipeline and 0.11
// for the global lists; 192: 0.11 / 0.15 / 0.14; tools/decode_bench.py N --image | --lists | --two-pass).
bool cfear_kstrong_cols_preferred(const uint8_t* d_src, const cfear_polar_desc* sd, const cfear_kstrong_params* par) {
  return sd->batch >= 128 && cfear_kstrong_cols_supported(d_src, sd, par);
}

int cfear_kstrong_cols_device(cfear_ctx* ctx, const uint8_t* d_src, const cfear_polar_desc* sd, const cfear_kstrong_params* par,
                              const cfear_kstrong_fused* fused, int route) {
  bool all_tiles = route == 3;
  if (!fused || !fused->row_keys || !fused->row_valid || !cfear_kstrong_cols_supported(d_src, sd, par))
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "fused decode: unsupported image geometry or outputs");
  KStrongArgs a{};
  a.polar = d_src;
  a.rows = sd->cols; a.cols = sd->rows; a.stride = sd->stride; a.batch = sd->batch;
  a.batch_stride = sd->batch > 1 ? sd->batch_stride : (int64_t)sd->rows * sd->stride;
  a.k = par->k_strongest;
  a.u_zmin = (int)(uint8_t)(int)par->z_min;                    // radar_driver.cpp:58, radar_filters.cpp:212
  a.row_keys = fused->row_keys;
  a.row_valid = fused->row_valid;
  a.min_range_bin = (int)std::ceil((double)par->min_distance / (double)par->range_res);   // radar_filters.cpp:315
  const bool mask = (a.cols % 16 != 0) || a.u_zmin == 0;
  const int nchunk = (a.cols + 1023) / 1024, tiles = a.rows / kColsTile;
  if (a.u_zmin == 0) all_tiles = true;                       // every bin is a candidate: the lists would only overflow
  // scratch: candidate counts | tile flags | work count | work list | candidate keys
  const size_t n_rows = (size_t)a.batch * a.rows, n_tiles = (size_t)a.batch * tiles;
  const size_t o_flag = n_rows * 4, o_wn = o_flag + n_tiles * 4, o_work = o_wn + 256, o_cand = (o_work + n_tiles * 4 + 255) / 256 * 256;
  uint32_t* work = nullptr;
  int32_t* work_n = nullptr;
  const int n_cu = ctx->n_cu;
  if (fused->cand_stats) CFEAR_HIP_CHECK(ctx, hipMemsetAsync(fused->cand_stats, 0, 64 * 4, ctx->stream));
  if (!all_tiles) {
    char* ws = (char*)cfear_workspace(ctx, 12, o_cand + n_rows * kCandCap * 4);
    if (!ws) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
    int32_t* cand_cnt = (int32_t*)ws;
    uint32_t* tile_flag = (uint32_t*)(ws + o_flag);
    uint32_t* cand = (uint32_t*)(ws + o_cand);
    work_n = (int32_t*)(ws + o_wn);
    work = (uint32_t*)(ws + o_work);
    const int segs = a.rows / 16;
    const uint32_t magic = (uint32_t)(((uint64_t)1 << 32) / (uint32_t)segs) + 1u;   // p / segs = umulhi(p, magic) while p * segs < 2^32; segs = 1 is taken apart in the kernel
    const uint32_t n_pieces = (uint32_t)a.cols * (uint32_t)segs;
    // image-per-workgroup route: its LDS holds rows * (1 + lds_cap) dwords
    const size_t img_fixed = (size_t)((a.rows + 3) & ~3) * 4 + (size_t)kImgWaves * (kCandCap + 8) * 4;
    const size_t img_budget = 80 * 1024 - 512;
    const int lds_cap = img_fixed >= img_budget ? 0 : (int)std::min<size_t>(40, (img_budget - img_fixed) / ((size_t)a.rows * 4));
    const bool per_image = route == 2 || (route == 0 && lds_cap >= 16);   // (the global lists: azimuth counts whose lists do not fit the LDS)
    if (per_image && lds_cap < 1) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "fused decode: too many azimuths for the image route");
    if (per_image) {
      CFEAR_HIP_CHECK(ctx, hipMemsetAsync(ws + o_flag, 0, o_work - o_flag, ctx->stream));
      const size_t lds = img_fixed + (size_t)a.rows * lds_cap * 4;
      { const int rc_lds = cfear_allow_lds(ctx, (const void*)kstrong_image_kernel, 160 * 1024); if (rc_lds != CFEAR_OK) return rc_lds; }
      ProfScope ps(ctx, "kstrong_image");
      hipLaunchKernelGGL(kstrong_image_kernel, dim3((unsigned)std::min(a.batch, 2 * n_cu)), dim3(64 * kImgWaves), lds, ctx->stream, a,
                         magic, segs, tiles, lds_cap, cand, tile_flag, work_n, work, fused->cand_stats);
      CFEAR_HIP_CHECK(ctx, hipGetLastError());
    } else {
      CFEAR_HIP_CHECK(ctx, hipMemsetAsync(ws, 0, o_work, ctx->stream));
      for (int b0 = 0; b0 < a.batch; b0 += 65535) {             // gridDim.y limit
        a.batch0 = b0;
        const unsigned by = (unsigned)std::min(65535, a.batch - b0);
        {
          ProfScope ps(ctx, "kstrong_extract");
          hipLaunchKernelGGL(kstrong_extract_kernel, dim3((n_pieces + 256 * kExtractPieces - 1) / (256 * kExtractPieces), by), dim3(256), 0,
                             ctx->stream, a, magic, segs, cand_cnt, cand);
        }
        {
          ProfScope ps(ctx, "kstrong_select");
          hipLaunchKernelGGL(kstrong_select_kernel, dim3((a.rows + kRowsPerBlock - 1) / kRowsPerBlock, by), dim3(256), 0, ctx->stream, a,
                             tiles, cand_cnt, cand, tile_flag, work_n, work, fused->cand_stats);
        }
        CFEAR_HIP_CHECK(ctx, hipGetLastError());
      }
      a.batch0 = 0;
    }
  }
  const int kpad = std::max((a.k + 3) & ~3, 64);
  const size_t lds = (size_t)kColsTile * ((a.cols + 15) & ~15) +
                     (size_t)kColsWaves * (kstrong_scratch_bytes(nchunk <= 1 ? 1 : (nchunk <= 2 ? 2 : 4)) + kpad * 4);
  // persistent: two workgroups per CU (LDS), a multiple of the XCD count; fewer when the batch is small
  const long long per_xcd = (long long)tiles * ((a.batch + kXcds - 1) / kXcds);
  const int slots = (int)std::max<long long>(1, std::min<long long>(per_xcd, std::max(1, n_cu * 2 / kXcds)));
  const dim3 grid((unsigned)(slots * kXcds));
  ProfScope ps(ctx, "kstrongest_cols");
  auto launch = [&](auto fn) {
    (void)cfear_allow_lds(ctx, (const void*)fn, 160 * 1024);
    hipLaunchKernelGGL(fn, grid, dim3(64 * kColsWaves), lds, ctx->stream, a, tiles, (const uint32_t*)work, (const int32_t*)work_n);
  };
  if (nchunk <= 1) { if (mask) launch(kstrongest_cols_kernel<1, true>); else launch(kstrongest_cols_kernel<1, false>); }
  else if (nchunk <= 2) { if (mask) launch(kstrongest_cols_kernel<2, true>); else launch(kstrongest_cols_kernel<2, false>); }
  else { if (mask) launch(kstrongest_cols_kernel<4, true>); else launch(kstrongest_cols_kernel<4, false>); }
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}

extern "C" int cfear_filter_kstrongest_rowkeys(cfear_ctx* ctx, const uint8_t* polar, const cfear_polar_desc* desc,
                                               const cfear_kstrong_params* par, int32_t flags, uint32_t* row_keys,
                                               int32_t* row_counts) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!polar || !par || !row_keys || !row_counts) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  const bool bins_major = (flags & CFEAR_ROWKEYS_BINS_MAJOR) != 0;
  if (!desc || desc->rows <= 0 || desc->cols <= 0 || desc->stride < desc->cols || desc->batch <= 0 ||
      (desc->batch > 1 && desc->batch_stride < (int64_t)desc->rows * desc->stride) ||
      (bins_major ? desc->rows : desc->cols) > kMaxCols)
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "bad polar descriptor");
  if (par->k_strongest < 1 || par->k_strongest > 64)
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "row keys need k_strongest in [1,64]");
  if (!(par->range_res > 0.f)) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "range_res must be > 0");
  if (!cfear_is_device_ptr(polar) || !cfear_is_device_ptr(row_keys) || !cfear_is_device_ptr(row_counts))
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "images and outputs must be device memory");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  cfear_kstrong_params kp = *par;
  kp.want_peaks = 0;
  cfear_kstrong_fused fz;
  fz.row_keys = row_keys;
  fz.row_valid = row_counts;
  cfear_kstrong_out none{};
  if (!bins_major) return cfear_kstrong_device(ctx, polar, desc, &kp, &none, false, &fz);
  const bool routed = (flags & (CFEAR_ROWKEYS_TILE_SWEEP | CFEAR_ROWKEYS_ROUTE_LISTS | CFEAR_ROWKEYS_ROUTE_IMAGE)) != 0;
  if (!(flags & CFEAR_ROWKEYS_TWO_PASS) &&
      (routed ? cfear_kstrong_cols_supported(polar, desc, &kp) : cfear_kstrong_cols_preferred(polar, desc, &kp)))
    return cfear_kstrong_cols_device(ctx, polar, desc, &kp, &fz, (flags & CFEAR_ROWKEYS_TILE_SWEEP) ? 3 : ((flags >> 4) & 3));
  cfear_polar_desc rd{};                                      // the rotated images: rows = azimuths
  rd.rows = desc->cols; rd.cols = desc->rows; rd.stride = (desc->rows + 15) & ~15; rd.batch = desc->batch;
  rd.batch_stride = (int64_t)rd.rows * rd.stride;
  uint8_t* rot = (uint8_t*)cfear_workspace(ctx, 0, (size_t)rd.batch_stride * rd.batch);
  if (!rot) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  const int rc = cfear_rotate_ccw_device(ctx, polar, desc, rot, rd.stride, rd.batch_stride);
  if (rc != CFEAR_OK) return rc;
  return cfear_kstrong_device(ctx, rot, &rd, &kp, &none, true, &fz);
}

extern "C" int cfear_filter_kstrongest(cfear_ctx* ctx, const uint8_t* polar, const cfear_polar_desc* desc,
                                       const cfear_kstrong_params* par, const cfear_kstrong_out* out) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!polar || !par || !out) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  int rc = check_desc(ctx, desc);
  if (rc != CFEAR_OK) return rc;
  if (par->k_strongest < 1 || par->k_strongest > kMaxK)
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "k_strongest must be in [1,%d]", kMaxK);
  if (!(par->range_res > 0.f))
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "range_res must be > 0");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int rows = desc->rows, k = par->k_strongest, batch = desc->batch;
  const size_t nsel = (size_t)batch * rows * k;
  const bool dev = cfear_is_device_ptr(polar);
  const bool want_cloud = out->xyzi || out->n_points;
  const bool want_pk = par->want_peaks && (out->is_peak || out->xyzi_peaks || out->n_peaks);
  // device buffers: caller's (device mode) or workspace
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const bool need_ws_sel = !dev || !out->sel_range || !out->sel_intensity || !out->sel_count || (want_pk && !out->is_peak);
  size_t o_sr = 0, o_si = 0, o_sc = 0, o_pk = 0, o_xyz = 0, o_np = 0, o_xyzp = 0, o_npp = 0;
  if (need_ws_sel || !dev) {
    o_sr = carve(nsel * 4); o_si = carve(nsel); o_sc = carve((size_t)batch * rows * 4); o_pk = carve(nsel);
  }
  if (!dev) {
    o_xyz = carve(nsel * 16); o_np = carve((size_t)batch * 4);
    o_xyzp = carve(nsel * 16); o_npp = carve((size_t)batch * 4);
  }
  char* ws = off ? (char*)cfear_workspace(ctx, 1, off) : nullptr;
  if (off && !ws) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed (%zu bytes)", off);
  cfear_kstrong_out d;
  d.sel_range = (dev && out->sel_range) ? out->sel_range : (int32_t*)(ws + o_sr);
  d.sel_intensity = (dev && out->sel_intensity) ? out->sel_intensity : (uint8_t*)(ws + o_si);
  d.sel_count = (dev && out->sel_count) ? out->sel_count : (int32_t*)(ws + o_sc);
  d.is_peak = want_pk ? ((dev && out->is_peak) ? out->is_peak : (uint8_t*)(ws + o_pk)) : nullptr;
  if (dev) {
    d.xyzi = out->xyzi; d.n_points = out->n_points;
    d.xyzi_peaks = want_pk ? out->xyzi_peaks : nullptr; d.n_peaks = want_pk ? out->n_peaks : nullptr;
  } else {
    d.xyzi = want_cloud ? (float*)(ws + o_xyz) : nullptr;
    d.n_points = want_cloud ? (int32_t*)(ws + o_np) : nullptr;
    const bool pc = want_pk && (out->xyzi_peaks || out->n_peaks);
    d.xyzi_peaks = pc ? (float*)(ws + o_xyzp) : nullptr;
    d.n_peaks = pc ? (int32_t*)(ws + o_npp) : nullptr;
  }
  const uint8_t* d_polar = polar;
  cfear_polar_desc dd = *desc;
  if (!dev) {
    // stage the images densely: [batch][rows][stride]
    const size_t img_bytes = (size_t)rows * desc->stride;
    uint8_t* st = (uint8_t*)cfear_workspace(ctx, 0, img_bytes * batch);
    if (!st) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
    const int64_t bs = batch > 1 ? desc->batch_stride : (int64_t)img_bytes;
    for (int b = 0; b < batch; b++)
      CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(st + (size_t)b * img_bytes, polar + (size_t)b * bs, img_bytes,
                                          hipMemcpyHostToDevice, ctx->stream));
    d_polar = st;
    dd.batch_stride = (int64_t)img_bytes;
  }
  rc = cfear_kstrong_device(ctx, d_polar, &dd, par, &d);
  if (rc != CFEAR_OK) return rc;
  if (!dev) {
    auto back = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
      return dst ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream) : hipSuccess;
    };
    CFEAR_HIP_CHECK(ctx, back(out->sel_range, d.sel_range, nsel * 4));
    CFEAR_HIP_CHECK(ctx, back(out->sel_intensity, d.sel_intensity, nsel));
    CFEAR_HIP_CHECK(ctx, back(out->sel_count, d.sel_count, (size_t)batch * rows * 4));
    if (want_pk) CFEAR_HIP_CHECK(ctx, back(out->is_peak, d.is_peak, nsel));
    if (d.xyzi) CFEAR_HIP_CHECK(ctx, back(out->xyzi, d.xyzi, nsel * 16));
    if (d.n_points) CFEAR_HIP_CHECK(ctx, back(out->n_points, d.n_points, (size_t)batch * 4));
    if (d.xyzi_peaks) CFEAR_HIP_CHECK(ctx, back(out->xyzi_peaks, d.xyzi_peaks, nsel * 16));
    if (d.n_peaks) CFEAR_HIP_CHECK(ctx, back(out->n_peaks, d.n_peaks, (size_t)batch * 4));
    CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return CFEAR_OK;
}


extern "C" int cfear_filter_kstrongest_legacy(cfear_ctx* ctx, const uint8_t* polar, const cfear_polar_desc* desc, int32_t k_strongest,
                                              double z_min, double range_res, double min_distance, float* xyzi, int32_t* n_points,
                                              int32_t cap_points) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!polar || !xyzi || !n_points || cap_points <= 0) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  int rc = check_desc(ctx, desc);
  if (rc != CFEAR_OK) return rc;
  if (k_strongest < 1 || k_strongest > kMaxK) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "k_strongest must be in [1,%d]", kMaxK);
  if (!(range_res > 0.0)) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "range_res must be > 0");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int rows = desc->rows, cols = desc->cols, batch = desc->batch, k = k_strongest;
  const bool dev = cfear_is_device_ptr(polar);
  if (dev != cfear_is_device_ptr(xyzi) || dev != cfear_is_device_ptr(n_points))
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "polar, xyzi and n_points must all be host or all be device memory");
  const size_t img_bytes = (size_t)rows * desc->stride;
  const uint8_t* d_polar = polar;
  long long bs = batch > 1 ? desc->batch_stride : (long long)img_bytes;
  if (!dev) {
    uint8_t* st = (uint8_t*)cfear_workspace(ctx, 0, img_bytes * batch);
    if (!st) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
    for (int b = 0; b < batch; b++)
      CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(st + (size_t)b * img_bytes, polar + (size_t)b * bs, img_bytes, hipMemcpyHostToDevice, ctx->stream));
    d_polar = st;
    bs = (long long)img_bytes;
  }
  // workspace: reversed masked images | first-bin records | sel arrays | float trig tables | (host mode) cloud
  const int rev_stride = (cols + 15) / 16 * 16;
  const size_t nsel = (size_t)batch * rows * k;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_rev = carve((size_t)batch * rows * rev_stride), o_first = carve((size_t)batch * rows * 8);
  const size_t o_sr = carve(nsel * 4), o_si = carve(nsel), o_sc = carve((size_t)batch * rows * 4), o_trig = carve((size_t)rows * 8);
  const size_t o_xyz = carve(dev ? 0 : (size_t)batch * cap_points * 16), o_np = carve((size_t)batch * 4);
  char* ws = (char*)cfear_workspace(ctx, 1, off);
  if (!ws) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed (%zu bytes)", off);
  {
    std::vector<float> h(2 * (size_t)rows);                // host cosf / sinf of the FLOAT theta: bit-exact with glibc
    for (int bearing = 0; bearing < rows; bearing++) {
      const float theta = ((float)(bearing + 1) / rows) * 2 * M_PI;            // radar_filters.cpp:52
      h[bearing] = std::cos(theta);
      h[rows + bearing] = std::sin(theta);
    }
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(ws + o_trig, h.data(), h.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  int u_z = (int)std::ceil(z_min);                         // uchar v < z_min  <=>  v < ceil(z_min)
  u_z = std::max(0, std::min(256, u_z));
  {
    ProfScope ps(ctx, "kstrong_legacy_prepare");
    hipLaunchKernelGGL(legacy_prepare_kernel, dim3((rows + 3) / 4, batch), dim3(256), 0, ctx->stream, d_polar, rows, cols, desc->stride, bs,
                       u_z, (uint8_t*)(ws + o_rev), rev_stride, (int32_t*)(ws + o_first));
  }
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  cfear_polar_desc rd{rows, cols, rev_stride, batch, (int64_t)rows * rev_stride};
  cfear_kstrong_params kp{k, 1.0f, 1.0f, 0.0f, 0};
  cfear_kstrong_out o{};
  o.sel_range = (int32_t*)(ws + o_sr); o.sel_intensity = (uint8_t*)(ws + o_si); o.sel_count = (int32_t*)(ws + o_sc);
  rc = cfear_kstrong_device(ctx, (const uint8_t*)(ws + o_rev), &rd, &kp, &o);
  if (rc != CFEAR_OK) return rc;
  float* d_xyzi = dev ? xyzi : (float*)(ws + o_xyz);
  int32_t* d_np = dev ? n_points : (int32_t*)(ws + o_np);
  {
    ProfScope ps(ctx, "kstrong_legacy_cloud");
    hipLaunchKernelGGL(legacy_cloud_kernel, dim3(batch), dim3(256), (size_t)(rows + 1) * 4, ctx->stream, o.sel_range, o.sel_intensity,
                       o.sel_count, (const int32_t*)(ws + o_first), (const float*)(ws + o_trig), (const float*)(ws + o_trig) + rows, rows, cols, k,
                       range_res, min_distance * min_distance, d_xyzi, d_np, cap_points);
  }
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  if (!dev) {
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(n_points, d_np, (size_t)batch * 4, hipMemcpyDeviceToHost, ctx->stream));
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(xyzi, d_xyzi, (size_t)batch * cap_points * 16, hipMemcpyDeviceToHost, ctx->stream));
    CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < batch; b++)
      if (n_points[b] > cap_points) return cfear_set_error(ctx, CFEAR_ERR_CAPACITY, "image %d: %d points > cap_points %d", b, n_points[b], cap_points);
  }
  return CFEAR_OK;
}

// The kernel arguments that follow from the filter's parameters alone (rows = azimuths, cols = bins of the ROTATED image);
// D / DL / nch = the chunk geometry (cacfar_rows_kernel's template arguments).
static void cfar_derive(CfarArgs& a, const cfear_cacfar_params* par, int rows, int cols, int& D, int& DL, int& nch) {
  a.rows = rows; a.cols = cols;
  a.window = par->window_size; a.guard = par->nb_guard_cells;
  const double false_alarm_rate_ = (double)par->false_alarm_rate;
  const double N = par->window_size * 2;                                     // cfar.cpp:32
  a.scaling = N * (std::pow(false_alarm_rate_, -1. / N) - 1.);               // cfar.cpp:12-16
  a.range_res = (double)par->range_res;
  a.static_threshold = (double)par->z_min;
  a.min_distance = (double)par->min_distance;
  a.max_distance = par->max_distance;
  {
    // the candidate pre-test in integers.  intensity > static_threshold for integer intensities: the smallest passing value;
    // range > min_distance && range < max_distance (cfar.cpp:43-45, range = range_res * bin in double): the bin interval,
    // found with the reference's own expression
    const double st = a.static_threshold;
    a.thr_i = st < 0.0 ? 0 : (st >= 255.0 ? 256 : (int)std::floor(st) + 1);
    int lo = 0, hi = cols;
    while (lo < cols && !(a.range_res * (double)lo > a.min_distance)) lo++;
    while (hi > 0 && !(a.range_res * (double)(hi - 1) < a.max_distance)) hi--;
    a.bin_lo = lo; a.bin_hi = hi;
    if (a.thr_i >= 256 || hi <= lo) a.bin_hi = a.bin_lo = 0;                  // nothing passes the static threshold / the range window
    // bins the arithmetic of the bins in [bin_lo, bin_hi) can reach
    const long long reach = a.bin_hi > 0 ? std::min<long long>(cols, (long long)a.bin_hi - 1 + a.guard + a.window) : 0;
    a.need_cols = (int)((reach + 15) / 16 * 16);
    a.lut_ok = cfar_build_lut(a, a.lut) ? 1 : 0;
    // lower-bound pre-filter: the aligned quads inside the windows of every bin of an 8-bin block
    auto floor_div = [](int x, int y) { return x >= 0 ? x / y : -((-x + y - 1) / y); };
    auto ceil_div = [&](int x, int y) { return -floor_div(-x, y); };
    a.pa0 = ceil_div(7 - a.guard - a.window, 4); a.pa1 = floor_div(-a.guard, 4);
    a.pb0 = ceil_div(7 + a.guard, 4); a.pb1 = floor_div(a.guard + a.window, 4);
    if (a.pa1 < a.pa0) a.pa1 = a.pa0;
    if (a.pb1 < a.pb0) a.pb1 = a.pb0;
    a.pre_on = a.lut_ok && a.guard + a.window <= 1024 && (a.pa1 > a.pa0 || a.pb1 > a.pb0);
    if (!a.pre_on) a.pa0 = a.pa1 = a.pb0 = a.pb1 = 0;
    a.pad_lo = a.pre_on ? (std::max(0, -a.pa0) + 3) / 4 * 4 : 0;
    a.pad_hi = a.pre_on ? std::max(0, a.pb1) + 2 : 0;
    a.kappa_lb = (float)(a.scaling / (2.0 * (double)a.window) * (1.0 - 3e-5));
  }
  // chunk geometry: D dwords per lane and chunk.  Per-chunk overhead ~ 60 wave instructions, per dword of a lane ~ 25:
  // the D in {4, 6, 8} with the cheapest cover of the reachable bins (without the pre-filter only D = 4 is built)
  D = 4; DL = 4; nch = std::max(1, (a.need_cols + 1023) / 1024);
  if (a.pre_on) {
    // measured on the Kvarntorp rows: ~10 us per chunk and ~9 us per dword of a lane (per 204 800 rows)
    auto cost = [](int n, int d, int dl) { return (long long)n * 10 + (long long)((n - 1) * d + dl) * 9; };
    long long best = cost(nch, 4, 4);
    for (int d : {6, 8}) {
      const int n = std::max(1, (a.need_cols + 256 * d - 1) / (256 * d));
      if (cost(n, d, d) < best) { best = cost(n, d, d); D = d; DL = d; nch = n; }
      if (n == 2)                                          // two chunks: the second may be shorter
        for (int dl = 2; dl < d; dl += 2)
          if (256 * d + 256 * dl >= a.need_cols && cost(2, d, dl) < best) { best = cost(2, d, dl); D = d; DL = dl; nch = 2; }
    }
  }
  a.colsp = (nch - 1) * 256 * D + 256 * DL;
}

static size_t cfar_cols_lds(const CfarArgs& a, int D) {
  return 1024 + (size_t)kCfarTile * ((size_t)a.need_cols + 16) + (size_t)kCfarColsWaves * cfar_cols_wave_lds(a.colsp, a.pad_lo, a.pad_hi, 256 * D);
}

// [range bins][azimuths] sources through cacfar_cols_kernel: sd = the SOURCE images (rows = bins, cols = azimuths).
bool cfear_cacfar_cols_supported(const uint8_t* d_src, const cfear_polar_desc* sd, const cfear_cacfar_params* par) {
  if (sd->cols % kCfarTile != 0 || sd->rows % 16 != 0 || sd->stride % 16 != 0 || (uintptr_t)d_src % 16 != 0 ||
      (sd->batch > 1 && sd->batch_stride % 16 != 0) || (int64_t)sd->rows * sd->stride >= ((int64_t)1 << 31))
    return false;
  CfarArgs a;
  memset(&a, 0, sizeof(a));
  int D, DL, nch;
  cfar_derive(a, par, sd->cols, sd->rows, D, DL, nch);
  return a.colsp <= 4096 && cfar_cols_lds(a, D) <= 160 * 1024 - 256;
}

// Device-side CA-CFAR entry (also used by the odometry pipeline).  With `fused` the rows kernel leaves per-row key lists for
// surface_prep_kernel (cfear_cacfar_fused, common.hpp) and no cloud is built here; fused->bins_major: desc describes the
// [range bins][azimuths] SOURCE images and the decode is fused into the filter (cacfar_cols_kernel).
int cfear_cacfar_device(cfear_ctx* ctx, const uint8_t* d_polar, const cfear_polar_desc* desc,
                        const cfear_cacfar_params* par, float* d_xyzi, int32_t* d_n_points,
                        int32_t cap_points, uint8_t* d_det_mask, const cfear_cacfar_fused* fused) {
  const bool keys = fused && fused->row_keys;
  const bool cols_route = keys && fused->bins_major;
  if (cols_route && !cfear_cacfar_cols_supported(d_polar, desc, par))
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "fused CA-CFAR decode: unsupported image geometry");
  const int rows = cols_route ? desc->cols : desc->rows, cols = cols_route ? desc->rows : desc->cols, batch = desc->batch;
  const int words = (cols + 63) / 64;
  CfarArgs a;
  memset(&a, 0, sizeof(a));
  a.polar = d_polar; a.stride = desc->stride; a.batch = batch;
  a.batch_stride = batch > 1 ? desc->batch_stride : (int64_t)desc->rows * desc->stride;
  a.total_rows = (long long)batch * rows;
  a.words = words;
  int D, DL, nch;
  cfar_derive(a, par, rows, cols, D, DL, nch);
  if (keys) {
    a.row_keys = fused->row_keys; a.row_cnt = fused->row_cnt; a.kcap = fused->kcap;
  } else {
    size_t bits_bytes = (size_t)batch * rows * words * 8, cnt_bytes = (size_t)batch * rows * 4;
    char* ws = (char*)cfear_workspace(ctx, 3, bits_bytes + cnt_bytes + 256);
    if (!ws) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
    a.det_bits = (unsigned long long*)ws;
    a.det_count = (int32_t*)(ws + (bits_bytes + 255) / 256 * 256);
  }
  a.list_cap = cols_route ? cfar_cols_list_cap(256 * D) : 256 * D + kCfarListSlack;
  if (cols_route) {
    using ColsFn = void (*)(const CfarArgs, int);
    const bool pre = a.pre_on != 0;
    ColsFn fn = nullptr;
    if (D == 4) {
      fn = pre ? cacfar_cols_kernel<4, 4, 4, true> : cacfar_cols_kernel<4, 4, 4, false>;
    } else if (D == 6) {
      static const ColsFn f6s[2] = {cacfar_cols_kernel<6, 2, 2, true>, cacfar_cols_kernel<6, 2, 4, true>};
      fn = DL != D ? f6s[DL / 2 - 1] : (ColsFn)cacfar_cols_kernel<6, 2, 6, true>;   // (colsp <= 4096: at most two chunks of 1536)
    } else {
      static const ColsFn f8s[3] = {cacfar_cols_kernel<8, 2, 2, true>, cacfar_cols_kernel<8, 2, 4, true>, cacfar_cols_kernel<8, 2, 6, true>};
      fn = DL != D ? f8s[DL / 2 - 1] : (ColsFn)cacfar_cols_kernel<8, 2, 8, true>;
    }
    const size_t lds = cfar_cols_lds(a, D);
    { const int rc_lds = cfear_allow_lds(ctx, (const void*)fn, lds); if (rc_lds != CFEAR_OK) return rc_lds; }
    const int n_cu = ctx->n_cu;
    const int tiles = rows / kCfarTile;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, 160 * 1024 / lds));
    const long long per_xcd = (long long)tiles * ((batch + kXcds - 1) / kXcds);
    const int slots = (int)std::max<long long>(1, std::min<long long>(per_xcd, std::max(1, n_cu * per_cu / kXcds)));
    ProfScope ps(ctx, "cacfar_cols");
    hipLaunchKernelGGL(fn, dim3((unsigned)(slots * kXcds)), dim3(64 * kCfarColsWaves), lds, ctx->stream, a, tiles);
    CFEAR_HIP_CHECK(ctx, hipGetLastError());
    return CFEAR_OK;
  }
  {
    const bool pre = a.pre_on != 0;
    using KernelFn = void (*)(const CfarArgs);
    KernelFn fn = nullptr;
    if (D == 4) {
      const bool wide = nch > 4;
      static const KernelFn f4[2][2][2] = {   // [wide][keys][pre]
          {{cacfar_rows_kernel<4, 4, 4, false, false>, cacfar_rows_kernel<4, 4, 4, false, true>}, {cacfar_rows_kernel<4, 4, 4, true, false>, cacfar_rows_kernel<4, 4, 4, true, true>}},
          {{cacfar_rows_kernel<4, 8, 4, false, false>, cacfar_rows_kernel<4, 8, 4, false, true>}, {cacfar_rows_kernel<4, 8, 4, true, false>, cacfar_rows_kernel<4, 8, 4, true, true>}}};
      fn = f4[wide ? 1 : 0][keys ? 1 : 0][pre ? 1 : 0];
    } else if (D == 6) {
      static const KernelFn f6[2][2] = {{cacfar_rows_kernel<6, 2, 6, false, true>, cacfar_rows_kernel<6, 2, 6, true, true>},
                                        {cacfar_rows_kernel<6, 6, 6, false, true>, cacfar_rows_kernel<6, 6, 6, true, true>}};
      static const KernelFn f6s[2][2] = {{cacfar_rows_kernel<6, 2, 2, false, true>, cacfar_rows_kernel<6, 2, 2, true, true>},
                                         {cacfar_rows_kernel<6, 2, 4, false, true>, cacfar_rows_kernel<6, 2, 4, true, true>}};
      fn = DL != D ? f6s[DL / 2 - 1][keys ? 1 : 0] : f6[nch > 2 ? 1 : 0][keys ? 1 : 0];
    } else {
      static const KernelFn f8[2][2] = {{cacfar_rows_kernel<8, 2, 8, false, true>, cacfar_rows_kernel<8, 2, 8, true, true>},
                                        {cacfar_rows_kernel<8, 4, 8, false, true>, cacfar_rows_kernel<8, 4, 8, true, true>}};
      static const KernelFn f8s[3][2] = {{cacfar_rows_kernel<8, 2, 2, false, true>, cacfar_rows_kernel<8, 2, 2, true, true>},
                                         {cacfar_rows_kernel<8, 2, 4, false, true>, cacfar_rows_kernel<8, 2, 4, true, true>},
                                         {cacfar_rows_kernel<8, 2, 6, false, true>, cacfar_rows_kernel<8, 2, 6, true, true>}};
      fn = DL != D ? f8s[DL / 2 - 1][keys ? 1 : 0] : f8[nch > 2 ? 1 : 0][keys ? 1 : 0];
    }
    const size_t rows_lds = 1024 + (size_t)kRowsPerBlock * cfar_wave_lds(a.colsp, a.pad_lo, a.pad_hi, keys, 256 * D);
    if (rows_lds > 64 * 1024)
      { const int rc_lds = cfear_allow_lds(ctx, (const void*)fn, rows_lds); if (rc_lds != CFEAR_OK) return rc_lds; }
    // persistent wavefronts: as many workgroups as the chip holds at this LDS footprint (160 KiB per CU)
    const int n_cu = ctx->n_cu;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, 160 * 1024 / rows_lds));
    const long long want = (a.total_rows + kRowsPerBlock - 1) / kRowsPerBlock;
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>(want, (long long)n_cu * per_cu));
    ProfScope ps(ctx, "cacfar_rows");
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), rows_lds, ctx->stream, a);
  }
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  if (keys) return CFEAR_OK;
  double *d_cos = nullptr, *d_sin = nullptr;
  int rc = upload_trig(ctx, rows, &d_cos, &d_sin);
  if (rc != CFEAR_OK) return rc;
  CfarCloudArgs c;
  c.polar = d_polar; c.rows = rows; c.cols = cols; c.stride = desc->stride; c.batch_stride = a.batch_stride;
  c.det_bits = a.det_bits; c.det_count = a.det_count; c.words = words;
  c.cos_t = d_cos; c.sin_t = d_sin; c.range_res = a.range_res;
  c.xyzi = d_xyzi; c.n_points = d_n_points; c.cap_points = cap_points; c.det_mask = d_det_mask;
  {
    ProfScope ps(ctx, "cacfar_cloud");
    hipLaunchKernelGGL(cacfar_cloud_kernel, dim3(batch, kCfarCloudSplit), dim3(256), (size_t)(rows + 1) * 4 + 4 * 64 * 64 * 2, ctx->stream, c);
  }
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}

extern "C" int cfear_filter_cacfar(cfear_ctx* ctx, const uint8_t* polar, const cfear_polar_desc* desc,
                                   const cfear_cacfar_params* par, float* xyzi, int32_t* n_points,
                                   int32_t cap_points, uint8_t* det_mask) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!polar || !par || !xyzi || !n_points || cap_points <= 0)
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  int rc = check_desc(ctx, desc);
  if (rc != CFEAR_OK) return rc;
  if (par->window_size < 1 || par->nb_guard_cells < 0 || !(par->range_res > 0.f))
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "bad CFAR parameters");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int rows = desc->rows, cols = desc->cols, batch = desc->batch;
  const bool dev = cfear_is_device_ptr(polar);
  if (dev) return cfear_cacfar_device(ctx, polar, desc, par, xyzi, n_points, cap_points, det_mask);
  const size_t img_bytes = (size_t)rows * desc->stride;
  const size_t xyz_bytes = (size_t)batch * cap_points * 16, mask_bytes = det_mask ? (size_t)batch * rows * cols : 0;
  char* st = (char*)cfear_workspace(ctx, 0, img_bytes * batch);
  char* ws = (char*)cfear_workspace(ctx, 1, xyz_bytes + mask_bytes + 1024);
  if (!st || !ws) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  const int64_t bs = batch > 1 ? desc->batch_stride : (int64_t)img_bytes;
  for (int b = 0; b < batch; b++)
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(st + (size_t)b * img_bytes, polar + (size_t)b * bs, img_bytes,
                                        hipMemcpyHostToDevice, ctx->stream));
  cfear_polar_desc dd = *desc;
  dd.batch_stride = (int64_t)img_bytes;
  float* d_xyzi = (float*)ws;
  int32_t* d_np = (int32_t*)(ws + (xyz_bytes + 255) / 256 * 256);
  uint8_t* d_mask = det_mask ? (uint8_t*)(ws + (xyz_bytes + 255) / 256 * 256 + 256) : nullptr;
  rc = cfear_cacfar_device(ctx, (const uint8_t*)st, &dd, par, d_xyzi, d_np, cap_points, d_mask);
  if (rc != CFEAR_OK) return rc;
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(n_points, d_np, (size_t)batch * 4, hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(xyzi, d_xyzi, xyz_bytes, hipMemcpyDeviceToHost, ctx->stream));
  if (det_mask) CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(det_mask, d_mask, mask_bytes, hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (int b = 0; b < batch; b++)
    if (n_points[b] > cap_points)
      return cfear_set_error(ctx, CFEAR_ERR_CAPACITY, "image %d: %d detections > cap_points %d", b, n_points[b], cap_points);
  return CFEAR_OK;
}

// ---- image decode of the non-Oxford sensors: rotate 90 degrees counter-clockwise ------------------------
// radarDriver::Callback (radar_driver.cpp:74-90): Navtech drivers other than Oxford's publish the sweep as
// [range bins][azimuths]; cv::rotate(ROTATE_90_COUNTERCLOCKWISE) turns it into the rows = azimuth layout that
// Process() and every filter expect:  dst[i][j] = src[j][src.cols - 1 - i].
// HBM-bound byte transpose: tiles through LDS, 16-byte loads along the source rows, 4 x 4 byte blocks
// transposed in registers, dword stores along the destination rows (byte accesses only on ragged edges /
// unaligned strides).
namespace {

constexpr int kRotTile = 64;                  // general kernel (wide sources): 64 x 64 byte tiles
constexpr int kRotPitchW = kRotTile / 4 + 1;  // LDS row pitch in dwords: odd, spreads the 4 x 4 block reads over the banks

struct RotArgs {
  const uint8_t* src; uint8_t* dst;
  int rows_in, cols_in, stride_in, stride_out;
  long long bs_in, bs_out;
  int batch0, vec_in, vec_out, vec16_out;
};

// General kernel: one 64 x 64 byte tile per workgroup.  Load: 16-byte pieces along the source rows -> LDS.  Each thread then owns
// 4 x 4 byte blocks: 4 dword reads down a column of blocks, a byte transpose in registers (v_perm_b32) and 4 dword
// stores; 16 consecutive lanes write 64 contiguous bytes of one destination row.
__global__ __launch_bounds__(256) void rotate_ccw_kernel(const RotArgs a) {
  __shared__ uint32_t tile[kRotTile][kRotPitchW];
  const int t = threadIdx.x;
  const int C0 = blockIdx.x * kRotTile, R0 = blockIdx.y * kRotTile;
  const size_t img = (size_t)a.batch0 + blockIdx.z;
  const uint8_t* src = a.src + img * (size_t)a.bs_in;
  uint8_t* dst = a.dst + img * (size_t)a.bs_out;
  constexpr int kSeg = kRotTile / 16;            // 16-byte pieces per tile row
#pragma unroll
  for (int q = 0; q < kRotTile * kSeg / 256; q++) {
    const int idx = t + 256 * q, r = idx / kSeg, s = idx % kSeg;
    const int row = R0 + r, col = C0 + 16 * s;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < a.rows_in && col < a.cols_in) {
      const uint8_t* p = src + (size_t)row * a.stride_in + col;
      if (a.vec_in && col + 16 <= a.cols_in) {
        v = *(const uint4*)p;
      } else {
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; k++)
          if (col + k < a.cols_in) w[k >> 2] |= (uint32_t)p[k] << (8 * (k & 3));
        v = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    tile[r][4 * s] = v.x; tile[r][4 * s + 1] = v.y; tile[r][4 * s + 2] = v.z; tile[r][4 * s + 3] = v.w;
  }
  __syncthreads();
  constexpr int kBlk = kRotTile / 4;             // 4 x 4 blocks per tile side
#pragma unroll
  for (int q = 0; q < kBlk * kBlk / 256; q++) {
    const int blk = t + 256 * q, br = blk % kBlk, bc = blk / kBlk;   // source rows 4 br .. +3, source columns 4 bc .. +3
    const int j0 = R0 + 4 * br;                  // destination columns j0 .. j0 + 3 = source rows
    if (j0 >= a.rows_in || C0 + 4 * bc >= a.cols_in) continue;
    const uint32_t w0 = tile[4 * br][bc], w1 = tile[4 * br + 1][bc], w2 = tile[4 * br + 2][bc], w3 = tile[4 * br + 3][bc];
    const uint32_t t0 = __builtin_amdgcn_perm(w1, w0, 0x05010400u), t1 = __builtin_amdgcn_perm(w1, w0, 0x07030602u);
    const uint32_t t2 = __builtin_amdgcn_perm(w3, w2, 0x05010400u), t3 = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
    uint32_t colw[4];                            // colw[b] = source column 4 bc + b as {row0, row1, row2, row3}
    colw[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u); colw[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    colw[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u); colw[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int c = C0 + 4 * bc + b;             // source column -> destination row cols_in - 1 - c
      if (c < a.cols_in) {
        uint8_t* p = dst + (size_t)(a.cols_in - 1 - c) * a.stride_out + j0;
        if (a.vec_out && j0 + 4 <= a.rows_in) {
          *(uint32_t*)p = colw[b];
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) if (j0 + k < a.rows_in) p[k] = (uint8_t)(colw[b] >> (8 * k));
        }
      }
    }
  }
}

// Narrow sources (cols <= 512: a radar sweep has 400 azimuths): the tile spans whole source rows, so the loads walk
// 128 x cols contiguous bytes and every destination row receives 128 contiguous bytes -- full cache lines both ways.
constexpr int kRotRows = 128;
constexpr int kRotMaxCols = 512;

__global__ __launch_bounds__(256) void rotate_ccw_rows_kernel(const RotArgs a) {
  extern __shared__ uint32_t rtile[];            // [kRotRows][pitch]
  const int t = threadIdx.x;
  const int R0 = blockIdx.x * kRotRows;
  const size_t img = (size_t)a.batch0 + blockIdx.y;
  const uint8_t* src = a.src + img * (size_t)a.bs_in;
  uint8_t* dst = a.dst + img * (size_t)a.bs_out;
  const int ppr = (a.cols_in + 15) >> 4;         // 16-byte pieces per row
  const int pitch = (ppr * 4) | 1;               // dwords, odd
  const int rows_here = min(kRotRows, a.rows_in - R0);
  for (int idx = t; idx < rows_here * ppr; idx += 256) {
    const int r = idx / ppr, s = idx - r * ppr, col = 16 * s;
    const uint8_t* p = src + (size_t)(R0 + r) * a.stride_in + col;
    uint4 v;
    if (a.vec_in && col + 16 <= a.cols_in) {
      { const u32x4 nv = __builtin_nontemporal_load((const u32x4*)p); v = make_uint4(nv.x, nv.y, nv.z, nv.w); }
    } else {
      uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 16; k++)
        if (col + k < a.cols_in) w[k >> 2] |= (uint32_t)p[k] << (8 * (k & 3));
      v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    uint32_t* o = rtile + (size_t)r * pitch + 4 * s;
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  __syncthreads();
  const int nbc = (a.cols_in + 3) >> 2;          // 4 x 4 blocks along the source columns
  if (a.vec16_out && rows_here == kRotRows) {
    // full tile, 16-byte aligned destination: a thread transposes 16 source rows x 4 source columns and writes four
    // 16-byte pieces (8 lanes fill a 128-byte line of a destination row) -- a quarter of the store instructions
    for (int blk = t; blk < (kRotRows / 16) * nbc; blk += 256) {
      const int br = blk & (kRotRows / 16 - 1), bc = blk / (kRotRows / 16);
      const uint32_t* c0 = rtile + (size_t)(16 * br) * pitch + bc;
      uint32_t colw[4][4];                       // [row group][source column]
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const uint32_t w0 = c0[(4 * g) * pitch], w1 = c0[(4 * g + 1) * pitch], w2 = c0[(4 * g + 2) * pitch], w3 = c0[(4 * g + 3) * pitch];
        const uint32_t t0 = __builtin_amdgcn_perm(w1, w0, 0x05010400u), t1 = __builtin_amdgcn_perm(w1, w0, 0x07030602u);
        const uint32_t t2 = __builtin_amdgcn_perm(w3, w2, 0x05010400u), t3 = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
        colw[g][0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u); colw[g][1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
        colw[g][2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u); colw[g][3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
      }
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int c = 4 * bc + b;
        if (c < a.cols_in)
        {
          u32x4 ov; ov.x = colw[0][b]; ov.y = colw[1][b]; ov.z = colw[2][b]; ov.w = colw[3][b];
          __builtin_nontemporal_store(ov, (u32x4*)(dst + (size_t)(a.cols_in - 1 - c) * a.stride_out + R0 + 16 * br));
        }
      }
    }
    return;
  }
  for (int blk = t; blk < (kRotRows / 4) * nbc; blk += 256) {
    const int br = blk & (kRotRows / 4 - 1), bc = blk / (kRotRows / 4);
    const int j0 = R0 + 4 * br;
    if (j0 >= a.rows_in) continue;
    const uint32_t* c0 = rtile + (size_t)(4 * br) * pitch + bc;
    // rows beyond rows_here were never written: mask them instead of reading garbage into valid bytes
    const uint32_t w0 = c0[0], w1 = (4 * br + 1 < rows_here) ? c0[pitch] : 0u;
    const uint32_t w2 = (4 * br + 2 < rows_here) ? c0[2 * pitch] : 0u, w3 = (4 * br + 3 < rows_here) ? c0[3 * pitch] : 0u;
    const uint32_t t0 = __builtin_amdgcn_perm(w1, w0, 0x05010400u), t1 = __builtin_amdgcn_perm(w1, w0, 0x07030602u);
    const uint32_t t2 = __builtin_amdgcn_perm(w3, w2, 0x05010400u), t3 = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
    uint32_t colw[4];
    colw[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u); colw[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    colw[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u); colw[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int c = 4 * bc + b;
      if (c < a.cols_in) {
        uint8_t* p = dst + (size_t)(a.cols_in - 1 - c) * a.stride_out + j0;
        if (a.vec_out && j0 + 4 <= a.rows_in) {
          *(uint32_t*)p = colw[b];
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) if (j0 + k < a.rows_in) p[k] = (uint8_t)(colw[b] >> (8 * k));
        }
      }
    }
  }
}

}  // namespace

// src_desc describes the SOURCE images (rows = range bins, cols = azimuths); dst images are cols x rows.
int cfear_rotate_ccw_device(cfear_ctx* ctx, const uint8_t* d_src, const cfear_polar_desc* sd, uint8_t* d_dst,
                            int dst_stride, int64_t dst_batch_stride) {
  RotArgs a;
  a.src = d_src; a.dst = d_dst;
  a.rows_in = sd->rows; a.cols_in = sd->cols; a.stride_in = sd->stride; a.stride_out = dst_stride;
  a.bs_in = sd->batch > 1 ? sd->batch_stride : 0; a.bs_out = sd->batch > 1 ? dst_batch_stride : 0;
  a.vec_in = ((uintptr_t)d_src % 16 == 0) && (sd->stride % 16 == 0) && (a.bs_in % 16 == 0);
  a.vec_out = ((uintptr_t)d_dst % 4 == 0) && (dst_stride % 4 == 0) && (a.bs_out % 4 == 0);
  a.vec16_out = ((uintptr_t)d_dst % 16 == 0) && (dst_stride % 16 == 0) && (a.bs_out % 16 == 0);
  ProfScope ps(ctx, "rotate_ccw");
  if (sd->cols <= kRotMaxCols) {
    const int ppr = (sd->cols + 15) >> 4;
    const size_t lds = (size_t)kRotRows * ((ppr * 4) | 1) * 4;
    { const int rc_lds = cfear_allow_lds(ctx, (const void*)rotate_ccw_rows_kernel, 160 * 1024); if (rc_lds != CFEAR_OK) return rc_lds; }
    for (int b0 = 0; b0 < sd->batch; b0 += 65535) {
      a.batch0 = b0;
      const dim3 grid((sd->rows + kRotRows - 1) / kRotRows, std::min(65535, sd->batch - b0));
      hipLaunchKernelGGL(rotate_ccw_rows_kernel, grid, dim3(256), lds, ctx->stream, a);
    }
    CFEAR_HIP_CHECK(ctx, hipGetLastError());
    return CFEAR_OK;
  }
  for (int b0 = 0; b0 < sd->batch; b0 += 65535) {
    a.batch0 = b0;
    const dim3 grid((sd->cols + kRotTile - 1) / kRotTile, (sd->rows + kRotTile - 1) / kRotTile, std::min(65535, sd->batch - b0));
    hipLaunchKernelGGL(rotate_ccw_kernel, grid, dim3(256), 0, ctx->stream, a);
  }
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}

extern "C" int cfear_polar_rotate_ccw(cfear_ctx* ctx, const uint8_t* src, const cfear_polar_desc* src_desc, uint8_t* dst,
                                      int32_t dst_stride, int64_t dst_batch_stride) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!src || !dst || !src_desc) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  const cfear_polar_desc& d = *src_desc;
  if (d.rows <= 0 || d.cols <= 0 || d.stride < d.cols || d.batch <= 0 || dst_stride < d.rows ||
      (d.batch > 1 && (d.batch_stride < (int64_t)d.rows * d.stride || dst_batch_stride < (int64_t)d.cols * dst_stride)))
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "bad polar descriptor");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const bool dev = cfear_is_device_ptr(src);
  if (dev != cfear_is_device_ptr(dst))
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "src and dst must both be host or both be device memory");
  if (dev) return cfear_rotate_ccw_device(ctx, src, src_desc, dst, dst_stride, dst_batch_stride);
  // host images: stage densely, rotate, copy back row by row into the caller's pitch
  const size_t in_bytes = (size_t)d.rows * d.stride;
  const int ostride = (d.rows + 15) / 16 * 16;
  const size_t out_bytes = (size_t)d.cols * ostride;
  uint8_t* st = (uint8_t*)cfear_workspace(ctx, 0, in_bytes * d.batch);
  uint8_t* ot = (uint8_t*)cfear_workspace(ctx, 1, out_bytes * d.batch);
  if (!st || !ot) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  const int64_t bs = d.batch > 1 ? d.batch_stride : (int64_t)in_bytes;
  for (int b = 0; b < d.batch; b++)
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(st + (size_t)b * in_bytes, src + (size_t)b * bs, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  cfear_polar_desc dd = d;
  dd.batch_stride = (int64_t)in_bytes;
  const int rc = cfear_rotate_ccw_device(ctx, st, &dd, ot, ostride, (int64_t)out_bytes);
  if (rc != CFEAR_OK) return rc;
  const int64_t obs = d.batch > 1 ? dst_batch_stride : 0;
  for (int b = 0; b < d.batch; b++)
    CFEAR_HIP_CHECK(ctx, hipMemcpy2DAsync(dst + (size_t)b * obs, dst_stride, ot + (size_t)b * out_bytes, ostride, d.rows, d.cols,
                                          hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CFEAR_OK;
}
