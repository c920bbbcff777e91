// gridsort.hpp -- block-wide LDS sort of (grid cell, point index) pairs, shared by the kernels that
// replace a PCL voxel grid / kd-tree radius search with a sort-based uniform grid (surface.hip, coral.hip).
#pragma once
#include "common.hpp"

constexpr int kGridSortThreads = 1024;
constexpr int kGridSortMaxPoints = 16384;        // 64-bit keys: 128 KiB of LDS
constexpr int kGridSortRadixMaxPoints = 8192;    // radix path: 2 x 32 KiB key buffers + 32 KiB counters

#if defined(__HIPCC__)
// Two-level variant for the common case (n <= 8192, no crowded grid row): a counting sort by grid ROW with LDS
// atomics, then every element finds its rank among the elements of its row.  With 1 m cells a row of two merged peak
// clouds holds ~80 points: ~80 LDS reads per element replace five radix passes.  The order is the unique order of the
// (cell, index) keys, so the result equals grid_sort_block's bit for bit.  cell_xy(i, ix, iy) -> column / row of point
// i (ix < dbx <= 65535, iy < dby <= 4096); rowcnt: dby + 1 uint32 of LDS outside the 128 KiB key region; red_i, red_m:
// 16 ints each.  Returns npad (like grid_sort_block), or 0 without having sorted when the variant does not apply.
constexpr int kGridRowSortMaxN = 8192;
template <typename CellXY>
__device__ int grid_sort_rows_block(uint8_t* smem, int n, int dbx, int dby, uint32_t* rowcnt, int* red_i, int* red_m,
                                    int max_row, CellXY cell_xy) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (n > kGridRowSortMaxN || dbx > 65535 || dby > 4096) return 0;
  unsigned long long* keys = (unsigned long long*)smem;            // [npad] result
  uint32_t* tmp = (uint32_t*)(smem + 64 * 1024);                    // [n] (ix << 13 | i), grouped by row
  unsigned short* rowid = (unsigned short*)(smem + 96 * 1024);      // [n] grid row of tmp[j]
  for (int y = tid; y <= dby; y += kGridSortThreads) rowcnt[y] = 0;
  __syncthreads();
  int cx[kGridRowSortMaxN / kGridSortThreads], cy[kGridRowSortMaxN / kGridSortThreads];
#pragma unroll
  for (int q = 0; q < kGridRowSortMaxN / kGridSortThreads; q++) {
    const int i = tid + q * kGridSortThreads;
    cx[q] = cy[q] = 0;
    if (i < n) {
      cell_xy(i, cx[q], cy[q]);
      atomicAdd(&rowcnt[cy[q]], 1u);
    }
  }
  __syncthreads();
  bool ok;
  {                                                                 // exclusive scan over the rows + largest row
    const int per_t = (dby + kGridSortThreads) / kGridSortThreads;  // rows per thread (<= 5)
    const int y0 = tid * per_t;
    int tot = 0, mx = 0;
    for (int y = y0; y < min(dby, y0 + per_t); y++) { const int c = (int)rowcnt[y]; tot += c; mx = max(mx, c); }
    const int incl = wave_incl_scan_i32(tot);
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
    if (lane == 63) red_i[wave] = incl;
    if (lane == 0) red_m[wave] = mx;
    __syncthreads();
    int run = incl - tot;
    for (int wv = 0; wv < wave; wv++) run += red_i[wv];
    int mxall = 0;
    for (int wv = 0; wv < 16; wv++) mxall = max(mxall, red_m[wv]);
    ok = mxall <= max_row;
    for (int y = y0; y < min(dby, y0 + per_t); y++) { const int c = (int)rowcnt[y]; rowcnt[y] = (uint32_t)run; run += c; }
    __syncthreads();
  }
  if (!ok) return 0;
#pragma unroll
  for (int q = 0; q < kGridRowSortMaxN / kGridSortThreads; q++) {
    const int i = tid + q * kGridSortThreads;
    if (i < n) {
      const uint32_t pos = atomicAdd(&rowcnt[cy[q]], 1u);           // afterwards rowcnt[y] = end of row y
      tmp[pos] = ((uint32_t)cx[q] << 13) | (uint32_t)i;
      rowid[pos] = (unsigned short)cy[q];
    }
  }
  __syncthreads();
  const int npad = (n + kGridSortThreads - 1) / kGridSortThreads * kGridSortThreads;
  for (int j = tid; j < npad; j += kGridSortThreads) {
    if (j >= n) { keys[j] = ~0ull; continue; }
    const int y = rowid[j];
    const int s = y > 0 ? (int)rowcnt[y - 1] : 0, e = (int)rowcnt[y];
    const uint32_t key = tmp[j];
    int rank = 0;
    for (int i = s; i < e; i++) rank += tmp[i] < key;
    keys[s + rank] = ((unsigned long long)((uint32_t)y * (uint32_t)dbx + (key >> 13)) << 32) | (key & 8191u);
  }
  __syncthreads();
  return max(npad, kGridSortThreads);
}

// Sorts the n points of a 1024-thread workgroup by (cell, index): cells ascending, points of a cell in input
// order (stable).  cell_of(i) -> uint32 cell id < n_cells.  On return smem holds npad (>= n, a multiple of
// 1024) 64-bit keys (cell << 32 | index), padding = ~0; thread t owns elements [t * npad/1024, ...).
// red_i: 16 ints of LDS outside the key region.  Returns npad.  Block-wide collective (ends with a barrier).
//   n <= 8192 and cell bits + index bits <= 32: packed 32-bit keys, stable LSD radix sort on the cell digits
//   (4 bits per pass, thread-contiguous chunks keep the input order, per-thread u16 digit counters);
//   otherwise: bitonic sort of the 64-bit keys.
template <typename CellFn>
__device__ int grid_sort_block(uint8_t* smem, int n, long long n_cells, int* red_i, CellFn cell_of) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  unsigned long long* keys = (unsigned long long*)smem;
  int npad = (n + kGridSortThreads - 1) / kGridSortThreads * kGridSortThreads;   // radix path: any multiple of 1024
  if (npad < kGridSortThreads) npad = kGridSortThreads;
  int ib = 10;                                             // index bits: 2^ib >= npad
  while ((1 << ib) < npad) ib++;
  int vb = 1;                                              // cell-index bits
  while (((long long)1 << vb) < n_cells) vb++;
  const bool radix = (npad <= kGridSortRadixMaxPoints) && (vb + ib <= 32);
  if (!radix) {                                            // the bitonic network needs a power of two
    npad = 1024;
    while (npad < n) npad <<= 1;
  }
  if (radix) {
    uint32_t* kA = (uint32_t*)smem;
    uint32_t* kB = kA + npad;
    unsigned short* cnt = (unsigned short*)(kB + npad);   // [16][1024]
    const int per = npad / kGridSortThreads;               // 1..8 consecutive elements per thread
    for (int i = tid; i < npad; i += kGridSortThreads) {
      uint32_t key = 0xFFFFFFFFu;
      if (i < n) key = (cell_of(i) << ib) | (uint32_t)i;
      kA[i] = key;
    }
    __syncthreads();
    uint32_t* src = kA;
    uint32_t* dst = kB;
    for (int shift = ib; shift < ib + vb; shift += 4) {
#pragma unroll
      for (int d = 0; d < 16; d++) cnt[d * kGridSortThreads + tid] = 0;
      for (int q = 0; q < per; q++) {
        const uint32_t dg = (src[tid * per + q] >> shift) & 15u;
        cnt[dg * kGridSortThreads + tid]++;
      }
      __syncthreads();
      // exclusive scan of the 16 x 1024 counters in (digit, thread) order: 16 consecutive per thread
      unsigned short local[16];
      int tot = 0;
#pragma unroll
      for (int q = 0; q < 16; q++) { local[q] = cnt[tid * 16 + q]; tot += local[q]; }
      const int inc = wave_incl_scan_i32(tot);
      if (lane == 63) red_i[wave] = inc;
      __syncthreads();
      int run = inc - tot;
      for (int wv = 0; wv < wave; wv++) run += red_i[wv];
#pragma unroll
      for (int q = 0; q < 16; q++) { cnt[tid * 16 + q] = (unsigned short)run; run += local[q]; }
      __syncthreads();
      for (int q = 0; q < per; q++) {
        const uint32_t key = src[tid * per + q];
        const uint32_t dg = (key >> shift) & 15u;
        const int pos = cnt[dg * kGridSortThreads + tid]++;
        dst[pos] = key;
      }
      __syncthreads();
      uint32_t* t = src; src = dst; dst = t;
    }
    // widen to the 64-bit (cell, index) form; registers bridge the overlap
    uint32_t mine32[kGridSortRadixMaxPoints / kGridSortThreads];
#pragma unroll
    for (int q = 0; q < kGridSortRadixMaxPoints / kGridSortThreads; q++) mine32[q] = (q < per) ? src[tid * per + q] : 0xFFFFFFFFu;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kGridSortRadixMaxPoints / kGridSortThreads; q++)
      if (q < per) {
        const uint32_t key = mine32[q];
        keys[tid * per + q] = key == 0xFFFFFFFFu ? ~0ull
                                                 : (((unsigned long long)(key >> ib)) << 32) | (key & ((1u << ib) - 1u));
      }
    __syncthreads();
  } else {
    // General path: bitonic sort of 64-bit (cell, index) keys.
    for (int i = tid; i < npad; i += kGridSortThreads) {
      unsigned long long key = ~0ull;
      if (i < n) key = ((unsigned long long)cell_of(i) << 32) | (unsigned)i;
      keys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= npad; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < (npad >> 1); t += kGridSortThreads) {
          const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int hi = lo | j;
          const bool asc = (lo & k) == 0;
          const unsigned long long a = keys[lo], b = keys[hi];
          if ((a > b) == asc) { keys[lo] = b; keys[hi] = a; }
        }
        __syncthreads();
      }
    }
  }
  return npad;
}
#endif
