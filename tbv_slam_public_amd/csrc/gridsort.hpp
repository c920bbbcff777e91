// gridsort.hpp -- block-wide LDS sort of (grid cell, point index) pairs, shared by the kernels that
// replace a PCL voxel grid / kd-tree radius search with a sort-based uniform grid (surface.hip, coral.hip).
#pragma once
#include "common.hpp"

constexpr int kGridSortThreads = 1024;
constexpr int kGridSortMaxPoints = 16384;        // 64-bit keys: 128 KiB of LDS
constexpr int kGridSortRadixMaxPoints = 8192;    // radix path: 2 x 32 KiB key buffers + 32 KiB counters

#if defined(__HIPCC__)
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
