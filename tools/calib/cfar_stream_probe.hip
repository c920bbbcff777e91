// cfar_stream_probe.hip -- does a column-STREAMING CA-CFAR beat cacfar_cols_kernel's tile transpose?  (VERDICT r05 #2.)
//
// Input layout of the non-Oxford drivers (radar_driver.cpp:74-90): [range bins][azimuths], 400 bytes per bin row.  The azimuth is
// the fast axis, so a lane that owns four adjacent azimuths (one dword per bin row) and walks the bins reads whole 400-byte rows
// with its neighbours -- every 128-byte line consumed once, no transpose, no over-fetch -- and CA-CFAR along range (cfar.cpp:46-60)
// becomes two sliding integer sums of squares per azimuth: with T(c) = [c-g-w, c-g), F(c) = [c+g, c+g+w)
//     S(c) = S(c-1) + I[c-g-1]^2 - I[c-g-w-1]^2 + I[c+g+w-1]^2 - I[c+g-1]^2        (exact in u32)
// The four old rows come from an LDS ring of 2 (g + w) = 100 rows x 400 B = 40 KB per (image, bin segment): at most FOUR segments
// = 400 dword-lanes = 7 wavefronts fit a CU's 160 KB.  This probe implements the main loop for real (ring, sliding sums, the exact
// integer decision 2 S + 1 < lut[I] of cacfar_rows_kernel's step C, key emission per azimuth) on the Kvarntorp geometry -- bins
// whose windows the row's ends cut and the ambiguous case 2 S + 1 == lut[I] are left out (4 % of the bins; they only add work)
// -- checks its detections against a brute-force kernel, and times it: the figure to hold against cacfar_cols (0.33 ms per 512
// sweeps) and cacfar_rows (0.20 ms on pre-rotated sweeps).
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/calib/cfar_stream_probe.hip -o /tmp/cfar_stream_probe && /tmp/cfar_stream_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int AZ = 400, LANES = AZ / 4, IMAGES = 512, BINS = 2352;       // reachable bins of a Kvarntorp row (need_cols)
constexpr int G = 10, W = 40, SPAN = 2 * (G + W);                        // guard, window, rows a window pair spans (= ring slots)
constexpr int BIN_LO = 64, BIN_HI = 2286;                                // decided bins (full windows only in this probe)
constexpr int NSEG = 2, SEGS_PER_WG = 4, THREADS = 448;                  // 2 segments per image, 4 segments (400 lanes) per workgroup
constexpr int SEG_LEN = (BIN_HI - BIN_LO + NSEG - 1) / NSEG;
constexpr int KCAP = 1024, U = 8;                                        // keys per azimuth; rows requested ahead per lane
constexpr int THR = 21;                                                  // intensity > 20 (static threshold)

__global__ void fill_kernel(uint8_t* img, size_t n) {                    // noise floor 10 + Exp(7), a bright bump every ~120 bins
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 32) * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const float u = (float)((h & 0xffffff) + 1) / 16777217.0f;
    float v = 10.0f - 7.0f * logf(u);
    const size_t bin = (i / AZ) % BINS, az = i % AZ, im = i / ((size_t)AZ * BINS);
    const int phase = (int)((bin + 7 * az + 31 * im) % 120);
    if (phase < 3) v += 120.0f + 30.0f * (float)((h >> 24) & 3);
    img[i] = (uint8_t)fminf(v, 255.0f);
  }
}

struct Args {
  const uint8_t* img;
  uint32_t* keys;          // [IMAGES][AZ][KCAP]
  int32_t* cnt;            // [IMAGES][AZ][NSEG]
  uint32_t lut[256];       // fires <=> 2 S + 1 < lut[I]
};

// brute force: one thread per (image, azimuth, bin)
__global__ void brute_kernel(const Args a, uint32_t* det_count, uint32_t* det_hash) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)IMAGES * AZ * (BIN_HI - BIN_LO);
  if (i >= total) return;
  const int az = (int)(i % AZ), bin = BIN_LO + (int)((i / AZ) % (BIN_HI - BIN_LO)), im = (int)(i / ((size_t)AZ * (BIN_HI - BIN_LO)));
  const uint8_t* col = a.img + (size_t)im * BINS * AZ + az;
  const uint32_t c = col[(size_t)bin * AZ];
  if (c < THR) return;
  uint32_t s = 0;
  for (int q = bin - G - W; q < bin - G; q++) s += (uint32_t)col[(size_t)q * AZ] * col[(size_t)q * AZ];
  for (int q = bin + G; q < bin + G + W; q++) s += (uint32_t)col[(size_t)q * AZ] * col[(size_t)q * AZ];
  if (2u * s + 1u < a.lut[c]) {
    atomicAdd(det_count, 1u);
    atomicAdd(det_hash, (uint32_t)(im * 1315423911u) ^ ((uint32_t)az * 2654435761u + (uint32_t)bin * 40503u + c));
  }
}

// PRE: the decision's hot path stays in registers -- S < I^2 ceil(2 w / scaling) + 1 is necessary for a detection (two 24-bit
// multiplies), the table in LDS is read for the survivors only (without it every bin above the static threshold waits for an LDS
// read: four dependent round trips per lane and step, with one or two wavefronts per SIMD to hide them)
template <bool PRE>
__global__ __launch_bounds__(THREADS, 1) void stream_kernel(const Args a, const uint32_t kceil, uint32_t* det_count, uint32_t* det_hash) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* lut = (uint32_t*)smem;
  uint32_t* ring = (uint32_t*)(smem + 1024);                 // [SEGS_PER_WG][SPAN][LANES] dwords
  if (threadIdx.x < 256) lut[threadIdx.x] = a.lut[threadIdx.x];
  for (int i = threadIdx.x; i < SEGS_PER_WG * SPAN * LANES; i += THREADS) ring[i] = 0u;
  __syncthreads();
  const int tid = threadIdx.x;
  if (tid >= SEGS_PER_WG * LANES) return;
  const int sw = tid / LANES, l = tid - sw * LANES;
  const int gseg = blockIdx.x * SEGS_PER_WG + sw;            // global segment: image = gseg / NSEG
  const int im = gseg / NSEG, seg = gseg - im * NSEG;
  const int x0 = BIN_LO + seg * SEG_LEN, x1 = min(x0 + SEG_LEN, BIN_HI);
  const uint32_t* col = (const uint32_t*)(a.img + (size_t)im * BINS * AZ) + l;      // row r at col[r * LANES]
  uint32_t* myring = ring + sw * SPAN * LANES + l;                                   // slot s at myring[s * LANES]
  uint32_t A[4] = {0, 0, 0, 0}, B[4] = {0, 0, 0, 0};
  int n[4] = {0, 0, 0, 0};
  uint32_t* kbase = a.keys + ((size_t)im * AZ + 4 * l) * KCAP + seg * (KCAP / NSEG);
  uint32_t lcount = 0, lhash = 0;
  // centre c runs from x0 - SPAN (warm-up: sums fill, nothing is decided) to x1 - 1; step c loads row c + G + W - 1
  const int c_begin = x0 - SPAN, steps = (x1 - c_begin + U - 1) / U * U;
  auto load_row = [&](int r) -> uint32_t { return (r >= 0 && r < BINS) ? __builtin_nontemporal_load(col + (size_t)r * LANES) : 0u; };
  uint32_t nxt[U];
#pragma unroll
  for (int u = 0; u < U; u++) nxt[u] = load_row(c_begin + u + G + W - 1);
  int s_new = ((c_begin + G + W - 1) % SPAN + SPAN) % SPAN;   // ring slot of the row that enters with step c (= slot of row c - G - W - 1)
  for (int it = 0; it < steps; it += U) {
    uint32_t cur[U];
#pragma unroll
    for (int u = 0; u < U; u++) cur[u] = nxt[u];
#pragma unroll
    for (int u = 0; u < U; u++) nxt[u] = load_row(c_begin + it + U + u + G + W - 1);
    // every ring read of the block first (32 independent LDS reads in flight), then the block's writes: the rows read by the
    // steps of a block (<= c0 + U + G - 2) are older than the rows it writes (>= c0 + G + W - 1), and a slot is read before the
    // row that replaces it is written
    uint32_t r_outT[U], r_inT[U], r_outF[U], r_cen[U];
    {
      int s = s_new;
#pragma unroll
      for (int u = 0; u < U; u++) {
        int s_inT = s - (2 * G + W); if (s_inT < 0) s_inT += SPAN;
        int s_outF = s - W; if (s_outF < 0) s_outF += SPAN;
        int s_cen = s - (G + W - 1); if (s_cen < 0) s_cen += SPAN;
        r_outT[u] = myring[s * LANES]; r_inT[u] = myring[s_inT * LANES]; r_outF[u] = myring[s_outF * LANES]; r_cen[u] = myring[s_cen * LANES];
        s = s + 1 == SPAN ? 0 : s + 1;
      }
      s = s_new;
#pragma unroll
      for (int u = 0; u < U; u++) { myring[s * LANES] = cur[u]; s = s + 1 == SPAN ? 0 : s + 1; }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int c = c_begin + it + u;
      const uint32_t outT = r_outT[u], inT = r_inT[u], outF = r_outF[u], cen = r_cen[u];
      const uint32_t inF = cur[u];
      const bool decide = c >= x0 && c < x1;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t pin = __builtin_amdgcn_perm(inT, inF, 0x0c0c0400u + 0x00000101u * k);   // bytes {inF_k, inT_k, 0, 0}
        const uint32_t pout = __builtin_amdgcn_perm(outT, outF, 0x0c0c0400u + 0x00000101u * k);
        A[k] = __builtin_amdgcn_udot4(pin, pin, A[k], false);
        B[k] = __builtin_amdgcn_udot4(pout, pout, B[k], false);
        if (decide) {
          const uint32_t cv = (cen >> (8 * k)) & 0xffu;
          const uint32_t S = A[k] - B[k];
          const bool maybe = PRE ? S < __umul24(__umul24(cv, cv), kceil) + 1u : cv >= THR;
          if (maybe && 2u * S + 1u < lut[cv]) {
            if (n[k] < KCAP / NSEG) kbase[k * KCAP + n[k]] = (cv << 24) | (uint32_t)c;
            n[k]++;
            lcount++;
            lhash += (uint32_t)(im * 1315423911u) ^ ((uint32_t)(4 * l + k) * 2654435761u + (uint32_t)c * 40503u + cv);
          }
        }
      }
      s_new = s_new + 1 == SPAN ? 0 : s_new + 1;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) a.cnt[((size_t)im * AZ + 4 * l + k) * NSEG + seg] = n[k];
  if (lcount) { atomicAdd(det_count, lcount); atomicAdd(det_hash, lhash); }
}

int main() {
  const size_t bytes = (size_t)IMAGES * BINS * AZ;
  Args a;
  uint8_t* img; uint32_t *keys, *res; int32_t* cnt;
  hipMalloc(&img, bytes); hipMalloc(&keys, (size_t)IMAGES * AZ * KCAP * 4); hipMalloc(&cnt, (size_t)IMAGES * AZ * NSEG * 4); hipMalloc(&res, 64);
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, img, bytes);
  const double N = 2.0 * W, scaling = N * (std::pow(0.01, -1.0 / N) - 1.0), cc = 2.0 * W / scaling;      // cfar.cpp:12-16, filter.hip cfar_build_lut
  for (int i = 0; i < 256; i++) {
    a.lut[i] = 0;
    if (i < THR) continue;
    const double Bv = (double)(i * i) * cc, T = std::ceil(Bv - 1e-3);
    a.lut[i] = (uint32_t)T << 1;
  }
  a.img = img; a.keys = keys; a.cnt = cnt;
  hipMemset(res, 0, 64);
  const size_t total = (size_t)IMAGES * AZ * (BIN_HI - BIN_LO);
  hipLaunchKernelGGL(brute_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, a, res, res + 1);
  const size_t lds = 1024 + (size_t)SEGS_PER_WG * SPAN * LANES * 4;
  const uint32_t kceil = (uint32_t)std::ceil(cc);
  const int wgs = IMAGES * NSEG / SEGS_PER_WG;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int variant = 0; variant < 2; variant++) {
    auto fn = variant ? stream_kernel<true> : stream_kernel<false>;
    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipMemset(res + 2, 0, 16);
    hipLaunchKernelGGL(fn, dim3(wgs), dim3(THREADS), lds, 0, a, kceil, res + 2, res + 3);
    hipDeviceSynchronize();
    printf("launch status: %s\n", hipGetErrorString(hipGetLastError()));
    const int reps = 20;
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(fn, dim3(wgs), dim3(THREADS), lds, 0, a, kceil, res + 4, res + 5);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint32_t h[8];
    hipMemcpy(h, res, 32, hipMemcpyDeviceToHost);
    printf("brute force: %u detections (hash %08x); streaming: %u detections (hash %08x) -> %s\n", h[0], h[1], h[2], h[3],
           (h[0] == h[2] && h[1] == h[3]) ? "EQUAL" : "DIFFERENT");
    printf("streaming CA-CFAR (%s), %d images [%d bins][%d azimuths], %d workgroups x %d threads, LDS %zu B: %.4f ms per launch "
           "(%.2f TB/s of the %zu reachable bytes; detections per image %.0f)\n", variant ? "register pre-test" : "table for every bin above the static threshold",
           IMAGES, BINS, AZ, wgs, THREADS, lds, ms / reps, (double)bytes / (ms / reps * 1e-3) / 1e12, bytes, (double)h[2] / IMAGES);
  }
  return 0;
}
