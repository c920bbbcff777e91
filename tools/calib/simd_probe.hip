// simd_probe.hip -- on which SIMD does wavefront 0 of a 4-wavefront workgroup run, and which workgroups share a CU?
// matcher_kernel's serial piece (the trust-region round) runs on wavefront 0 of every workgroup; if the dispatcher starts every
// workgroup's wavefront 0 on the SAME SIMD of its CU, the four registrations that share a CU queue their rounds on one SIMD while
// three idle.  Launch like the matcher's regular form (256 threads, 40 KB of LDS, 4096 workgroups) and record HW_ID per wavefront.
// build + run:  hipcc -O3 --offload-arch=gfx950 tools/calib/simd_probe.hip -o /tmp/simd_probe && /tmp/simd_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void probe(uint32_t* out, int spin) {
  extern __shared__ uint8_t smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
  uint32_t xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11));
  long long t0 = __builtin_readcyclecounter();
  volatile uint8_t* s = smem;
  float acc = 0.f;
  for (int i = 0; i < spin; i++) acc += (float)s[(i * 67 + threadIdx.x) & 1023];          // keep the workgroup resident for a while
  if (lane == 0) { out[(blockIdx.x * 4 + wave) * 4] = hw; out[(blockIdx.x * 4 + wave) * 4 + 1] = xcc; out[(blockIdx.x * 4 + wave) * 4 + 2] = (uint32_t)(t0 >> 8); out[(blockIdx.x * 4 + wave) * 4 + 3] = acc == 1.5f; }
}
int main() {
  const int nb = 4096;
  uint32_t* d;
  hipMalloc(&d, nb * 16 * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
  hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 40 * 1024, 0, d, 20000);
  hipDeviceSynchronize();
  std::vector<uint32_t> h(nb * 16);
  hipMemcpy(h.data(), d, nb * 16 * 4, hipMemcpyDeviceToHost);
  int simd_hist[4][4] = {};                                   // [wave][simd]
  std::map<uint32_t, std::vector<int>> cu_blocks;             // (xcc, se, sh, cu) -> blocks in launch order
  for (int b = 0; b < nb; b++) {
    for (int w = 0; w < 4; w++) { const uint32_t hw = h[(b * 4 + w) * 4]; simd_hist[w][(hw >> 4) & 3]++; }
    const uint32_t hw = h[b * 16], xcc = h[b * 16 + 1];
    const uint32_t key = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 15);
    cu_blocks[key].push_back(b);
  }
  for (int w = 0; w < 4; w++) printf("wavefront %d runs on SIMD 0/1/2/3 in %d / %d / %d / %d workgroups\n", w, simd_hist[w][0], simd_hist[w][1], simd_hist[w][2], simd_hist[w][3]);
  printf("%zu distinct CUs seen; first CUs' workgroups in arrival order:\n", cu_blocks.size());
  int shown = 0;
  for (auto& kv : cu_blocks) {
    if (shown++ >= 4) break;
    printf("  cu %06x:", kv.first);
    for (size_t i = 0; i < kv.second.size() && i < 16; i++) printf(" %d(w0 simd %u)", kv.second[i], (h[kv.second[i] * 16] >> 4) & 3);
    printf("\n");
  }
  return 0;
}
