// FETCH_SIZE calibration for the access pattern of cacfar_cols_kernel (MI355X_MICROARCH.md: "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Three kernels over the SAME buffer of 512 images x 2336 x 400
// bytes (478 MB > the 256 MB Infinity Cache):
//   wide    every byte once, 16 B per lane, coalesced                                   (the guide: FETCH_SIZE reports 1/2)
//   pieces  every byte once as 16-byte pieces at the source-row stride (400 B), one workgroup per 16-column tile, the tiles of
//           an image on one XCD (blockIdx % 8) -- cacfar_cols_kernel's loads without its arithmetic
//   pieces1 the same pieces, but ONE tile per image only (columns 0..15): a line is touched once -- what "no sharing" costs
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/calib/fetch_calib.hip -o /tmp/fetch_calib &&
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/calib -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 2336, COLS = 400, IMAGES = 512, TILES = COLS / 16;
__global__ __launch_bounds__(256) void wide(const u32x4* __restrict__ p, size_t n16, uint32_t* out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const u32x4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(512) void pieces(const uint8_t* __restrict__ p, int tiles_used, uint32_t* out) {
  const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8, slots = gridDim.x / 8;
  const int nq = (IMAGES / 8) * tiles_used;
  uint32_t acc = 0;
  for (int q = slot; q < nq; q += slots) {
    const int im = q / tiles_used, tile = q - im * tiles_used, b = im * 8 + xcd;
    const uint8_t* src = p + (size_t)b * ROWS * COLS + tile * 16;
    for (int r = threadIdx.x; r < ROWS; r += 512) { const u32x4 v = *(const u32x4*)(src + (size_t)r * COLS); acc += v.x ^ v.y ^ v.z ^ v.w; }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
int main() {
  const size_t bytes = (size_t)IMAGES * ROWS * COLS;
  uint8_t* d; uint32_t* out;
  hipMalloc(&d, bytes + 64); hipMalloc(&out, 64);
  hipMemset(d, 1, bytes + 64);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(wide, dim3(4096), dim3(256), 0, 0, (const u32x4*)d, bytes / 16, out);
    hipLaunchKernelGGL(pieces, dim3(512), dim3(512), 0, 0, d, TILES, out);
    hipLaunchKernelGGL(pieces, dim3(512), dim3(512), 0, 0, d, 1, out);
  }
  hipDeviceSynchronize();
  printf("bytes per launch: wide %zu, pieces %zu, pieces1 touches %zu bytes in %zu lines\n", bytes, bytes, (size_t)IMAGES * ROWS * 16, (size_t)IMAGES * ROWS);
  return 0;
}
