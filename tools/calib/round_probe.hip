// Isolated cost of one trust-region round (lm_round of matcher.hip): ONE wavefront alone on a CU walks the accept path
// REPS times from the same LDS state; prints cycles per round.  Under load (four registrations per CU) the same round is
// measured at ~4.2 k cycles by the -DCFEAR_REG_TIMING build (tools/reg_quick.sh): the difference is what the other
// wavefronts' instructions cost it.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/calib/round_probe.hip -o /tmp/round_probe -L tbv_slam_public_amd -lcfear_hip -Wl,-rpath,$PWD/tbv_slam_public_amd && /tmp/round_probe
#include "../../tbv_slam_public_amd/csrc/matcher.hip"
namespace {
__global__ __launch_bounds__(64) void probe(const double* init, const double* cnd_in, long long* out, int reps) {
  __shared__ double st[S_COUNT];
  __shared__ double st0[S_COUNT];
  if (threadIdx.x < S_COUNT) { st0[threadIdx.x] = init[threadIdx.x]; st[threadIdx.x] = init[threadIdx.x]; }
  __syncthreads();
  double cnd[10];
  for (int k = 0; k < 10; k++) cnd[k] = readlane_f64(cnd_in[k], 0);
  long long total = 0;
  for (int r = 0; r < reps; r++) {
    if (threadIdx.x < S_COUNT) st[threadIdx.x] = st0[threadIdx.x];
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    lm_round(st, cnd, true, 20);
    __builtin_amdgcn_s_waitcnt(0);
    const long long t1 = __builtin_readcyclecounter();
    total += t1 - t0;
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = total; out[1] = ((int*)(st + S_INTS))[SI_DONE]; out[2] = ((int*)(st + S_INTS))[SI_ITER]; }
}
}
int main() {
  double init[S_COUNT] = {0}, cnd[10];
  // a plausible state: pose (2.5, -0.1, 0.013), cost 31.0, H ~ diag(200, 200, 5e5), radius 1e4, model change 0.5
  init[S_X] = 2.5; init[S_X + 1] = -0.1; init[S_X + 2] = 0.013; init[S_XCOST] = 31.0;
  const double g[3] = {-3.0, 1.5, -40.0}, H[6] = {210.0, 0.0, -800.0, 190.0, 1200.0, 6.0e5};
  for (int k = 0; k < 3; k++) init[S_CUR + k] = g[k];
  for (int k = 0; k < 6; k++) init[S_CUR + 3 + k] = H[k];
  init[S_SCALE] = 1.0 / (1.0 + 14.5); init[S_SCALE + 1] = 1.0 / (1.0 + 13.8); init[S_SCALE + 2] = 1.0 / (1.0 + 775.0);
  init[S_XNORM] = 2.502; init[S_GMAX] = 40.0; init[S_RADIUS] = 1e4; init[S_DEC] = 2.0; init[S_MINCOST] = 31.0; init[S_MODEL] = 0.5;
  init[S_CAND] = 2.51; init[S_CAND + 1] = -0.11; init[S_CAND + 2] = 0.0131; init[S_ITCOST] = 31.0; init[S_INIT] = 33.0;
  int* si = (int*)(init + S_INTS);
  si[SI_ITER] = 2; si[SI_USABLE] = 1; si[SI_ITSUCC] = 1; si[SI_PUSHED] = 2;
  cnd[0] = 30.6; cnd[1] = -2.0; cnd[2] = 1.0; cnd[3] = -25.0;
  for (int k = 0; k < 6; k++) cnd[4 + k] = H[k] * 1.01;
  double *d_init, *d_cnd; long long* d_out;
  hipMalloc(&d_init, sizeof(init)); hipMalloc(&d_cnd, sizeof(cnd)); hipMalloc(&d_out, 64);
  hipMemcpy(d_init, init, sizeof(init), hipMemcpyHostToDevice); hipMemcpy(d_cnd, cnd, sizeof(cnd), hipMemcpyHostToDevice);
  const int reps = 2000;
  for (int it = 0; it < 2; it++) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_init, d_cnd, d_out, reps);
  long long out[3];
  hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost);
  printf("lm_round alone: %.0f cycles per round (done %lld, iteration %lld)\n", (double)out[0] / reps, out[1], out[2]);
  return 0;
}
