// Double-precision helpers shared by the surface-point and registration kernels (device code only).
#pragma once
#include <hip/hip_runtime.h>

namespace {

// A 64-bit literal pinned to a scalar register pair AT ITS USE.  gfx950's VOP3 encodings take no 64-bit literals, so the
// compiler materialises every fp64 constant in registers and hoists it out of the enclosing loops -- the polynomial
// coefficients below, inlined at a few sites of a long-running kernel, then hold ~30 VGPRs for the kernel's whole life
// (the matcher: the difference between two and three wavefronts per SIMD in round 4).  The empty volatile asm keeps the two
// s_mov_b32 where they are written; the scalar unit is idle there anyway, and v_fma_f64 takes the pair as an operand.
template <bool PIN>
__device__ __forceinline__ double kc(double c) {
  if (PIN) asm volatile("" : "+s"(c));
  return c;
}

// sin / cos of |a| <= 0.5 without libm's range reduction: Taylor series to x^17 / x^16 (truncation < 1e-22), Horner in
// fp64 -- within an ulp or two of libm, the level at which device and host libm differ anyway.
template <bool PIN = false>
__device__ __forceinline__ void sincos_small(const double a, double* s, double* c) {
  const double z = a * a;
  double ps = kc<PIN>(2.8114572543455206e-15);                 // 1/17!
  ps = fma(ps, z, kc<PIN>(-7.6471637318198164e-13));           // -1/15!
  ps = fma(ps, z, kc<PIN>(1.6059043836821613e-10));            // 1/13!
  ps = fma(ps, z, kc<PIN>(-2.5052108385441720e-08));           // -1/11!
  ps = fma(ps, z, kc<PIN>(2.7557319223985893e-06));            // 1/9!
  ps = fma(ps, z, kc<PIN>(-1.9841269841269841e-04));           // -1/7!
  ps = fma(ps, z, kc<PIN>(8.3333333333333332e-03));            // 1/5!
  ps = fma(ps, z, kc<PIN>(-1.6666666666666666e-01));           // -1/3!
  *s = fma(a * z, ps, a);
  double pc = kc<PIN>(4.7794773323873853e-14);                 // 1/16!
  pc = fma(pc, z, kc<PIN>(-1.1470745597729725e-11));           // -1/14!
  pc = fma(pc, z, kc<PIN>(2.0876756987868100e-09));            // 1/12!
  pc = fma(pc, z, kc<PIN>(-2.7557319223985888e-07));           // -1/10!
  pc = fma(pc, z, kc<PIN>(2.4801587301587302e-05));            // 1/8!
  pc = fma(pc, z, kc<PIN>(-1.3888888888888889e-03));           // -1/6!
  pc = fma(pc, z, kc<PIN>(4.1666666666666664e-02));            // 1/4!
  *c = fma(z * z, pc, fma(z, -0.5, 1.0));
}

// sin / cos for |a| <= 1e5: beyond the polynomial's range, Cody-Waite reduction by pi/2 in three parts (k < 2^17, so
// k * the 33-bit head is exact) and the same polynomials on |r| <= pi/4 (truncation < 2e-18).
template <bool PIN = false>
__device__ __forceinline__ void sincos_reduced(const double a, double* s, double* c) {
  if (fabs(a) <= 0.5) { sincos_small<PIN>(a, s, c); return; }
  const double k = rint(a * kc<PIN>(6.36619772367581382433e-01));               // 2 / pi
  double r = fma(-k, kc<PIN>(1.57079632673412561417e+00), a);                   // pi/2: first 33 bits
  r = fma(-k, kc<PIN>(6.07710050630396597660e-11), r);                          //       next 33 bits
  r = fma(-k, kc<PIN>(2.02226624879595063154e-21), r);                          //       the rest
  double sr, cr;
  sincos_small<PIN>(r, &sr, &cr);
  const int q = (int)k & 3;
  const double ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}

// 1 / d for a normal, finite d: v_rcp_f64 (about 23 good bits) and two Newton steps -- within an ulp or two of the IEEE
// quotient in 5 dependent instructions instead of the 11 of the division sequence.  For the trust-region bookkeeping
// of the matcher, whose decisions already see rounding-level noise from the order of the block sums.
__device__ __forceinline__ double rcp_newton(const double d) {
  double x = __builtin_amdgcn_rcp(d);
  double e = fma(-d, x, 1.0);
  x = fma(x, e, x);
  e = fma(-d, x, 1.0);
  return fma(x, e, x);
}

// sqrt(a) for a normal, finite a >= 0 (0 -> 0): v_rsq_f64 and two coupled Newton steps.
__device__ __forceinline__ double sqrt_newton(const double a) {
  if (!(a > 0.0)) return 0.0;
  const double y = __builtin_amdgcn_rsq(a);
  const double h = 0.5 * y;
  double s = a * y;
  s = fma(fma(-s, s, a), h, s);
  return fma(fma(-s, s, a), h, s);
}

// 1 / sqrt(a) for a normal, finite a > 0: the device library's own refinement of v_rsq_f64 (one coupled step,
// y0 + y0 e (1/2 + 3/8 e), e = 1 - a y0^2 -- the same bits) without its class test for zero / infinity / NaN.
__device__ __forceinline__ double rsqrt_newton(const double a) {
  const double y0 = __builtin_amdgcn_rsq(a);
  const double e = fma(y0 * -a, y0, 1.0);
  return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

}  // namespace
