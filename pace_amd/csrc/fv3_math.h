// fv3_math.h -- range-specific fp64 log / exp for the SIM1 Riemann solvers (fv3_nh.hip).
//
// Why: the solvers evaluate p = exp(gamma * log(rho R T)), dz ~ exp((kappa - 1) * log(p)), the layer-mean pressure
// dp / log(p2 / p1) and pk = exp(kappa * log(p)) on every level: 7 calls per level in riem_solver3, 5 in riem_solver_c.  The
// ocml fp64 log is ~90 VALU instructions per call (it carries the subnormal / special-value handling and a double-double
// reconstruction), exp ~35: 465 of the 690 VALU instructions per level of riem_solver3 (profiles/r03_final_c768_sq_a.md,
// instruction counts from `hipcc -S`).  The arguments here are pressures in Pa, ratios of pressures and their powers: positive,
// normal, finite.  fv3_log is the classic argument reduction to [sqrt(2)/2, sqrt(2)) + the degree-14 odd series in
// s = f / (2 + f) (Sun fdlibm / musl e_log.c coefficients, error < 1 ulp), ~40 instructions; anything outside the positive
// normal range is handled like libm does (special values, subnormals scaled up) in a branch the solvers never take.  fv3_exp: below.
// Same source on the device and in the host emulation: with -ffp-contract=off and the polynomials written with explicit fused multiply-adds
// (correctly rounded on both sides; -DFV3_MATH_NO_FMA: the separate multiply / add form, A/B) both give the same bits, so the host-emulation
// parity suite keeps checking the device arithmetic.  Accuracy is pinned against 80-bit logl over the solvers' argument range
// by tests/test_fast_math.py (<= 1 ulp).  -DFV3_LIBM_MATH selects the libm calls (A/B).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#include "fv3_common.h"

FV3_HD inline double fv3_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// x / y for finite, normal x, y with a normal quotient (what the solvers divide: pressures, air masses, heights, tridiagonal pivots): the
// compiler's own fp64 division sequence -- reciprocal estimate, two Newton steps, quotient, one correction -- without the operand scaling
// (`v_div_scale` x 2), the scaled final step (`v_div_fmas`) and the special-value fix-up (`v_div_fixup`) that exist for operands near the
// ends of the exponent range, infinities, zeros and NaNs.  With nothing scaled those are identities, so the result is the correctly
// rounded quotient, bit for bit what `/` gives (checked: the column form of the solver keeps `/` and is compared bitwise with the wave
// form on the device; the bench state checksums of a -DFV3_DIV_PLAIN build are identical).  8 instructions instead of 11 - 12.
FV3_HD inline double fv3_div(double x, double y) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FV3_DIV_PLAIN)
  const double r0 = __builtin_amdgcn_rcp(y);
  const double e0 = __builtin_fma(-y, r0, 1.0);
  const double r1 = __builtin_fma(r0, e0, r0);
  const double e1 = __builtin_fma(-y, r1, 1.0);
  const double r2 = __builtin_fma(r1, e1, r1);
  const double q = x * r2;
  const double e2 = __builtin_fma(-y, q, x);
  return __builtin_fma(e2, r2, q);
#else
  return x / y;
#endif
}
FV3_HD inline float fv3_div(float x, float y) { return x / y; }

FV3_HD inline double fv3_log_f64(double x) {
  uint64_t ix;
  memcpy(&ix, &x, sizeof(ix));
  uint32_t hx = (uint32_t)(ix >> 32);
  int k = 0;
  if (hx - 0x00100000u >= 0x7fe00000u) {  // zero, subnormal, negative, inf, nan (the solvers never get here)
    if (x != x || x < 0.0) return (x - x) / (x - x);  // nan
    if (x == 0.0) return -HUGE_VAL;
    if (hx >= 0x7ff00000u) return x;  // +inf
    x *= 0x1p54;                      // subnormal: scale up
    k = -54;
    memcpy(&ix, &x, sizeof(ix));
    hx = (uint32_t)(ix >> 32);
  }
  // x = 2^k * m, m in [sqrt(2)/2, sqrt(2))
  hx += 0x3ff00000u - 0x3fe6a09eu;
  k += (int)(hx >> 20) - 0x3ff;
  hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
  ix = ((uint64_t)hx << 32) | (ix & 0xffffffffu);
  double m;
  memcpy(&m, &ix, sizeof(m));
  const double f = m - 1.0;
  const double hfsq = 0.5 * f * f;
  const double s = fv3_div(f, 2.0 + f);
  const double z = s * s;
  const double w = z * z;
#if defined(FV3_MATH_NO_FMA)
  const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
  const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
  const double R = t2 + t1;
  const double dk = (double)k;
  return s * (hfsq + R) + dk * 1.90821492927058770002e-10 - hfsq + f + dk * 6.93147180369123816490e-01;
#else
  // (explicit fused multiply-adds: one correctly rounded operation on the device and in the host emulation alike, so the two still agree to
  //  the bit while the rest of the library is built with contraction off)
  const double t1 = w * fv3_fma(w, fv3_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fv3_fma(w, fv3_fma(w, fv3_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R = t2 + t1;
  const double dk = (double)k;
  return fv3_fma(dk, 6.93147180369123816490e-01, fv3_fma(s, hfsq + R, dk * 1.90821492927058770002e-10) - hfsq + f);
#endif
}

// exp(x) (the solvers stay within +-20): x = k ln2 + r, |r| <= ln2 / 2 (two-term ln2, the product k * ln2_hi is exact for
// |k| < 2^11), exp(r) by its Taylor polynomial of degree 13 in Horner form (remainder 4e-18 relative), scaled by 2^k with ldexp
// (which rounds into the subnormal range / overflows to inf by itself).  ~20 instructions (libm's: ~35).
FV3_HD inline double fv3_exp_f64(double x) {
  if (!(fabs(x) <= 745.0)) return x != x ? x : (x > 0.0 ? HUGE_VAL : 0.0);  // (ldexp below covers the gradual over / underflow up to there)
  const double kf = rint(x * 1.44269504088896338700e+00);
#if defined(FV3_MATH_NO_FMA)
  const double r = (x - kf * 6.93147180369123816490e-01) - kf * 1.90821492927058770002e-10;
  double p = 1.0 / 6227020800.0;
  p = p * r + 1.0 / 479001600.0;
  p = p * r + 1.0 / 39916800.0;
  p = p * r + 1.0 / 3628800.0;
  p = p * r + 1.0 / 362880.0;
  p = p * r + 1.0 / 40320.0;
  p = p * r + 1.0 / 5040.0;
  p = p * r + 1.0 / 720.0;
  p = p * r + 1.0 / 120.0;
  p = p * r + 1.0 / 24.0;
  p = p * r + 1.0 / 6.0;
  p = p * r + 0.5;
  // exp(r) = 1 + (r + r^2 p): the last addition carries the only rounding of order ulp(1)
  return ldexp(1.0 + (r + (r * r) * p), (int)kf);
#else
  const double r = fv3_fma(-kf, 1.90821492927058770002e-10, fv3_fma(-kf, 6.93147180369123816490e-01, x));
  double p = 1.0 / 6227020800.0;
  p = fv3_fma(p, r, 1.0 / 479001600.0);
  p = fv3_fma(p, r, 1.0 / 39916800.0);
  p = fv3_fma(p, r, 1.0 / 3628800.0);
  p = fv3_fma(p, r, 1.0 / 362880.0);
  p = fv3_fma(p, r, 1.0 / 40320.0);
  p = fv3_fma(p, r, 1.0 / 5040.0);
  p = fv3_fma(p, r, 1.0 / 720.0);
  p = fv3_fma(p, r, 1.0 / 120.0);
  p = fv3_fma(p, r, 1.0 / 24.0);
  p = fv3_fma(p, r, 1.0 / 6.0);
  p = fv3_fma(p, r, 0.5);
  // exp(r) = 1 + (r + r^2 p): the last addition carries the only rounding of order ulp(1)
  return ldexp(1.0 + fv3_fma(r * r, p, r), (int)kf);
#endif
}

#if defined(FV3_LIBM_MATH)
FV3_HD inline double fv3_log(double x) { return log(x); }
FV3_HD inline double fv3_exp(double x) { return exp(x); }
#else
FV3_HD inline double fv3_log(double x) { return fv3_log_f64(x); }
FV3_HD inline double fv3_exp(double x) { return fv3_exp_f64(x); }
#endif
FV3_HD inline float fv3_log(float x) { return logf(x); }
FV3_HD inline float fv3_exp(float x) { return expf(x); }
