/*
 * mrh_softmath.h — sinf / cosf / atan2f / asinf in plain fp32 operations, shared VERBATIM by the HIP kernels and the CPU
 * oracle.
 *
 * Why: the spherical camera model (camera.cuh:93-101 inverseProjection, :147-164 / :184-201 projectPoint[Approx]) calls
 * sinf / cosf / atan2f / asinf, and an integer pixel decision hangs on their results.  CUDA's, ROCm's and glibc's
 * implementations differ in the last ulps, so a CPU oracle and a device kernel can only agree bit for bit if both
 * evaluate ONE implementation: this one — float additions, multiplications, divisions, sqrtf and float<->int conversions
 * only, every expression parenthesised into a fixed order, compiled without FMA contraction on both sides
 * (oracle/Makefile, mrhash_amd/build.py).  Against the reference's CUDA intrinsics that is a documented deviation of
 * the same kind as rsqrtf -> 1 / sqrtf: the functions below are within ~2 ulp of the correctly rounded
 * results on the ranges the camera uses (|az| <= pi, |el| <= pi / 2), so a projected pixel can differ from a CUDA run
 * only where a coordinate lies within ~1e-6 of a rounding boundary.
 *
 * Polynomials and range reductions follow the single-precision Cephes routines (sinf.c, atanf.c, asinf.c; S. Moshier,
 * public domain algorithms), restated here; tests/test_softmath.py checks them against numpy on dense grids.
 */
#ifndef MRH_SOFTMATH_H
#define MRH_SOFTMATH_H

#if defined(__HIPCC__)
#define MRH_SM_FN __host__ __device__ static inline
#else
#include <math.h>
#define MRH_SM_FN static inline
#endif

#define MRH_SM_PI 3.14159265358979323846f
#define MRH_SM_PIO2 1.57079632679489661923f
#define MRH_SM_PIO4 0.78539816339744830962f

/* sin / cos of x for |x| <= 8192 (the camera uses |x| <= pi): octant reduction with a three-part pi / 4 (Cody-Waite),
 * degree-7 / degree-8 polynomials on [0, pi / 4]. */
MRH_SM_FN void mrh_sincosf(float x, float* s_out, float* c_out) {
  const float FOPI = 1.27323954473516f; /* 4 / pi */
  const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
  int sign_s = 1, sign_c = 1;
  float ax = x;
  if (x < 0.f) { sign_s = -1; ax = -x; }
  int j = (int) (FOPI * ax);
  float y = (float) j;
  if (j & 1) { j += 1; y += 1.0f; } /* map zeros to the origin */
  j &= 7;
  if (j > 3) { sign_s = -sign_s; sign_c = -sign_c; j -= 4; }
  if (j > 1) sign_c = -sign_c;
  const float r = ((ax - y * DP1) - y * DP2) - y * DP3;
  const float z = r * r;
  const float ps = ((((-1.9515295891e-4f * z) + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r) + r;
  const float pc = (((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z) + 1.0f;
  float s, c;
  if (j == 1 || j == 2) { s = pc; c = ps; }
  else { s = ps; c = pc; }
  *s_out = sign_s < 0 ? -s : s;
  *c_out = sign_c < 0 ? -c : c;
}

/* atan(x) for any finite x */
MRH_SM_FN float mrh_atanf(float xx) {
  float x = xx, y;
  int neg = 0;
  if (xx < 0.f) { neg = 1; x = -xx; }
  if (x > 2.414213562373095f) { /* tan(3 pi / 8) */
    y = MRH_SM_PIO2;
    x = -(1.0f / x);
  } else if (x > 0.4142135623730950f) { /* tan(pi / 8) */
    y = MRH_SM_PIO4;
    x = (x - 1.0f) / (x + 1.0f);
  } else {
    y = 0.f;
  }
  const float z = x * x;
  y = y + ((((((8.05374449538e-2f * z) - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x) + x);
  return neg ? -y : y;
}

/* atan2f(y, x), C convention (result in (-pi, pi]); both zero -> 0 */
MRH_SM_FN float mrh_atan2f(float y, float x) {
  if (x > 0.f) return mrh_atanf(y / x);
  if (x < 0.f) {
    const float a = mrh_atanf(y / x);
    return (y >= 0.f) ? (a + MRH_SM_PI) : (a - MRH_SM_PI);
  }
  if (y > 0.f) return MRH_SM_PIO2;
  if (y < 0.f) return -MRH_SM_PIO2;
  return 0.f;
}

/* asin(x) for |x| <= 1 (outside: the value at the clamped argument) */
MRH_SM_FN float mrh_asinf(float xx) {
  float a = xx, x, z;
  int neg = 0, flag = 0;
  if (xx < 0.f) { neg = 1; a = -xx; }
  if (a > 1.0f) a = 1.0f;
  if (a > 0.5f) {
    z = 0.5f * (1.0f - a);
    x = sqrtf(z);
    flag = 1;
  } else {
    x = a;
    z = x * x;
  }
  z = (((((((4.2163199048e-2f * z) + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * x) + x);
  if (flag) {
    z = z + z;
    z = MRH_SM_PIO2 - z;
  }
  return neg ? -z : z;
}

#endif /* MRH_SOFTMATH_H */
