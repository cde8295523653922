// Shared host/device arithmetic contract of the depth-filter path.
//
// The HIP kernels (rmd_kernels.hip) and the CPU oracle (oracle/remode_oracle.cpp)
// compile THIS header, so every operation whose result is not pinned by IEEE-754
// is evaluated by identical source on both sides and the two agree bit for bit.
// Both sides build with -ffp-contract=off; fp32 / and sqrt are correctly rounded
// (hipcc: -fhip-fp32-correctly-rounded-divide-sqrt, the default).
//
// What is defined here and why:
//   * rmd_rsqrtf      The reference calls rsqrtf (helper_math.h:1304-1313 normalize,
//                     epipolar_match.cu:123, seed_update.cu:36).  Its CUDA build gets a
//                     2-ulp hardware approximation; the value is pinned here as the
//                     IEEE expression 1/sqrt(x) (what helper_math.h:62-65 itself falls
//                     back to off-device).
//   * rmd_lerp        One linear-filter tap pair of the texture unit the reference
//                     reads images through (texture_memory.cuh:45-66; §5.1 of SURVEY.md):
//                     fma(t, b-a, a), exact texel at t == 0.
//   * rmd_expf / rmd_sinf / rmd_acosf
//                     seed_update.cu:36 (expf), triangulation.cu:63-66 (acosf, sinf).
//                     libm (host) and ocml (device) differ in the last ulp, so they are
//                     restated here: evaluated in fp64 (range reduction + Taylor series,
//                     truncation < 1e-17) and rounded once to fp32, i.e. correctly
//                     rounded except when the fp64 value lies within ~1e-16 of a
//                     rounding boundary.  glibc's own fp32 functions are within 1 ulp of
//                     these; oracle/ measures that difference (tests/test_oracle_pin.py).
//   * fminf/fmaxf     as the ternaries of helper_math.h:42-50 (NaN in the first
//                     argument yields the second), spelled out at the call sites.
#ifndef RMD_MATH_H
#define RMD_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define RMD_HD __host__ __device__ __forceinline__
#else
#define RMD_HD static inline
#endif

#define RMD_PI_D 3.141592653589793  // (double) M_PI, used in fp64 by seed_update.cu:36 and triangulation.cu:65

RMD_HD float rmd_rsqrtf(float x) { return 1.0f / sqrtf(x); }

RMD_HD float rmd_lerp(float t, float a, float b) { return fmaf(t, b - a, a); }

RMD_HD float rmd_nanf() {
  const uint32_t bits = 0x7fc00000u;
  float f;
  memcpy(&f, &bits, sizeof(f));
  return f;
}

RMD_HD double rmd_pow2_d(int k) {  // 2^k for k in [-1022, 1023]
  const uint64_t bits = static_cast<uint64_t>(k + 1023) << 52;
  double d;
  memcpy(&d, &bits, sizeof(d));
  return d;
}

// e^x, x fp32 -> fp32
RMD_HD float rmd_expf(float xf) {
  if (xf != xf) return xf;
  if (xf > 89.0f) return INFINITY;
  if (xf < -104.0f) return 0.0f;
  const double x = static_cast<double>(xf);
  const double kd = floor(x * 1.4426950408889634 + 0.5);
  // ln2 split so that kd * hi is exact (hi has 33 significant bits, |kd| <= 151)
  const double r = (x - kd * 0.6931471804855391) - kd * 7.440617110012397e-11;
  double p = 1.6059043836821613e-10;  // 1/13!
  p = p * r + 2.08767569878681e-09;
  p = p * r + 2.505210838544172e-08;
  p = p * r + 2.755731922398589e-07;
  p = p * r + 2.7557319223985893e-06;
  p = p * r + 2.48015873015873e-05;
  p = p * r + 0.0001984126984126984;
  p = p * r + 0.001388888888888889;
  p = p * r + 0.008333333333333333;
  p = p * r + 0.041666666666666664;
  p = p * r + 0.16666666666666666;
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  return static_cast<float>(p * rmd_pow2_d(static_cast<int>(kd)));
}

// sin(x), x fp32 -> fp32.  Cody-Waite reduction by pi/2 in fp64 (exact product for
// |x| < ~1e6, graceful beyond), Taylor kernels on |r| <= pi/4.
RMD_HD float rmd_sinf(float xf) {
  if (!(fabsf(xf) <= 3.4028234663852886e38f)) return xf - xf;  // Inf, NaN -> NaN
  const double x = static_cast<double>(xf);
  const double kd = floor(x * 0.6366197723675814 + 0.5);
  const double r = (x - kd * 1.5707963267341256) - kd * 6.077100506506192e-11;
  const double q = kd - 4.0 * floor(kd * 0.25);  // quadrant 0..3
  const double r2 = r * r;
  double v;
  if (q == 0.0 || q == 2.0) {
    double s = 2.8114572543455206e-15;  // 1/17!
    s = s * r2 - 7.647163731819816e-13;
    s = s * r2 + 1.6059043836821613e-10;
    s = s * r2 - 2.505210838544172e-08;
    s = s * r2 + 2.7557319223985893e-06;
    s = s * r2 - 0.0001984126984126984;
    s = s * r2 + 0.008333333333333333;
    s = s * r2 - 0.16666666666666666;
    v = r + r * (r2 * s);
  } else {
    double c = -1.5619206968586225e-16;  // -1/18!
    c = c * r2 + 4.779477332387385e-14;
    c = c * r2 - 1.1470745597729725e-11;
    c = c * r2 + 2.08767569878681e-09;
    c = c * r2 - 2.755731922398589e-07;
    c = c * r2 + 2.48015873015873e-05;
    c = c * r2 - 0.001388888888888889;
    c = c * r2 + 0.041666666666666664;
    c = c * r2 - 0.5;
    v = 1.0 + r2 * c;
  }
  if (q >= 2.0) v = -v;
  return static_cast<float>(v);
}

// asin(z)/z as a series in s = z*z, 0 <= s <= 0.25 (truncation 2e-18)
RMD_HD double rmd_asin_over_z(double s) {
  double p = 0.0022014739737101384;
  p = p * s + 0.002338091892111975;
  p = p * s + 0.0024894486782468836;
  p = p * s + 0.00265787063820729;
  p = p * s + 0.002846178401108942;
  p = p * s + 0.0030578216492580306;
  p = p * s + 0.003297059503473485;
  p = p * s + 0.0035692053938259347;
  p = p * s + 0.003880964558837669;
  p = p * s + 0.004240907093679363;
  p = p * s + 0.004660143486915096;
  p = p * s + 0.005153309682319905;
  p = p * s + 0.005740037670841924;
  p = p * s + 0.006447210311889649;
  p = p * s + 0.0073125258735988454;
  p = p * s + 0.008390335809616815;
  p = p * s + 0.009761609529194078;
  p = p * s + 0.011551800896139705;
  p = p * s + 0.01396484375;
  p = p * s + 0.017352764423076924;
  p = p * s + 0.022372159090909092;
  p = p * s + 0.030381944444444444;
  p = p * s + 0.044642857142857144;
  p = p * s + 0.075;
  p = p * s + 0.16666666666666666;
  p = p * s + 1.0;
  return p;
}

// acos(x), x fp32 -> fp32
RMD_HD float rmd_acosf(float xf) {
  if (xf != xf) return xf;
  const double x = static_cast<double>(xf);
  const double ax = fabs(x);
  if (ax > 1.0) return rmd_nanf();
  if (ax <= 0.5) {
    const double as = x * rmd_asin_over_z(x * x);
    return static_cast<float>(1.5707963267948966 - as);
  }
  const double s = (1.0 - ax) * 0.5;  // exact
  const double z = sqrt(s);
  const double t = 2.0 * (z * rmd_asin_over_z(s));
  return static_cast<float>(x > 0.0 ? t : RMD_PI_D - t);
}

#endif  // RMD_MATH_H
