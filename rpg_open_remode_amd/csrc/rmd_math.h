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
//                     libm (host) and ocml (device) differ in the last ulp, so the functions
//                     are restated here -- as the very algorithms of the C library the
//                     reference's own kernels link against when they are compiled for the
//                     host (glibc 2.35): bit-identical to it for every argument (see below),
//                     so that the device, the restated oracle and the UNMODIFIED reference
//                     built against the system libm all agree bit for bit.
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

// ------------------------------------------------------------------------------------------------------------------------
// expf / sinf / acosf: the functions of GNU libc 2.35 (x86-64), restated operation for operation.
//
// The reference calls expf (seed_update.cu:36), sinf and acosf (triangulation.cu:63-66).  Its CUDA build gets NVIDIA's fast-math
// intrinsics, which nothing but that build can reproduce; the only executable form of the reference -- its own kernels compiled for the
// host (the test suite's "Oracle A") -- gets the C library's.  So the contract is pinned to that library: the three functions below return,
// for EVERY float argument, the bits glibc 2.35 returns on an x86-64 host with FMA (the variants its ifunc resolvers select on every
// current CPU: __expf_fma, __sinf_fma; acosf has a single variant).  oracle/libm_exhaustive.cpp checks all 2^32 arguments of each against
// the host's libm (tests/test_math_contract.py runs a sample of it every time). With that, the reference built unmodified against the
// system's libm and the HIP kernels agree bit for bit.
//
// The algorithms are the published ones: expf and sinf are Szabolcs Nagy's / Wilco Dijkstra's routines from ARM's
// optimized-routines as adopted by glibc 2.27 / 2.28 (sysdeps/ieee754/flt-32/e_expf.c, s_sinf.c, sincosf.h, sincosf_data.c,
// e_exp2f_data.c; evaluated in double, table-driven reduction); acosf is Sun's fdlibm routine in Ian Lance Taylor's float port
// (sysdeps/ieee754/flt-32/e_acosf.c; evaluated in float).  Where the FMA build of glibc contracts a multiply-add, fma() is
// written out here, because the kernels and the oracle are compiled without contraction.

RMD_HD uint32_t rmd_f2u(float f) { uint32_t u; memcpy(&u, &f, sizeof(u)); return u; }
RMD_HD float rmd_u2f(uint32_t u) { float f; memcpy(&f, &u, sizeof(f)); return f; }
RMD_HD uint64_t rmd_d2u(double d) { uint64_t u; memcpy(&u, &d, sizeof(u)); return u; }
RMD_HD double rmd_u2d(uint64_t u) { double d; memcpy(&d, &u, sizeof(d)); return d; }

// e_exp2f_data.c: T[i] = bits(2^(i/32)) - (i << 52) / 32
static const uint64_t RMD_EXP2F_T[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
        0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull,
        0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull,
        0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull,
        0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull,
        0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

// T[i]: from the table on the host; on the device from immediates through a binary tree of selects on the five index bits -- a
// per-lane table read would be a memory round trip in the middle of the setup kernel's dependent chain
RMD_HD uint64_t rmd_exp2f_t(uint32_t i) {
#if defined(__HIP_DEVICE_COMPILE__)
#define RMD_T2(a) ((i & 1u) ? RMD_EXP2F_T[(a) + 1] : RMD_EXP2F_T[(a)])
#define RMD_T4(a) ((i & 2u) ? RMD_T2((a) + 2) : RMD_T2(a))
#define RMD_T8(a) ((i & 4u) ? RMD_T4((a) + 4) : RMD_T4(a))
#define RMD_T16(a) ((i & 8u) ? RMD_T8((a) + 8) : RMD_T8(a))
  return (i & 16u) ? RMD_T16(16) : RMD_T16(0);  // constant indices: the compiler folds the table reads into literals
#undef RMD_T2
#undef RMD_T4
#undef RMD_T8
#undef RMD_T16
#else
  return RMD_EXP2F_T[i];
#endif
}

// e^x, glibc 2.35 __expf (FMA variant)
RMD_HD float rmd_expf(float x) {
  const uint32_t ix = rmd_f2u(x);
  const uint32_t abstop = (ix >> 20) & 0x7ffu;
  const double xd = static_cast<double>(x);
  if (abstop >= 0x42bu) {  // |x| >= 88 or NaN
    if (ix == 0xff800000u) return 0.0f;                                  // -Inf
    if (abstop >= 0x7f8u) return x + x;                                  // +Inf, NaN
    if (x > 88.7228317f) return INFINITY;                              // x > log(0x1p128): overflow  [0x1.62e42ep6f]
    if (x < -103.972076f) return 0.0f;                                 // x < log(0x1p-150): underflow  [-0x1.9fe368p6f]
    // x < log(0x1p-149): may underflow (rounds to the smallest denormal)  [-0x1.9d1d9ep6f 0x1.4p-75f 0x1.4p-75f]
    if (x < -103.278923f) return 3.30872245e-23f * 3.30872245e-23f;
  }
  const double InvLn2N = 46.166241308446828, SHIFT = 6755399441055744.0;  // 0x1.71547652b82fep+5 0x1.8p+52
  // 0x1.c6af84b912394p-20 0x1.ebfce50fac4f3p-13 0x1.62e42ff0c52d6p-6
  const double C0 = 1.6938359250920212e-06, C1 = 0.00023459809789509004, C2 = 0.021660849396613134;
  // x * N / ln2 = k + r with r in [-1/2, 1/2] and integer k
  double kd = fma(InvLn2N, xd, SHIFT);
  const uint64_t ki = rmd_d2u(kd);
  kd -= SHIFT;
  const double r = fma(InvLn2N, xd, -kd);
  // exp(x) = 2^(k/N) * 2^(r/N) ~= s * (C0 r^3 + C1 r^2 + C2 r + 1)
  const uint64_t t = rmd_exp2f_t(static_cast<uint32_t>(ki) & 31u) + (ki << 47);
  const double s = rmd_u2d(t);
  const double z = fma(C0, r, C1);
  const double r2 = r * r;
  double y = fma(r, C2, 1.0);
  y = fma(z, r2, y);
  y = y * s;
  return static_cast<float>(y);
}

// sincosf_data.c, __sincosf_table[0]: 2/pi * 2^24, pi/2, the cosine and the sine coefficients.  (__sincosf_table[1], used for
// quadrants 2 and 3, is the same with every cosine coefficient negated, and sign[] = {1, -1, -1, 1}: both are applied arithmetically
// below -- multiplying by -1 and negating are exact, so the bits are the table's.)
#define RMD_SC_HPI_INV 10680707.430881744  // 0x1.45F306DC9C883p+23
#define RMD_SC_HPI 1.5707963267948966  // 0x1.921FB54442D18p0
#define RMD_SC_C0 1.0  // 0x1p0
#define RMD_SC_C1 (-0.49999999725108224)  // -0x1.ffffffd0c621cp-2
#define RMD_SC_C2 0.041666623324344516  // 0x1.55553e1068f19p-5
#define RMD_SC_C3 (-0.0013886763794376041)  // -0x1.6c087e89a359dp-10
#define RMD_SC_C4 2.4390450703564542e-05  // 0x1.99343027bf8c3p-16
#define RMD_SC_S1 (-0.16666654943701084)  // -0x1.555545995a603p-3
#define RMD_SC_S2 0.0083321781461388536  // 0x1.1107605230bc4p-7
#define RMD_SC_S3 (-0.00019517298981385725)  // -0x1.994eb3774cf24p-13
// 4/pi as a 768-bit fraction in overlapping 32-bit words (__inv_pio4)
static const uint32_t RMD_INV_PIO4[24] = {0xa2u,       0xa2f9u,     0xa2f983u,   0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u,
    0x4e441529u,
                                          0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu,
                                              0xf534ddc0u,
                                          0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u,
                                              0x3c439041u};

// sincosf.h sinf_poly on quadrant q = n + (sign of a large argument): sine polynomial for even n on x * sign[q & 3], cosine
// polynomial (of table q & 2) for odd n; rounded to float
RMD_HD float rmd_sinf_poly(double x, double x2, int n, int q) {
  if ((n & 1) == 0) {
    const double xs = (((q & 3) == 1) || ((q & 3) == 2)) ? -x : x;  // x * sign[q & 3]
    const double x3 = xs * x2;
    const double s1 = fma(RMD_SC_S3, x2, RMD_SC_S2);
    const double x7 = x3 * x2;
    const double s = fma(x3, RMD_SC_S1, xs);
    return static_cast<float>(fma(s1, x7, s));
  }
  const double x4 = x2 * x2;
  const double c2 = fma(RMD_SC_C4, x2, RMD_SC_C3);
  const double c1 = fma(RMD_SC_C1, x2, RMD_SC_C0);
  const double x6 = x4 * x2;
  const double c = fma(x4, RMD_SC_C2, c1);
  const float r = static_cast<float>(fma(c2, x6, c));
  return (q & 2) ? -r : r;  // table 1 = table 0 with the cosine coefficients negated: the negated result, bit for bit
}

// sin(x), glibc 2.35 __sinf (FMA variant)
RMD_HD float rmd_sinf(float y) {
  const uint32_t iy = rmd_f2u(y);
  const uint32_t abstop = (iy >> 20) & 0x7ffu;
  const double x = static_cast<double>(y);
  if (abstop < 0x3f4u) {  // |y| < pi/4
    if (abstop < 0x398u) return y;  // |y| < 2^-12
    return rmd_sinf_poly(x, x * x, 0, 0);
  }
  if (abstop < 0x42fu) {  // |y| < 120: reduce_fast
    const double r = x * RMD_SC_HPI_INV;
    const int n = (static_cast<int32_t>(r) + 0x800000) >> 24;
    const double xr = fma(-static_cast<double>(n), RMD_SC_HPI, x);
    return rmd_sinf_poly(xr, xr * xr, n, n);
  }
  if (abstop < 0x7f8u) {  // finite: reduce_large, 4/pi to 96 bits around the argument's exponent
    const int sign = static_cast<int>(iy >> 31);
    const uint32_t* arr = &RMD_INV_PIO4[(iy >> 26) & 15u];
    const int shift = static_cast<int>((iy >> 23) & 7u);
    uint32_t xi = (iy & 0xffffffu) | 0x800000u;
    xi <<= shift;
    uint64_t res0 = static_cast<uint64_t>(static_cast<uint32_t>(xi * arr[0]));
    const uint64_t res1 = static_cast<uint64_t>(xi) * arr[4];
    const uint64_t res2 = static_cast<uint64_t>(xi) * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t nn = (res0 + (1ull << 61)) >> 62;
    res0 -= nn << 62;
    const double xr = static_cast<double>(static_cast<int64_t>(res0)) * 3.4061215800865545e-19;  // 0x1.921FB54442D18p-62
    const int n = static_cast<int>(nn);
    return rmd_sinf_poly(xr, xr * xr, n, n + sign);
  }
  return (y - y) / (y - y);  // Inf, NaN -> NaN
}

// acos(x), glibc 2.35 __ieee754_acosf behind its wrapper (fdlibm, float arithmetic)
RMD_HD float rmd_acosf(float x) {
  const float one = 1.0f, pi = rmd_u2f(0x40490fdau), pio2_hi = rmd_u2f(0x3fc90fdau), pio2_lo = rmd_u2f(0x33a22168u);
  const float pS0 = rmd_u2f(0x3e2aaaabu), pS1 = -rmd_u2f(0x3ea6b090u), pS2 = rmd_u2f(0x3e4e0aa8u), pS3 = -rmd_u2f(0x3d241146u),
              pS4 = rmd_u2f(0x3a4f7f04u), pS5 = rmd_u2f(0x3811ef08u);
  const float qS1 = -rmd_u2f(0x4019d139u), qS2 = rmd_u2f(0x4001572du), qS3 = -rmd_u2f(0x3f303361u), qS4 = rmd_u2f(0x3d9dc62eu);
  const uint32_t hx = rmd_f2u(x), ix = hx & 0x7fffffffu;
  if (ix == 0x3f800000u) return (hx >> 31) ? pi + rmd_u2f(0x34222168u) : 0.0f;  // |x| == 1: 0 or pi + 2 pio2_lo
  if (ix > 0x3f800000u) return (x - x) / (x - x);                                // |x| > 1 or NaN
  if (ix < 0x3f000000u) {                                                        // |x| < 0.5
    if (ix <= 0x32800000u) return pio2_lo + pio2_hi;                             // |x| <= 2^-26
    const float z = x * x;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    return pio2_hi - (x - (pio2_lo - x * r));
  }
  if (hx >> 31) {                                                                // x <= -0.5
    const float z = (one + x) * 0.5f;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float s = sqrtf(z);
    const float r = p / q;
    const float w = r * s - pio2_lo;
    const float sw = w + s;
    return pi - (sw + sw);
  }
  const float z = (one - x) * 0.5f;                                              // x >= 0.5
  const float s = sqrtf(z);
  const float df = rmd_u2f(rmd_f2u(s) & 0xfffff000u);
  const float c = (z - df * df) / (s + df);
  const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  const float r = p / q;
  const float w = r * s + c;
  const float dw = w + df;
  return dw + dw;
}

#endif  // RMD_MATH_H
