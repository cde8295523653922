// Device-side building blocks of the depth-filter kernels: small-vector / camera / pose
// math and the software replacement for the texture unit.  Every expression is written in
// the operation order of the reference's headers so that results are bit-identical to the
// CPU oracle (see rmd_math.h for the arithmetic contract).
#ifndef RMD_DEVICE_HPP
#define RMD_DEVICE_HPP

#include <hip/hip_runtime.h>

#include "rmd_math.h"

namespace rmdk {

struct F2 { float x, y; };
struct F3 { float x, y, z; };
struct Cam { float fx, fy, cx, cy; };  // pinhole_camera.cuh:27-63
struct Pose { float d[12]; };           // se3.cuh: 3x4 row-major [R|t]

#define RMDK_D __device__ __forceinline__

RMDK_D float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // helper_math.h:1248-1251
RMDK_D float dot2(F2 a, F2 b) { return a.x * b.x + a.y * b.y; }              // helper_math.h:1244-1247
RMDK_D float norm3(F3 v) { return sqrtf(dot3(v, v)); }                       // helper_vector_types.cuh:23-28
RMDK_D float norm2(F2 v) { return sqrtf(dot2(v, v)); }
RMDK_D F3 normalize3(F3 v) {  // helper_math.h:1309-1313
  const float inv = rmd_rsqrtf(dot3(v, v));
  return F3{v.x * inv, v.y * inv, v.z * inv};
}
RMDK_D F2 normalize2(F2 v) {  // helper_math.h:1304-1308
  const float inv = rmd_rsqrtf(dot2(v, v));
  return F2{v.x * inv, v.y * inv};
}
RMDK_D F3 scale3(F3 v, float s) { return F3{v.x * s, v.y * s, v.z * s}; }

RMDK_D F3 cam2world(const Cam& c, float u, float v) { return F3{(u - c.cx) / c.fx, (v - c.cy) / c.fy, 1.0f}; }
RMDK_D F2 world2cam(const Cam& c, F3 p) { return F2{c.fx * p.x / p.z + c.cx, c.fy * p.y / p.z + c.cy}; }

RMDK_D F3 pose_rotate(const Pose& p, F3 v) {  // se3.cuh:109-115
  return F3{p.d[0] * v.x + p.d[1] * v.y + p.d[2] * v.z, p.d[4] * v.x + p.d[5] * v.y + p.d[6] * v.z,
            p.d[8] * v.x + p.d[9] * v.y + p.d[10] * v.z};
}
RMDK_D F3 pose_apply(const Pose& p, F3 v) {  // se3.cuh:164-168
  const F3 r = pose_rotate(p, v);
  return F3{r.x + p.d[3], r.y + p.d[7], r.z + p.d[11]};
}
RMDK_D F3 pose_translation(const Pose& p) { return F3{p.d[3], p.d[7], p.d[11]}; }

RMDK_D int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Wave64 reductions / scans on the VALU's data-parallel-primitive path (DPP: row shifts inside rows of 16 lanes, then the two row
// broadcasts of the gfx9 family), with ALL 64 lanes active.  __shfl_xor / __shfl_up go through the LDS crossbar (ds_bpermute: an LDS
// instruction and its latency per step); these are plain VALU instructions.  The inclusive scan leaves the total in lane 63.
enum : int { DPP_ROW_SHR = 0x110, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143 };
struct WaveAdd { static constexpr int identity = 0; RMDK_D static int op(int a, int b) { return a + b; } };
struct WaveMin { static constexpr int identity = 0x7fffffff; RMDK_D static int op(int a, int b) { return a < b ? a : b; } };
struct WaveMax { static constexpr int identity = -0x7fffffff - 1; RMDK_D static int op(int a, int b) { return a > b ? a : b; } };
template <typename OP>
RMDK_D int wave_scan_i32(int v) {  // lane i <- op over lanes 0..i
  v = OP::op(v, __builtin_amdgcn_update_dpp(OP::identity, v, DPP_ROW_SHR | 1, 0xf, 0xf, false));
  v = OP::op(v, __builtin_amdgcn_update_dpp(OP::identity, v, DPP_ROW_SHR | 2, 0xf, 0xf, false));
  v = OP::op(v, __builtin_amdgcn_update_dpp(OP::identity, v, DPP_ROW_SHR | 4, 0xf, 0xf, false));
  v = OP::op(v, __builtin_amdgcn_update_dpp(OP::identity, v, DPP_ROW_SHR | 8, 0xf, 0xf, false));
  v = OP::op(v, __builtin_amdgcn_update_dpp(OP::identity, v, DPP_ROW_BCAST15, 0xa, 0xf, false));
  v = OP::op(v, __builtin_amdgcn_update_dpp(OP::identity, v, DPP_ROW_BCAST31, 0xc, 0xf, false));
  return v;
}
template <typename OP>
RMDK_D int wave_reduce_i32(int v) {  // the same value in every lane (a scalar register)
  return __builtin_amdgcn_readlane(wave_scan_i32<OP>(v), 63);
}

// One fetch through a clamp-addressed, linear-filtered, unnormalised 2-D "texture"
// (texture_memory.cuh:45-66) from a pitched plane in global memory.  General form:
// any coordinate, clamping at the edges.  Used by the per-pixel matcher and as the
// fallback of the tile matcher when a search window does not fit in LDS.
RMDK_D float tex_linear_global(const float* __restrict__ plane, int w, int h, size_t stride, float x, float y) {
  if (!(isfinite(x) && isfinite(y))) return rmd_nanf();
  const float xb = x - 0.5f, yb = y - 0.5f;
  const float fi = floorf(xb), fj = floorf(yb);
  const float a = xb - fi, b = yb - fj;
  const float wmax = static_cast<float>(w - 1), hmax = static_cast<float>(h - 1);
  const int i0 = fi < 0.0f ? 0 : (fi > wmax ? w - 1 : static_cast<int>(fi));
  const int j0 = fj < 0.0f ? 0 : (fj > hmax ? h - 1 : static_cast<int>(fj));
  const float fi1 = fi + 1.0f, fj1 = fj + 1.0f;
  const int i1 = fi1 < 0.0f ? 0 : (fi1 > wmax ? w - 1 : static_cast<int>(fi1));
  const int j1 = fj1 < 0.0f ? 0 : (fj1 > hmax ? h - 1 : static_cast<int>(fj1));
  const float t00 = plane[j0 * stride + i0], t10 = plane[j0 * stride + i1];
  const float t01 = plane[j1 * stride + i0], t11 = plane[j1 * stride + i1];
  const float h0 = a == 0.0f ? t00 : rmd_lerp(a, t00, t10);
  const float h1 = a == 0.0f ? t01 : rmd_lerp(a, t01, t11);
  return b == 0.0f ? h0 : rmd_lerp(b, h0, h1);
}

// triangulation.cu:29-50
RMDK_D F3 triangulate(F3 f_ref, F3 f_cur, const Pose& T_ref_curr) {
  const F3 t = pose_translation(T_ref_curr);
  const F3 f2 = pose_rotate(T_ref_curr, f_cur);
  const float bx = dot3(t, f_ref), by = dot3(t, f2);
  const float A0 = dot3(f_ref, f_ref);
  const float A2 = dot3(f_ref, f2);
  const float A1 = -A2;
  const float A3 = dot3(F3{-f2.x, -f2.y, -f2.z}, f2);
  const float det = A0 * A3 - A1 * A2;
  const float l0 = (A3 * bx - A1 * by) / det;
  const float l1 = (-A2 * bx + A0 * by) / det;
  const F3 xm = F3{l0 * f_ref.x, l0 * f_ref.y, l0 * f_ref.z};
  const F3 xn = F3{t.x + l1 * f2.x, t.y + l1 * f2.y, t.z + l1 * f2.z};
  return F3{(xm.x + xn.x) / 2.0f, (xm.y + xn.y) / 2.0f, (xm.z + xn.z) / 2.0f};
}

// triangulation.cu:52-68
RMDK_D float triangulation_uncertainty(float z, F3 f_ref, F3 t, float one_pix_angle) {
  const F3 a = F3{f_ref.x * z - t.x, f_ref.y * z - t.y, f_ref.z * z - t.z};
  const float t_norm = norm3(t);
  const float a_norm = norm3(a);
  const float alpha = rmd_acosf(dot3(f_ref, t) / t_norm);
  const float beta = rmd_acosf((-dot3(a, t)) / (t_norm * a_norm));
  const float beta_plus = beta + one_pix_angle;
  const float gamma_plus = static_cast<float>(RMD_PI_D - static_cast<double>(alpha) - static_cast<double>(beta_plus));
  const float z_plus = t_norm * rmd_sinf(beta_plus) / rmd_sinf(gamma_plus);
  return z_plus - z;
}

// seed_update.cu:30-37
RMDK_D float normpdf(float x, float mu, float sigma_sq) {
  const float e = rmd_expf(-(x - mu) * (x - mu) / (2.0f * sigma_sq));
  const float two_pi_ss = static_cast<float>(static_cast<double>(2.0f) * RMD_PI_D * static_cast<double>(sigma_sq));
  return e * rmd_rsqrtf(two_pi_ss);
}

}  // namespace rmdk

#endif  // RMD_DEVICE_HPP
