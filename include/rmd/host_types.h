// Plain host-side value types that appear in the reference's public headers (CUDA's float2/float3/dim3).
// Defined here only when no GPU runtime header has already provided them, so the reference's host code
// (src/depthmap.cpp, src/publisher.cpp, the gtest sources) compiles with a plain C++ compiler.
#ifndef RMD_HOST_TYPES_H
#define RMD_HOST_TYPES_H

#if !defined(__HIP_PLATFORM_AMD__) && !defined(HIP_INCLUDE_HIP_HIP_VECTOR_TYPES_H) && !defined(__VECTOR_TYPES_H__)
struct float2 { float x, y; };
struct float3 { float x, y, z; };
inline float2 make_float2(float x, float y) { float2 v; v.x = x; v.y = y; return v; }
inline float3 make_float3(float x, float y, float z) { float3 v; v.x = x; v.y = y; v.z = z; return v; }
struct dim3 {
  unsigned int x, y, z;
  dim3(unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1) : x(x_), y(y_), z(z_) {}
};
#endif

// The float3 arithmetic the reference's HOST sources use (src/publisher.cpp:73-74 through cuda_toolkit/helper_math.h:
// operator* :814-821, dot :1248-1251, length :1291-1294, normalize :1309-1313 with the off-device rsqrtf of :62-65,
// i.e. v * (1.0f / sqrtf(dot(v, v)))).  One fp32 rounding per operation, in that order.
#if !defined(__HIP_PLATFORM_AMD__) && !defined(RMD_HOST_FLOAT3_MATH)
#define RMD_HOST_FLOAT3_MATH
#include <cmath>
inline float3 operator+(float3 a, float3 b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline float3 operator-(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline float3 operator*(float3 a, float s) { return make_float3(a.x * s, a.y * s, a.z * s); }
inline float3 operator*(float s, float3 a) { return make_float3(s * a.x, s * a.y, s * a.z); }
inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(float3 v) { return sqrtf(dot(v, v)); }
inline float3 normalize(float3 v) {
  const float inv_len = 1.0f / sqrtf(dot(v, v));
  return v * inv_len;
}
#endif

#endif  // RMD_HOST_TYPES_H
