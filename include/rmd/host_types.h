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

#endif  // RMD_HOST_TYPES_H
