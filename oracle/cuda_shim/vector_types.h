// TEST INFRASTRUCTURE ONLY -- CPU stand-in for CUDA's built-in vector types.
// See cuda_runtime.h in this directory for what this shim is and is not.
#ifndef RMD_ORACLE_CUDA_SHIM_VECTOR_TYPES_H
#define RMD_ORACLE_CUDA_SHIM_VECTOR_TYPES_H

#define RMD_SHIM_VEC2(T, N) struct N##2 { T x, y; }; \
  static inline N##2 make_##N##2(T x, T y) { N##2 v; v.x = x; v.y = y; return v; }
#define RMD_SHIM_VEC3(T, N) struct N##3 { T x, y, z; }; \
  static inline N##3 make_##N##3(T x, T y, T z) { N##3 v; v.x = x; v.y = y; v.z = z; return v; }
#define RMD_SHIM_VEC4(T, N) struct N##4 { T x, y, z, w; }; \
  static inline N##4 make_##N##4(T x, T y, T z, T w) { N##4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }

RMD_SHIM_VEC2(float, float)
RMD_SHIM_VEC3(float, float)
RMD_SHIM_VEC4(float, float)
RMD_SHIM_VEC2(int, int)
RMD_SHIM_VEC3(int, int)
RMD_SHIM_VEC4(int, int)
RMD_SHIM_VEC2(unsigned int, uint)
RMD_SHIM_VEC3(unsigned int, uint)
RMD_SHIM_VEC4(unsigned int, uint)

struct dim3 {
  unsigned int x, y, z;
  dim3(unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1) : x(x_), y(y_), z(z_) {}
};

// Integer min/max are compiler built-ins under nvcc.
static inline int max(int a, int b) { return a > b ? a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline unsigned int max(unsigned int a, unsigned int b) { return a > b ? a : b; }
static inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }

#endif  // RMD_ORACLE_CUDA_SHIM_VECTOR_TYPES_H
