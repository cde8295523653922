// TEST INFRASTRUCTURE ONLY -- never shipped, never linked into the product.
//
// A minimal CPU stand-in for the slice of the CUDA runtime API that the
// reference's device code (/root/reference/src/*.cu, include/rmd/*.cuh) touches, so
// that those files compile UNMODIFIED with plain g++ and run one "CUDA thread"
// at a time on the host.  It exists to turn the reference's own kernels into an
// executable oracle ("Oracle A", see oracle/README.md).  Nothing here is a
// port of reference code: it is an emulation of the *platform* underneath it.
//
// Choices that are ours (the real platform is unavailable) and therefore part
// of the parity contract (DESIGN.md "Arithmetic contract"):
//   * tex2D, linear filter: unnormalised coords, clamp-to-edge, CUDA's
//     xB = x - 0.5, i = floor(xB), alpha = frac(xB) rule, evaluated in fp32 as
//     a separable lerp  h = fma(alpha, T[i+1]-T[i], T[i]);  v = fma(beta, h1-h0, h0)
//     (hardware uses 9-bit fixed-point weights; not reproducible here).
//     A weight that is exactly 0 returns the texel untouched (so a read at a
//     texel centre is a plain load even if the neighbour holds NaN/Inf).
//     A NaN / Inf coordinate returns NaN.
//   * rsqrtf(x) = 1.0f / sqrtf(x), IEEE; fminf/fmaxf from libm.
//   * expf/sinf/acosf/atan2f: glibc.
//   * no FMA contraction (build with -ffp-contract=off).
#ifndef RMD_ORACLE_CUDA_SHIM_RUNTIME_H
#define RMD_ORACLE_CUDA_SHIM_RUNTIME_H

// Make vendored helper_math.h skip its own host re-definitions of
// fminf/fmaxf/rsqrtf (they clash with <math.h>); we provide rsqrtf below.
#ifndef __CUDACC__
#define __CUDACC__ 1
#endif

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <limits>

#define __host__
#define __device__
#define __global__
#define __constant__
#define __shared__
#define __forceinline__ inline __attribute__((always_inline))

#include "vector_types.h"

// ---------------------------------------------------------------- errors
enum cudaError { cudaSuccess = 0, cudaErrorMemoryAllocation = 2, cudaErrorInvalidValue = 11 };
typedef cudaError cudaError_t;
inline const char* cudaGetErrorString(cudaError e) {
  return e == cudaSuccess ? "no error" : "emulated CUDA error";
}

// ---------------------------------------------------------------- memory
enum cudaMemcpyKind {
  cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1,
  cudaMemcpyDeviceToHost = 2, cudaMemcpyDeviceToDevice = 3
};

template <typename T>
inline cudaError cudaMalloc(T** p, size_t n) {
  *p = static_cast<T*>(calloc(n ? n : 1, 1));
  return *p ? cudaSuccess : cudaErrorMemoryAllocation;
}
template <typename T>
inline cudaError cudaMallocPitch(T** p, size_t* pitch, size_t width_bytes, size_t height) {
  // The real allocator pads rows to 512 B on current parts; any multiple of the
  // element size is legal, so keep a non-trivial pitch to catch stride bugs.
  const size_t align = 512;
  *pitch = (width_bytes + align - 1) / align * align;
  *p = static_cast<T*>(calloc((*pitch) * (height ? height : 1), 1));
  return *p ? cudaSuccess : cudaErrorMemoryAllocation;
}
inline cudaError cudaFree(void* p) { free(p); return cudaSuccess; }
inline cudaError cudaMemcpy(void* dst, const void* src, size_t n, cudaMemcpyKind) {
  memcpy(dst, src, n);
  return cudaSuccess;
}
inline cudaError cudaMemcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch,
                              size_t width_bytes, size_t height, cudaMemcpyKind) {
  for (size_t r = 0; r < height; ++r)
    memcpy(static_cast<char*>(dst) + r * dpitch, static_cast<const char*>(src) + r * spitch, width_bytes);
  return cudaSuccess;
}
inline cudaError cudaMemset2D(void* dst, size_t pitch, int value, size_t width_bytes, size_t height) {
  for (size_t r = 0; r < height; ++r) memset(static_cast<char*>(dst) + r * pitch, value, width_bytes);
  return cudaSuccess;
}
template <typename T>
inline cudaError cudaMemcpyToSymbol(T& symbol, const void* src, size_t n) {
  memcpy(&symbol, src, n);
  return cudaSuccess;
}
inline cudaError cudaDeviceSynchronize() { return cudaSuccess; }

// ---------------------------------------------------------------- execution
// One emulated CUDA thread runs at a time per host thread; the driver sets
// these before calling a kernel function.
extern thread_local uint3 threadIdx;
extern thread_local uint3 blockIdx;
extern thread_local dim3 blockDim;
extern thread_local dim3 gridDim;
// Defined by the driver: yields the current fibre when a block is being run
// cooperatively (kernels with barriers), otherwise a no-op.
void __syncthreads();

// ---------------------------------------------------------------- math
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

// ---------------------------------------------------------------- textures
enum cudaTextureFilterMode { cudaFilterModePoint = 0, cudaFilterModeLinear = 1 };
enum cudaTextureAddressMode { cudaAddressModeWrap = 0, cudaAddressModeClamp = 1 };
enum cudaTextureReadMode { cudaReadModeElementType = 0, cudaReadModeNormalizedFloat = 1 };
#define cudaTextureType2D 2

struct cudaChannelFormatDesc { int x, y, z, w, f; };
template <typename T>
inline cudaChannelFormatDesc cudaCreateChannelDesc() {
  cudaChannelFormatDesc d = {static_cast<int>(8 * sizeof(T)), 0, 0, 0, 0};
  return d;
}

template <typename T, int Dim = cudaTextureType2D, cudaTextureReadMode Mode = cudaReadModeElementType>
struct texture {
  cudaTextureAddressMode addressMode[3];
  cudaTextureFilterMode filterMode;
  int normalized;
  const char* base;
  size_t width, height, pitch;
  const T& texel(long ix, long iy) const {
    return *reinterpret_cast<const T*>(base + static_cast<size_t>(iy) * pitch + static_cast<size_t>(ix) * sizeof(T));
  }
};

template <typename T, int D, cudaTextureReadMode M>
inline cudaError cudaBindTexture2D(size_t* offset, texture<T, D, M>& tex, const void* dev_ptr,
                                   const cudaChannelFormatDesc&, size_t width, size_t height, size_t pitch) {
  if (offset) *offset = 0;
  tex.base = static_cast<const char*>(dev_ptr);
  tex.width = width;
  tex.height = height;
  tex.pitch = pitch;
  return cudaSuccess;
}

namespace rmd_shim {
inline long clamp_index(float f, size_t n) {
  // f is an integer-valued float (result of floorf) or +-Inf is excluded by the caller.
  if (f < 0.0f) return 0;
  const float hi = static_cast<float>(n - 1);
  if (f > hi) return static_cast<long>(n - 1);
  return static_cast<long>(f);
}
inline float lerp1(float t, float a, float b) { return t == 0.0f ? a : fmaf(t, b - a, a); }
inline bool finite2(float x, float y) { return isfinite(x) && isfinite(y); }
}  // namespace rmd_shim

template <int D, cudaTextureReadMode M>
inline float tex2D(const texture<float, D, M>& t, float x, float y) {
  if (!rmd_shim::finite2(x, y)) return std::numeric_limits<float>::quiet_NaN();
  if (t.filterMode == cudaFilterModePoint)
    return t.texel(rmd_shim::clamp_index(floorf(x), t.width), rmd_shim::clamp_index(floorf(y), t.height));
  const float xb = x - 0.5f, yb = y - 0.5f;
  const float fi = floorf(xb), fj = floorf(yb);
  const float a = xb - fi, b = yb - fj;
  const long i0 = rmd_shim::clamp_index(fi, t.width), i1 = rmd_shim::clamp_index(fi + 1.0f, t.width);
  const long j0 = rmd_shim::clamp_index(fj, t.height), j1 = rmd_shim::clamp_index(fj + 1.0f, t.height);
  const float h0 = rmd_shim::lerp1(a, t.texel(i0, j0), t.texel(i1, j0));
  const float h1 = rmd_shim::lerp1(a, t.texel(i0, j1), t.texel(i1, j1));
  return rmd_shim::lerp1(b, h0, h1);
}

template <int D, cudaTextureReadMode M>
inline float2 tex2D(const texture<float2, D, M>& t, float x, float y) {
  float2 r;
  if (!rmd_shim::finite2(x, y)) {
    r.x = r.y = std::numeric_limits<float>::quiet_NaN();
    return r;
  }
  if (t.filterMode == cudaFilterModePoint)
    return t.texel(rmd_shim::clamp_index(floorf(x), t.width), rmd_shim::clamp_index(floorf(y), t.height));
  const float xb = x - 0.5f, yb = y - 0.5f;
  const float fi = floorf(xb), fj = floorf(yb);
  const float a = xb - fi, b = yb - fj;
  const long i0 = rmd_shim::clamp_index(fi, t.width), i1 = rmd_shim::clamp_index(fi + 1.0f, t.width);
  const long j0 = rmd_shim::clamp_index(fj, t.height), j1 = rmd_shim::clamp_index(fj + 1.0f, t.height);
  const float2 t00 = t.texel(i0, j0), t10 = t.texel(i1, j0), t01 = t.texel(i0, j1), t11 = t.texel(i1, j1);
  r.x = rmd_shim::lerp1(b, rmd_shim::lerp1(a, t00.x, t10.x), rmd_shim::lerp1(a, t01.x, t11.x));
  r.y = rmd_shim::lerp1(b, rmd_shim::lerp1(a, t00.y, t10.y), rmd_shim::lerp1(a, t01.y, t11.y));
  return r;
}

// Integer textures are never filtered (the reference binds them POINT).
template <int D, cudaTextureReadMode M>
inline int tex2D(const texture<int, D, M>& t, float x, float y) {
  if (!rmd_shim::finite2(x, y)) return 0;
  return t.texel(rmd_shim::clamp_index(floorf(x), t.width), rmd_shim::clamp_index(floorf(y), t.height));
}

#endif  // RMD_ORACLE_CUDA_SHIM_RUNTIME_H
