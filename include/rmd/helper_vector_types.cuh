// norm() of the reference's include/rmd/helper_vector_types.cuh:23-29 for the host-side float3 of host_types.h.
#ifndef RMD_HELPER_VECTOR_TYPES_CUH
#define RMD_HELPER_VECTOR_TYPES_CUH

#include <cmath>

#include <rmd/host_types.h>

template <typename VectorType>
inline float norm(const VectorType& v) {
  return sqrtf(dot(v, v));
}

#endif  // RMD_HELPER_VECTOR_TYPES_CUH
