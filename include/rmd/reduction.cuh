// rmd::ImageReducer<T> (reference: include/rmd/reduction.cuh:26-62, src/reduction.cu) over the C ABI.
#ifndef RMD_REDUCTION_CUH
#define RMD_REDUCTION_CUH

#include <rmd/device_image.cuh>

namespace rmd {

template <typename T>
class ImageReducer {
 public:
  // the reference's launch shape arguments are accepted for source compatibility; the HIP reducers size themselves
  ImageReducer(dim3 /*num_threads_per_block*/ = dim3(16, 16), dim3 /*num_blocks_per_grid*/ = dim3(4, 4)) {}

  T sum(const DeviceImage<T>& in_img);
  size_t countEqual(const DeviceImage<int>& in_img, int value) {
    size_t n = 0;
    detail::throw_on_error(rmd_hip_reduce_count_eq_i32(in_img.handle(), value, &n), "countEqual: reduction failed");
    return n;
  }
};

template <>
inline float ImageReducer<float>::sum(const DeviceImage<float>& in_img) {
  float s = 0.0f;
  detail::throw_on_error(rmd_hip_reduce_sum_f32(in_img.handle(), &s), "sum: reduction failed");
  return s;
}

}  // namespace rmd

#endif  // RMD_REDUCTION_CUH
