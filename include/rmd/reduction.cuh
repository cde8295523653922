// rmd::ImageReducer<T> (reference: include/rmd/reduction.cuh:26-62, src/reduction.cu) over the C ABI.
// Instantiated for int and float like the reference (reduction.cu:186-187).
#ifndef RMD_REDUCTION_CUH
#define RMD_REDUCTION_CUH

#include <rmd/device_image.cuh>

namespace rmd {

template <typename T>
class ImageReducer {
 public:
  // the reference's launch shape arguments are accepted for source compatibility; the HIP reducers size themselves
  ImageReducer(dim3 /*num_threads_per_block*/ = dim3(16, 16), dim3 /*num_blocks_per_grid*/ = dim3(4, 4)) {}

  // Sum image by reduction: device pointer + row stride in elements (reduction.cuh:33-36), or a DeviceImage (:38)
  T sum(const T* in_img_data, size_t in_img_stride, size_t in_img_width, size_t in_img_height);
  T sum(const DeviceImage<T>& in_img);

  // Count elements equal to 'value' (reduction.cuh:41-48)
  size_t countEqual(const int* in_img_data, size_t in_img_stride, size_t in_img_width, size_t in_img_height, int value) {
    size_t n = 0;
    detail::throw_on_error(rmd_hip_reduce_count_eq_i32_raw(in_img_data, in_img_stride, in_img_width, in_img_height, value, &n),
                           "countEqual: reduction failed");
    return n;
  }
  size_t countEqual(const DeviceImage<int>& in_img, int value) {
    size_t n = 0;
    detail::throw_on_error(rmd_hip_reduce_count_eq_i32(in_img.handle(), value, &n), "countEqual: reduction failed");
    return n;
  }
};

template <>
inline float ImageReducer<float>::sum(const float* d, size_t stride, size_t w, size_t h) {
  float s = 0.0f;
  detail::throw_on_error(rmd_hip_reduce_sum_f32_raw(d, stride, w, h, &s), "sum: reduction failed");
  return s;
}
template <>
inline float ImageReducer<float>::sum(const DeviceImage<float>& in_img) {
  float s = 0.0f;
  detail::throw_on_error(rmd_hip_reduce_sum_f32(in_img.handle(), &s), "sum: reduction failed");
  return s;
}
template <>
inline int ImageReducer<int>::sum(const int* d, size_t stride, size_t w, size_t h) {
  int s = 0;
  detail::throw_on_error(rmd_hip_reduce_sum_i32_raw(d, stride, w, h, &s), "sum: reduction failed");
  return s;
}
template <>
inline int ImageReducer<int>::sum(const DeviceImage<int>& in_img) {
  int s = 0;
  detail::throw_on_error(rmd_hip_reduce_sum_i32(in_img.handle(), &s), "sum: reduction failed");
  return s;
}

}  // namespace rmd

#endif  // RMD_REDUCTION_CUH
