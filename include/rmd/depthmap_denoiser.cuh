// rmd::DepthmapDenoiser (reference: include/rmd/depthmap_denoiser.cuh:27-54, src/depthmap_denoiser.cu) over the C ABI.
#ifndef RMD_DEPTHMAP_DENOISER_CUH
#define RMD_DEPTHMAP_DENOISER_CUH

#include <iostream>

#include <rmd/device_image.cuh>

namespace rmd {

class DepthmapDenoiser {
 public:
  DepthmapDenoiser(size_t width, size_t height) : handle_(NULL) {
    detail::throw_on_error(rmd_hip_denoiser_create(static_cast<int>(width), static_cast<int>(height), &handle_), "DepthmapDenoiser: unable to create");
  }
  ~DepthmapDenoiser() { rmd_hip_denoiser_destroy(handle_); }

  void denoise(const rmd::DeviceImage<float>& mu, const rmd::DeviceImage<float>& sigma_sq, const rmd::DeviceImage<float>& a,
               const rmd::DeviceImage<float>& b, float* host_denoised, float lambda, int iterations) {
    const int rc = rmd_hip_denoiser_denoise(handle_, mu.handle(), sigma_sq.handle(), a.handle(), b.handle(), host_denoised, lambda, iterations);
    if (rc == RMD_HIP_ERR_NOT_READY) {  // the reference prints and returns (depthmap_denoiser.cu:189-193)
      std::cerr << "ERROR: setLargeSigmaSq must be called before this method" << std::endl;
      return;
    }
    detail::throw_on_error(rc, "DepthmapDenoiser: denoise failed");
  }
  void setLargeSigmaSq(float depth_range) {
    detail::throw_on_error(rmd_hip_denoiser_set_large_sigma_sq(handle_, depth_range), "DepthmapDenoiser: setLargeSigmaSq failed");
  }
  // Extension: the device-resident output of the last denoise() (valid until the next one), for SeedMatrix::downloadPointCloud.
  const rmd_hip_image_t* resultHandle() const {
    const rmd_hip_image_t* v = NULL;
    detail::throw_on_error(rmd_hip_denoiser_result(handle_, &v), "DepthmapDenoiser: result failed");
    return v;
  }

 private:
  DepthmapDenoiser(const DepthmapDenoiser&);
  DepthmapDenoiser& operator=(const DepthmapDenoiser&);
  rmd_hip_denoiser_t* handle_;
};

}  // namespace rmd

#endif  // RMD_DEPTHMAP_DENOISER_CUH
