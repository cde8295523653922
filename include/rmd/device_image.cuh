// rmd::DeviceImage<T>: a pitched 2-D device buffer (reference: include/rmd/device_image.cuh:34-180),
// here a thin RAII wrapper over an rmd_hip_image_t handle of the C ABI.  Same public surface:
// width / height / pitch / stride / data fields, setDevData, getDevData, zero, operator=.
#ifndef DEVICE_IMAGE_CUH
#define DEVICE_IMAGE_CUH

#include <cassert>
#include <cstddef>

#include <rmd/cuda_exception.cuh>
#include <rmd/host_types.h>

namespace rmd {

struct Size {
  int width;
  int height;
};

namespace detail {
template <typename T> struct ImageKind;
template <> struct ImageKind<float> { static const int value = RMD_HIP_KIND_F32; };
template <> struct ImageKind<int> { static const int value = RMD_HIP_KIND_I32; };
template <> struct ImageKind<float2> { static const int value = RMD_HIP_KIND_F32X2; };
}  // namespace detail

template <typename ElementType>
struct DeviceImage {
  DeviceImage(size_t width_, size_t height_) : width(width_), height(height_), pitch(0), stride(0), data(NULL), dev_ptr(NULL), handle_(NULL), owns_(true) {
    detail::throw_on_error(rmd_hip_image_create(detail::ImageKind<ElementType>::value, static_cast<int>(width_), static_cast<int>(height_), &handle_),
                           "Image: unable to allocate pitched memory.");
    refresh();
  }
  // non-owning view of a plane that lives inside another object (what SeedMatrix::getMu() & co. return)
  explicit DeviceImage(const rmd_hip_image_t* view) : width(0), height(0), pitch(0), stride(0), data(NULL), dev_ptr(NULL),
                                                     handle_(const_cast<rmd_hip_image_t*>(view)), owns_(false) {
    refresh();
  }
  ~DeviceImage() {
    if (owns_ && handle_) rmd_hip_image_destroy(handle_);  // never throws (the reference's destructor does)
  }

  void setDevData(const ElementType* aligned_data_row_major) {
    detail::throw_on_error(rmd_hip_image_upload(handle_, aligned_data_row_major), "Image: unable to copy data from host to device.");
  }
  void getDevData(ElementType* aligned_data_row_major) const {
    detail::throw_on_error(rmd_hip_image_download(handle_, aligned_data_row_major), "Image: unable to copy data from device to host.");
  }
  void zero() { detail::throw_on_error(rmd_hip_image_zero(handle_), "Image: unable to zero."); }
  DeviceImage<ElementType>& operator=(const DeviceImage<ElementType>& other) {
    if (this != &other) {
      assert(width == other.width && height == other.height);
      detail::throw_on_error(rmd_hip_image_copy(handle_, other.handle_), "Image, operator '=': unable to copy data from another image.");
    }
    return *this;
  }

  const rmd_hip_image_t* handle() const { return handle_; }

  // fields, as in the reference (dev_ptr, the device-side copy of the descriptor, has no counterpart: kernels get
  // their parameters as kernel arguments; it is kept as a null member for source compatibility)
  size_t width;
  size_t height;
  size_t pitch;
  size_t stride;
  ElementType* data;
  DeviceImage<ElementType>* dev_ptr;

 private:
  DeviceImage(const DeviceImage&);  // copying a device buffer implicitly is a bug; use operator= on two allocated images
  void refresh() {
    int k = 0, w = 0, h = 0;
    void* p = NULL;
    detail::throw_on_error(rmd_hip_image_info(handle_, &k, &w, &h, &pitch, &stride, &p), "Image: bad handle.");
    width = static_cast<size_t>(w);
    height = static_cast<size_t>(h);
    data = static_cast<ElementType*>(p);
  }
  rmd_hip_image_t* handle_;
  bool owns_;
};

}  // namespace rmd

#endif  // DEVICE_IMAGE_CUH
