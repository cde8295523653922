// Host-side pinhole model with the interface of the reference's rmd::PinholeCamera (include/rmd/pinhole_camera.cuh:27-63).
// The device code of this library has its own copy of the projection arithmetic (csrc/rmd_device.hpp: Cam, cam2world,
// world2cam); this header only serves host code written against the reference's class.
#ifndef RMD_PINHOLE_CAMERA_CUH_
#define RMD_PINHOLE_CAMERA_CUH_

#include <cmath>

#include <rmd/host_types.h>

namespace rmd {

struct PinholeCamera {
  // focal lengths and principal point in pixels (fy may be negative: the test dataset's image y axis points up)
  float fx, fy;
  float cx, cy;

  PinholeCamera();
  PinholeCamera(float focal_x, float focal_y, float centre_x, float centre_y);

  float3 cam2world(const float2& uv) const;   // pixel -> ray through it, not normalised (z = 1)
  float2 world2cam(const float3& xyz) const;  // point in the camera frame -> pixel
  float getOnePixAngle() const;               // angle subtended by one pixel at the image centre
};

inline PinholeCamera::PinholeCamera() : fx(0.0f), fy(0.0f), cx(0.0f), cy(0.0f) {}

inline PinholeCamera::PinholeCamera(float focal_x, float focal_y, float centre_x, float centre_y)
    : fx(focal_x), fy(focal_y), cx(centre_x), cy(centre_y) {}

inline float3 PinholeCamera::cam2world(const float2& uv) const {
  const float ray_x = (uv.x - cx) / fx;
  const float ray_y = (uv.y - cy) / fy;
  return make_float3(ray_x, ray_y, 1.0f);
}

inline float2 PinholeCamera::world2cam(const float3& xyz) const {
  const float u = fx * xyz.x / xyz.z + cx;  // (fx * x) / z + cx, one rounding per operation as in the reference
  const float v = fy * xyz.y / xyz.z + cy;
  return make_float2(u, v);
}

inline float PinholeCamera::getOnePixAngle() const {
  const float half_pixel_angle = std::atan2(1.0f, 2.0f * fx);
  return half_pixel_angle * 2.0f;
}

}  // namespace rmd

#endif  // RMD_PINHOLE_CAMERA_CUH_
