// rmd::PinholeCamera (reference: include/rmd/pinhole_camera.cuh:27-63), host side.
#ifndef RMD_PINHOLE_CAMERA_CUH_
#define RMD_PINHOLE_CAMERA_CUH_

#include <cmath>

#include <rmd/host_types.h>

namespace rmd {

struct PinholeCamera {
  PinholeCamera() : fx(0.0f), fy(0.0f), cx(0.0f), cy(0.0f) {}
  PinholeCamera(float fx_, float fy_, float cx_, float cy_) : fx(fx_), fy(fy_), cx(cx_), cy(cy_) {}

  // pixel -> bearing with z = 1
  float3 cam2world(const float2& uv) const { return make_float3((uv.x - cx) / fx, (uv.y - cy) / fy, 1.0f); }
  // point in the camera frame -> pixel
  float2 world2cam(const float3& p) const { return make_float2(fx * p.x / p.z + cx, fy * p.y / p.z + cy); }
  // angle subtended by one pixel
  float getOnePixAngle() const { return std::atan2(1.0f, 2.0f * fx) * 2.0f; }

  float fx, fy;
  float cx, cy;
};

}  // namespace rmd

#endif  // RMD_PINHOLE_CAMERA_CUH_
