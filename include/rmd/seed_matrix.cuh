// rmd::SeedMatrix (reference: include/rmd/seed_matrix.cuh:45-109, src/seed_matrix.cu) as an inline wrapper over
// librmd_hip.so.  Same constructor, methods, return conventions (bool always true, exceptions on runtime errors).
// RMD_CORR_PATCH_SIDE / RMD_MAX_EXTENT_EPIPOLAR_SEARCH keep their role as compile-time knobs of the consumer
// (CMakeLists.txt:50-53); they are forwarded to the library, which holds all four patch sizes.
#ifndef SEED_MATRIX_CUH
#define SEED_MATRIX_CUH

#include <rmd/device_image.cuh>
#include <rmd/pinhole_camera.cuh>
#include <rmd/reduction.cuh>
#include <rmd/se3.cuh>

#ifndef RMD_CORR_PATCH_SIDE
#define RMD_CORR_PATCH_SIDE 5
#endif
#define RMD_CORR_PATCH_OFFSET -RMD_CORR_PATCH_SIDE / 2
#define RMD_CORR_PATCH_AREA RMD_CORR_PATCH_SIDE* RMD_CORR_PATCH_SIDE
#ifndef RMD_MAX_EXTENT_EPIPOLAR_SEARCH
#define RMD_MAX_EXTENT_EPIPOLAR_SEARCH 100
#endif

namespace rmd {

namespace ConvergenceStates {
enum ConvergenceState { UPDATE = 0, CONVERGED, BORDER, DIVERGED, NO_MATCH, NOT_VISIBLE };
}
typedef ConvergenceStates::ConvergenceState ConvergenceState;

class SeedMatrix {
 public:
  SeedMatrix(const size_t& width, const size_t& height, const PinholeCamera& cam)
      : handle_(NULL), owns_(true), mu_(NULL), sigma_(NULL), a_(NULL), b_(NULL), conv_(NULL) {
    detail::throw_on_error(rmd_hip_seeds_create(static_cast<int>(width), static_cast<int>(height), cam.fx, cam.fy, cam.cx, cam.cy,
                                                RMD_CORR_PATCH_SIDE, RMD_MAX_EXTENT_EPIPOLAR_SEARCH, &handle_),
                           "SeedMatrix: unable to create");
    make_views();
  }
  ~SeedMatrix() {
    delete mu_; delete sigma_; delete a_; delete b_; delete conv_;
    if (owns_) rmd_hip_seeds_destroy(handle_);
  }

  bool setReferenceImage(float* host_ref_img_align_row_maj, const SE3<float>& T_curr_world, const float& min_depth, const float& max_depth) {
    detail::throw_on_error(rmd_hip_seeds_set_reference(handle_, host_ref_img_align_row_maj, T_curr_world.data.data, min_depth, max_depth),
                           "SeedMatrix: setReferenceImage failed");
    return true;
  }
  bool update(float* host_curr_img_align_row_maj, const SE3<float>& T_curr_world) {
    detail::throw_on_error(rmd_hip_seeds_update(handle_, host_curr_img_align_row_maj, T_curr_world.data.data), "SeedMatrix: update failed");
    return true;
  }
  // not in the reference: an 8-bit frame the caller keeps in pinned host memory (rmd_hip_host_alloc) is read where it lies; it must stay
  // unchanged until pinnedFramesDone() has reached the ticket returned here (include/rmd_hip.h)
  unsigned long long updatePinned(const unsigned char* pinned_gray_align_row_maj, const SE3<float>& T_curr_world) {
    unsigned long long ticket = 0;
    detail::throw_on_error(rmd_hip_seeds_update_u8_pinned(handle_, pinned_gray_align_row_maj, T_curr_world.data.data, &ticket),
                           "SeedMatrix: updatePinned failed");
    return ticket;
  }
  unsigned long long pinnedFramesDone() {
    unsigned long long done = 0;
    detail::throw_on_error(rmd_hip_seeds_pinned_frames_done(handle_, &done), "SeedMatrix: pinnedFramesDone failed");
    return done;
  }

  void downloadDepthmap(float* host_depthmap_align_row_maj) const { download(RMD_HIP_PLANE_MU, host_depthmap_align_row_maj); }
  void downloadConvergence(int* host_align_row_maj) const { download(RMD_HIP_PLANE_CONVERGENCE, host_align_row_maj); }

  const DeviceImage<float>& getMu() const { return *mu_; }
  const DeviceImage<float>& getSigmaSq() const { return *sigma_; }
  const DeviceImage<float>& getA() const { return *a_; }
  const DeviceImage<float>& getB() const { return *b_; }
  const DeviceImage<int>& getConvergence() const { return *conv_; }

  size_t getConvergedCount() const {
    size_t n = 0;
    detail::throw_on_error(rmd_hip_seeds_converged_count(handle_, &n), "SeedMatrix: getConvergedCount failed");
    return n;
  }
  float getDistFromRef() const {
    float d = 0.0f;
    detail::throw_on_error(rmd_hip_seeds_dist_from_ref(handle_, &d), "SeedMatrix: getDistFromRef failed");
    return d;
  }

  // Extension (not in the reference class): the coloured convergence map of Publisher::publishConvergenceMap (publisher.cpp:112-147) on the
  // device -- gray reference image as BGR, blue = 255 where CONVERGED, red = 255 where DIVERGED; W x H x 3 bytes, packed (the data of a
  // cv::Mat CV_8UC3): a 3-byte-per-pixel download instead of downloadConvergence + a host loop.
  void downloadConvergenceBGR8(unsigned char* host_bgr) const {
    detail::throw_on_error(rmd_hip_seeds_convergence_bgr8(handle_, host_bgr), "SeedMatrix: downloadConvergenceBGR8 failed");
  }

  // Extension (not in the reference class): the loop of Publisher::publishPointCloud (publisher.cpp:54-104) on the device.
  // World-frame (x, y, z, intensity) of every CONVERGED seed, row-major pixel order; `depth` = NULL uses the seeds' own mu,
  // otherwise e.g. DepthmapDenoiser::resultHandle().  Returns the number of converged seeds; at most `capacity` points are written.
  size_t downloadPointCloud(const rmd_hip_image_t* depth, float* host_xyzi, size_t capacity) const {
    size_t n = 0;
    detail::throw_on_error(rmd_hip_seeds_point_cloud(handle_, depth, host_xyzi, capacity, &n), "SeedMatrix: downloadPointCloud failed");
    return n;
  }
  size_t downloadPointCloud(const DeviceImage<float>& depth, float* host_xyzi, size_t capacity) const {
    return downloadPointCloud(depth.handle(), host_xyzi, capacity);
  }

  // Extension (not in the reference class): publication off the update stream -- what DepthmapNode::denoiseAndPublishResults /
  // publishConvergenceMap (depthmap_node.cpp:165-182) compute between two messages, requested without waiting for it.  publishAsync snapshots
  // the state (one device-to-device kernel) and queues TV-L1 denoising (lambda, iterations; setLargeSigmaSq(depth_range)), the point cloud of
  // the converged seeds, the coloured convergence map and / or the convergence plane (RMD_HIP_PUBLISH_* bits) on the handle's second stream;
  // setReferenceImage / update may follow at once.  collectPublication hands over the oldest publication not collected yet: true and the
  // products in the caller's buffers (NULL: not wanted), false when it is still in flight and `wait` is false.  At most
  // RMD_HIP_PUBLISH_SLOTS publications may be uncollected.
  int publishAsync(unsigned int what, float depth_range, float lambda, int iterations) {
    int ticket = 0;
    detail::throw_on_error(rmd_hip_seeds_publish_async(handle_, what, depth_range, lambda, iterations, &ticket), "SeedMatrix: publishAsync failed");
    return ticket;
  }
  bool collectPublication(bool wait, unsigned int* what, int* ticket, float* host_depth, float* host_xyzi, size_t capacity, size_t* n_points,
                          unsigned char* host_bgr, int* host_convergence) {
    const int rc = rmd_hip_seeds_publish_collect(handle_, wait ? 1 : 0, what, ticket, host_depth, host_xyzi, capacity, n_points, host_bgr, host_convergence);
    if (rc == RMD_HIP_BUSY) return false;
    detail::throw_on_error(rc, "SeedMatrix: collectPublication failed");
    return true;
  }

#if RMD_BUILD_TESTS
  void downloadSigmaSq(float* host_align_row_maj) const { download(RMD_HIP_PLANE_SIGMA_SQ, host_align_row_maj); }
  void downloadA(float* host_align_row_maj) const { download(RMD_HIP_PLANE_A, host_align_row_maj); }
  void downloadB(float* host_align_row_maj) const { download(RMD_HIP_PLANE_B, host_align_row_maj); }
  void downloadSumTempl(float* host_align_row_maj) const { download(RMD_HIP_PLANE_SUM_TEMPL, host_align_row_maj); }
  void downloadConstTemplDenom(float* host_align_row_maj) const { download(RMD_HIP_PLANE_CONST_TEMPL_DENOM, host_align_row_maj); }
  void downloadEpipolarMatches(float2* host_align_row_maj) const { download(RMD_HIP_PLANE_EPIPOLAR_MATCHES, host_align_row_maj); }
#endif

  rmd_hip_seeds_t* handle() const { return handle_; }

 private:
  friend class SeedMatrixBatch;
  // a member of a SeedMatrixBatch (seed_matrix_batch.cuh): the batch owns the handle
  explicit SeedMatrix(rmd_hip_seeds_t* member) : handle_(member), owns_(false), mu_(NULL), sigma_(NULL), a_(NULL), b_(NULL), conv_(NULL) { make_views(); }
  void make_views() {
    mu_ = view<float>(RMD_HIP_PLANE_MU);
    sigma_ = view<float>(RMD_HIP_PLANE_SIGMA_SQ);
    a_ = view<float>(RMD_HIP_PLANE_A);
    b_ = view<float>(RMD_HIP_PLANE_B);
    conv_ = view<int>(RMD_HIP_PLANE_CONVERGENCE);
  }
  SeedMatrix(const SeedMatrix&);
  SeedMatrix& operator=(const SeedMatrix&);
  void download(int plane, void* dst) const {
    detail::throw_on_error(rmd_hip_seeds_download(handle_, plane, dst), "SeedMatrix: download failed");
  }
  template <typename T>
  DeviceImage<T>* view(int plane) const {
    const rmd_hip_image_t* v = NULL;
    detail::throw_on_error(rmd_hip_seeds_plane(handle_, plane, &v), "SeedMatrix: plane view failed");
    return new DeviceImage<T>(v);
  }
  rmd_hip_seeds_t* handle_;
  bool owns_;
  DeviceImage<float>*mu_, *sigma_, *a_, *b_;
  DeviceImage<int>* conv_;
};

}  // namespace rmd

#endif  // SEED_MATRIX_CUH
