// rmd::SeedMatrixBatch -- NOT in the reference: several independent rmd::SeedMatrix objects of one size on one GPU whose update()
// calls are issued together, as one launch pair per step (BASELINE configs[3] / north_star "a batched mode shards independent image
// sequences").  A single 640x480 sequence cannot fill an MI355X; four of them stepped together run at twice the aggregate rate.
// Each member is a full rmd::SeedMatrix (setReferenceImage, downloads, getters, getConvergedCount, use with DepthmapDenoiser) whose
// results are bit-identical to the same sequence in a stand-alone SeedMatrix (seed_matrix.cu:120-158 per member); only update() is
// replaced by the batch's.  Inline wrapper over librmd_hip.so (rmd_hip_batch_*).
#ifndef SEED_MATRIX_BATCH_CUH
#define SEED_MATRIX_BATCH_CUH

#include <vector>

#include <rmd/seed_matrix.cuh>

namespace rmd {

class SeedMatrixBatch {
 public:
  SeedMatrixBatch(const size_t& n, const size_t& width, const size_t& height, const PinholeCamera& cam) : handle_(NULL) {
    detail::throw_on_error(rmd_hip_batch_create(static_cast<int>(n), static_cast<int>(width), static_cast<int>(height), cam.fx, cam.fy, cam.cx, cam.cy,
                                                RMD_CORR_PATCH_SIDE, RMD_MAX_EXTENT_EPIPOLAR_SEARCH, &handle_),
                           "SeedMatrixBatch: unable to create");
    try {  // (the destructor does not run for a partly constructed object: release what exists so far, then rethrow)
      for (size_t i = 0; i < n; ++i) {
        rmd_hip_seeds_t* m = NULL;
        detail::throw_on_error(rmd_hip_batch_member(handle_, static_cast<int>(i), &m), "SeedMatrixBatch: member");
        members_.push_back(new SeedMatrix(m));
      }
      poses_.resize(12 * n);
    } catch (...) {
      for (size_t i = 0; i < members_.size(); ++i) delete members_[i];
      members_.clear();
      rmd_hip_batch_destroy(handle_);
      handle_ = NULL;
      throw;
    }
  }
  ~SeedMatrixBatch() {
    for (size_t i = 0; i < members_.size(); ++i) delete members_[i];
    rmd_hip_batch_destroy(handle_);
  }
  size_t size() const { return members_.size(); }
  SeedMatrix& operator[](size_t i) { return *members_[i]; }
  const SeedMatrix& operator[](size_t i) const { return *members_[i]; }

  // SeedMatrix::update for every member i with host_curr_imgs[i] != NULL (a member without a frame in this step is left alone);
  // T_curr_world: one pose per member
  bool update(float* const* host_curr_imgs_align_row_maj, const SE3<float>* T_curr_world) {
    pack(T_curr_world);
    detail::throw_on_error(rmd_hip_batch_update(handle_, host_curr_imgs_align_row_maj, &poses_[0]), "SeedMatrixBatch: update failed");
    return true;
  }
  // the same for 8-bit gray frames (what rmd::Depthmap::inputImage receives, depthmap.cpp:95-106: x(1/255) on the device)
  bool updateU8(const unsigned char* const* host_gray, const SE3<float>* T_curr_world) {
    pack(T_curr_world);
    detail::throw_on_error(rmd_hip_batch_update_u8(handle_, host_gray, &poses_[0]), "SeedMatrixBatch: update failed");
    return true;
  }
  void synchronize() { detail::throw_on_error(rmd_hip_batch_sync(handle_), "SeedMatrixBatch: synchronize failed"); }
  // DepthmapDenoiser::denoise for every member in one launch sequence (depthmap_denoiser.cu:179-224 per member, bit for bit): depth_range[i]
  // = what DepthmapDenoiser::setLargeSigmaSq gets for member i; host_denoised[i] = W x H floats or NULL
  void denoise(const float* depth_range, float* const* host_denoised, float lambda, int iterations) {
    detail::throw_on_error(rmd_hip_batch_denoise(handle_, depth_range, lambda, iterations, host_denoised), "SeedMatrixBatch: denoise failed");
  }
  rmd_hip_batch_t* handle() const { return handle_; }

 private:
  SeedMatrixBatch(const SeedMatrixBatch&);
  SeedMatrixBatch& operator=(const SeedMatrixBatch&);
  void pack(const SE3<float>* T) {
    for (size_t i = 0; i < members_.size(); ++i)
      for (int k = 0; k < 12; ++k) poses_[12 * i + k] = T[i].data.data[k];
  }
  rmd_hip_batch_t* handle_;
  std::vector<SeedMatrix*> members_;
  std::vector<float> poses_;
};

}  // namespace rmd

#endif  // SEED_MATRIX_BATCH_CUH
