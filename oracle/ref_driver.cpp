// TEST INFRASTRUCTURE ONLY ("Oracle A") -- never shipped, never linked into the product.
//
// Builds the reference's OWN host orchestration and device kernels
//   /root/reference/src/seed_matrix.cu  (+ seed_init.cu, seed_check.cu, epipolar_match.cu,
//                                          seed_update.cu, triangulation.cu)
//   /root/reference/src/depthmap_denoiser.cu
//   /root/reference/src/reduction.cu    (+ reduction_kernels.cu)
// for the CPU, from where they lie, through the emulation shim in oracle/cuda_shim/.
// The three macros below name pipes set up by oracle/Makefile; each pipe carries
// one of the files above with the single non-C++ construct `k<<<g,b>>>(...)`
// rewritten to `RMD_SHIM_LAUNCH(k, g, b)(...)`.  No reference source is copied
// into this repository; the only outputs are shared objects under oracle/_ref/.
//
// The extern "C" surface below is what tests/ and bench.py's cpu_baseline leg
// call through ctypes.  It forwards 1:1 to the reference classes
// rmd::SeedMatrix (seed_matrix.cuh:45-109), rmd::DepthmapDenoiser
// (depthmap_denoiser.cuh:27-54) and rmd::ImageReducer<T> (reduction.cuh:26-62).
// Standard headers first: the access hack below must not reach into libstdc++.
#include <assert.h>
#include <float.h>
#include <iomanip>
#include <iostream>
#include <ostream>
#include <sstream>
#include <string>

#include <cuda_runtime.h>
#include <shim_launch.h>

#ifdef RMD_REF_USE_SHARED_MATH
// Variant "ref_rmd": the reference's sources with the three libm transcendentals on the path
// (seed_update.cu:36 expf; triangulation.cu:63-66 acosf, sinf) taken from csrc/rmd_math.h -- the
// header the HIP kernels compile -- instead of glibc.  Nothing else changes.  This is the oracle the
// HIP path must equal bit for bit; the plain build ("ref") measures what that substitution moves.
#include "rmd_math.h"
#define expf rmd_expf
#define sinf rmd_sinf
#define acosf rmd_acosf
#endif

#define private public  // the driver reaches into the reference classes' buffers (test hooks)
// reduction.cu first: it explicitly specialises ImageReducer<int>::countEqual, which
// seed_matrix.cu uses (separate translation units in the reference's own build).
#include RMD_REF_REDUCTION_CU
#include RMD_REF_SEED_MATRIX_CU
#include RMD_REF_DENOISER_CU
#undef private

#include <omp.h>
#include <new>

// Dynamic shared memory of the reduction kernels (reduction_kernels.cu:37-55 declares
// these `extern __shared__`); fibre-mode blocks run serially so one copy suffices.
namespace rmd {
int s_int[4096];
float s_float[4096];
}  // namespace rmd

namespace {
rmd::SE3<float> se3_from_rowmajor(const float* T) {
  float r[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
  float t[3] = {T[3], T[7], T[11]};
  return rmd::SE3<float>(r, t);
}
}  // namespace

extern "C" {

int ref_patch_side(void) { return RMD_CORR_PATCH_SIDE; }
int ref_max_threads(void) { return omp_get_max_threads(); }
void ref_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int ref_max_extent(void) { return RMD_MAX_EXTENT_EPIPOLAR_SEARCH; }
int ref_uses_shared_math(void) {
#ifdef RMD_REF_USE_SHARED_MATH
  return 1;
#else
  return 0;
#endif
}

void* ref_seeds_create(int w, int h, float fx, float fy, float cx, float cy) {
  return new rmd::SeedMatrix(static_cast<size_t>(w), static_cast<size_t>(h), rmd::PinholeCamera(fx, fy, cx, cy));
}
void ref_seeds_destroy(void* s) { delete static_cast<rmd::SeedMatrix*>(s); }

int ref_seeds_set_reference(void* s, const float* img, const float* T_curr_world, float min_depth, float max_depth) {
  return static_cast<rmd::SeedMatrix*>(s)->setReferenceImage(const_cast<float*>(img), se3_from_rowmajor(T_curr_world),
                                                            min_depth, max_depth) ? 0 : -1;
}
int ref_seeds_update(void* s, const float* img, const float* T_curr_world) {
  return static_cast<rmd::SeedMatrix*>(s)->update(const_cast<float*>(img), se3_from_rowmajor(T_curr_world)) ? 0 : -1;
}

// plane ids shared with include/rmd_hip.h (RMD_HIP_PLANE_*)
int ref_seeds_download(void* sp, int plane, void* dst) {
  rmd::SeedMatrix* s = static_cast<rmd::SeedMatrix*>(sp);
  switch (plane) {
    case 0: s->downloadDepthmap(static_cast<float*>(dst)); return 0;
    case 1: s->downloadSigmaSq(static_cast<float*>(dst)); return 0;
    case 2: s->downloadA(static_cast<float*>(dst)); return 0;
    case 3: s->downloadB(static_cast<float*>(dst)); return 0;
    case 4: s->downloadConvergence(static_cast<int*>(dst)); return 0;
    case 5: s->downloadSumTempl(static_cast<float*>(dst)); return 0;
    case 6: s->downloadConstTemplDenom(static_cast<float*>(dst)); return 0;
    case 7: s->downloadEpipolarMatches(static_cast<float2*>(dst)); return 0;
    default: return -1;
  }
}
// test hook: overwrite a state plane (mu, sigma_sq, a, b) so single stages can be
// driven with arbitrary seed states
int ref_seeds_upload(void* sp, int plane, const float* src) {
  rmd::SeedMatrix* s = static_cast<rmd::SeedMatrix*>(sp);
  switch (plane) {
    case 0: s->mu_.setDevData(src); return 0;
    case 1: s->sigma_.setDevData(src); return 0;
    case 2: s->a_.setDevData(src); return 0;
    case 3: s->b_.setDevData(src); return 0;
    default: return -1;
  }
}
unsigned long ref_seeds_converged_count(void* s) { return static_cast<rmd::SeedMatrix*>(s)->getConvergedCount(); }
float ref_seeds_dist_from_ref(void* s) { return static_cast<rmd::SeedMatrix*>(s)->getDistFromRef(); }

void* ref_denoiser_create(int w, int h) {
  rmd::DepthmapDenoiser* d = new rmd::DepthmapDenoiser(static_cast<size_t>(w), static_cast<size_t>(h));
  // depthmap_denoiser.cu:124-141 leaves large_sigma_sq uninitialised; callers
  // must set it (depthmap.cpp:69 always does).  Pin it so a forgotten call is loud.
  d->host_ptr->large_sigma_sq = -1.0f;
  return d;
}
void ref_denoiser_destroy(void* d) { delete static_cast<rmd::DepthmapDenoiser*>(d); }
void ref_denoiser_set_large_sigma_sq(void* d, float depth_range) {
  static_cast<rmd::DepthmapDenoiser*>(d)->setLargeSigmaSq(depth_range);
}
int ref_denoiser_denoise(void* d, void* sp, float* host_out, float lambda, int iterations) {
  rmd::SeedMatrix* s = static_cast<rmd::SeedMatrix*>(sp);
  static_cast<rmd::DepthmapDenoiser*>(d)->denoise(s->getMu(), s->getSigmaSq(), s->getA(), s->getB(), host_out, lambda,
                                                  iterations);
  return 0;
}
void ref_denoiser_constants(void* dp, float* out4) {
  rmd::DepthmapDenoiser* d = static_cast<rmd::DepthmapDenoiser*>(dp);
  out4[0] = d->host_ptr->L; out4[1] = d->host_ptr->tau; out4[2] = d->host_ptr->sigma; out4[3] = d->host_ptr->theta;
}

float ref_reduce_sum_f32(const float* host_img, int w, int h) {
  rmd::DeviceImage<float> img(static_cast<size_t>(w), static_cast<size_t>(h));
  img.setDevData(host_img);
  rmd::ImageReducer<float> red(dim3(16, 16), dim3(4, 4));
  return red.sum(img);
}
unsigned long ref_reduce_count_eq_i32(const int* host_img, int w, int h, int value) {
  rmd::DeviceImage<int> img(static_cast<size_t>(w), static_cast<size_t>(h));
  img.setDevData(host_img);
  rmd::ImageReducer<int> red(dim3(16, 16), dim3(4, 4));
  return red.countEqual(img, value);
}

}  // extern "C"
