// librmd_hip.so -- rmd::DepthmapDenoiser (depthmap_denoiser.cu:124-229): rmd_hip_denoiser_*, and the launch sequence shared with
// rmd_hip_batch_denoise.
#include "rmd_host.hpp"
#include "rmd_tv_kernels.hpp"

using namespace rmdh;

namespace rmdh {

// tv_prepare + `iterations` primal-dual iterations of TvParams P for `n_z` depth maps (grid z; 1 = the single denoiser, P.members null) on
// `stream`, ping-ponging between the two sets of iterate planes; *result_index = the set that holds the result.  ev0 (may be null) is
// recorded between the preparation and the first iteration.
int tv_run(const rmdk::TvParams& P, float* const u[2], float* const uh[2], float2* const p[2], int n_z, int iterations,
    int opt_iters_per_launch, int opt_geometry,
           hipStream_t stream, hipEvent_t ev0, int* result_index, long* launches) {
  const unsigned int nz = static_cast<unsigned int>(n_z);
  {
    const dim3 block(64, 4), grid((P.w + 63) / 64, (P.h + 3) / 4, nz);
    hipLaunchKernelGGL(rmdk::tv_prepare_kernel, grid, block, 0, stream, P, u[0], uh[0], p[0]);
    HIP_TRY(hipGetLastError());
  }
  int cur_buf = 0;
  if (ev0) HIP_TRY(hipEventRecord(ev0, stream));
  long n_launches = 0;
  if (opt_iters_per_launch == 1) {
    const dim3 block(rmdk::TV_TX, rmdk::TV_TY);
    const dim3 grid((P.w + rmdk::TV_TX - 1) / rmdk::TV_TX, (P.h + rmdk::TV_TY - 1) / rmdk::TV_TY, nz);
    for (int it = 0; it < iterations; ++it) {
      const int nxt = cur_buf ^ 1;
      hipLaunchKernelGGL(rmdk::tv_iterate_kernel, grid, block, 0, stream, P, u[cur_buf], uh[cur_buf], p[cur_buf], u[nxt], uh[nxt], p[nxt]);
      cur_buf = nxt;
      ++n_launches;
    }
  } else {
    // Temporally blocked kernel: tile geometry and blocking depth K by image size (K iterations per launch; the halo
    // grows with K, so small tiles pay more redundant work per iteration, but a VGA launch is latency-bound: fewer, fatter
    // launches win there).  opt_geometry (experiments): 0 = by size, 1..n = a fixed entry of the table below.
    auto run = [&](auto geom, int kmax, auto kernel) {
      using G = decltype(geom);
      const int k = opt_iters_per_launch == 0 ? kmax : (opt_iters_per_launch < kmax ? opt_iters_per_launch : kmax);
      const dim3 block(G::THREADS), grid((P.w + G::BX - 1) / G::BX, (P.h + G::BY - 1) / G::BY, nz);
      for (int done = 0; done < iterations; done += k) {
        const int now = iterations - done < k ? iterations - done : k;
        const int nxt = cur_buf ^ 1;
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, P, u[cur_buf], uh[cur_buf], p[cur_buf], u[nxt], uh[nxt], p[nxt], now);
        cur_buf = nxt;
        ++n_launches;
      }
    };
    // Measured (tools/denoise_sweep.py, MI355X): 16x16 tiles with K = 4 are the fastest at 640x480 (3.4 us per iteration, 50
    // launches for 200 iterations, latency-bound) AND at 1920x1080 (11.5 us per iteration = 7.2 TB/s of algorithmic traffic);
    // deeper blocking (K = 8) loses more to the redundant halo work than it saves in launches.
    int geometry = opt_geometry;
    if (geometry == 0) geometry = opt_iters_per_launch == 2 ? 1 : 4;
    switch (geometry) {
      case 1: run(rmdk::TvBlocked<32, 8, 2>(), 2, rmdk::tv_iterate_blocked_kernel<32, 8, 2>); break;
      case 2: run(rmdk::TvBlocked<64, 16, 4>(), 4, rmdk::tv_iterate_blocked_kernel<64, 16, 4>); break;
      case 3: run(rmdk::TvBlocked<32, 16, 4>(), 4, rmdk::tv_iterate_blocked_kernel<32, 16, 4>); break;
      case 4: run(rmdk::TvBlocked<16, 16, 4>(), 4, rmdk::tv_iterate_blocked_kernel<16, 16, 4>); break;
      case 5: run(rmdk::TvBlocked<16, 16, 8>(), 8, rmdk::tv_iterate_blocked_kernel<16, 16, 8>); break;
      default: return fail(RMD_HIP_ERR_INVALID_ARG, "denoise: unknown geometry %d", geometry);
    }
  }
  HIP_TRY(hipGetLastError());
  *result_index = cur_buf;
  *launches = n_launches;
  return RMD_HIP_OK;
}

}  // namespace rmdh

extern "C" {

// ---- DepthmapDenoiser -----------------------------------------------------------------------
int rmd_hip_denoiser_destroy(rmd_hip_denoiser_t* d) {
  if (!d) return RMD_HIP_OK;
  (void)hipSetDevice(d->device);
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  d->timer.destroy();
  if (d->ev0) (void)hipEventDestroy(d->ev0);
  if (d->ev1) (void)hipEventDestroy(d->ev1);
  if (d->h_staging) (void)hipHostFree(d->h_staging);
  rmd_hip_image* all[] = {&d->u[0], &d->u[1], &d->u_head[0], &d->u_head[1], &d->p[0], &d->p[1], &d->g};
  for (auto* im : all)
    if (im->owns && im->data) (void)hipFree(im->data);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
  return RMD_HIP_OK;
}

int rmd_hip_denoiser_create(int width, int height, rmd_hip_denoiser_t** out) {
  if (!out) return fail(RMD_HIP_ERR_INVALID_ARG, "denoiser_create: null output");
  *out = nullptr;
  if (width <= 0 || height <= 0) return fail(RMD_HIP_ERR_INVALID_ARG, "denoiser_create: bad size %dx%d", width, height);
  int ndev = 0;
  TRY(rmd_hip_device_count(&ndev));
  rmd_hip_denoiser* d = new (std::nothrow) rmd_hip_denoiser();
  if (!d) return fail(RMD_HIP_ERR_RUNTIME, "denoiser_create: out of host memory");
  d->width = width; d->height = height;
  (void)hipGetDevice(&d->device);
  // denoise::DeviceData constructor, depthmap_denoiser.cu:124-141
  d->L = sqrtf(8.0f);
  d->tau = 0.02f;
  d->sigma = (1 / (d->L * d->L)) / d->tau;
  d->theta = 0.5f;
  auto bail = [&](int rc) { rmd_hip_denoiser_destroy(d); return rc; };
  if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(RMD_HIP_ERR_RUNTIME, "denoiser_create: hipStreamCreate failed"));
  rmd_hip_image* f32[] = {&d->u[0], &d->u[1], &d->u_head[0], &d->u_head[1], &d->g};
  for (auto* im : f32) {
    const int rc = image_alloc(im, RMD_HIP_KIND_F32, width, height);
    if (rc != RMD_HIP_OK) return bail(rc);
    im->owner_stream = d->stream;
  }
  for (int k = 0; k < 2; ++k) {
    const int rc = image_alloc(&d->p[k], RMD_HIP_KIND_F32X2, width, height);
    if (rc != RMD_HIP_OK) return bail(rc);
    d->p[k].owner_stream = d->stream;
  }
  if (hipHostMalloc(reinterpret_cast<void**>(&d->h_staging), static_cast<size_t>(width) * height * 4) != hipSuccess)
    return bail(fail(RMD_HIP_ERR_RUNTIME, "denoiser_create: pinned staging buffer"));
  if (hipDeviceSynchronize() != hipSuccess) return bail(fail(RMD_HIP_ERR_RUNTIME, "denoiser_create: device synchronisation failed"));
  *out = d;
  return RMD_HIP_OK;
}

int rmd_hip_denoiser_set_large_sigma_sq(rmd_hip_denoiser_t* d, float depth_range) {
  if (!d) return fail(RMD_HIP_ERR_INVALID_ARG, "set_large_sigma_sq: null handle");
  d->large_sigma_sq = depth_range * depth_range / 72.0f;
  return RMD_HIP_OK;
}

int rmd_hip_denoiser_constants(const rmd_hip_denoiser_t* d, float* out4) {
  if (!d || !out4) return fail(RMD_HIP_ERR_INVALID_ARG, "denoiser_constants: null argument");
  out4[0] = d->L; out4[1] = d->tau; out4[2] = d->sigma; out4[3] = d->theta;
  return RMD_HIP_OK;
}

int rmd_hip_denoiser_set_option(rmd_hip_denoiser_t* d, int option, int value) {
  if (!d) return fail(RMD_HIP_ERR_INVALID_ARG, "denoiser_set_option: null handle");
  switch (option) {
    case RMD_HIP_DENOISE_OPT_TIMING: d->opt_timing = value != 0; return RMD_HIP_OK;
    case RMD_HIP_DENOISE_OPT_ITERS_PER_LAUNCH:
      if (value < 0 || value > 8) return fail(RMD_HIP_ERR_INVALID_ARG, "iters_per_launch %d outside 0..8", value);
      d->opt_iters_per_launch = value;
      return RMD_HIP_OK;
    case RMD_HIP_DENOISE_OPT_GEOMETRY:
      if (value < 0 || value > 5) return fail(RMD_HIP_ERR_INVALID_ARG, "geometry %d outside 0..5", value);
      d->opt_geometry = value;
      return RMD_HIP_OK;
    default: return fail(RMD_HIP_ERR_INVALID_ARG, "denoiser_set_option: unknown option %d", option);
  }
}

int rmd_hip_denoiser_denoise(rmd_hip_denoiser_t* d, const rmd_hip_image_t* mu, const rmd_hip_image_t* sigma_sq,
                             const rmd_hip_image_t* a, const rmd_hip_image_t* b, float* host_denoised, float lambda,
                             int iterations) {
  if (!d || !mu || !sigma_sq || !a || !b) return fail(RMD_HIP_ERR_INVALID_ARG, "denoise: null argument");
  if (iterations < 0) return fail(RMD_HIP_ERR_INVALID_ARG, "denoise: negative iteration count");
  if (d->large_sigma_sq < 0.0f) return fail(RMD_HIP_ERR_NOT_READY, "denoise: setLargeSigmaSq must be called before this method");
  const rmd_hip_image_t* ins[4] = {mu, sigma_sq, a, b};
  for (auto* im : ins) {
    if (im->kind != RMD_HIP_KIND_F32 || im->width != d->width || im->height != d->height || im->stride != mu->stride)
      return fail(RMD_HIP_ERR_INVALID_ARG, "denoise: input planes must be f32 %dx%d with one common stride", d->width, d->height);
  }
  int cur = -1;
  HIP_TRY(hipGetDevice(&cur));
  if (cur != d->device) HIP_TRY(hipSetDevice(d->device));
  // inputs may still be written by their owner (the SeedMatrix's last kernel is left in flight)
  for (auto* im : ins)
    if (im->owner_seeds || (im->owner_stream && im->owner_stream != d->stream)) TRY(image_settle(im));
  HIP_TRY(hipStreamSynchronize(d->stream));
  d->timer.drain();
  d->timer.reset();

  rmdk::TvParams P;
  P.w = d->width; P.h = d->height;
  P.stride = static_cast<int>(d->g.stride);
  P.stride2 = static_cast<int>(d->p[0].stride);
  P.mu = static_cast<const float*>(mu->data);
  P.sigma_sq = static_cast<const float*>(sigma_sq->data);
  P.a = static_cast<const float*>(a->data);
  P.b = static_cast<const float*>(b->data);
  P.in_stride = static_cast<int>(mu->stride);
  P.g = static_cast<float*>(d->g.data);
  P.large_sigma_sq = d->large_sigma_sq;
  P.tau = d->tau; P.sigma = d->sigma; P.theta = d->theta; P.lambda = lambda;
  P.members = nullptr; P.member_stride = 0; P.member_stride2 = 0;

  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (d->opt_timing) {  // one event pair around the whole iteration loop (per-launch markers would serialise it)
    if (!d->ev0) HIP_TRY(hipEventCreate(&d->ev0));
    if (!d->ev1) HIP_TRY(hipEventCreate(&d->ev1));
    ev0 = d->ev0; ev1 = d->ev1;
  }
  int cur_buf = 0;
  long n_launches = 0;
  float* us[2] = {static_cast<float*>(d->u[0].data), static_cast<float*>(d->u[1].data)};
  float* uhs[2] = {static_cast<float*>(d->u_head[0].data), static_cast<float*>(d->u_head[1].data)};
  float2* ps[2] = {static_cast<float2*>(d->p[0].data), static_cast<float2*>(d->p[1].data)};
  TRY(tv_run(P, us, uhs, ps, 1, iterations, d->opt_iters_per_launch, d->opt_geometry, d->stream, ev0, &cur_buf, &n_launches));
  HIP_TRY(hipGetLastError());
  if (d->opt_timing) HIP_TRY(hipEventRecord(ev1, d->stream));
  d->result_index = cur_buf;
  const size_t out_bytes = static_cast<size_t>(d->width) * d->height * 4;
  if (host_denoised) {
    const rmd_hip_image& r = d->u[cur_buf];
    const size_t row = static_cast<size_t>(r.width) * 4;
    HIP_TRY(hipMemcpy2DAsync(d->h_staging, row, r.data, r.pitch, row, r.height, hipMemcpyDeviceToHost, d->stream));
  }
  HIP_TRY(hipStreamSynchronize(d->stream));
  if (host_denoised) memcpy(host_denoised, d->h_staging, out_bytes);
  if (d->opt_timing) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) { d->timer.total_ms = ms; d->timer.launches = n_launches; }
  }
  return RMD_HIP_OK;
}

int rmd_hip_denoiser_result(const rmd_hip_denoiser_t* d, const rmd_hip_image_t** view) {
  if (!d || !view) return fail(RMD_HIP_ERR_INVALID_ARG, "denoiser_result: null argument");
  *view = &d->u[d->result_index];
  return RMD_HIP_OK;
}

int rmd_hip_denoiser_timing(const rmd_hip_denoiser_t* d, double* total_ms, long* launches) {
  if (!d) return fail(RMD_HIP_ERR_INVALID_ARG, "denoiser_timing: null handle");
  if (total_ms) *total_ms = d->timer.total_ms;
  if (launches) *launches = d->timer.launches;
  return RMD_HIP_OK;
}

}  // extern "C"

