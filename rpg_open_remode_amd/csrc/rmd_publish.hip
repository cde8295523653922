// librmd_hip.so -- the two steps AFTER the path that the reference runs on the host over downloaded images: the
// CONVERGED-masked world-frame point cloud (Publisher::publishPointCloud, src/publisher.cpp:54-104) and the coloured convergence map
// (Publisher::publishConvergenceMap, src/publisher.cpp:112-147), with their kernels.
#include "rmd_host.hpp"

using namespace rmdh;

namespace rmdk {  // (kernels of this unit only: each kernel of the library has ONE home translation unit)

// ------------------------------------------------------------------------------------------
// CONVERGED-masked back-projection to a world-frame XYZI point cloud: src/publisher.cpp:54-104 (a host loop over two
// downloaded images in the reference).  Per pixel (x, y), row-major:  f = normalize(((x-cx)/fx, (y-cy)/fy, 1));
// xyz = T_world_ref * (f * depth(x, y));  kept iff convergence(x, y) == CONVERGED;  intensity = 8-bit reference image.
// Three launches keep the reference's point order (order-preserving compaction): per-block counts, one-block exclusive
// scan, write at block offset + rank.  Only the points cross the bus afterwards, not two W x H images.
struct PointCloudParams {
  int w, h;
  int stride;        // conv, ref planes
  int depth_stride;  // depth plane (the SeedMatrix's mu or a denoiser's output)
  const float* depth;
  const int* conv;
  const float* ref;  // reference image as uploaded: u8 * (1/255)
  Cam cam;
  Pose T_world_ref;
};
constexpr int PC_BLOCK = 256;

RMDK_D bool pc_pixel(const PointCloudParams& P, int i, int& x, int& y) {
  if (i >= P.w * P.h) return false;
  y = i / P.w;
  x = i - y * P.w;
  return P.conv[static_cast<size_t>(y) * P.stride + x] == ST_CONVERGED;
}

static __global__ __launch_bounds__(PC_BLOCK) void pc_count_kernel(PointCloudParams P, unsigned int* __restrict__ block_counts) {
  __shared__ unsigned int wave_part[PC_BLOCK / 64];
  int x, y;
  const bool keep = pc_pixel(P, blockIdx.x * PC_BLOCK + threadIdx.x, x, y);
  const unsigned int n = static_cast<unsigned int>(__popcll(__ballot(keep)));
  if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_part[0] + wave_part[1] + wave_part[2] + wave_part[3];
}

// in place: counts -> exclusive offsets; total to *total
static __global__ __launch_bounds__(1024) void pc_scan_kernel(unsigned int* __restrict__ counts, int n, unsigned int* __restrict__ total) {
  __shared__ unsigned int wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (n + 1023) / 1024;
  const int first = tid * per;
  unsigned int mine = 0;
  for (int q = 0; q < per; ++q)
    if (first + q < n) mine += counts[first + q];
  unsigned int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  unsigned int base = 0, all = 0;
  for (int wv = 0; wv < 16; ++wv) {
    base += wv < wave ? wave_tot[wv] : 0u;
    all += wave_tot[wv];
  }
  base += incl - mine;
  for (int q = 0; q < per; ++q) {
    if (first + q < n) {
      const unsigned int c = counts[first + q];
      counts[first + q] = base;
      base += c;
    }
  }
  if (tid == 0) *total = all;
}

static __global__ __launch_bounds__(PC_BLOCK) void pc_write_kernel(PointCloudParams P, const unsigned int* __restrict__ block_offsets,
                                                            float4* __restrict__ out, unsigned int capacity) {
  __shared__ unsigned int wave_part[PC_BLOCK / 64];
  int x = 0, y = 0;
  const bool keep = pc_pixel(P, blockIdx.x * PC_BLOCK + threadIdx.x, x, y);
  const unsigned long long mask = __ballot(keep);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_part[wave] = static_cast<unsigned int>(__popcll(mask));
  __syncthreads();
  if (!keep) return;
  unsigned int rank = static_cast<unsigned int>(__popcll(mask & ((1ull << lane) - 1ull)));
  for (int wv = 0; wv < wave; ++wv) rank += wave_part[wv];
  const unsigned int idx = block_offsets[blockIdx.x] + rank;
  if (idx >= capacity) return;
  const F3 f = normalize3(cam2world(P.cam, static_cast<float>(x), static_cast<float>(y)));
  const F3 xyz = pose_apply(P.T_world_ref, scale3(f, P.depth[static_cast<size_t>(y) * P.depth_stride + x]));
  // the 8-bit intensity back from the float image the path works on: rint(v * 255) is exact for v = k * (1/255)f
  const float intensity = rintf(P.ref[static_cast<size_t>(y) * P.stride + x] * 255.0f);
  out[idx] = make_float4(xyz.x, xyz.y, xyz.z, intensity);
}

// ------------------------------------------------------------------------------------------
// The coloured convergence map of Publisher::publishConvergenceMap (src/publisher.cpp:112-147): cv::cvtColor(ref_img, GRAY2BGR), then
// channel 0 (blue) = 255 where the seed has CONVERGED, channel 2 (red) = 255 where it has DIVERGED -- a host loop over two images in the
// reference, after a W x H x int32 download.  Here: 1 + 4 bytes in, 3 bytes out per pixel, and only the 3 cross the bus.  The 8-bit
// reference image comes back from the float plane the path works on (rint(v * 255) is exact for v = k * (1/255)f, like pc_write_kernel).
// Four pixels per lane: twelve output bytes = three dwords (rows of the packed W x 3 output need not be dword-aligned: the output is
// addressed as ONE array of W * H * 3 bytes, groups of four pixels counted over the whole image, the last group may be short).
static __global__ __launch_bounds__(256) void convergence_bgr8_kernel(const float* __restrict__ ref, const int* __restrict__ conv, int w,
    int h, int stride,
                                                               unsigned char* __restrict__ out) {
  const long long n = static_cast<long long>(w) * h;
  const long long p0 = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) * 4;
  if (p0 >= n) return;
  unsigned int b[12];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const long long p = p0 + q < n ? p0 + q : n - 1;
    const int y = static_cast<int>(p / w), x = static_cast<int>(p - static_cast<long long>(y) * w);
    const size_t i = static_cast<size_t>(y) * stride + x;
    // saturates like cv::saturate_cast<uchar>
    const unsigned int g = static_cast<unsigned int>(fminf(fmaxf(rintf(ref[i] * 255.0f), 0.0f), 255.0f));
    const int st = conv[i];
    b[3 * q] = st == ST_CONVERGED ? 255u : g;
    b[3 * q + 1] = g;
    b[3 * q + 2] = st == ST_DIVERGED ? 255u : g;
  }
  if (p0 + 3 < n) {
    unsigned int* o = reinterpret_cast<unsigned int*>(out + p0 * 3);  // p0 * 3 is a multiple of 12
    o[0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
    o[1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
    o[2] = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
  } else {
    for (long long k = 0; k < (n - p0) * 3; ++k) out[p0 * 3 + k] = static_cast<unsigned char>(b[k]);
  }
}

// ------------------------------------------------------------------------------------------ The snapshot a publication works on
// (rmd_hip_seeds_publish_async): up to six planes of one pitch copied as flat arrays of 16-byte vectors -- the planes' rows are padded to
// 256 bytes, padding travels along.  One launch on the update stream: 24 + 24 bytes per pixel at most.
typedef unsigned int snap_vec_t __attribute__((ext_vector_type(4)));
struct SnapshotArgs {
  const snap_vec_t* src[6];
  snap_vec_t* dst[6];
  int n_planes;
  unsigned int n_vec;  // 16-byte vectors per plane
};
static __global__ __launch_bounds__(256) void snapshot_kernel(SnapshotArgs A) {
  // (indexed in place: no copy of the pointer arrays into scratch)
  const SnapshotArgs* const a = (const SnapshotArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  (void)A;
  const unsigned int step = gridDim.x * blockDim.x;
  for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < a->n_vec; i += step) {
    for (int q = 0; q < a->n_planes; ++q) a->dst[q][i] = __builtin_nontemporal_load(a->src[q] + i);
  }
}

}  // namespace rmdk

extern "C" {

int rmd_hip_seeds_point_cloud(rmd_hip_seeds_t* s, const rmd_hip_image_t* depth, float* out_xyzi, size_t capacity, size_t* n_points) {
  if (!s || !n_points || (!out_xyzi && capacity)) return fail(RMD_HIP_ERR_INVALID_ARG, "point_cloud: null argument");
  if (!s->has_reference) return fail(RMD_HIP_ERR_NOT_READY, "point_cloud: no reference image set");
  if (depth && (depth->kind != RMD_HIP_KIND_F32 || depth->width != s->width || depth->height != s->height))
    return fail(RMD_HIP_ERR_INVALID_ARG, "point_cloud: depth must be an f32 %dx%d image", s->width, s->height);
  TRY(seeds_bind_device(s));
  TRY(seeds_flush(s));
  if (depth && !(depth->owner_seeds == s) && (depth->owner_seeds || (depth->owner_stream && depth->owner_stream != s->stream)))
    TRY(image_settle(depth));
  const int n_pix = s->width * s->height;
  const int n_blocks = (n_pix + rmdk::PC_BLOCK - 1) / rmdk::PC_BLOCK;
  // The points go straight into pinned host memory (posted writes over the host link, 16 bytes per lane) and from there into the caller's
  // buffer: their number is not known when the work is queued, and a device-to-host copy into PAGEABLE memory goes through the runtime's
  // own staging (measured at 0.3 GB/s on one box of the pool: 3.5 ms for a cloud of 60 000 points).
  if (!s->d_pc_counts) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_pc_counts), (static_cast<size_t>(n_blocks) + 1) * sizeof(unsigned int)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_pc_points), static_cast<size_t>(n_pix) * sizeof(float4), hipHostMallocMapped));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_pc_total), 64, hipHostMallocMapped));
  }
  void *d_points = nullptr, *d_total = nullptr;
  HIP_TRY(hipHostGetDevicePointer(&d_points, s->h_pc_points, 0));
  HIP_TRY(hipHostGetDevicePointer(&d_total, s->h_pc_total, 0));
  rmdk::PointCloudParams P;
  P.w = s->width; P.h = s->height;
  P.stride = s->P.stride;
  P.depth = depth ? static_cast<const float*>(depth->data) : s->P.mu;
  P.depth_stride = depth ? static_cast<int>(depth->stride) : s->P.stride;
  P.conv = s->P.conv;
  P.ref = s->P.ref;
  P.cam = s->P.cam;
  P.T_world_ref = s->T_world_ref;
  hipLaunchKernelGGL(rmdk::pc_count_kernel, dim3(n_blocks), dim3(rmdk::PC_BLOCK), 0, s->stream, P, s->d_pc_counts);
  hipLaunchKernelGGL(rmdk::pc_scan_kernel, dim3(1), dim3(1024), 0, s->stream, s->d_pc_counts, n_blocks,
      static_cast<unsigned int*>(d_total));
  hipLaunchKernelGGL(rmdk::pc_write_kernel, dim3(n_blocks), dim3(rmdk::PC_BLOCK), 0, s->stream, P, s->d_pc_counts,
      static_cast<float4*>(d_points),
                     static_cast<unsigned int>(n_pix));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s->stream));
  const size_t total = *static_cast<volatile unsigned int*>(s->h_pc_total);
  *n_points = total;
  const size_t n_copy = total < capacity ? total : capacity;
  if (n_copy) memcpy(out_xyzi, s->h_pc_points, n_copy * sizeof(float4));
  return RMD_HIP_OK;
}

int rmd_hip_seeds_convergence_bgr8(rmd_hip_seeds_t* s, unsigned char* host_bgr) {
  if (!s || !host_bgr) return fail(RMD_HIP_ERR_INVALID_ARG, "convergence_bgr8: null argument");
  if (!s->has_reference) return fail(RMD_HIP_ERR_NOT_READY, "convergence_bgr8: no reference image set");
  TRY(seeds_bind_device(s));
  // No flush of the deferred finalisation: it only ever turns UPDATE into NO_MATCH, and neither has a colour (publisher.cpp:124-134);
  // CONVERGED / DIVERGED were settled by the update's seed_check.  Stream order puts the kernel behind the update.
  const size_t bytes = static_cast<size_t>(s->width) * s->height * 3;
  if (!s->d_bgr) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_bgr), (bytes + 15) & ~static_cast<size_t>(15)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_bgr), bytes));
  }
  const long long groups = (static_cast<long long>(s->width) * s->height + 3) / 4;
  hipLaunchKernelGGL(rmdk::convergence_bgr8_kernel, dim3(static_cast<unsigned int>((groups + 255) / 256)), dim3(256), 0, s->stream,
      s->P.ref, s->P.conv, s->width,
                     s->height, s->P.stride, s->d_bgr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(s->h_bgr, s->d_bgr, bytes, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  TRY(ingest_error_check(s->batch ? s->batch->group_of(s->batch_index).h_progress : s->h_progress));
  memcpy(host_bgr, s->h_bgr, bytes);
  return RMD_HIP_OK;
}

// ---- publication off the update stream (include/rmd_hip.h: rmd_hip_seeds_publish_async / _collect) -------------------------------------
int rmd_hip_seeds_publish_async(rmd_hip_seeds_t* s, unsigned int what, float depth_range, float lambda, int iterations, int* ticket) {
  if (!s) return fail(RMD_HIP_ERR_INVALID_ARG, "publish_async: null handle");
  const unsigned int known = RMD_HIP_PUBLISH_DEPTH | RMD_HIP_PUBLISH_CLOUD | RMD_HIP_PUBLISH_CONVERGENCE_BGR | RMD_HIP_PUBLISH_CONVERGENCE;
  if (what == 0u || (what & ~known)) return fail(RMD_HIP_ERR_INVALID_ARG, "publish_async: unknown products 0x%x", what);
  if (what & RMD_HIP_PUBLISH_CLOUD) what |= RMD_HIP_PUBLISH_DEPTH;
  if ((what & RMD_HIP_PUBLISH_DEPTH) && iterations < 0) return fail(RMD_HIP_ERR_INVALID_ARG, "publish_async: negative iteration count");
  if (!s->has_reference) return fail(RMD_HIP_ERR_NOT_READY, "publish_async: no reference image set");
  if (s->pub_pending >= RMD_HIP_PUBLISH_SLOTS)
    return fail(RMD_HIP_ERR_NOT_READY, "publish_async: %d publications are waiting to be collected (rmd_hip_seeds_publish_collect)",
        s->pub_pending);
  TRY(seeds_bind_device(s));
  const int w = s->width, h = s->height;
  const size_t n_pix = static_cast<size_t>(w) * h;
  if (!s->pub_stream) {
    // its own priority level: never on the hardware queue of the update or the copy stream (create_stream)
    HIP_TRY(create_stream(&s->pub_stream, 1));
    rmd_hip_image* f32[] = {&s->pub_u[0], &s->pub_u[1], &s->pub_u_head[0], &s->pub_u_head[1], &s->pub_g};
    for (auto* im : f32) TRY(image_alloc(im, RMD_HIP_KIND_F32, w, h));
    for (int k = 0; k < 2; ++k) TRY(image_alloc(&s->pub_p[k], RMD_HIP_KIND_F32X2, w, h));
    const int n_blocks = static_cast<int>((n_pix + rmdk::PC_BLOCK - 1) / rmdk::PC_BLOCK);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->pub_pc_counts), (static_cast<size_t>(n_blocks) + 1) * sizeof(unsigned int)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->pub_d_bgr), (n_pix * 3 + 15) & ~static_cast<size_t>(15)));
  }
  const int slot = (s->pub_oldest + s->pub_pending) % RMD_HIP_PUBLISH_SLOTS;
  if (!s->pub[slot]) {
    rmd_hip_seeds::Publication* np = new (std::nothrow) rmd_hip_seeds::Publication();
    if (!np) return fail(RMD_HIP_ERR_RUNTIME, "publish_async: out of host memory");
    s->pub[slot] = np;  // (whatever a failure below leaves half-built is released with the handle)
    rmd_hip_image* f32[] = {&np->mu, &np->sigma_sq, &np->a, &np->b, &np->ref};
    for (auto* im : f32) TRY(image_alloc(im, RMD_HIP_KIND_F32, w, h));
    TRY(image_alloc(&np->conv, RMD_HIP_KIND_I32, w, h));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&np->h_depth), n_pix * sizeof(float)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&np->h_points), n_pix * sizeof(float4), hipHostMallocMapped));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&np->h_total), 64, hipHostMallocMapped));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&np->h_bgr), n_pix * 3));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&np->h_conv), n_pix * sizeof(int)));
    HIP_TRY(hipEventCreateWithFlags(&np->snapped, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&np->done, hipEventDisableTiming));
  }
  rmd_hip_seeds::Publication& pb = *s->pub[slot];
  const bool want_depth = (what & RMD_HIP_PUBLISH_DEPTH) != 0u, want_cloud = (what & RMD_HIP_PUBLISH_CLOUD) != 0u;
  const bool want_bgr = (what & RMD_HIP_PUBLISH_CONVERGENCE_BGR) != 0u, want_conv = (what & RMD_HIP_PUBLISH_CONVERGENCE) != 0u;
  // the state must be final where the products read it: mu, sigma_sq, a, b and the NO_MATCH states come from the deferred finalisation; the
  // colours do not (CONVERGED / DIVERGED were settled by the update's seed_check: rmd_hip_seeds_convergence_bgr8)
  if (want_depth || want_conv) TRY(seeds_flush(s));
  rmdk::SnapshotArgs A;
  memset(&A, 0, sizeof(A));
  auto add = [&](const void* src, rmd_hip_image& dst) { A.src[A.n_planes] = static_cast<const rmdk::snap_vec_t*>(src); A.dst[A.n_planes] =
      static_cast<rmdk::snap_vec_t*>(dst.data); ++A.n_planes; };
  if (want_depth) { add(s->P.mu, pb.mu); add(s->P.sigma_sq, pb.sigma_sq); add(s->P.a, pb.a); add(s->P.b, pb.b); }
  if (want_cloud || want_bgr || want_conv) add(s->P.conv, pb.conv);
  if (want_cloud || want_bgr) add(s->P.ref, pb.ref);
  A.n_vec = static_cast<unsigned int>(pb.mu.pitch * static_cast<size_t>(h) / 16);  // (every f32 / i32 plane of the handle has this pitch)
  hipLaunchKernelGGL(rmdk::snapshot_kernel, dim3((A.n_vec + 255) / 256 < 2048u ? (A.n_vec + 255) / 256 : 2048u), dim3(256), 0, s->stream,
      A);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(pb.snapped, s->stream));
  pb.T_world_ref = s->T_world_ref;
  HIP_TRY(hipStreamWaitEvent(s->pub_stream, pb.snapped, 0));
  const size_t row = static_cast<size_t>(w) * 4;
  const rmd_hip_image* depth_img = nullptr;
  if (want_depth) {
    rmdk::TvParams P;
    P.w = w; P.h = h;
    P.stride = static_cast<int>(s->pub_g.stride);
    P.stride2 = static_cast<int>(s->pub_p[0].stride);
    P.mu = static_cast<const float*>(pb.mu.data); P.sigma_sq = static_cast<const float*>(pb.sigma_sq.data);
    P.a = static_cast<const float*>(pb.a.data); P.b = static_cast<const float*>(pb.b.data);
    P.in_stride = static_cast<int>(pb.mu.stride);
    P.g = static_cast<float*>(s->pub_g.data);
    P.large_sigma_sq = depth_range * depth_range / 72.0f;  // DepthmapDenoiser::setLargeSigmaSq, depthmap_denoiser.cu:226-229
    const float L = sqrtf(8.0f);                          // denoise::DeviceData constructor, depthmap_denoiser.cu:124-141
    P.tau = 0.02f; P.sigma = (1 / (L * L)) / P.tau; P.theta = 0.5f; P.lambda = lambda;
    P.members = nullptr; P.member_stride = 0; P.member_stride2 = 0;
    float* us[2] = {static_cast<float*>(s->pub_u[0].data), static_cast<float*>(s->pub_u[1].data)};
    float* uhs[2] = {static_cast<float*>(s->pub_u_head[0].data), static_cast<float*>(s->pub_u_head[1].data)};
    float2* ps[2] = {static_cast<float2*>(s->pub_p[0].data), static_cast<float2*>(s->pub_p[1].data)};
    int cur_buf = 0;
    long n_launches = 0;
    const bool same = pb.tv_exec && pb.tv_range == depth_range && pb.tv_lambda == lambda && pb.tv_iterations == iterations;
    if (s->pub_graphs && !same) {  // (re)capture this slot's launch sequence
      if (pb.tv_exec) { (void)hipGraphExecDestroy(pb.tv_exec); pb.tv_exec = nullptr; }
      hipGraph_t graph = nullptr;
      bool ok = hipStreamBeginCapture(s->pub_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
      if (ok) {
        const int rc = tv_run(P, us, uhs, ps, 1, iterations, 0, 0, s->pub_stream, nullptr, &cur_buf, &n_launches);
        ok = hipStreamEndCapture(s->pub_stream, &graph) == hipSuccess && rc == RMD_HIP_OK && graph != nullptr;
      }
      if (ok) ok = hipGraphInstantiate(&pb.tv_exec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (graph) (void)hipGraphDestroy(graph);
      if (ok) { pb.tv_range = depth_range; pb.tv_lambda = lambda; pb.tv_iterations = iterations; pb.tv_result = cur_buf; }
      else { pb.tv_exec = nullptr; s->pub_graphs = false; (void)hipGetLastError(); }
    }
    if (pb.tv_exec) {
      HIP_TRY(hipGraphLaunch(pb.tv_exec, s->pub_stream));
      cur_buf = pb.tv_result;
    } else {
      TRY(tv_run(P, us, uhs, ps, 1, iterations, 0, 0, s->pub_stream, nullptr, &cur_buf, &n_launches));
    }
    depth_img = &s->pub_u[cur_buf];
    HIP_TRY(hipMemcpy2DAsync(pb.h_depth, row, depth_img->data, depth_img->pitch, row, h, hipMemcpyDeviceToHost, s->pub_stream));
  }
  if (want_cloud) {
    rmdk::PointCloudParams P;
    P.w = w; P.h = h;
    P.stride = static_cast<int>(pb.conv.stride);
    P.depth = static_cast<const float*>(depth_img->data);
    P.depth_stride = static_cast<int>(depth_img->stride);
    P.conv = static_cast<const int*>(pb.conv.data);
    P.ref = static_cast<const float*>(pb.ref.data);
    P.cam = s->P.cam;
    P.T_world_ref = pb.T_world_ref;
    const int n_blocks = static_cast<int>((n_pix + rmdk::PC_BLOCK - 1) / rmdk::PC_BLOCK);
    void *d_points = nullptr, *d_total = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&d_points, pb.h_points, 0));
    HIP_TRY(hipHostGetDevicePointer(&d_total, pb.h_total, 0));
    hipLaunchKernelGGL(rmdk::pc_count_kernel, dim3(n_blocks), dim3(rmdk::PC_BLOCK), 0, s->pub_stream, P, s->pub_pc_counts);
    hipLaunchKernelGGL(rmdk::pc_scan_kernel, dim3(1), dim3(1024), 0, s->pub_stream, s->pub_pc_counts, n_blocks,
        static_cast<unsigned int*>(d_total));
    // the points go straight into pinned host memory (posted writes over the host link, 16 bytes per lane): their number is not known when
    // the transfers are queued, and a copy of all W x H slots would move the unconverged ones too
    hipLaunchKernelGGL(rmdk::pc_write_kernel, dim3(n_blocks), dim3(rmdk::PC_BLOCK), 0, s->pub_stream, P, s->pub_pc_counts,
        static_cast<float4*>(d_points),
                       static_cast<unsigned int>(n_pix));
    HIP_TRY(hipGetLastError());
  }
  if (want_bgr) {
    const long long groups = (static_cast<long long>(n_pix) + 3) / 4;
    hipLaunchKernelGGL(rmdk::convergence_bgr8_kernel, dim3(static_cast<unsigned int>((groups + 255) / 256)), dim3(256), 0, s->pub_stream,
                       static_cast<const float*>(pb.ref.data), static_cast<const int*>(pb.conv.data), w, h,
                           static_cast<int>(pb.conv.stride), s->pub_d_bgr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(pb.h_bgr, s->pub_d_bgr, n_pix * 3, hipMemcpyDeviceToHost, s->pub_stream));
  }
  if (want_conv) HIP_TRY(hipMemcpy2DAsync(pb.h_conv, row, pb.conv.data, pb.conv.pitch, row, h, hipMemcpyDeviceToHost, s->pub_stream));
  HIP_TRY(hipEventRecord(pb.done, s->pub_stream));
  pb.what = what;
  pb.ticket = ++s->pub_tickets;
  pb.pending = true;
  ++s->pub_pending;
  if (ticket) *ticket = pb.ticket;
  return RMD_HIP_OK;
}

int rmd_hip_seeds_publish_peek(rmd_hip_seeds_t* s, int wait, unsigned int* what, int* ticket, const float** depth, const float** xyzi,
    size_t* n_points,
                               const unsigned char** bgr, const int** convergence) {
  if (!s) return fail(RMD_HIP_ERR_INVALID_ARG, "publish_peek: null handle");
  if (s->pub_pending == 0) return fail(RMD_HIP_ERR_NOT_READY, "publish_peek: no publication in flight");
  TRY(seeds_bind_device(s));
  rmd_hip_seeds::Publication& pb = *s->pub[s->pub_oldest];
  if (wait) HIP_TRY(hipEventSynchronize(pb.done));
  else {
    const hipError_t e = hipEventQuery(pb.done);
    if (e == hipErrorNotReady) return RMD_HIP_BUSY;
    HIP_TRY(e);
  }
  if (what) *what = pb.what;
  if (ticket) *ticket = pb.ticket;
  if (depth) *depth = (pb.what & RMD_HIP_PUBLISH_DEPTH) ? pb.h_depth : nullptr;
  if (xyzi) *xyzi = (pb.what & RMD_HIP_PUBLISH_CLOUD) ? reinterpret_cast<const float*>(pb.h_points) : nullptr;
  if (n_points) *n_points = (pb.what & RMD_HIP_PUBLISH_CLOUD) ? *static_cast<volatile unsigned int*>(pb.h_total) : 0u;
  if (bgr) *bgr = (pb.what & RMD_HIP_PUBLISH_CONVERGENCE_BGR) ? pb.h_bgr : nullptr;
  if (convergence) *convergence = (pb.what & RMD_HIP_PUBLISH_CONVERGENCE) ? pb.h_conv : nullptr;
  return RMD_HIP_OK;
}

int rmd_hip_seeds_publish_release(rmd_hip_seeds_t* s) {
  if (!s) return fail(RMD_HIP_ERR_INVALID_ARG, "publish_release: null handle");
  if (s->pub_pending == 0) return fail(RMD_HIP_ERR_NOT_READY, "publish_release: no publication in flight");
  TRY(seeds_bind_device(s));
  rmd_hip_seeds::Publication& pb = *s->pub[s->pub_oldest];
  HIP_TRY(hipEventSynchronize(pb.done));  // (a slot is never handed back while the device still writes it)
  pb.pending = false;
  s->pub_oldest = (s->pub_oldest + 1) % RMD_HIP_PUBLISH_SLOTS;
  --s->pub_pending;
  return RMD_HIP_OK;
}

int rmd_hip_seeds_publish_collect(rmd_hip_seeds_t* s, int wait, unsigned int* what, int* ticket, float* host_depth, float* host_xyzi,
    size_t capacity,
                                  size_t* n_points, unsigned char* host_bgr, int* host_convergence) {
  const float *depth = nullptr, *xyzi = nullptr;
  const unsigned char* bgr = nullptr;
  const int* conv = nullptr;
  size_t total = 0;
  const int rc = rmd_hip_seeds_publish_peek(s, wait, what, ticket, &depth, &xyzi, &total, &bgr, &conv);
  if (rc != RMD_HIP_OK) return rc;
  const size_t n_pix = static_cast<size_t>(s->width) * s->height;
  if (depth && host_depth) memcpy(host_depth, depth, n_pix * sizeof(float));
  const size_t n_copy = total < capacity ? total : capacity;
  if (xyzi && host_xyzi && n_copy) memcpy(host_xyzi, xyzi, n_copy * sizeof(float4));
  if (n_points) *n_points = total;
  if (bgr && host_bgr) memcpy(host_bgr, bgr, n_pix * 3);
  if (conv && host_convergence) memcpy(host_convergence, conv, n_pix * sizeof(int));
  return rmd_hip_seeds_publish_release(s);
}

}  // extern "C"

namespace rmdh {

void publish_release(rmd_hip_seeds* s) {
  if (s->pub_stream) (void)hipStreamSynchronize(s->pub_stream);
  for (auto*& pb : s->pub) {
    if (!pb) continue;
    rmd_hip_image* planes[] = {&pb->mu, &pb->sigma_sq, &pb->a, &pb->b, &pb->conv, &pb->ref};
    for (auto* im : planes)
      if (im->owns && im->data) (void)hipFree(im->data);
    void* pinned[] = {pb->h_depth, pb->h_points, pb->h_total, pb->h_bgr, pb->h_conv};
    for (void* q : pinned)
      if (q) (void)hipHostFree(q);
    if (pb->tv_exec) (void)hipGraphExecDestroy(pb->tv_exec);
    if (pb->snapped) (void)hipEventDestroy(pb->snapped);
    if (pb->done) (void)hipEventDestroy(pb->done);
    delete pb;
    pb = nullptr;
  }
  rmd_hip_image* ws[] = {&s->pub_u[0], &s->pub_u[1], &s->pub_u_head[0], &s->pub_u_head[1], &s->pub_p[0], &s->pub_p[1], &s->pub_g};
  for (auto* im : ws)
    if (im->owns && im->data) { (void)hipFree(im->data); im->data = nullptr; }
  if (s->pub_pc_counts) (void)hipFree(s->pub_pc_counts);
  if (s->pub_d_bgr) (void)hipFree(s->pub_d_bgr);
  if (s->pub_stream) (void)hipStreamDestroy(s->pub_stream);
  s->pub_pc_counts = nullptr; s->pub_d_bgr = nullptr; s->pub_stream = nullptr;
  s->pub_pending = 0;
}

}  // namespace rmdh
