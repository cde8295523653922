// librmd_hip.so -- the two steps AFTER the path that the reference runs on the host over downloaded images: the CONVERGED-masked
// world-frame point cloud (Publisher::publishPointCloud, src/publisher.cpp:54-104) and the coloured convergence map
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
static __global__ __launch_bounds__(256) void convergence_bgr8_kernel(const float* __restrict__ ref, const int* __restrict__ conv, int w, int h, int stride,
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
    const unsigned int g = static_cast<unsigned int>(fminf(fmaxf(rintf(ref[i] * 255.0f), 0.0f), 255.0f));  // saturates like cv::saturate_cast<uchar>
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
  if (!s->d_pc_counts) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_pc_counts), (static_cast<size_t>(n_blocks) + 1) * sizeof(unsigned int)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_pc_points), static_cast<size_t>(n_pix) * sizeof(float4)));
  }
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
  hipLaunchKernelGGL(rmdk::pc_scan_kernel, dim3(1), dim3(1024), 0, s->stream, s->d_pc_counts, n_blocks, s->d_pc_counts + n_blocks);
  hipLaunchKernelGGL(rmdk::pc_write_kernel, dim3(n_blocks), dim3(rmdk::PC_BLOCK), 0, s->stream, P, s->d_pc_counts, s->d_pc_points,
                     static_cast<unsigned int>(n_pix));
  HIP_TRY(hipGetLastError());
  unsigned int total = 0;
  HIP_TRY(hipMemcpyAsync(&total, s->d_pc_counts + n_blocks, sizeof(total), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  *n_points = total;
  const size_t n_copy = total < capacity ? total : capacity;
  if (n_copy) HIP_TRY(hipMemcpy(out_xyzi, s->d_pc_points, n_copy * sizeof(float4), hipMemcpyDeviceToHost));
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
  hipLaunchKernelGGL(rmdk::convergence_bgr8_kernel, dim3(static_cast<unsigned int>((groups + 255) / 256)), dim3(256), 0, s->stream, s->P.ref, s->P.conv, s->width,
                     s->height, s->P.stride, s->d_bgr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(s->h_bgr, s->d_bgr, bytes, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  TRY(ingest_error_check(s->batch ? s->batch->group_of(s->batch_index).h_progress : s->h_progress));
  memcpy(host_bgr, s->h_bgr, bytes);
  return RMD_HIP_OK;
}

}  // extern "C"
