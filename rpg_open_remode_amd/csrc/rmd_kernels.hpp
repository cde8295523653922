// HIP kernels of the depth-filter path (gfx950 / CDNA4, wave64).
//
//   seed_init_kernel          <- src/seed_init.cu:27-61
//   seed_update_pixel_kernel  <- src/seed_check.cu:28-67 + src/epipolar_match.cu:37-140 +
//                                src/seed_update.cu:39-121, fused, one lane per pixel
//   (seed_update_tile_kernel, the production matcher, lives in rmd_matcher.hpp)
//   count_eq_kernel / sum_*   <- src/reduction_kernels.cu:57-159
//   tv_prepare_kernel         <- src/depthmap_denoiser.cu:45-59 + :215-217
//   tv_iterate_kernel         <- src/depthmap_denoiser.cu:61-118
//
// No textures, no __constant__ symbols, no process-global state: everything a kernel needs
// arrives as a by-value parameter block in kernarg SGPRs.
#ifndef RMD_KERNELS_HPP
#define RMD_KERNELS_HPP

#include <float.h>

#include "rmd_device.hpp"

namespace rmdk {

enum : int { ST_UPDATE = 0, ST_CONVERGED = 1, ST_BORDER = 2, ST_DIVERGED = 3, ST_NO_MATCH = 4, ST_NOT_VISIBLE = 5 };

// Parameter block of the seed kernels (the reference's mvs::DeviceData, mvs_device_data.cuh:46-106,
// minus the pointer-to-descriptor indirection).
struct SeedParams {
  int w, h;
  int stride;   // elements, shared by every f32/i32 plane of one SeedMatrix
  int stride2;  // float2 elements, epipolar_matches plane
  int cur_stride;  // elements, current image (may be a caller-owned device buffer)
  const float* ref;
  const float* cur;
  float* sum_templ;
  float* denom;
  float* mu;
  float* sigma_sq;
  float* a;
  float* b;
  int* conv;
  float2* match;
  Cam cam;
  float one_pix_angle;
  float avg_depth, depth_range, sigma_sq_max;
  float eta_inlier, eta_outlier, epsilon;
  float max_extent;
  Pose T_curr_ref;
  Pose T_ref_curr;
  unsigned long long* stats;  // [0] live seeds, [1] steps visited, [2] NCC evaluations; may be null
  unsigned long long* trace;  // diagnostics: in-kernel timeline probes (100 MHz wall clock) of the update pipeline; may be null
};

// ------------------------------------------------------------------------------------------
// seed_init.cu:27-61 -- NCC template statistics + prior.  Runs once per reference frame.
template <int SIDE>
__global__ __launch_bounds__(256) void seed_init_kernel(SeedParams P) {
  constexpr int OFFSET = -SIDE / 2;
  constexpr int AREA = SIDE * SIDE;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= P.w || y >= P.h) return;
  float sum_t = 0.0f, sum_t_sq = 0.0f;
#pragma unroll
  for (int py = 0; py < SIDE; ++py) {
    const int yy = clampi(y + OFFSET + py, 0, P.h - 1);
#pragma unroll
    for (int px = 0; px < SIDE; ++px) {
      const float t = P.ref[yy * P.stride + clampi(x + OFFSET + px, 0, P.w - 1)];
      sum_t += t;
      sum_t_sq += t * t;
    }
  }
  const int i = y * P.stride + x;
  P.sum_templ[i] = sum_t;
  P.denom[i] = static_cast<float>(static_cast<double>(AREA) * sum_t_sq - static_cast<double>(sum_t) * sum_t);
  P.mu[i] = P.avg_depth;
  P.sigma_sq[i] = P.sigma_sq_max;
  P.a[i] = 10.0f;
  P.b[i] = 10.0f;
}

// ------------------------------------------------------------------------------------------
// Frame ingest (reference: Depthmap::inputImage, src/depthmap.cpp:95-106 -- cv::Mat::convertTo(CV_32F, 1.0f/255.0f) on the
// host): 8-bit gray -> f32 plane on the device.  One fp32 multiply per pixel, identical bits to the host conversion.
// 4 pixels per lane: one 32-bit load, one 128-bit store.
static __global__ __launch_bounds__(256) void ingest_u8_kernel(const unsigned char* __restrict__ src, int src_pitch, float* __restrict__ dst,
                                                        int dst_stride, int w, int h) {
  const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x4 >= w || y >= h) return;
  const unsigned char* row = src + static_cast<size_t>(y) * src_pitch;
  float* out = dst + static_cast<size_t>(y) * dst_stride;
  if (x4 + 3 < w) {
    const unsigned int v = *reinterpret_cast<const unsigned int*>(row + x4);  // src_pitch and x4 are multiples of 4
    float4 f;
    f.x = static_cast<float>(v & 0xffu) * (1.0f / 255.0f);
    f.y = static_cast<float>((v >> 8) & 0xffu) * (1.0f / 255.0f);
    f.z = static_cast<float>((v >> 16) & 0xffu) * (1.0f / 255.0f);
    f.w = static_cast<float>(v >> 24) * (1.0f / 255.0f);
    *reinterpret_cast<float4*>(out + x4) = f;
  } else {
    for (int x = x4; x < w; ++x) out[x] = static_cast<float>(row[x]) * (1.0f / 255.0f);
  }
}

// The same with lens undistortion in front (Depthmap::inputImage with is_distorted_, depthmap.cpp:95-106): cv::remap of the
// 8-bit frame through the fixed-point maps of cv::initUndistortRectifyMap(..., CV_16SC2) with INTER_LINEAR and
// BORDER_CONSTANT 0, then x(1/255).  map1 = integer source position, map2 = 5-bit fractions (fy << 5 | fx); the bilinear
// weights are OpenCV's 15-bit table entries (32 - fy)(32 - fx) * 32 etc., the result (sum + 2^14) >> 15.  Integer arithmetic.
// one destination pixel; `src` may be read with any load flavour through LOAD(ptr)
template <typename LOAD>
RMDK_D float remap_u8_pixel(const unsigned char* src, int src_pitch, short2 m, int f, int w, int h, LOAD load) {
  const int fx = f & 31, fy = f >> 5;
  const int sx = m.x, sy = m.y;
  int v = 0;
  if (!(sx >= w || sx + 1 < 0 || sy >= h || sy + 1 < 0)) {
    const bool x0 = sx >= 0, x1 = sx + 1 < w, y0 = sy >= 0, y1 = sy + 1 < h;
    const unsigned char* r0 = src + static_cast<ptrdiff_t>(sy) * src_pitch;
    const unsigned char* r1 = r0 + src_pitch;
    const int v00 = (x0 && y0) ? load(r0 + sx) : 0, v01 = (x1 && y0) ? load(r0 + sx + 1) : 0;
    const int v10 = (x0 && y1) ? load(r1 + sx) : 0, v11 = (x1 && y1) ? load(r1 + sx + 1) : 0;
    const int sum = v00 * ((32 - fy) * (32 - fx) * 32) + v01 * ((32 - fy) * fx * 32) + v10 * (fy * (32 - fx) * 32) + v11 * (fy * fx * 32);
    v = (sum + (1 << 14)) >> 15;
  }
  return static_cast<float>(v) * (1.0f / 255.0f);
}

static __global__ __launch_bounds__(256) void ingest_u8_remap_kernel(const unsigned char* __restrict__ src, int src_pitch,
                                                              const short2* __restrict__ map1, const unsigned short* __restrict__ map2,
                                                              float* __restrict__ dst, int dst_stride, int w, int h) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const short2 m = map1[static_cast<size_t>(y) * w + x];
  const int f = map2[static_cast<size_t>(y) * w + x] & 1023;
  dst[static_cast<size_t>(y) * dst_stride + x] = remap_u8_pixel(src, src_pitch, m, f, w, h, [](const unsigned char* p) { return static_cast<int>(*p); });
}

// ------------------------------------------------------------------------------------------
// seed_check.cu:28-67 as a function of one pixel's state
RMDK_D int seed_check(const SeedParams& P, int x, int y, float sigma_sq, float a, float b, int side) {
  // the reference compares in size_t: width - SIDE - 1 wraps for tiny images
  const size_t x_hi = static_cast<size_t>(P.w) - side - 1, y_hi = static_cast<size_t>(P.h) - side - 1;
  if (static_cast<size_t>(x) > x_hi || static_cast<size_t>(y) > y_hi || x < side || y < side) return ST_BORDER;
  if ((a / (a + b)) > P.eta_inlier && sigma_sq < P.epsilon) return ST_CONVERGED;
  if ((a - 1) / (a + b - 2) < P.eta_outlier) return ST_DIVERGED;
  return ST_UPDATE;
}

// Epipolar segment of one seed in the current frame (epipolar_match.cu:59-75)
struct Segment {
  F2 mean;   // projection of mu
  F2 dir;    // unit direction (NaN for a zero-length segment)
  float half_length;
};
RMDK_D Segment epipolar_segment(const SeedParams& P, int x, int y, float mu, float sigma_sq) {
  const float sigma = sqrtf(sigma_sq);
  const F3 f_ref = normalize3(cam2world(P.cam, static_cast<float>(x), static_cast<float>(y)));
  Segment s;
  s.mean = world2cam(P.cam, pose_apply(P.T_curr_ref, scale3(f_ref, mu)));
  const float d_lo = mu - 3.0f * sigma;
  const F2 px_min = world2cam(P.cam, pose_apply(P.T_curr_ref, scale3(f_ref, d_lo > 0.01f ? d_lo : 0.01f)));
  const F2 px_max = world2cam(P.cam, pose_apply(P.T_curr_ref, scale3(f_ref, mu + (3.0f * sigma))));
  const F2 line = F2{px_max.x - px_min.x, px_max.y - px_min.y};
  s.dir = normalize2(line);
  const float len = norm2(line);
  s.half_length = 0.5f * (len < P.max_extent ? len : P.max_extent);
  return s;
}

// seed_update.cu:58-119 for one pixel whose state (after matching) is `state`, on values in registers.
// Returns 0: nothing changes, 1: (mu, sigma_sq, a, b) have been replaced by the posterior, 2: only b has changed (NO_MATCH).
#ifdef RMD_PROFILE_ROUNDS
RMDK_D unsigned long long prof_clock_k() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return t;
}
#define RMD_PROF_STAMP(k) do { if (prof) prof[k] = prof_clock_k(); } while (0)
#else
#define RMD_PROF_STAMP(k) do { } while (0)
#endif
// T_ref_curr: the pose of the frame the match was found in (P.T_ref_curr, or the previous frame's when that frame's finalisation runs
// fused into the next frame's setup kernel).
RMDK_D int seed_fuse_values(const SeedParams& P, const Pose& T_ref_curr, int x, int y, int state, float& mu, float& sigma_sq, float& a, float& b, F2 match,
                            unsigned long long* prof = nullptr) {
  (void)prof;
  if (state == ST_UPDATE) {
    const F3 f_ref = normalize3(cam2world(P.cam, static_cast<float>(x), static_cast<float>(y)));
    const F3 f_epi = normalize3(cam2world(P.cam, match.x, match.y));
    const F3 pt = triangulate(f_ref, f_epi, T_ref_curr);
    RMD_PROF_STAMP(0);
    if (pt.z < 0.0f) return 0;
    const float depth = norm3(pt);
    const float tau = triangulation_uncertainty(depth, f_ref, pose_translation(T_ref_curr), P.one_pix_angle);
    RMD_PROF_STAMP(1);
    const float tau_sq = tau * tau;
    const float s_sq = (tau_sq * sigma_sq) / (tau_sq + sigma_sq);
    const float m = s_sq * (mu / sigma_sq + depth / tau_sq);
    float c1 = (a / (a + b)) * normpdf(depth, mu, sigma_sq + tau_sq);
    RMD_PROF_STAMP(2);
    float c2 = (b / (a + b)) * (1.0f / P.depth_range);
    const float norm_const = c1 + c2;
    c1 = c1 / norm_const;
    c2 = c2 / norm_const;
    const float f = c1 * ((a + 1.0f) / (a + b + 1.0f)) + c2 * (a / (a + b + 1.0f));
    const float e = c1 * (((a + 1.0f) * (a + 2.0f)) / ((a + b + 1.0f) * (a + b + 2.0f))) +
                    c2 * (a * (a + 1.0f) / ((a + b + 1.0f) * (a + b + 2.0f)));
    if (isnan(c1 * m)) return 0;
    const float mu_prime = c1 * m + c2 * mu;
    const float sigma_sq_prime = c1 * (s_sq + m * m) + c2 * (sigma_sq + mu * mu) - mu_prime * mu_prime;
    const float a_prime = (e - f) / (f - e / f);
    sigma_sq = sigma_sq_prime;
    mu = mu_prime;
    a = a_prime;
    b = a_prime * (1.0f - f) / f;
    return 1;
  }
  if (state == ST_NO_MATCH) {
    b = b + 1.0f;
    return 2;
  }
  return 0;
}

// ... and on the planes of the SeedMatrix
RMDK_D void seed_fuse(const SeedParams& P, int x, int y, int i, int state, float mu, float sigma_sq, float a, float b,
                      F2 match) {
  const int what = seed_fuse_values(P, P.T_ref_curr, x, y, state, mu, sigma_sq, a, b, match);
  if (what == 1) {
    P.sigma_sq[i] = sigma_sq;
    P.mu[i] = mu;
    P.a[i] = a;
    P.b[i] = b;
  } else if (what == 2) {
    P.b[i] = b;
  }
}

// ------------------------------------------------------------------------------------------
// Fused check + match + update, one lane per pixel, all image reads straight from global
// memory (L1/L2).  Simple and exact; kept as the A/B baseline and small-image path.
template <int SIDE>
__global__ __launch_bounds__(256) void seed_update_pixel_kernel(SeedParams P) {
  constexpr int OFFSET = -SIDE / 2;
  constexpr float AREA = static_cast<float>(SIDE * SIDE);
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= P.w || y >= P.h) return;
  const int i = y * P.stride + x;
  const float mu = P.mu[i], sigma_sq = P.sigma_sq[i], a = P.a[i], b = P.b[i];
  int state = seed_check(P, x, y, sigma_sq, a, b, SIDE);
  if (state != ST_UPDATE) {
    P.conv[i] = state;
    return;
  }
  const Segment seg = epipolar_segment(P, x, y, mu, sigma_sq);
  const float sum_templ = P.sum_templ[i], denom = P.denom[i];
  float best_ncc = -1.0f;
  F2 best_px = F2{0.0f, 0.0f};
  unsigned steps = 0, evals = 0;
  for (float l = -seg.half_length; l <= seg.half_length; l += 0.7f) {
    ++steps;
    const F2 px = F2{seg.mean.x + l * seg.dir.x, seg.mean.y + l * seg.dir.y};
    if (px.x >= static_cast<float>(P.w - SIDE) || px.y >= static_cast<float>(P.h - SIDE) ||
        px.x < static_cast<float>(SIDE) || px.y < static_cast<float>(SIDE))
      continue;
    ++evals;
    float sum_img = 0.0f, sum_img_sq = 0.0f, sum_img_templ = 0.0f;
    for (int py = 0; py < SIDE; ++py) {
      const float cy = px.y + static_cast<float>(OFFSET + py) + 0.5f;
      const float* ref_row = P.ref + clampi(y + OFFSET + py, 0, P.h - 1) * P.stride;
      for (int pxi = 0; pxi < SIDE; ++pxi) {
        const float templ = ref_row[clampi(x + OFFSET + pxi, 0, P.w - 1)];
        const float img = tex_linear_global(P.cur, P.w, P.h, P.cur_stride, px.x + static_cast<float>(OFFSET + pxi) + 0.5f, cy);
        sum_img += img;
        sum_img_sq += img * img;
        sum_img_templ += img * templ;
      }
    }
    const float num = AREA * sum_img_templ - sum_img * sum_templ;
    const float den = (AREA * sum_img_sq - sum_img * sum_img) * denom;
    const float ncc = num * rmd_rsqrtf(den + FLT_MIN);
    if (ncc > best_ncc) {
      best_px = px;
      best_ncc = ncc;
    }
  }
  if (best_ncc < 0.5f) {
    state = ST_NO_MATCH;
  } else {
    P.match[y * P.stride2 + x] = make_float2(best_px.x, best_px.y);
  }
  P.conv[i] = state;
  if (P.stats) {
    atomicAdd(&P.stats[0], 1ull);
    atomicAdd(&P.stats[1], static_cast<unsigned long long>(steps));
    atomicAdd(&P.stats[2], static_cast<unsigned long long>(evals));
  }
  seed_fuse(P, x, y, i, state, mu, sigma_sq, a, b, best_px);
}

// ------------------------------------------------------------------------------------------
// Reductions (reduction_kernels.cu:57-159): wave-level shuffles, one atomic / one partial per block.
RMDK_D unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
RMDK_D double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

static __global__ __launch_bounds__(256) void count_eq_kernel(const int* __restrict__ img, int w, int h, int stride, int value,
                                                       unsigned long long* __restrict__ out) {
  __shared__ unsigned long long wave_part[4];
  unsigned long long c = 0;
  for (int y = blockIdx.y; y < h; y += gridDim.y) {
    const int* row = img + static_cast<size_t>(y) * stride;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x) c += (row[x] == value);
  }
  c = wave_sum_u64(c);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_part[wave] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = wave_part[0] + wave_part[1] + wave_part[2] + wave_part[3];
    if (t) atomicAdd(out, t);
  }
}

// pass 1: one fp64 partial per block, in a fixed order; pass 2: one block folds the partials.
static __global__ __launch_bounds__(256) void sum_partial_kernel(const float* __restrict__ img, int w, int h, int stride,
                                                          double* __restrict__ partials) {
  __shared__ double wave_part[4];
  double acc = 0.0;
  for (int y = blockIdx.y; y < h; y += gridDim.y) {
    const float* row = img + static_cast<size_t>(y) * stride;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x) acc += static_cast<double>(row[x]);
  }
  acc = wave_sum_f64(acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    partials[blockIdx.y * gridDim.x + blockIdx.x] = ((wave_part[0] + wave_part[1]) + wave_part[2]) + wave_part[3];
}
// integer image sum (ImageReducer<int>::sum, reduction.cu:186): exact in 64 bits, the caller truncates to int like the
// reference's int accumulation wraps
static __global__ __launch_bounds__(256) void sum_i32_kernel(const int* __restrict__ img, int w, int h, int stride, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long wave_part[4];
  unsigned long long acc = 0;
  for (int y = blockIdx.y; y < h; y += gridDim.y) {
    const int* row = img + static_cast<size_t>(y) * stride;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x) acc += static_cast<unsigned long long>(static_cast<long long>(row[x]));
  }
  acc = wave_sum_u64(acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, wave_part[0] + wave_part[1] + wave_part[2] + wave_part[3]);
}

static __global__ __launch_bounds__(64) void sum_final_kernel(const double* __restrict__ partials, int n, float* __restrict__ out) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) acc += partials[i];
  acc = wave_sum_f64(acc);
  if (threadIdx.x == 0) *out = static_cast<float>(acc);
}

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
    const unsigned int g = static_cast<unsigned int>(rintf(ref[i] * 255.0f)) & 0xffu;
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

// ------------------------------------------------------------------------------------------
// TV-L1 denoiser.
struct TvParams {
  int w, h;
  int stride;   // f32 planes
  int stride2;  // float2 plane
  const float* mu;
  const float* sigma_sq;
  const float* a;
  const float* b;
  int in_stride;  // stride of the four input planes (they belong to the SeedMatrix)
  float* g;
  float large_sigma_sq;
  float tau, sigma, theta, lambda;
  // Several depth maps of one size denoised by ONE launch sequence (rmd_hip_batch_denoise): grid z = member.  The denoiser's own planes
  // (g, u, u_head, p) hold the members back to back, `member_stride` / `member_stride2` elements apart; the input planes belong to the
  // members' SeedMatrix objects and come from a table in device memory, TV_MEMBER_WORDS 64-bit words per member: the addresses of mu,
  // sigma_sq, a, b, then large_sigma_sq (the bits of the float in the low half).  null: one depth map, the fields above.
  const unsigned long long* members;
  size_t member_stride, member_stride2;
};
constexpr int TV_MEMBER_WORDS = 5;

// the parameters and iterate planes of member blockIdx.z (no-op for a single depth map); the table is read through the scalar path
// (nobody writes it while a denoise is in flight)
RMDK_D void tv_select_member(TvParams& P) {
  if (!P.members) return;
  typedef const __attribute__((address_space(4))) unsigned long long* const_u64_ptr;
  const const_u64_ptr t = (const_u64_ptr)(P.members) + static_cast<size_t>(blockIdx.z) * TV_MEMBER_WORDS;
  P.mu = reinterpret_cast<const float*>(t[0]);
  P.sigma_sq = reinterpret_cast<const float*>(t[1]);
  P.a = reinterpret_cast<const float*>(t[2]);
  P.b = reinterpret_cast<const float*>(t[3]);
  P.large_sigma_sq = __uint_as_float(static_cast<unsigned int>(t[4]));
  P.g += static_cast<size_t>(blockIdx.z) * P.member_stride;
}
template <typename T>
RMDK_D T* tv_member_plane(const TvParams& P, T* plane, size_t member_stride) {
  return P.members ? plane + static_cast<size_t>(blockIdx.z) * member_stride : plane;
}

// depthmap_denoiser.cu:45-59 (weights) fused with the re-initialisation at :215-217
static __global__ __launch_bounds__(256) void tv_prepare_kernel(TvParams P, float* __restrict__ u, float* __restrict__ u_head,
                                                         float2* __restrict__ p) {
  tv_select_member(P);
  u = tv_member_plane(P, u, P.member_stride); u_head = tv_member_plane(P, u_head, P.member_stride); p = tv_member_plane(P, p, P.member_stride2);
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= P.w || y >= P.h) return;
  const int ii = y * P.in_stride + x;
  const float a = P.a[ii], b = P.b[ii];
  const float E_pi = a / (a + b);
  const float v = (E_pi * P.sigma_sq[ii] + (1.0f - E_pi) * P.large_sigma_sq) / P.large_sigma_sq;
  const int i = y * P.stride + x;
  P.g[i] = v > 1.0f ? v : 1.0f;
  const float m = P.mu[ii];
  u[i] = m;
  u_head[i] = m;
  p[y * P.stride2 + x] = make_float2(0.0f, 0.0f);
}

// dual step of one pixel (depthmap_denoiser.cu:73-83), reading the previous iterate
RMDK_D float2 tv_dual(const TvParams& P, const float* __restrict__ u, const float* __restrict__ u_head,
                      const float2* __restrict__ p, int x, int y) {
  const int i = y * P.stride + x;
  const float g = P.g[i], cu = u[i];
  const int xe = x + 1 < P.w - 1 ? x + 1 : P.w - 1;
  const int ys = y + 1 < P.h - 1 ? y + 1 : P.h - 1;
  const float gx = u_head[y * P.stride + xe] - cu;
  const float gy = u_head[ys * P.stride + x] - cu;
  const float2 po = p[y * P.stride2 + x];
  const float tx = g * gx * P.sigma + po.x;
  const float ty = g * gy * P.sigma + po.y;
  const float mag = sqrtf(tx * tx + ty * ty);
  const float den = 1.0f > mag ? 1.0f : mag;
  return make_float2(tx / den, ty / den);
}

// One primal-dual iteration (depthmap_denoiser.cu:61-118) with "all duals, then all primals"
// semantics: the block evaluates the dual for its TX x TY tile plus the one-pixel west column
// and north row it needs, keeps them in LDS, then runs the primal step.  Iterates are
// ping-ponged between (u,u_head,p)_in and _out so there is no inter-block race.
constexpr int TV_TX = 64, TV_TY = 4;
static __global__ __launch_bounds__(TV_TX* TV_TY) void tv_iterate_kernel(TvParams P, const float* __restrict__ u_in,
                                                                  const float* __restrict__ uh_in,
                                                                  const float2* __restrict__ p_in, float* __restrict__ u_out,
                                                                  float* __restrict__ uh_out, float2* __restrict__ p_out) {
  __shared__ float2 sp[TV_TY + 1][TV_TX + 1];
  tv_select_member(P);
  u_in = tv_member_plane(P, u_in, P.member_stride); uh_in = tv_member_plane(P, uh_in, P.member_stride); p_in = tv_member_plane(P, p_in, P.member_stride2);
  u_out = tv_member_plane(P, u_out, P.member_stride); uh_out = tv_member_plane(P, uh_out, P.member_stride); p_out = tv_member_plane(P, p_out, P.member_stride2);
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int x0 = blockIdx.x * TV_TX, y0 = blockIdx.y * TV_TY;
  const int x = x0 + tx, y = y0 + ty;
  const bool inside = x < P.w && y < P.h;
  if (inside) sp[ty + 1][tx + 1] = tv_dual(P, u_in, uh_in, p_in, x, y);
  const int t = ty * TV_TX + tx;
  if (t < TV_TY) {  // west halo column
    const int yy = y0 + t;
    if (x0 > 0 && yy < P.h) sp[t + 1][0] = tv_dual(P, u_in, uh_in, p_in, x0 - 1, yy);
  } else if (t >= 64 && t < 64 + TV_TX) {  // north halo row (a different wave than the west column)
    const int xx = x0 + (t - 64);
    if (y0 > 0 && xx < P.w) sp[0][t - 64 + 1] = tv_dual(P, u_in, uh_in, p_in, xx, y0 - 1);
  }
  __syncthreads();
  if (!inside) return;
  const int i = y * P.stride + x;
  const float noisy = P.mu[y * P.in_stride + x], old_u = u_in[i], g = P.g[i];
  float2 cp = sp[ty + 1][tx + 1];
  float wpx = sp[ty + 1][tx].x;
  float npy = sp[ty][tx + 1].y;
  if (x == 0) wpx = 0.0f;
  else if (x >= P.w - 1) cp.x = 0.0f;
  if (y == 0) npy = 0.0f;
  else if (y >= P.h - 1) cp.y = 0.0f;
  const float divergence = cp.x - wpx + cp.y - npy;
  const float temp_u = old_u + P.tau * g * divergence;
  float nu;
  if ((temp_u - noisy) > (P.tau * P.lambda)) nu = temp_u - P.tau * P.lambda;
  else if ((temp_u - noisy) < (-P.tau * P.lambda)) nu = temp_u + P.tau * P.lambda;
  else nu = noisy;
  u_out[i] = nu;
  uh_out[i] = nu + P.theta * (nu - old_u);
  p_out[y * P.stride2 + x] = sp[ty + 1][tx + 1];
}

// ------------------------------------------------------------------------------------------
// Temporally blocked TV-L1: `iters` (<= KMAX) primal-dual iterations per launch.  A workgroup loads its BX x BY output
// tile plus a halo of `iters` pixels (clipped to the image) of (u, u_head, p, g, mu) into LDS and iterates there; the
// region in which the LDS copy equals the true iterate shrinks by one pixel per iteration on every side that is not an
// image border (the dual looks east/south, the primal west/north), so after `iters` iterations exactly the output tile
// is valid.  Per pixel the arithmetic is that of tv_iterate_kernel, hence the same bits; HBM traffic per iteration
// drops from 40 B/pixel to (24*(BX+2K)(BY+2K) + 16*BX*BY)/(K*BX*BY) ~= 14 B/pixel at K = 4.
template <int BX_, int BY_, int KMAX>
struct TvBlocked {
  static constexpr int BX = BX_, BY = BY_, THREADS = 256;
  static constexpr int EW = BX + 2 * KMAX, EH = BY + 2 * KMAX, EN = EW * EH;
  static constexpr int SLOTS = (EN + THREADS - 1) / THREADS;
};

template <int BX_, int BY_, int KMAX>
__global__ __launch_bounds__(256) void tv_iterate_blocked_kernel(TvParams P, const float* __restrict__ u_in,
                                                                 const float* __restrict__ uh_in, const float2* __restrict__ p_in,
                                                                 float* __restrict__ u_out, float* __restrict__ uh_out,
                                                                 float2* __restrict__ p_out, int iters) {
  using G = TvBlocked<BX_, BY_, KMAX>;
  __shared__ float su[G::EN], suh[G::EN], spx[G::EN], spy[G::EN], sg[G::EN], smu[G::EN];
  tv_select_member(P);
  u_in = tv_member_plane(P, u_in, P.member_stride); uh_in = tv_member_plane(P, uh_in, P.member_stride); p_in = tv_member_plane(P, p_in, P.member_stride2);
  u_out = tv_member_plane(P, u_out, P.member_stride); uh_out = tv_member_plane(P, uh_out, P.member_stride); p_out = tv_member_plane(P, p_out, P.member_stride2);
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * G::BX, y0 = blockIdx.y * G::BY;
  const int ex0 = max(x0 - iters, 0), ey0 = max(y0 - iters, 0);
  const int ex1 = min(x0 + G::BX + iters, P.w), ey1 = min(y0 + G::BY + iters, P.h);  // exclusive
  const int ew = ex1 - ex0, eh = ey1 - ey0, en = ew * eh;
  // this lane's pixels of the extended region: local index (row * EW + col) and whether the slot is used
  int lidx[G::SLOTS];
  short lxs[G::SLOTS], lys[G::SLOTS];
  const float inv_ew = 1.0f / static_cast<float>(ew);
#pragma unroll
  for (int q = 0; q < G::SLOTS; ++q) {
    const int e = tid + q * G::THREADS;
    int ly = static_cast<int>(static_cast<float>(e) * inv_ew);
    int lx = e - ly * ew;
    if (lx < 0) { --ly; lx += ew; } else if (lx >= ew) { ++ly; lx -= ew; }
    lxs[q] = static_cast<short>(lx); lys[q] = static_cast<short>(ly);
    lidx[q] = e < en ? ly * G::EW + lx : -1;
  }
#pragma unroll
  for (int q = 0; q < G::SLOTS; ++q) {
    if (lidx[q] < 0) continue;
    const int gx = ex0 + lxs[q], gy = ey0 + lys[q];
    const int gi = gy * P.stride + gx;
    su[lidx[q]] = u_in[gi];
    suh[lidx[q]] = uh_in[gi];
    const float2 pv = p_in[gy * P.stride2 + gx];
    spx[lidx[q]] = pv.x; spy[lidx[q]] = pv.y;
    sg[lidx[q]] = P.g[gi];
    smu[lidx[q]] = P.mu[gy * P.in_stride + gx];
  }
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    // dual (depthmap_denoiser.cu:73-83); neighbour indices clamp to the region, which is the image clamp wherever the
    // region ends at the image border and only touches the (discarded) rim elsewhere
#pragma unroll
    for (int q = 0; q < G::SLOTS; ++q) {
      if (lidx[q] < 0) continue;
      const int lx = lxs[q], ly = lys[q], i = lidx[q];
      const float g = sg[i], cu = su[i];
      const int xe = lx + 1 < ew - 1 ? lx + 1 : ew - 1;
      const int ys = ly + 1 < eh - 1 ? ly + 1 : eh - 1;
      const float gxv = suh[ly * G::EW + xe] - cu;
      const float gyv = suh[ys * G::EW + lx] - cu;
      const float tx = g * gxv * P.sigma + spx[i];
      const float ty = g * gyv * P.sigma + spy[i];
      const float mag = sqrtf(tx * tx + ty * ty);
      const float den = 1.0f > mag ? 1.0f : mag;
      spx[i] = tx / den;
      spy[i] = ty / den;
    }
    __syncthreads();
    // primal (depthmap_denoiser.cu:87-115); the boundary rules use image coordinates
#pragma unroll
    for (int q = 0; q < G::SLOTS; ++q) {
      if (lidx[q] < 0) continue;
      const int lx = lxs[q], ly = lys[q], i = lidx[q];
      const int gx = ex0 + lx, gy = ey0 + ly;
      const float noisy = smu[i], old_u = su[i], g = sg[i];
      float cpx = spx[i], cpy = spy[i];
      float wpx = spx[ly * G::EW + (lx - 1 > 0 ? lx - 1 : 0)];
      float npy = spy[(ly - 1 > 0 ? ly - 1 : 0) * G::EW + lx];
      if (gx == 0) wpx = 0.0f;
      else if (gx >= P.w - 1) cpx = 0.0f;
      if (gy == 0) npy = 0.0f;
      else if (gy >= P.h - 1) cpy = 0.0f;
      const float divergence = cpx - wpx + cpy - npy;
      const float temp_u = old_u + P.tau * g * divergence;
      float nu;
      if ((temp_u - noisy) > (P.tau * P.lambda)) nu = temp_u - P.tau * P.lambda;
      else if ((temp_u - noisy) < (-P.tau * P.lambda)) nu = temp_u + P.tau * P.lambda;
      else nu = noisy;
      su[i] = nu;
      suh[i] = nu + P.theta * (nu - old_u);
    }
    __syncthreads();
  }
  // store the output tile
#pragma unroll
  for (int q = 0; q < (G::BX * G::BY) / G::THREADS; ++q) {
    const int t = tid + q * G::THREADS;
    const int ty = t / G::BX, tx = t - ty * G::BX;
    const int gx = x0 + tx, gy = y0 + ty;
    if (gx >= P.w || gy >= P.h) continue;
    const int i = (gy - ey0) * G::EW + (gx - ex0);
    const int gi = gy * P.stride + gx;
    u_out[gi] = su[i];
    uh_out[gi] = suh[i];
    p_out[gy * P.stride2 + gx] = make_float2(spx[i], spy[i]);
  }
}

// ------------------------------------------------------------------------------------------
// device side of the arithmetic contract, for the self test
// self test of the DPP wave primitives of rmd_device.hpp against the shuffle forms: mismatching lanes -> *bad
static __global__ __launch_bounds__(64) void wave_primitives_selftest_kernel(unsigned int seed, unsigned int* bad) {
  const int lane = threadIdx.x;
  unsigned int h = seed * 2654435761u + static_cast<unsigned int>(lane) * 40503u + blockIdx.x * 97u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  const int v = static_cast<int>(h % 2001u) - 1000;
  int add = v, mn = v, mx = v;
  for (int off = 32; off > 0; off >>= 1) {
    add += __shfl_xor(add, off, 64);
    mn = min(mn, __shfl_xor(mn, off, 64));
    mx = max(mx, __shfl_xor(mx, off, 64));
  }
  int incl = v;
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  const bool ok = wave_reduce_i32<WaveAdd>(v) == add && wave_reduce_i32<WaveMin>(v) == mn && wave_reduce_i32<WaveMax>(v) == mx &&
                  wave_scan_i32<WaveAdd>(v) == incl;
  if (!ok) atomicAdd(bad, 1u);
}

static __global__ void math_eval_kernel(int op, const float* x, const float* y, const float* z, float* out, size_t n) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r;
  switch (op) {
    case 0: r = rmd_expf(x[i]); break;
    case 1: r = rmd_sinf(x[i]); break;
    case 2: r = rmd_acosf(x[i]); break;
    case 3: r = rmd_rsqrtf(x[i]); break;
    case 4: r = sqrtf(x[i]); break;
    case 5: r = x[i] / y[i]; break;
    default: r = rmd_lerp(x[i], y[i], z[i]); break;
  }
  out[i] = r;
}

}  // namespace rmdk

#endif  // RMD_KERNELS_HPP
