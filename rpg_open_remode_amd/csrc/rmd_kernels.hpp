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
// T_ref_curr: the pose of the frame the match was found in (P.T_ref_curr, or the previous frame's when that frame's finalisation runs
// fused into the next frame's setup kernel).
RMDK_D int seed_fuse_values(const SeedParams& P, const Pose& T_ref_curr, int x, int y, int state, float& mu, float& sigma_sq, float& a,
    float& b, F2 match) {
  if (state == ST_UPDATE) {
    const F3 f_ref = normalize3(cam2world(P.cam, static_cast<float>(x), static_cast<float>(y)));
    const F3 f_epi = normalize3(cam2world(P.cam, match.x, match.y));
    const F3 pt = triangulate(f_ref, f_epi, T_ref_curr);
    if (pt.z < 0.0f) return 0;
    const float depth = norm3(pt);
    const float tau = triangulation_uncertainty(depth, f_ref, pose_translation(T_ref_curr), P.one_pix_angle);
    const float tau_sq = tau * tau;
    const float s_sq = (tau_sq * sigma_sq) / (tau_sq + sigma_sq);
    const float m = s_sq * (mu / sigma_sq + depth / tau_sq);
    float c1 = (a / (a + b)) * normpdf(depth, mu, sigma_sq + tau_sq);
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

}  // namespace rmdk

#endif  // RMD_KERNELS_HPP
