// TV-L1 kernels (src/depthmap_denoiser.cu:45-118): tv_prepare_kernel <- :45-59 + :215-217, tv_iterate_kernel <- :61-118 (one iteration per
// launch), tv_iterate_blocked_kernel (K iterations per launch in LDS).  Included by ONE translation unit, rmd_denoise.hip: the kernels
// exist in one code object (TvParams and the member table live in rmd_kernels.hpp, which the host units share).
#ifndef RMD_TV_KERNELS_HPP
#define RMD_TV_KERNELS_HPP

#include "rmd_kernels.hpp"

namespace rmdk {

// depthmap_denoiser.cu:45-59 (weights) fused with the re-initialisation at :215-217
static __global__ __launch_bounds__(256) void tv_prepare_kernel(TvParams P, float* __restrict__ u, float* __restrict__ u_head,
                                                         float2* __restrict__ p) {
  tv_select_member(P);
  u = tv_member_plane(P, u, P.member_stride); u_head = tv_member_plane(P, u_head, P.member_stride); p = tv_member_plane(P, p,
      P.member_stride2);
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
static __global__ __launch_bounds__(TV_TX * TV_TY) void tv_iterate_kernel(TvParams P, const float* __restrict__ u_in,
                                                                  const float* __restrict__ uh_in,
                                                                  const float2* __restrict__ p_in, float* __restrict__ u_out,
                                                                  float* __restrict__ uh_out, float2* __restrict__ p_out) {
  __shared__ float2 sp[TV_TY + 1][TV_TX + 1];
  tv_select_member(P);
  u_in = tv_member_plane(P, u_in, P.member_stride); uh_in = tv_member_plane(P, uh_in, P.member_stride); p_in = tv_member_plane(P, p_in,
      P.member_stride2);
  u_out = tv_member_plane(P, u_out, P.member_stride); uh_out = tv_member_plane(P, uh_out, P.member_stride); p_out = tv_member_plane(P,
      p_out, P.member_stride2);
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
  u_in = tv_member_plane(P, u_in, P.member_stride); uh_in = tv_member_plane(P, uh_in, P.member_stride); p_in = tv_member_plane(P, p_in,
      P.member_stride2);
  u_out = tv_member_plane(P, u_out, P.member_stride); uh_out = tv_member_plane(P, uh_out, P.member_stride); p_out = tv_member_plane(P,
      p_out, P.member_stride2);
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

}  // namespace rmdk

#endif  // RMD_TV_KERNELS_HPP
