// Tile-cooperative fused seed update: seed_check (src/seed_check.cu:28-67) + epipolar NCC search
// (src/epipolar_match.cu:37-140) + triangulation and Bayesian fusion (src/seed_update.cu:39-121,
// src/triangulation.cu) in ONE launch per frame.  This is the production kernel.
//
// Why not one lane per pixel (the reference's shape, kept in rmd_kernels.hpp as the A/B baseline):
// the per-seed trip count of the search varies from 0 (converged / diverged / border seeds) to 143
// steps, so a wave64 idles at the pace of its longest lane, and every lane gathers its own
// (SIDE+1)^2 texels per step through the vector memory path.  Here a 256-thread workgroup owns a
// 16x16 tile of seeds and
//   1. every lane sets up ITS seed (state check, epipolar segment, the contiguous run of steps
//      whose patch lies inside the image) and the workgroup prefix-sums the step counts;
//   2. the bounding box of all sample positions of the tile (plus patch halo) is staged ONCE from
//      the current image into LDS with coalesced row loads, next to the reference-image tile;
//   3. the tile's (seed, step) pairs are dealt round-robin to the 256 lanes, so lanes stay busy
//      whatever the per-seed trip counts are, and neighbouring lanes walk neighbouring steps of
//      the same segment (LDS-friendly).  One work item = one NCC evaluation with the separable
//      bilinear filter of rmd_math.h: (SIDE+1)^2 LDS reads feed SIDE*(SIDE+1) horizontal and
//      SIDE^2 vertical lerps instead of 4*SIDE^2 texel fetches;
//   4. the per-seed arg-max (ties -> lowest step, as the reference's strict '>' does) is an LDS
//      64-bit atomic max on {orderable(ncc), ~step};
//   5. every lane finishes ITS seed: match coordinates, triangulation, posterior update.
// Results are bit-identical to the reference semantics (tests/test_hip_parity.py).
#ifndef RMD_MATCHER_HPP
#define RMD_MATCHER_HPP

#include "rmd_kernels.hpp"

namespace rmdk {

constexpr int TILE_W = 16, TILE_H = 16, TILE_PIX = TILE_W * TILE_H;
constexpr int MAX_STEPS = 160;  // > floor(100 / 0.7) + 1 = 143 iterations of the search loop

struct MatcherWorkspace {
  // device: [0] max window width seen, [1] max window height seen, [2] tiles that fell back to global reads
  unsigned int* d_feedback = nullptr;
  unsigned int* h_feedback = nullptr;  // pinned mirror, read one frame late
  int allocate(int, int) {
    if (hipMalloc(reinterpret_cast<void**>(&d_feedback), 4 * sizeof(unsigned int)) != hipSuccess) return -1;
    if (hipHostMalloc(reinterpret_cast<void**>(&h_feedback), 4 * sizeof(unsigned int)) != hipSuccess) return -1;
    (void)hipMemset(d_feedback, 0, 4 * sizeof(unsigned int));
    h_feedback[0] = h_feedback[1] = h_feedback[2] = h_feedback[3] = 0;
    return 0;
  }
  void release() {
    if (d_feedback) (void)hipFree(d_feedback);
    if (h_feedback) (void)hipHostFree(h_feedback);
    d_feedback = nullptr;
    h_feedback = nullptr;
  }
};

RMDK_D unsigned int orderable_f32(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
RMDK_D float from_orderable_f32(unsigned int o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

RMDK_D bool px_outside(const SeedParams& P, F2 px, int side) {  // the guard at epipolar_match.cu:91-97
  return px.x >= static_cast<float>(P.w - side) || px.y >= static_cast<float>(P.h - side) ||
         px.x < static_cast<float>(side) || px.y < static_cast<float>(side);
}

// Per-axis sample parameters of one step: for patch column/row k the reference evaluates
//   c = (p + (float)(OFFSET + k)) + 0.5f      (epipolar_match.cu:111-114)
// and the linear filter then uses  b = c - 0.5f,  i = floor(b),  w = b - i.
// The roundings differ from column to column near binade boundaries, so they are replayed exactly.
template <int SIDE>
RMDK_D bool axis_params(float p, int (&idx)[SIDE], float (&wgt)[SIDE]) {
  constexpr int OFFSET = -SIDE / 2;
  bool regular = true;
#pragma unroll
  for (int k = 0; k < SIDE; ++k) {
    const float c = p + static_cast<float>(OFFSET + k) + 0.5f;
    const float b = c - 0.5f;
    const float f = floorf(b);
    wgt[k] = b - f;
    idx[k] = static_cast<int>(f);
    regular = regular && (idx[k] == idx[0] + k);
  }
  return regular;
}

// Sums of one NCC evaluation over a REGULAR footprint: texel rows j0 .. j0+SIDE, columns i0 .. i0+SIDE,
// read from `base` (pointing at texel (i0, j0)) with row stride CT_STRIDE (compile time, LDS window)
// or rt_stride (run time, global memory).  Separable filter: SIDE+1 horizontal lerps per texel row are
// shared by the two patch rows that straddle it.
template <int SIDE, int CT_STRIDE>
RMDK_D void ncc_sums_regular(const float* __restrict__ base, int rt_stride, const float (&ax)[SIDE], const float (&ay)[SIDE],
                             const float* __restrict__ ref_patch, int ref_stride, float& sum_img, float& sum_img_sq,
                             float& sum_img_templ) {
  float hprev[SIDE], hcur[SIDE];
#pragma unroll
  for (int r = 0; r <= SIDE; ++r) {
    const float* row = CT_STRIDE > 0 ? base + r * CT_STRIDE : base + r * rt_stride;
    float t[SIDE + 1];
#pragma unroll
    for (int c = 0; c <= SIDE; ++c) t[c] = row[c];
#pragma unroll
    for (int k = 0; k < SIDE; ++k) hcur[k] = rmd_lerp(ax[k], t[k], t[k + 1]);
    if (r > 0) {
      const float by = ay[r - 1];
#pragma unroll
      for (int k = 0; k < SIDE; ++k) {
        const float img = rmd_lerp(by, hprev[k], hcur[k]);
        const float templ = ref_patch[(r - 1) * ref_stride + k];
        sum_img += img;
        sum_img_sq += img * img;
        sum_img_templ += img * templ;
      }
    }
#pragma unroll
    for (int k = 0; k < SIDE; ++k) hprev[k] = hcur[k];
  }
}

// One NCC evaluation at px.  Three sources for the current-image samples, same arithmetic in all:
//   1. the LDS window staged for this tile, when the footprint lies inside it;
//   2. global memory (L1/L2) with the same regular footprint, for seeds that wandered off the window;
//   3. per-sample fetches, only when the replayed roundings make the footprint irregular.
template <int SIDE, int WS>
RMDK_D float ncc_at(const SeedParams& P, F2 px, const float* __restrict__ win, int wx0, int wy0, int wx1, int wy1,
                    const float* __restrict__ ref_patch, int ref_stride, float sum_templ, float denom) {
  constexpr int OFFSET = -SIDE / 2;
  constexpr float AREA = static_cast<float>(SIDE * SIDE);
  float sum_img = 0.0f, sum_img_sq = 0.0f, sum_img_templ = 0.0f;
  int ix[SIDE], iy[SIDE];
  float ax[SIDE], ay[SIDE];
  const bool reg_x = axis_params<SIDE>(px.x, ix, ax);
  const bool reg_y = axis_params<SIDE>(px.y, iy, ay);
  if (reg_x && reg_y) {
    // the guard keeps px in [SIDE, dim-SIDE), so rows iy[0]..iy[0]+SIDE and columns ix[0]..ix[0]+SIDE are in the image
    const bool in_window = ix[0] >= wx0 && iy[0] >= wy0 && ix[0] + SIDE <= wx1 && iy[0] + SIDE <= wy1;
    if (in_window) {
      ncc_sums_regular<SIDE, WS>(win + (iy[0] - wy0) * WS + (ix[0] - wx0), 0, ax, ay, ref_patch, ref_stride, sum_img,
                                 sum_img_sq, sum_img_templ);
    } else {
      ncc_sums_regular<SIDE, 0>(P.cur + iy[0] * P.stride + ix[0], P.stride, ax, ay, ref_patch, ref_stride, sum_img,
                                sum_img_sq, sum_img_templ);
    }
  } else {
    for (int m = 0; m < SIDE; ++m) {
      const float cy = px.y + static_cast<float>(OFFSET + m) + 0.5f;
      for (int k = 0; k < SIDE; ++k) {
        const float cx = px.x + static_cast<float>(OFFSET + k) + 0.5f;
        const float img = tex_linear_global(P.cur, P.w, P.h, P.stride, cx, cy);
        const float templ = ref_patch[m * ref_stride + k];
        sum_img += img;
        sum_img_sq += img * img;
        sum_img_templ += img * templ;
      }
    }
  }
  const float num = AREA * sum_img_templ - sum_img * sum_templ;
  const float den = (AREA * sum_img_sq - sum_img * sum_img) * denom;
  return num * rmd_rsqrtf(den + FLT_MIN);
}

template <int SIDE, int WS, int WROWS>
struct TileSmem {
  static constexpr int HALF = SIDE / 2;
  static constexpr int REF_W = TILE_W + SIDE - 1, REF_H = TILE_H + SIDE - 1;
  float win[WROWS * WS];
  float ref[REF_H * REF_W];
  float mean_x[TILE_PIX], mean_y[TILE_PIX], dir_x[TILE_PIX], dir_y[TILE_PIX];
  float l_first[TILE_PIX];
  float sum_templ[TILE_PIX], denom[TILE_PIX];
  unsigned long long best[TILE_PIX];
  int prefix[TILE_PIX + 1];
  unsigned short i_first[TILE_PIX];
  float red_f[4][8];
  int red_i[4];
  int win_box[4];  // wx0, wy0, wx1, wy1 (inclusive texel bounds of the staged window; wx1 < wx0 if none)
};

template <int SIDE, int WS, int WROWS>
__global__ __launch_bounds__(TILE_PIX) void seed_update_tile_kernel(SeedParams P, unsigned int* __restrict__ feedback) {
  using Smem = TileSmem<SIDE, WS, WROWS>;
  constexpr int HALF = SIDE / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& S = *reinterpret_cast<Smem*>(smem_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tx = tid & (TILE_W - 1), ty = tid >> 4;
  const int x0 = blockIdx.x * TILE_W, y0 = blockIdx.y * TILE_H;
  const int x = x0 + tx, y = y0 + ty;
  const bool in_image = x < P.w && y < P.h;
  const int gi = y * P.stride + x;

  // ---- phase 0: per-seed setup ----------------------------------------------------------------
  float mu = 0.0f, sigma_sq = 0.0f, a = 0.0f, b = 0.0f;
  int state = ST_BORDER;
  if (in_image) {
    mu = P.mu[gi]; sigma_sq = P.sigma_sq[gi]; a = P.a[gi]; b = P.b[gi];
    state = seed_check(P, x, y, sigma_sq, a, b, SIDE);
  }
  const bool live = in_image && state == ST_UPDATE;
  Segment seg;
  seg.mean = F2{0.0f, 0.0f}; seg.dir = F2{0.0f, 0.0f}; seg.half_length = 0.0f;
  int n_valid = 0, i_first = 0;
  float l_first = 0.0f;
  unsigned int n_steps = 0, n_evals = 0;
  float bb_x0 = INFINITY, bb_y0 = INFINITY, bb_x1 = -INFINITY, bb_y1 = -INFINITY;
  if (live) {
    seg = epipolar_segment(P, x, y, mu, sigma_sq);
    const bool finite = isfinite(seg.mean.x) && isfinite(seg.mean.y) && isfinite(seg.dir.x) && isfinite(seg.dir.y);
    // walk the search loop once (epipolar_match.cu:88): count the steps and find the run of steps that
    // pass the in-image guard.  The guard region is convex and px is monotone in l, so that run is contiguous.
    int i = 0;
    for (float l = -seg.half_length; l <= seg.half_length; l += 0.7f, ++i) {
      const F2 px = F2{seg.mean.x + l * seg.dir.x, seg.mean.y + l * seg.dir.y};
      if (!px_outside(P, px, SIDE)) {
        ++n_evals;  // what the reference would evaluate (NaN positions pass its guard too)
        if (finite) {
          if (n_valid == 0) { i_first = i; l_first = l; }
          ++n_valid;
          bb_x0 = fminf(bb_x0, px.x); bb_x1 = fmaxf(bb_x1, px.x);
          bb_y0 = fminf(bb_y0, px.y); bb_y1 = fmaxf(bb_y1, px.y);
        }
      }
    }
    n_steps = static_cast<unsigned int>(i);
    // a non-finite position yields NaN sums in the reference and never becomes a candidate: no work
  }
  S.mean_x[tid] = seg.mean.x; S.mean_y[tid] = seg.mean.y;
  S.dir_x[tid] = seg.dir.x; S.dir_y[tid] = seg.dir.y;
  S.l_first[tid] = l_first;
  S.i_first[tid] = static_cast<unsigned short>(i_first);
  S.best[tid] = 0ull;
  if (live) { S.sum_templ[tid] = P.sum_templ[gi]; S.denom[tid] = P.denom[gi]; }

  // ---- phase 1: workgroup scan of step counts; where do the tile's samples fall? ---------------
  int incl = n_valid;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  // step-weighted centroid of the seeds' sample positions (placement heuristic only; not part of any result)
  float cw = static_cast<float>(n_valid);
  float cxw = n_valid ? cw * 0.5f * (bb_x0 + bb_x1) : 0.0f;
  float cyw = n_valid ? cw * 0.5f * (bb_y0 + bb_y1) : 0.0f;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    bb_x0 = fminf(bb_x0, __shfl_xor(bb_x0, off, 64)); bb_y0 = fminf(bb_y0, __shfl_xor(bb_y0, off, 64));
    bb_x1 = fmaxf(bb_x1, __shfl_xor(bb_x1, off, 64)); bb_y1 = fmaxf(bb_y1, __shfl_xor(bb_y1, off, 64));
    cw += __shfl_xor(cw, off, 64); cxw += __shfl_xor(cxw, off, 64); cyw += __shfl_xor(cyw, off, 64);
  }
  if (lane == 63) S.red_i[wave] = incl;
  if (lane == 0) {
    S.red_f[wave][0] = bb_x0; S.red_f[wave][1] = bb_y0; S.red_f[wave][2] = bb_x1; S.red_f[wave][3] = bb_y1;
    S.red_f[wave][4] = cw; S.red_f[wave][5] = cxw; S.red_f[wave][6] = cyw;
  }
  __syncthreads();
  int wave_off = 0;
#pragma unroll
  for (int wv = 0; wv < 4; ++wv) wave_off += (wv < wave) ? S.red_i[wv] : 0;
  const int total = S.red_i[0] + S.red_i[1] + S.red_i[2] + S.red_i[3];
  S.prefix[tid] = wave_off + incl - n_valid;  // exclusive
  if (tid == 0) {
    S.prefix[TILE_PIX] = total;
    int wx0 = 0, wy0 = 0, wx1 = -1, wy1 = -1;
    if (total > 0) {
      const float fx0 = fminf(fminf(S.red_f[0][0], S.red_f[1][0]), fminf(S.red_f[2][0], S.red_f[3][0]));
      const float fy0 = fminf(fminf(S.red_f[0][1], S.red_f[1][1]), fminf(S.red_f[2][1], S.red_f[3][1]));
      const float fx1 = fmaxf(fmaxf(S.red_f[0][2], S.red_f[1][2]), fmaxf(S.red_f[2][2], S.red_f[3][2]));
      const float fy1 = fmaxf(fmaxf(S.red_f[0][3], S.red_f[1][3]), fmaxf(S.red_f[2][3], S.red_f[3][3]));
      // texels touched by a sample at p: floor(p) - HALF .. floor(p) + HALF + 1, +1 for the replayed roundings
      wx0 = max(static_cast<int>(floorf(fx0)) - HALF - 1, 0);
      wy0 = max(static_cast<int>(floorf(fy0)) - HALF - 1, 0);
      wx1 = min(static_cast<int>(floorf(fx1)) + HALF + 2, P.w - 1);
      wy1 = min(static_cast<int>(floorf(fy1)) + HALF + 2, P.h - 1);
      const int need_w = wx1 - wx0 + 1, need_h = wy1 - wy0 + 1;
      if (feedback) {
        atomicMax(&feedback[0], static_cast<unsigned int>(need_w));
        atomicMax(&feedback[1], static_cast<unsigned int>(need_h));
      }
      if (need_w > WS || need_h > WROWS) {
        // a few seeds wandered off: centre the window on where most samples are; the rest read global memory
        const float w_sum = S.red_f[0][4] + S.red_f[1][4] + S.red_f[2][4] + S.red_f[3][4];
        const float cx = (S.red_f[0][5] + S.red_f[1][5] + S.red_f[2][5] + S.red_f[3][5]) / w_sum;
        const float cy = (S.red_f[0][6] + S.red_f[1][6] + S.red_f[2][6] + S.red_f[3][6]) / w_sum;
        if (need_w > WS) {
          wx0 = min(max(static_cast<int>(cx) - WS / 2, 0), max(P.w - WS, 0));
          wx1 = min(wx0 + WS - 1, P.w - 1);
        }
        if (need_h > WROWS) {
          wy0 = min(max(static_cast<int>(cy) - WROWS / 2, 0), max(P.h - WROWS, 0));
          wy1 = min(wy0 + WROWS - 1, P.h - 1);
        }
        if (feedback) atomicAdd(&feedback[2], 1u);
      }
    }
    S.win_box[0] = wx0; S.win_box[1] = wy0; S.win_box[2] = wx1; S.win_box[3] = wy1;
  }
  __syncthreads();

  // ---- phase 2: stage the current-image window and the reference tile into LDS ----------------
  const int wx0 = S.win_box[0], wy0 = S.win_box[1], wx1 = S.win_box[2], wy1 = S.win_box[3];
  if (total > 0) {
    const int ww = wx1 - wx0 + 1, wh = wy1 - wy0 + 1;
    for (int r = wave; r < wh; r += 4) {  // one wave per row, lanes along the row: coalesced dword loads
      const float* src = P.cur + (wy0 + r) * P.stride + wx0;
      for (int c = lane; c < ww; c += 64) S.win[r * WS + c] = src[c];
    }
    for (int i = tid; i < Smem::REF_H * Smem::REF_W; i += TILE_PIX) {
      const int ry = i / Smem::REF_W, rx = i - ry * Smem::REF_W;
      S.ref[i] = P.ref[clampi(y0 - HALF + ry, 0, P.h - 1) * P.stride + clampi(x0 - HALF + rx, 0, P.w - 1)];
    }
  }
  __syncthreads();

  // ---- phase 3: the tile's (seed, step) work items, dealt round-robin to the lanes -------------
  const int rounds = (total + TILE_PIX - 1) / TILE_PIX;
  for (int rd = 0; rd < rounds; ++rd) {
    const int k = rd * TILE_PIX + tid;
    int p = -1;
    unsigned long long key = 0ull;
    if (k < total) {
      int lo = 0, hi = TILE_PIX;  // last p with prefix[p] <= k
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int mid = (lo + hi) >> 1;
        if (S.prefix[mid] <= k) lo = mid; else hi = mid;
      }
      p = lo;
      const int j = k - S.prefix[p];
      float l = S.l_first[p];
      for (int q = 0; q < j; ++q) l += 0.7f;  // the reference accumulates l; replay it
      const F2 px = F2{S.mean_x[p] + l * S.dir_x[p], S.mean_y[p] + l * S.dir_y[p]};
      const int ptx = p & (TILE_W - 1), pty = p >> 4;
      const float ncc = ncc_at<SIDE, WS>(P, px, S.win, wx0, wy0, wx1, wy1, S.ref + pty * Smem::REF_W + ptx, Smem::REF_W,
                                        S.sum_templ[p], S.denom[p]);
      if (ncc > -1.0f) {  // NaN and anything the reference's "ncc > best_ncc" would never accept are dropped
        const unsigned int step = static_cast<unsigned int>(S.i_first[p]) + static_cast<unsigned int>(j);
        key = (static_cast<unsigned long long>(orderable_f32(ncc + 0.0f)) << 32) | (0xffffffffu - step);
      }
    }
    // seeds occupy runs of consecutive lanes: segmented max towards the run's first lane, then one LDS atomic per run
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long okey = __shfl_down(key, off, 64);
      const int op = __shfl_down(p, off, 64);
      if (lane + off < 64 && op == p && okey > key) key = okey;
    }
    const int prev_p = __shfl_up(p, 1, 64);
    if (p >= 0 && key != 0ull && (lane == 0 || prev_p != p)) atomicMax(&S.best[p], key);
  }
  __syncthreads();

  // ---- phase 4: every lane finishes its own seed ---------------------------------------------
  if (P.stats) {  // diagnostics: one atomic triple per wave
    const unsigned long long s_live = wave_sum_u64(live ? 1ull : 0ull);
    const unsigned long long s_steps = wave_sum_u64(static_cast<unsigned long long>(n_steps));
    const unsigned long long s_evals = wave_sum_u64(static_cast<unsigned long long>(n_evals));
    if (lane == 0 && s_live) {
      atomicAdd(&P.stats[0], s_live);
      atomicAdd(&P.stats[1], s_steps);
      atomicAdd(&P.stats[2], s_evals);
    }
  }
  if (!in_image) return;
  F2 best_px = F2{0.0f, 0.0f};
  if (live) {
    const unsigned long long key = S.best[tid];
    float best_ncc = -1.0f;
    if (key != 0ull) {
      best_ncc = from_orderable_f32(static_cast<unsigned int>(key >> 32));
      const int step = static_cast<int>(0xffffffffu - static_cast<unsigned int>(key & 0xffffffffu));
      float l = l_first;
      for (int q = i_first; q < step; ++q) l += 0.7f;
      best_px = F2{seg.mean.x + l * seg.dir.x, seg.mean.y + l * seg.dir.y};
    }
    if (best_ncc < 0.5f) {
      state = ST_NO_MATCH;
    } else {
      P.match[y * P.stride2 + x] = make_float2(best_px.x, best_px.y);
    }
  }
  P.conv[gi] = state;
  if (live) seed_fuse(P, x, y, gi, state, mu, sigma_sq, a, b, best_px);
}

template <int SIDE>
inline void launch_seed_update_tile(const SeedParams& P, MatcherWorkspace& ws, hipStream_t stream) {
  // window geometry: rows of 133 floats (covers 16 + 100 + SIDE + 3 texels; odd stride spreads LDS banks)
  constexpr int WS = 133, WROWS = 104;
  using Smem = TileSmem<SIDE, WS, WROWS>;
  const dim3 block(TILE_PIX), grid((P.w + TILE_W - 1) / TILE_W, (P.h + TILE_H - 1) / TILE_H);
  auto kernel = seed_update_tile_kernel<SIDE, WS, WROWS>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(sizeof(Smem)));
  hipLaunchKernelGGL(kernel, grid, block, sizeof(Smem), stream, P, ws.d_feedback);
}

}  // namespace rmdk

#endif  // RMD_MATCHER_HPP
