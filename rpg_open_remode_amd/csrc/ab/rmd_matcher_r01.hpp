// Round-1 tile pipeline (seed_setup_kernel / seed_plan_kernel / seed_search_kernel with a tile-wide 66 KB LDS window): RETIRED
// from the product.  Compiled only into A/B builds of the library (-DRMD_AB_MATCHERS, tools/ab_make.sh), where
// RMD_HIP_OPT_MATCHER 1 selects it; kept because it is the baseline the compact pipeline of rmd_frame.hpp was measured
// against (61.6 vs 47 us per 640x480 update, profiles/r02_matcher_ab.txt).  Bit-identical results.
#ifndef RMD_MATCHER_R01_HPP
#define RMD_MATCHER_R01_HPP

#include "../rmd_matcher.hpp"

namespace rmdk {

// One NCC evaluation at px.  Three sources for the current-image samples, same arithmetic in all:
//   0. the LDS window staged for this tile, when the footprint lies inside it;
//   1. global memory (L1/L2) with the same regular footprint, for seeds that wandered off the window;
//   2. per-sample fetches, only when the replayed roundings make the footprint irregular.
template <int SIDE, int WS>
RMDK_D float ncc_at(const SeedParams& P, F2 px, const float* __restrict__ win, int wx0, int wy0, int wx1, int wy1,
                    const float* __restrict__ ref_patch, int ref_stride, float sum_templ, float denom, int& path) {
  constexpr int OFFSET = -SIDE / 2;
  constexpr float AREA = static_cast<float>(SIDE * SIDE);
  float sum_img = 0.0f, sum_img_sq = 0.0f, sum_img_templ = 0.0f;
  int ix[SIDE], iy[SIDE];
  float ax[SIDE], ay[SIDE];
  bool reg_x = true, reg_y = true;
  if (__all(axis_is_uniform<SIDE>(px.x) && axis_is_uniform<SIDE>(px.y))) {  // wave-uniform branch
    const float fx = floorf(px.x), fy = floorf(px.y);
    const float wx = px.x - fx, wy = px.y - fy;
    ix[0] = static_cast<int>(fx) + OFFSET; iy[0] = static_cast<int>(fy) + OFFSET;
#pragma unroll
    for (int k = 0; k < SIDE; ++k) { ax[k] = wx; ay[k] = wy; }
  } else {
    reg_x = axis_params<SIDE>(px.x, ix, ax);
    reg_y = axis_params<SIDE>(px.y, iy, ay);
  }
  if (reg_x && reg_y) {
    // the guard keeps px in [SIDE, dim-SIDE), so rows iy[0]..iy[0]+SIDE and columns ix[0]..ix[0]+SIDE are in the image
    const bool in_window = ix[0] >= wx0 && iy[0] >= wy0 && ix[0] + SIDE <= wx1 && iy[0] + SIDE <= wy1;
    path = in_window ? 0 : 1;
    if (in_window) {
      ncc_sums_lds_pipelined<SIDE>(win + (iy[0] - wy0) * WS + (ix[0] - wx0), WS, ax, ay, ref_patch, ref_stride, sum_img,
                                   sum_img_sq, sum_img_templ);
    } else {
      ncc_sums_regular<SIDE, 0>(P.cur + iy[0] * P.cur_stride + ix[0], P.cur_stride, ax, ay, ref_patch, ref_stride, sum_img,
                                sum_img_sq, sum_img_templ);
    }
  } else {
    path = 2;
    for (int m = 0; m < SIDE; ++m) {
      const float cy = px.y + static_cast<float>(OFFSET + m) + 0.5f;
      for (int k = 0; k < SIDE; ++k) {
        const float cx = px.x + static_cast<float>(OFFSET + k) + 0.5f;
        const float img = tex_linear_global(P.cur, P.w, P.h, P.cur_stride, cx, cy);
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

// ------------------------------------------------------------------------------------------------
// stage 1: per-tile setup
// FUSE_PREV: the previous frame's seed_finalize has been deferred (nobody looked at the state in between): run it here,
// in the lane that owns the same seed, right before that seed's check for the new frame.  Saves one launch per frame
// in streaming use.  Its intermediate convergence value (UPDATE / NO_MATCH of the previous frame) is never observable
// -- any observer forces the stand-alone seed_finalize_kernel first -- and is overwritten below, so it is not stored.
template <int SIDE, int WS, int WROWS, bool FUSE_PREV>
__global__ __launch_bounds__(TILE_PIX) void seed_setup_kernel(SeedParams P, MatcherArgs M, Pose T_ref_curr_prev) {
  constexpr int HALF = SIDE / 2;
  __shared__ float red_f[4][8];
  __shared__ int red_i[4];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tx = tid & (TILE_W - 1), ty = tid >> 4;
  const int x = blockIdx.x * TILE_W + tx, y = blockIdx.y * TILE_H + ty;
  const bool in_image = x < P.w && y < P.h;
  const int gi = y * P.stride + x;
  const unsigned long long trace_t0 = P.trace ? wall_clock64() : 0ull;

  if (FUSE_PREV) {
    if (in_image && P.conv[gi] == ST_UPDATE) {
      SeedParams Pprev = P;
      Pprev.T_ref_curr = T_ref_curr_prev;
      (void)finalize_seed(Pprev, M, x, y, gi);
    }
  }
  int state = ST_BORDER;
  float mu = 0.0f, sigma_sq = 0.0f;
  if (in_image) {
    mu = P.mu[gi]; sigma_sq = P.sigma_sq[gi];
    state = seed_check(P, x, y, sigma_sq, P.a[gi], P.b[gi], SIDE);
    P.conv[gi] = state;  // final for BORDER / CONVERGED / DIVERGED; UPDATE seeds are settled by seed_finalize
  }

  const bool live = in_image && state == ST_UPDATE;
  int n_valid = 0, i_first = 0;
  float l_first = 0.0f;
  unsigned int n_steps = 0, n_evals = 0;
  float bb_x0 = INFINITY, bb_y0 = INFINITY, bb_x1 = -INFINITY, bb_y1 = -INFINITY;
  if (live) {
    const Segment seg = epipolar_segment(P, x, y, mu, sigma_sq);
    const float half = seg.half_length;
    const ValidRun run = find_valid_run(P, seg, SIDE);
    n_valid = run.n_valid; i_first = run.i_first; l_first = run.l_first;
    if (n_valid) {
      bb_x0 = fminf(run.px_first.x, run.px_last.x); bb_x1 = fmaxf(run.px_first.x, run.px_last.x);
      bb_y0 = fminf(run.px_first.y, run.px_last.y); bb_y1 = fmaxf(run.px_first.y, run.px_last.y);
    }
    if (P.stats) {  // diagnostics only: the full walk, counting what the reference would visit / evaluate
      int i = 0;
      for (float l = -half; l <= half; l += 0.7f, ++i) {
        const F2 px = F2{seg.mean.x + l * seg.dir.x, seg.mean.y + l * seg.dir.y};
        if (!px_outside(P, px, SIDE)) ++n_evals;  // NaN positions pass the reference's guard too
      }
      n_steps = static_cast<unsigned int>(i);
    }
    M.best[gi] = 0ull;
    if (n_valid > 0) {
      M.mean[gi] = make_float2(seg.mean.x, seg.mean.y);
      M.dir[gi] = make_float2(seg.dir.x, seg.dir.y);
      M.lfirst[gi] = l_first;
    }
  }
  if (in_image) M.packed[gi] = (static_cast<unsigned int>(i_first) << 16) | static_cast<unsigned int>(n_valid);

  // tile totals: work, bounding box and step-weighted centroid of the sample positions
  int tot = n_valid;
  float cw = static_cast<float>(n_valid);
  float cxw = n_valid ? cw * 0.5f * (bb_x0 + bb_x1) : 0.0f;
  float cyw = n_valid ? cw * 0.5f * (bb_y0 + bb_y1) : 0.0f;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    tot += __shfl_xor(tot, off, 64);
    bb_x0 = fminf(bb_x0, __shfl_xor(bb_x0, off, 64)); bb_y0 = fminf(bb_y0, __shfl_xor(bb_y0, off, 64));
    bb_x1 = fmaxf(bb_x1, __shfl_xor(bb_x1, off, 64)); bb_y1 = fmaxf(bb_y1, __shfl_xor(bb_y1, off, 64));
    cw += __shfl_xor(cw, off, 64); cxw += __shfl_xor(cxw, off, 64); cyw += __shfl_xor(cyw, off, 64);
  }
  if (lane == 0) {
    red_i[wave] = tot;
    red_f[wave][0] = bb_x0; red_f[wave][1] = bb_y0; red_f[wave][2] = bb_x1; red_f[wave][3] = bb_y1;
    red_f[wave][4] = cw; red_f[wave][5] = cxw; red_f[wave][6] = cyw;
  }
  if (P.stats) {  // diagnostics
    const unsigned long long s_live = wave_sum_u64(live ? 1ull : 0ull);
    const unsigned long long s_steps = wave_sum_u64(static_cast<unsigned long long>(n_steps));
    const unsigned long long s_evals = wave_sum_u64(static_cast<unsigned long long>(n_evals));
    if (lane == 0 && s_live) {
      atomicAdd(&P.stats[0], s_live);
      atomicAdd(&P.stats[1], s_steps);
      atomicAdd(&P.stats[2], s_evals);
    }
  }
  __syncthreads();
  if (tid != 0) return;
  const int total = red_i[0] + red_i[1] + red_i[2] + red_i[3];
  TileInfo ti;
  ti.total = total; ti.wx0 = 0; ti.wy0 = 0; ti.wx1 = -1; ti.wy1 = -1;
  const int tile = blockIdx.y * M.tiles_x + blockIdx.x;
  if (total > 0) {
    const float fx0 = fminf(fminf(red_f[0][0], red_f[1][0]), fminf(red_f[2][0], red_f[3][0]));
    const float fy0 = fminf(fminf(red_f[0][1], red_f[1][1]), fminf(red_f[2][1], red_f[3][1]));
    const float fx1 = fmaxf(fmaxf(red_f[0][2], red_f[1][2]), fmaxf(red_f[2][2], red_f[3][2]));
    const float fy1 = fmaxf(fmaxf(red_f[0][3], red_f[1][3]), fmaxf(red_f[2][3], red_f[3][3]));
    // texels touched by a sample at p: floor(p) - HALF .. floor(p) + HALF + 1, +1 for the replayed roundings
    ti.wx0 = max(static_cast<int>(floorf(fx0)) - HALF - 1, 0);
    ti.wy0 = max(static_cast<int>(floorf(fy0)) - HALF - 1, 0);
    ti.wx1 = min(static_cast<int>(floorf(fx1)) + HALF + 2, P.w - 1);
    ti.wy1 = min(static_cast<int>(floorf(fy1)) + HALF + 2, P.h - 1);
    const int need_w = ti.wx1 - ti.wx0 + 1, need_h = ti.wy1 - ti.wy0 + 1;
    if (need_w > WS || need_h > WROWS) {
      // a few seeds wandered off: centre the window on where most samples are; the rest read global memory
      const float w_sum = red_f[0][4] + red_f[1][4] + red_f[2][4] + red_f[3][4];
      const float cx = (red_f[0][5] + red_f[1][5] + red_f[2][5] + red_f[3][5]) / w_sum;
      const float cy = (red_f[0][6] + red_f[1][6] + red_f[2][6] + red_f[3][6]) / w_sum;
      if (need_w > WS) {
        ti.wx0 = min(max(static_cast<int>(cx) - WS / 2, 0), max(P.w - WS, 0));
        ti.wx1 = min(ti.wx0 + WS - 1, P.w - 1);
      }
      if (need_h > WROWS) {
        ti.wy0 = min(max(static_cast<int>(cy) - WROWS / 2, 0), max(P.h - WROWS, 0));
        ti.wy1 = min(ti.wy0 + WROWS - 1, P.h - 1);
      }
    }
  }
  M.tiles[tile] = ti;  // no atomics here: seed_plan turns the per-tile totals into the unit queue
  M.tile_plan[tile] = static_cast<unsigned int>(total);
  if (P.trace) trace_record(P.trace, tile, trace_t0, wall_clock64());
}

// ------------------------------------------------------------------------------------------------
// stage 1b: one workgroup turns the per-tile totals into the work-unit list (exclusive scan)
constexpr int PLAN_THREADS = 256;
constexpr int PLAN_TILES_IN_REGS = 8;  // per thread: 2048 tiles (e.g. 640x480 -> 1200) stay in registers; larger images re-read L2
__global__ __launch_bounds__(PLAN_THREADS) void seed_plan_kernel(MatcherArgs M, int n_tiles, int target_units) {
  __shared__ int wave_tot[PLAN_THREADS / 64];
  __shared__ int red_s[PLAN_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long trace_t0 = M.trace ? wall_clock64() : 0ull;
  // thread t owns the contiguous run of tiles [t*c, (t+1)*c): one batch of independent loads, then two block-wide
  // reductions (total work -> unit size; exclusive scan of the unit counts -> unit list)
  const int c_tiles = (n_tiles + PLAN_THREADS - 1) / PLAN_THREADS;
  const int t_first = tid * c_tiles;
  int tot[PLAN_TILES_IN_REGS];
  int items = 0;
#pragma unroll
  for (int q = 0; q < PLAN_TILES_IN_REGS; ++q) {
    const int t = t_first + q;
    tot[q] = (q < c_tiles && t < n_tiles) ? static_cast<int>(M.tile_plan[t]) : 0;
    items += tot[q];
  }
  for (int q = PLAN_TILES_IN_REGS; q < c_tiles; ++q) {
    const int t = t_first + q;
    if (t >= n_tiles) break;
    items += static_cast<int>(M.tile_plan[t]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) items += __shfl_xor(items, off, 64);
  if (lane == 0) red_s[wave] = items;
  __syncthreads();
  items = red_s[0] + red_s[1] + red_s[2] + red_s[3];
  // unit size: one round of the 256 lanes when there is little work (latency: more workgroups, each short), up to
  // MAX_UNIT_ROUNDS rounds when there is plenty (amortises the per-unit staging).  Decided here, on the device, from this
  // frame's own total: a host that enqueues frames in bursts runs many frames ahead of anything it could read back.
  int unit_rounds = (items + target_units * TILE_PIX - 1) / (target_units * TILE_PIX);
  unit_rounds = min(max(unit_rounds, 1), MAX_UNIT_ROUNDS);
  const int unit_items = unit_rounds * TILE_PIX;
  int mine = 0;
#pragma unroll
  for (int q = 0; q < PLAN_TILES_IN_REGS; ++q) mine += units_of(tot[q], unit_rounds);
  for (int q = PLAN_TILES_IN_REGS; q < c_tiles; ++q) {
    const int t = t_first + q;
    if (t >= n_tiles) break;
    mine += units_of(static_cast<int>(M.tile_plan[t]), unit_rounds);
  }
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int wave_off = 0, block_tot = 0;
#pragma unroll
  for (int wv = 0; wv < PLAN_THREADS / 64; ++wv) {
    const int v = wave_tot[wv];
    wave_off += wv < wave ? v : 0;
    block_tot += v;
  }
  if (tid == 0) {
    M.queue[0] = static_cast<unsigned int>(block_tot);
    M.queue[1] = 0u;  // hand-out counter of this frame's search
    M.queue[5] = static_cast<unsigned int>(unit_items);
  }
  int base = wave_off + incl - mine;
#pragma unroll
  for (int q = 0; q < PLAN_TILES_IN_REGS; ++q) {
    const int n_u = units_of(tot[q], unit_rounds);  // 0 for tiles beyond this thread's run
    for (int u = 0; u < n_u; ++u) reinterpret_cast<uint2*>(M.units)[base + u] = make_uint2(static_cast<unsigned int>(t_first + q), static_cast<unsigned int>(u * unit_items));
    base += n_u;
  }
  for (int q = PLAN_TILES_IN_REGS; q < c_tiles; ++q) {
    const int t = t_first + q;
    if (t >= n_tiles) break;
    const int n_u = units_of(static_cast<int>(M.tile_plan[t]), unit_rounds);
    for (int u = 0; u < n_u; ++u) reinterpret_cast<uint2*>(M.units)[base + u] = make_uint2(static_cast<unsigned int>(t), static_cast<unsigned int>(u * unit_items));
    base += n_u;
  }
  if (M.trace && tid == 0)  // the unit size rides in the top byte of the end time stamp
    trace_record(M.trace, n_tiles, trace_t0, wall_clock64() | (static_cast<unsigned long long>(unit_rounds) << 56));
}

// ------------------------------------------------------------------------------------------------
// stage 2: persistent search workgroups
template <int SIDE, int WS, int WROWS>
struct SearchSmem {
  static constexpr int HALF = SIDE / 2;
  static constexpr int REF_W = TILE_W + SIDE - 1, REF_H = TILE_H + SIDE - 1;
  float win[WROWS * WS];
  float ref[REF_H * REF_W];
  float mean_x[TILE_PIX], mean_y[TILE_PIX], dir_x[TILE_PIX], dir_y[TILE_PIX];
  float l_first[TILE_PIX];
  float sum_templ[TILE_PIX], denom[TILE_PIX];
  int prefix[TILE_PIX + 1];
  unsigned short i_first[TILE_PIX];
  int red_i[4];
  unsigned int unit[2];
};

template <int SIDE, int WS, int WROWS>
__global__ __launch_bounds__(TILE_PIX) void seed_search_kernel(SeedParams P, MatcherArgs M) {
  using Smem = SearchSmem<SIDE, WS, WROWS>;
  constexpr int HALF = SIDE / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& S = *reinterpret_cast<Smem*>(smem_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tx = tid & (TILE_W - 1), ty = tid >> 4;
  const unsigned int n_units = M.queue[0];
  const int unit_items = static_cast<int>(M.queue[5]);
  unsigned int n_path[3] = {0, 0, 0};
  unsigned long long t_stage = 0, t_search = 0;
  unsigned int my_units = 0;
  const unsigned long long trace_t0 = P.trace ? wall_clock64() : 0ull;

  // unit blockIdx.x is ours for free; further units come from the shared counter (one returning atomic each)
  unsigned int u = blockIdx.x;
  for (;; ) {
    const long long t0 = P.stats ? clock64() : 0;
    if (u >= n_units) break;
    ++my_units;
    const uint2 unit = reinterpret_cast<const uint2*>(M.units)[u];
    const int tile = static_cast<int>(unit.x);
    const int first = static_cast<int>(unit.y);
    const TileInfo ti = M.tiles[tile];
    const int tile_y = tile / M.tiles_x, tile_x = tile - tile_y * M.tiles_x;
    const int x0 = tile_x * TILE_W, y0 = tile_y * TILE_H;
    const int x = x0 + tx, y = y0 + ty;
    const bool in_image = x < P.w && y < P.h;
    const int gi = y * P.stride + x;

    // per-seed descriptors of the tile -> LDS; exclusive prefix of the step counts
    const unsigned int packed = in_image ? M.packed[gi] : 0u;
    const int n_valid = static_cast<int>(packed & 0xffffu);
    if (n_valid > 0) {
      const float2 m = M.mean[gi], d = M.dir[gi];
      S.mean_x[tid] = m.x; S.mean_y[tid] = m.y; S.dir_x[tid] = d.x; S.dir_y[tid] = d.y;
      S.l_first[tid] = M.lfirst[gi];
      S.sum_templ[tid] = P.sum_templ[gi]; S.denom[tid] = P.denom[gi];
    }
    S.i_first[tid] = static_cast<unsigned short>(packed >> 16);
    int incl = n_valid;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off, 64);
      if (lane >= off) incl += v;
    }
    if (lane == 63) S.red_i[wave] = incl;

    // stage the window of the current image (batches of independent loads) and the reference tile
    const int wx0 = ti.wx0, wy0 = ti.wy0, wx1 = ti.wx1, wy1 = ti.wy1;
    const int ww = wx1 - wx0 + 1, wh = wy1 - wy0 + 1;
    {
      const int n_el = ww * wh;
      const float inv_ww = 1.0f / static_cast<float>(ww);
      // all of a thread's loads are issued before its first LDS store (chunks of 8, skipped wave-uniformly beyond the
      // window's end): one memory round trip per window, two for the largest, instead of one per 8 texels
      constexpr int BATCH = 8;  // 16 or 32 in flight per thread measured slower (register pressure)
      for (int e0 = tid; e0 < n_el; e0 += TILE_PIX * BATCH) {
        float v[BATCH];
        int dst[BATCH];
        const int e_wg = e0 - tid;  // same for the whole workgroup
#pragma unroll
        for (int c = 0; c < BATCH; c += 8) {
          if (e_wg + c * TILE_PIX < n_el) {
#pragma unroll
            for (int q = c; q < c + 8; ++q) {
              const int e = e0 + q * TILE_PIX;
              int r = static_cast<int>(static_cast<float>(e) * inv_ww);  // e / ww, fixed up below (e < 2^14)
              int cc = e - r * ww;
              if (cc < 0) { --r; cc += ww; } else if (cc >= ww) { ++r; cc -= ww; }
              dst[q] = r * WS + cc;
              v[q] = e < n_el ? P.cur[(wy0 + r) * P.cur_stride + wx0 + cc] : 0.0f;
            }
          }
        }
#pragma unroll
        for (int c = 0; c < BATCH; c += 8) {
          if (e_wg + c * TILE_PIX < n_el) {
#pragma unroll
            for (int q = c; q < c + 8; ++q)
              if (e0 + q * TILE_PIX < n_el) S.win[dst[q]] = v[q];
          }
        }
      }
      for (int i = tid; i < Smem::REF_H * Smem::REF_W; i += TILE_PIX) {
        const int ry = i / Smem::REF_W, rx = i - ry * Smem::REF_W;
        S.ref[i] = P.ref[clampi(y0 - HALF + ry, 0, P.h - 1) * P.stride + clampi(x0 - HALF + rx, 0, P.w - 1)];
      }
    }
    __syncthreads();
    int wave_off = 0;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) wave_off += (wv < wave) ? S.red_i[wv] : 0;
    S.prefix[tid] = wave_off + incl - n_valid;  // exclusive
    if (tid == 0) S.prefix[TILE_PIX] = ti.total;
    __syncthreads();
    const long long t1 = P.stats ? clock64() : 0;

    // the unit's (seed, step) work items, dealt round-robin to the lanes
    const int last = min(first + unit_items, ti.total);
    for (int k0 = first; k0 < last; k0 += TILE_PIX) {
      const int k = k0 + tid;
      int p = -1;
      unsigned long long key = 0ull;
      if (k < last) {
        int lo = 0, hi = TILE_PIX;  // last p with prefix[p] <= k
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int mid = (lo + hi) >> 1;
          if (S.prefix[mid] <= k) lo = mid; else hi = mid;
        }
        p = lo;
        const int j = k - S.prefix[p];
        const float l = replay_l(S.l_first[p], j);  // the reference accumulates l; replay it
        const F2 px = F2{S.mean_x[p] + l * S.dir_x[p], S.mean_y[p] + l * S.dir_y[p]};
        const int ptx = p & (TILE_W - 1), pty = p >> 4;
        int path = 0;
        const float ncc = ncc_at<SIDE, WS>(P, px, S.win, wx0, wy0, wx1, wy1, S.ref + pty * Smem::REF_W + ptx, Smem::REF_W,
                                          S.sum_templ[p], S.denom[p], path);
        n_path[0] += path == 0; n_path[1] += path == 1; n_path[2] += path == 2;
        if (ncc > -1.0f) {  // NaN and anything the reference's "ncc > best_ncc" would never accept are dropped
          const unsigned int step = static_cast<unsigned int>(S.i_first[p]) + static_cast<unsigned int>(j);
          key = (static_cast<unsigned long long>(orderable_f32(ncc + 0.0f)) << 32) | (0xffffffffu - step);
        }
      }
      // seeds occupy runs of consecutive lanes: segmented max towards the run's first lane, then one atomic per run
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long okey = __shfl_down(key, off, 64);
        const int op = __shfl_down(p, off, 64);
        if (lane + off < 64 && op == p && okey > key) key = okey;
      }
      const int prev_p = __shfl_up(p, 1, 64);
      if (p >= 0 && key != 0ull && (lane == 0 || prev_p != p)) {
        const int py = y0 + (p >> 4), pxx = x0 + (p & (TILE_W - 1));
        atomicMax(&M.best[py * P.stride + pxx], key);
      }
    }
    if (P.stats) {
      const long long t2 = clock64();
      t_stage += static_cast<unsigned long long>(t1 - t0);
      t_search += static_cast<unsigned long long>(t2 - t1);
    }
    if (n_units <= gridDim.x) break;  // light frame: every unit had its own workgroup, nothing to hand out
    __syncthreads();  // this unit's LDS is no longer read
    if (tid == 0) S.unit[0] = gridDim.x + atomicAdd(&M.queue[1], 1u);
    __syncthreads();
    u = S.unit[0];
  }
  if (P.stats) {
    const unsigned long long s0 = wave_sum_u64(n_path[0]), s1 = wave_sum_u64(n_path[1]), s2 = wave_sum_u64(n_path[2]);
    if (lane == 0) {
      if (s0) atomicAdd(&P.stats[3], s0);
      if (s1) atomicAdd(&P.stats[4], s1);
      if (s2) atomicAdd(&P.stats[5], s2);
    }
    if (tid == 0) {
      atomicAdd(&P.stats[8], t_stage);
      atomicAdd(&P.stats[9], t_search);
      atomicMax(&P.stats[11], t_stage + t_search);
      atomicAdd(&P.stats[13], static_cast<unsigned long long>(my_units));
    }
  }
  if (P.trace && tid == 0) {
    if (blockIdx.x < TRACE_MAX_SEARCH_WGS)
      trace_record(P.trace, M.tiles_x * ((P.h + TILE_H - 1) / TILE_H) + 1 + static_cast<int>(blockIdx.x), trace_t0, wall_clock64());
  }
}


// setup (+ the deferred finalisation of the previous frame when fuse_prev) -> plan -> search.  The frame's own
// finalisation is NOT launched: the caller either fuses it into the next frame's setup or runs launch_seed_finalize.
template <int SIDE, int WS, int WROWS>
inline hipError_t launch_seed_pipeline(const SeedParams& P, MatcherWorkspace& ws, hipStream_t stream, int num_cus, bool& attr_set,
                                       bool fuse_prev, const Pose& T_ref_curr_prev) {
  using Smem = SearchSmem<SIDE, WS, WROWS>;
  MatcherArgs M = matcher_args(ws);
  M.trace = P.trace;
  // Persistent grid: as many workgroups as fit the chip; those without a unit leave within a microsecond.
  // the LDS the device really has per CU (160 KB on gfx950; read at handle creation): a window that does not fit is an error the
  // caller sees (the default matcher's 38 KB always fits), never a silent launch failure
  if (sizeof(Smem) > static_cast<size_t>(ws.lds_bytes)) return hipErrorInvalidConfiguration;
  const int by_lds = static_cast<int>(static_cast<size_t>(ws.lds_bytes) / sizeof(Smem));
  const int wg_per_cu = by_lds < 4 ? by_lds : 4;  // >4 x 256 threads gain nothing at this register count
  const int resident = num_cus * wg_per_cu;
  const dim3 tiles(ws.tiles_x, ws.tiles_y);
  if (fuse_prev) hipLaunchKernelGGL((seed_setup_kernel<SIDE, WS, WROWS, true>), tiles, dim3(TILE_PIX), 0, stream, P, M, T_ref_curr_prev);
  else hipLaunchKernelGGL((seed_setup_kernel<SIDE, WS, WROWS, false>), tiles, dim3(TILE_PIX), 0, stream, P, M, T_ref_curr_prev);
  hipLaunchKernelGGL(seed_plan_kernel, dim3(1), dim3(PLAN_THREADS), 0, stream, M, ws.tiles_x * ws.tiles_y, resident);
  auto search = seed_search_kernel<SIDE, WS, WROWS>;
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(search), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(sizeof(Smem)));
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int grid = resident;
  hipLaunchKernelGGL(search, dim3(grid), dim3(TILE_PIX), sizeof(Smem), stream, P, M);
  return hipGetLastError();
}

// Two LDS window geometries for the search kernel (odd row strides spread the LDS banks):
//   large 133 x 104 texels (default): holds a 16x16 tile's worst case (100 px search, any direction)
//   small  69 x 64 texels: searches up to ~40 px; selectable for experiments (RMD_HIP_OPT_WINDOW = 1)
// Samples outside a tile's window are read from global memory.  Picking the geometry per frame from a read-back of the
// previous frames' window sizes was tried and removed: a host that submits frames in bursts decides from stale numbers,
// and at the search kernel's register count the small window does not buy more resident waves.
template <int SIDE>
inline hipError_t launch_seed_update_tile(const SeedParams& P, MatcherWorkspace& ws, hipStream_t stream, int num_cus,
                                          int force_window, bool fuse_prev, const Pose& T_ref_curr_prev) {
  if (force_window == 1) return launch_seed_pipeline<SIDE, 69, 64>(P, ws, stream, num_cus, ws.attr_set_small, fuse_prev, T_ref_curr_prev);
  return launch_seed_pipeline<SIDE, 133, 104>(P, ws, stream, num_cus, ws.attr_set_large, fuse_prev, T_ref_curr_prev);
}

}  // namespace rmdk

#endif  // RMD_MATCHER_R01_HPP
