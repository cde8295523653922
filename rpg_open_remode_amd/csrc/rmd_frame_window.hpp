// The LDS window of the current image that the search kernel samples from (rmd_frame_search.hpp) and the setup kernel sizes
// (rmd_frame_setup.hpp): the workgroup's shared-memory block, the sheared band and its row table, the window policy's reductions,
// LDS-direct staging, and ONE NCC evaluation at a run-time window (ncc_at_dyn; the arithmetic block itself is rmd_matcher.hpp's).  Who
// writes / reads what:
//   FrameSmem::win, row_start   written by frame_stage_window (every wave its own rows, LDS-direct: drained with drain_vmem() before the
//                               barrier that publishes them), read by ncc_at_dyn of every lane after that barrier
//   FrameSmem::red              per-wave partial results of the block_bbox reductions, published by the barrier between block_bbox and
//                               block_bbox_read
// Part of the seed update of one frame: rmd_frame.hpp.
#ifndef RMD_FRAME_WINDOW_HPP
#define RMD_FRAME_WINDOW_HPP

#include "rmd_matcher.hpp"
#include "rmd_lab.hpp"

namespace rmdk {

constexpr int FR_MIN_WAVES = 3;   // __launch_bounds__ of the search kernel (five workgroups per CU at 96 VGPRs: measured, slower -- LAB.md)
constexpr int FR_WIN_CAP = 5632;  // texels of the LDS window (22 KB); any shape with window_stride(width) * height <= FR_WIN_CAP
constexpr int FR_UNIT_ROUNDS = 4, FR_UNIT_ITEMS = FR_UNIT_ROUNDS * TILE_PIX;
constexpr int FR_MAX_ROWS = 255;   // rows of an LDS window (one entry of the row table each; 8 bits in a unit entry)
constexpr int FR_MAX_WIDTH = 511;  // texels per window row (9 bits in a unit entry)
// The LDS window is a SHEARED band, not a box: window row q (image row y0 + q) holds the image columns x0 + shear_of(y0 + q - yref, m) ...
// + ww - 1, where m / 2048 is the band's slope in columns per row (|m| < 16384) and yref the tile's first row.  The samples of a tile lie
// along nearly parallel epipolar segments up to max_extent pixels long: a box around a diagonal bundle of 100-pixel segments is 97 x 68
// texels -- more than the window holds -- although the bundle itself covers a band 16-35 texels wide (tests/sim_window_policy.py: on the
// benchmark sequence three work units in four of the light updates 24..55 / ~100 / ~160 had no window that held them: 8-15 % of those
// updates' evaluations read their texels from L2, a round took 7.3 instead of 4.6 us).  m = 0 is the box.
constexpr int FR_SHEAR_BITS = 11;
constexpr unsigned int TILE_WANTS_BAND = 0x10000u;  // flag in a tile's word of MatcherArgs::tile_live (its low half: seeds in state UPDATE)
RMDK_D int shear_of(int q, int m) { return (q * m) >> FR_SHEAR_BITS; }  // (arithmetic shift: floor, q may be negative)
// Row stride of a window of `ww` texels per row: odd, so that the rows of a vertical bundle of samples start in different LDS banks.  (Lab
// builds try others: LAB_WINDOW_STRIDE, rmd_lab.hpp.)
RMDK_D constexpr int window_stride(int ww) { return LAB_WINDOW_STRIDE(ww); }

// ---- agent-scope accessors (global address space, sc1) ---------------------------------------------------------------
RMDK_D unsigned int ld_agent(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RMDK_D unsigned long long ld_agent(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RMDK_D unsigned int ld_system(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
RMDK_D unsigned long long ld_system(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED,
    __HIP_MEMORY_SCOPE_SYSTEM); }
RMDK_D void st_agent(unsigned int* p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RMDK_D void st_agent(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RMDK_D void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int SIDE>
struct FrameSmem {
  static constexpr int REF_W = TILE_W + SIDE - 1, REF_H = TILE_H + SIDE - 1;
  float win[FR_WIN_CAP];
  float ref[REF_H * REF_W];
  float mean_x[TILE_PIX], mean_y[TILE_PIX], dir_x[TILE_PIX], dir_y[TILE_PIX];
  float l_first[TILE_PIX];
  float sum_templ[TILE_PIX], denom[TILE_PIX];
  unsigned long long best[TILE_PIX];
  int prefix[TILE_PIX + 1];
  unsigned int packed[TILE_PIX];   // state << 16 | first in-image step << 8 | number of in-image steps
  int red[4][12];
  int row_start[FR_MAX_ROWS + 1];  // the window's row table: image texel (X, y0 + q) is win[row_start[q] + X] (frame_stage_window)
  alignas(16) unsigned int bcast[8];  // [0] the workgroup's next unit
  LAB_PROF(unsigned long long prof[8];)  // lab builds (rmd_lab.hpp): per-phase ticks of the workgroup
};

// do all texels an IRREGULAR footprint at px may touch -- the regular one widened by a texel on every side -- lie inside the window?
template <int SIDE>
RMDK_D bool irregular_in_window(F2 px, const int* __restrict__ row_start, int ws, int wy0, int rows, int ww) {
  constexpr int OFFSET = -SIDE / 2;
  if (!(isfinite(px.x) && isfinite(px.y))) return false;
  const int x_lo = static_cast<int>(floorf(px.x)) + OFFSET - 1, q_lo = static_cast<int>(floorf(px.y)) + OFFSET - 1 - wy0,
      q_hi = q_lo + SIDE + 2;
  if (q_lo < 0 || q_hi >= rows) return false;
  const int c_a = x_lo + row_start[q_lo] - q_lo * ws, c_b = x_lo + row_start[q_hi] - q_hi * ws;
  return min(c_a, c_b) >= 0 && max(c_a, c_b) + SIDE + 2 < ww;
}

// One NCC evaluation at px; the LDS window has a run-time row stride.  Two sources for the current-image samples, same
// arithmetic in both: the LDS window when the (regular) footprint lies inside it -- practically always, the window is cut to
// the work -- and per-sample fetches from L2 otherwise (irregular footprints from the replayed roundings, ~1e-6 of the
// steps, and samples outside a clamped window).  The second path is deliberately compact (rolled loops): the kernel's code
// has to stay resident in the instruction cache while workgroups are in all of its phases at once.
template <int SIDE>
RMDK_D float ncc_at_dyn(const SeedParams& P, F2 px, const float* __restrict__ win, const int* __restrict__ row_start, int ws, int wy0,
    int rows, int ww,
                        const float* __restrict__ ref_patch, int ref_stride, float sum_templ, float denom, unsigned int& n_fallback) {
  constexpr int OFFSET = -SIDE / 2;
  constexpr float AREA = static_cast<float>(SIDE * SIDE);
  float sum_img = 0.0f, sum_img_sq = 0.0f, sum_img_templ = 0.0f;
  int ix[SIDE], iy[SIDE];
  float ax[SIDE], ay[SIDE];
  bool reg_x = true, reg_y = true;
  // wave-uniform branches, one per axis (a wave whose x positions straddle a power of two replays the roundings of x only).  The rows
  // first: they say where in the window's row table the footprint starts
  if (__all(axis_is_uniform<SIDE>(px.y))) {
    const float fy = floorf(px.y);
    const float wy = px.y - fy;
    iy[0] = static_cast<int>(fy) + OFFSET;
#pragma unroll
    for (int k = 0; k < SIDE; ++k) ay[k] = wy;
  } else {
    reg_y = axis_params<SIDE>(px.y, iy, ay);
  }
  // Inside the window: texel rows q0 .. q0 + SIDE exist and the footprint's columns lie inside the first and the last of them (the shear is
  // monotone, so inside every row between).  The window's row table says where a texel row starts in the LDS (a sheared band has no
  // constant row stride): three entries are read here -- first row, second row, last row -- at a clamped index whatever q0 is (no branch
  // around the reads), requested BEFORE the columns' parameters are worked out; the evaluation block reads the others two rows ahead of
  // their use. (All SIDE + 1 entries held in registers across the block cost 15 VGPRs: 116 instead of 101, and at 120 allocated registers
  // four search waves leave no room on a SIMD for a setup wave of another stream group of a batch -- a batch of 8 lost 5 %.)
  const int q0 = iy[0] - wy0;
  const int qc = max(0, min(q0, rows - 1 - SIDE));
  const int qws = __mul24(qc, ws);  // (qc < 256, ws < 512: the full-rate 24-bit multiply)
  int off_first = row_start[qc], off_second = row_start[qc + 1], off_last = row_start[qc + SIDE];
  __builtin_amdgcn_sched_barrier(0);
  if (__all(axis_is_uniform<SIDE>(px.x))) {
    const float fx = floorf(px.x);
    const float wx = px.x - fx;
    ix[0] = static_cast<int>(fx) + OFFSET;
#pragma unroll
    for (int k = 0; k < SIDE; ++k) ax[k] = wx;
  } else {
    reg_x = axis_params<SIDE>(px.x, ix, ax);
  }
  asm volatile("" : "+v"(off_first), "+v"(off_second), "+v"(off_last));
  const bool regular = reg_x && reg_y;
  const int c_first = ix[0] + off_first - qws, c_last = ix[0] + off_last - (qws + SIDE * ws);  // columns within their window rows
  const bool in_window = regular && q0 == qc && min(c_first, c_last) >= 0 && max(c_first, c_last) + SIDE < ww;
  if (in_window) {
    ncc_sums_lds_pipelined<SIDE>(win + ix[0], row_start + q0, off_first, off_second, ax, ay, ref_patch, ref_stride, sum_img, sum_img_sq,
        sum_img_templ);
  } else if (regular) {
    // outside the window (a clamped window, a seed that wandered off): the same separable filter on texel rows from L2, fully
    // unrolled so that the loads overlap -- a rolled loop costs a memory round trip per row and made the few such evaluations
    // the longest thing in a late frame
    ++n_fallback;
    ncc_sums_regular<SIDE, 0>(P.cur + iy[0] * P.cur_stride + ix[0], P.cur_stride, ax, ay, ref_patch, ref_stride, sum_img, sum_img_sq,
                              sum_img_templ);
  } else if (irregular_in_window<SIDE>(px, row_start, ws, wy0, rows, ww)) {
    // An irregular footprint (the replayed roundings moved a column or a row by a texel: ~1e-6 of the steps) whose neighbourhood lies
    // inside the window: sample by sample like the path below, but from the LDS.  From L2 such an evaluation is 81 x 4 dependent round
    // trips -- 15-20 us in ONE lane, and with a quarter of a million evaluations per light update every third or fourth update has one.
#pragma unroll 1
    for (int m = 0; m < SIDE; ++m) {
      const float cy = px.y + static_cast<float>(OFFSET + m) + 0.5f;
      const float yb = cy - 0.5f, fj = floorf(yb), b = yb - fj;  // (tex_linear_global's arithmetic, the row part once per row)
      const int q = static_cast<int>(fj) - wy0;
      const float* const r0 = win + row_start[q];
      const float* const r1 = win + row_start[q + 1];
#pragma unroll 3
      for (int k = 0; k < SIDE; ++k) {
        const float cx = px.x + static_cast<float>(OFFSET + k) + 0.5f;
        const float xb = cx - 0.5f, fi = floorf(xb), a = xb - fi;
        const int i0 = static_cast<int>(fi);
        const float t00 = r0[i0], t10 = r0[i0 + 1], t01 = r1[i0], t11 = r1[i0 + 1];
        const float h0 = a == 0.0f ? t00 : rmd_lerp(a, t00, t10);
        const float h1 = a == 0.0f ? t01 : rmd_lerp(a, t01, t11);
        const float img = b == 0.0f ? h0 : rmd_lerp(b, h0, h1);
        const float templ = ref_patch[m * ref_stride + k];
        sum_img += img;
        sum_img_sq += img * img;
        sum_img_templ += img * templ;
      }
    }
  } else {
    ++n_fallback;
#pragma unroll 1
    for (int m = 0; m < SIDE; ++m) {
      const float cy = px.y + static_cast<float>(OFFSET + m) + 0.5f;
#pragma unroll 1
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

// The band (see shear_of) around one sample position: a sample at p touches texels floor(p) - HALF .. floor(p) + HALF + 1 in both axes (one
// more for the replayed roundings, another to spare); in band coordinates u = X - shear_of(Y - yref, m) its columns span
// [xlo - max shear, xhi - min shear] over its rows.  Accumulates into (u0, y0, u1, y1).
RMDK_D void band_add_point(float px, float py, int half, int m, int yref, int& u0, int& y0, int& u1, int& y1) {
  const int fx = static_cast<int>(floorf(px)), fy = static_cast<int>(floorf(py));
  const int ylo = fy - half - 1, yhi = fy + half + 2;
  const int sa = shear_of(ylo - yref, m), sb = shear_of(yhi - yref, m);
  u0 = min(u0, fx - half - 1 - max(sa, sb)); u1 = max(u1, fx + half + 2 - min(sa, sb));
  y0 = min(y0, ylo); y1 = max(y1, yhi);
}
constexpr int BAND_NONE_LO = 0x3fffffff, BAND_NONE_HI = -0x3fffffff;  // identities of the reductions below (band columns may be negative)

// Band of the samples of steps [j0, j1] (relative to the first in-image step) of this lane's seed.  The positions come from
// l = l_first + 0.7 j, which is within 1e-3 px of the replayed value; u is linear along a segment, so its two ends bound it.  A band
// that misses a sample costs speed, not correctness (ncc_at_dyn reads L2 then).  Rows are clamped to the image.
template <int SIDE>
RMDK_D void seed_range_band(const SeedParams& P, const FrameSmem<SIDE>& S, int tid, bool has, int j0, int j1, int m, int yref, int& u0,
    int& y0, int& u1, int& y1) {
  constexpr int HALF = SIDE / 2;
  u0 = BAND_NONE_LO; y0 = BAND_NONE_LO; u1 = BAND_NONE_HI; y1 = BAND_NONE_HI;
  if (!has) return;
  const float lf = S.l_first[tid], mx = S.mean_x[tid], my = S.mean_y[tid], dx = S.dir_x[tid], dy = S.dir_y[tid];
  const float la = lf + 0.7f * static_cast<float>(j0), lb = lf + 0.7f * static_cast<float>(j1);
  band_add_point(mx + la * dx, my + la * dy, HALF, m, yref, u0, y0, u1, y1);
  band_add_point(mx + lb * dx, my + lb * dy, HALF, m, yref, u0, y0, u1, y1);
  y0 = max(y0, 0); y1 = min(y1, P.h - 1);
}

// min / max of four ints over the workgroup (all 256 threads call; result uniform)
template <int SIDE>
RMDK_D void block_bbox(FrameSmem<SIDE>& S, int tid, int& x0, int& y0, int& x1, int& y1, int slot) {
  x0 = wave_reduce_i32<WaveMin>(x0); y0 = wave_reduce_i32<WaveMin>(y0);
  x1 = wave_reduce_i32<WaveMax>(x1); y1 = wave_reduce_i32<WaveMax>(y1);
  const int wave = tid >> 6;
  if ((tid & 63) == 0) { S.red[wave][slot] = x0; S.red[wave][slot + 1] = y0; S.red[wave][slot + 2] = x1; S.red[wave][slot + 3] = y1; }
}
template <int SIDE>
RMDK_D void block_bbox_read(const FrameSmem<SIDE>& S, int& x0, int& y0, int& x1, int& y1, int slot) {
  x0 = min(min(S.red[0][slot], S.red[1][slot]), min(S.red[2][slot], S.red[3][slot]));
  y0 = min(min(S.red[0][slot + 1], S.red[1][slot + 1]), min(S.red[2][slot + 1], S.red[3][slot + 1]));
  x1 = max(max(S.red[0][slot + 2], S.red[1][slot + 2]), max(S.red[2][slot + 2], S.red[3][slot + 2]));
  y1 = max(max(S.red[0][slot + 3], S.red[1][slot + 3]), max(S.red[2][slot + 3], S.red[3][slot + 3]));
}
RMDK_D bool window_fits(int u0, int y0, int u1, int y1) {  // inclusive band columns and rows
  const int ww = u1 - u0 + 1, rows = y1 - y0 + 1;
  return u1 >= u0 && y1 >= y0 && rows <= FR_MAX_ROWS && ww <= FR_MAX_WIDTH && window_stride(ww) * rows <= FR_WIN_CAP;
}

// The LDS window of the current image, workgroup-uniform.
struct FrameWindow {
  int x0, y0;    // window row q = image row y0 + q; its LDS column 0 = image column x0 + shear_of(y0 + q - yref, m)
  int ww, rows;  // texels per row, rows
  int ws;        // row stride in LDS (window_stride(ww))
  int m, yref;   // the band's shear (m / 2048 columns per row; 0: a box) and the image row it is counted from (the tile's first row)
  bool valid;    // staged and covering every sample of the tile in LDS
  RMDK_D void set(int u0, int y0_, int u1, int y1) { x0 = u0; y0 = y0_; ww = u1 - u0 + 1; rows = y1 - y0_ + 1; ws = window_stride(ww); }
  RMDK_D void clear() { x0 = y0 = 0; ww = rows = 0; ws = 1; valid = false; }
};

// A band that does not fit the LDS window, cut down around its centre (the evaluations whose footprint falls outside read L2).  The cut
// favours a shape the row-wise staging below fetches with few instructions: at most 64 columns -- one column chunk, every lane of a row's
// load in use -- by up to 86 rows when the band is tall, the full width (a few chunks of a few rows) when it is flat.
RMDK_D void clamp_window(FrameWindow& W) {
  constexpr int FLAT = 43;  // rows: (128 | 1) * 43 <= FR_WIN_CAP
  const int ww = W.ww, wh = W.rows;
  const int nw = wh > FLAT ? min(ww, 64) : min(ww, min((FR_WIN_CAP / wh - 1) | 1, FR_MAX_WIDTH));
  const int nh = min(min(wh, FR_WIN_CAP / window_stride(nw)), FR_MAX_ROWS);
  W.x0 += (ww - nw) / 2; W.y0 += (wh - nh) / 2;  // (x0 is the band's column origin at ANY row: the shear is anchored at yref, not at y0)
  W.ww = nw; W.rows = nh; W.ws = window_stride(nw);
}

// Stage the band W of the current image into the LDS window, ROW-WISE and LDS-DIRECT: wave v takes rows v, v + 4, ...; one
// global_load_lds_dword per row and 64-column chunk brings 64 consecutive texels straight into the window (the instruction writes to a
// wave-uniform LDS base + lane x 4 bytes: exactly a row of the window; lanes past the row's end -- or outside the image, where a band may
// reach but no footprint does -- are masked out), no vector register and no ds_write in between -- so ALL rows of a wave are in flight
// together and the window arrives in ONE memory round trip whatever its shape; the row's first column is scalar arithmetic.  Also writes
// the window's row table.
// No barrier; the loads are still in flight when this returns.  A wave reads window rows that OTHER waves transferred, so every wave drains
// its own transfers (drain_vmem: s_waitcnt vmcnt(0)) before the workgroup barrier that precedes the first read: a workgroup-scope release
// only guarantees lgkmcnt(0), and the compiler tracks LDS-direct transfers per wave.  tests/test_kernel_budget.py checks the disassembly.
template <int SIDE>
RMDK_D void frame_stage_window(const SeedParams& P, FrameSmem<SIDE>& S, int tid, const FrameWindow& W) {
  typedef const __attribute__((address_space(1))) float* gptr_t;
  typedef __attribute__((address_space(3))) float* lptr_t;
  constexpr int WAVES = TILE_PIX / 64;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t stride = static_cast<size_t>(P.cur_stride);
  // Does the band stay inside the image's columns on every row?  (The shear is monotone: its first and last row decide.)  Nearly always --
  // and then a row is scalar arithmetic plus ONE vector instruction, the load itself: the lane's column offset is the same on every row.
  const int xs_a = W.x0 + shear_of(W.y0 - W.yref, W.m), xs_b = W.x0 + shear_of(W.y0 + W.rows - 1 - W.yref, W.m);
  const bool inside = min(xs_a, xs_b) >= 0 && max(xs_a, xs_b) + W.ww <= P.w;
  for (int c0 = 0; c0 < W.ww; c0 += 64) {  // (uniform: at most eight column chunks, nearly always one)
    const int c = c0 + lane;
    if (c < W.ww) {
      if (inside) {
        const float* row = P.cur + static_cast<size_t>(W.y0 + wave) * stride;  // scalar
        float* dst = S.win + wave * W.ws + c0;                                 // uniform over the wave
        for (int r = wave; r < W.rows; r += WAVES, row += WAVES * stride, dst += WAVES * W.ws) {
          const int xs = W.x0 + shear_of(W.y0 + r - W.yref, W.m);  // scalar
          __builtin_amdgcn_global_load_lds((gptr_t)(row + xs + c), (lptr_t)dst, 4, 0, 0);
        }
      } else {  // a band that reaches past the image's left or right edge (no footprint does): those lanes are masked out, row by row
        for (int r = wave; r < W.rows; r += WAVES) {
          const int y = W.y0 + r, xs = W.x0 + shear_of(y - W.yref, W.m), col = xs + c;
          if (col >= 0 && col < P.w)
            __builtin_amdgcn_global_load_lds((gptr_t)(P.cur + static_cast<size_t>(y) * stride + col), (lptr_t)(S.win + r * W.ws + c0), 4, 0,
                0);
        }
      }
    }
  }
  if (tid <= W.rows) S.row_start[tid] = __mul24(tid, W.ws) - (W.x0 + shear_of(W.y0 + tid - W.yref, W.m));
}

}  // namespace rmdk

#endif  // RMD_FRAME_WINDOW_HPP
