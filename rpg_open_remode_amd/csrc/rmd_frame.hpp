// The seed update of one frame -- seed_check (src/seed_check.cu:28-67), the epipolar NCC search (src/epipolar_match.cu:37-140),
// triangulation + Bayesian fusion (src/seed_update.cu:39-121, src/triangulation.cu) -- as TWO launches on one stream, for one
// sequence or for up to MAX_GROUP_SEQ independent sequences of one size at once (the path shards perfectly across sequences, and a
// single 640x480 frame cannot occupy 256 CUs):
//
//   seed_setup_compact   one workgroup per 16x16 tile (and sequence), one lane per seed: the deferred finalisation of the
//                        previous frame (decode the arg-max, triangulate, fuse) on values the lane loads anyway; state check;
//                        epipolar segment; the contiguous run of in-image search steps (find_valid_run); the per-seed search
//                        descriptor to global memory; the tile reserves its work units (1..4 rounds of 256 (seed, step) pairs)
//                        in one of 16 sharded lists with ONE returning atomic -- there is no planning step, the unit size
//                        comes from the previous frame's total work.  It also counts the seeds it found CONVERGED (per tile).
//                        A few extra workgroups bring frames that were handed over in host memory into the current-image plane.
//   seed_search_compact  persistent workgroups (4 per CU) take unit blockIdx.x of the sixteen lists read as one list and pull
//                        further units with one returning atomic each.  The LDS window of the current image (5 632 texels) is a BAND
//                        sheared along the tile's epipolar direction (a box when the samples fit one): the window of all samples
//                        of the tile when the setup kernel found one, else cut to the unit's own samples; one NCC evaluation per
//                        (seed, step) pair and lane with the separable, software-pipelined filter block of rmd_matcher.hpp; arg-max
//                        per seed with a 64-bit LDS atomic max on {orderable(ncc), ~step}, handed to the global key plane when the
//                        workgroup changes tile.  Its last workgroup adds up the per-tile CONVERGED counts and mirrors them to
//                        pinned host memory (getConvergedCount without a device synchronisation).
//
// Tiles are numbered sequence-major (tile_global = seq * n_tiles + tile): unit lists, shard counters and the persistent search
// workgroups are shared by all sequences of a launch; everything per sequence (planes, poses, the pending finalisation, the
// staged host frame) is one SeqArgs block of the kernel arguments, selected by blockIdx.z in the setup kernel and by the unit's
// tile number in the search kernel.  Results per sequence are bit-identical to that sequence run alone.
#ifndef RMD_FRAME_HPP
#define RMD_FRAME_HPP

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
// builds try others: LAB_WINDOW_STRIDE, rmd_lab.hpp; `ww | 3` and `(ww + 2) | 1` measured within the noise of `ww | 1`: profiles/r06_ab_window_stride.txt.)
RMDK_D constexpr int window_stride(int ww) { return LAB_WINDOW_STRIDE(ww); }

// ---- agent-scope accessors (global address space, sc1) ---------------------------------------------------------------
RMDK_D unsigned int ld_agent(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RMDK_D unsigned long long ld_agent(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RMDK_D unsigned int ld_system(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
RMDK_D unsigned long long ld_system(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
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
  const int x_lo = static_cast<int>(floorf(px.x)) + OFFSET - 1, q_lo = static_cast<int>(floorf(px.y)) + OFFSET - 1 - wy0, q_hi = q_lo + SIDE + 2;
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
RMDK_D float ncc_at_dyn(const SeedParams& P, F2 px, const float* __restrict__ win, const int* __restrict__ row_start, int ws, int wy0, int rows, int ww,
                        const float* __restrict__ ref_patch, int ref_stride, float sum_templ, float denom, unsigned int& n_fallback) {
  constexpr int OFFSET = -SIDE / 2;
  constexpr float AREA = static_cast<float>(SIDE * SIDE);
  float sum_img = 0.0f, sum_img_sq = 0.0f, sum_img_templ = 0.0f;
  int ix[SIDE], iy[SIDE];
  float ax[SIDE], ay[SIDE];
  bool reg_x = true, reg_y = true;
  // wave-uniform branches, one per axis (a wave whose x positions straddle a power of two replays the roundings of x only).  The rows first:
  // they say where in the window's row table the footprint starts
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
  // monotone, so inside every row between).  The window's row table says where a texel row starts in the LDS (a sheared band has no constant
  // row stride): three entries are read here -- first row, second row, last row -- at a clamped index whatever q0 is (no branch around the
  // reads), requested BEFORE the columns' parameters are worked out; the evaluation block reads the others two rows ahead of their use.
  // (All SIDE + 1 entries held in registers across the block cost 15 VGPRs: 116 instead of 101, and at 120 allocated registers four search
  // waves leave no room on a SIMD for a setup wave of another stream group of a batch -- a batch of 8 lost 5 %.)
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
    ncc_sums_lds_pipelined<SIDE>(win + ix[0], row_start + q0, off_first, off_second, ax, ay, ref_patch, ref_stride, sum_img, sum_img_sq, sum_img_templ);
  } else if (regular) {
    // outside the window (a clamped window, a seed that wandered off): the same separable filter on texel rows from L2, fully
    // unrolled so that the loads overlap -- a rolled loop costs a memory round trip per row and made the few such evaluations
    // the longest thing in a late frame
    ++n_fallback;
    ncc_sums_regular<SIDE, 0>(P.cur + iy[0] * P.cur_stride + ix[0], P.cur_stride, ax, ay, ref_patch, ref_stride, sum_img, sum_img_sq,
                              sum_img_templ);
  } else if (irregular_in_window<SIDE>(px, row_start, ws, wy0, rows, ww)) {
    // An irregular footprint (the replayed roundings moved a column or a row by a texel: ~1e-6 of the steps) whose neighbourhood lies inside the
    // window: sample by sample like the path below, but from the LDS.  From L2 such an evaluation is 81 x 4 dependent round trips -- 15-20 us
    // in ONE lane, and with a quarter of a million evaluations per light update every third or fourth update had one: its workgroup ended
    // at 30 us where the others ended at 17 (profiles/r04_timeline_light_frames.txt: the "slowest workgroups" of updates 31, 34).
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
RMDK_D void seed_range_band(const SeedParams& P, const FrameSmem<SIDE>& S, int tid, bool has, int j0, int j1, int m, int yref, int& u0, int& y0, int& u1, int& y1) {
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
// the window's row table.  History: element-wise staging (a division of the element index by the run-time width per texel: 25 vector
// instructions per texel row) -> row-wise through registers, 12 rows per lane in flight (a full 64 x 86 window: two round trips, 4-7 us of
// an unboxed unit's 23) -> LDS-direct boxes (round 4: one sequence 43.4 -> 40.0 us per update) -> sheared bands (round 5).
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
            __builtin_amdgcn_global_load_lds((gptr_t)(P.cur + static_cast<size_t>(y) * stride + col), (lptr_t)(S.win + r * W.ws + c0), 4, 0, 0);
        }
      }
    }
  }
  if (tid <= W.rows) S.row_start[tid] = __mul24(tid, W.ws) - (W.x0 + shear_of(W.y0 + tid - W.yref, W.m));
}

// Rounds of 256 NCC evaluations over work items [k0, k1) of the tile in LDS with window W; arg-max keys accumulate in S.best.
// No barrier.
template <int SIDE>
RMDK_D void frame_rounds(const SeedParams& P, FrameSmem<SIDE>& S, int tid, int k0, int k1, const FrameWindow& W, unsigned int& n_fallback) {
  using Smem = FrameSmem<SIDE>;
  // ncc_at_dyn's in-window test reads the row table at q0, q0 + 1, q0 + SIDE with q0 clamped to [0, rows - 1 - SIDE]: it needs a window of
  // at least SIDE + 1 rows.  Every window cut to in-image samples has SIDE + 3 or more (a sample's rows floor(y) - HALF - 1 .. floor(y) +
  // HALF + 2 lie inside the image: the guard of epipolar_match.cu:91-97 keeps y in [SIDE, h - SIDE)), and clamp_window keeps min(rows, 43);
  // a window that broke that rule is treated as empty -- zero usable columns, every evaluation reads L2 -- instead of trusted (scalar, once per call).
  const int ww_usable = W.rows > SIDE ? W.ww : 0;
  for (int r0 = k0; r0 < k1; r0 += TILE_PIX) {
    LAB_PROF(
    const unsigned long long prof_t0 = prof_clock();
    const unsigned int prof_fb0 = n_fallback;
    )
    const int kk = r0 + tid;
    int p = -1;
    unsigned long long key = 0ull;
    if (kk < k1) {
      int lo = 0, hi = TILE_PIX;  // last p with prefix[p] <= kk
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int mid = (lo + hi) >> 1;
        if (S.prefix[mid] <= kk) lo = mid; else hi = mid;
      }
      p = lo;
      const int j = kk - S.prefix[p];
      const float l = replay_l(S.l_first[p], j);  // the reference accumulates l; replay it
      const F2 px = F2{S.mean_x[p] + l * S.dir_x[p], S.mean_y[p] + l * S.dir_y[p]};
      const int ptx = p & (TILE_W - 1), pty = p >> 4;
      const float ncc = ncc_at_dyn<SIDE>(P, px, S.win, S.row_start, W.ws, W.y0, W.rows, ww_usable, S.ref + pty * Smem::REF_W + ptx, Smem::REF_W,
                                        S.sum_templ[p], S.denom[p], n_fallback);
      if (ncc > -1.0f) {  // NaN and anything the reference's "ncc > best_ncc" would never accept are dropped
        const unsigned int step = ((S.packed[p] >> 8) & 0xffu) + static_cast<unsigned int>(j);
        key = (static_cast<unsigned long long>(orderable_f32(ncc + 0.0f)) << 32) | (0xffffffffu - step);
      }
    }
    // one LDS atomic per evaluation: lanes of one seed hit one address and serialise inside the LDS, which is cheaper than a
    // segmented wave reduction first (12 ds_bpermute round trips)
    if (key != 0ull) atomicMax(&S.best[p], key);
    LAB_PROF(
    {  // diagnostics build only: ticks and count of this wave's rounds with / without an evaluation that left the LDS window
      const bool fb = __any(n_fallback != prof_fb0);
      const unsigned long long dt = prof_clock() - prof_t0;
      if ((tid & 63) == 0) S.prof[tid >> 6] += fb ? (dt << 24) | (1ull << 48) : dt | (1ull << 56);
    }
    )
  }
}

// The NCC search over work items [k_begin, k_end) of the tile whose descriptors are in LDS.  With a valid tile-wide window
// these are just the rounds; otherwise the window follows the work: the box of all remaining samples if that fits, else of
// the next 4 rounds, else of one round, else a clamped box (the rest reads L2).  All 256 threads call; ends with a barrier.
template <int SIDE>
RMDK_D void frame_search(const SeedParams& P, FrameSmem<SIDE>& S, int tid, int k_begin, int k_end, const FrameWindow& tile_win,
                         unsigned int& n_fallback, unsigned int& n_windows) {
  const unsigned int pk = S.packed[tid];
  const int my_n = static_cast<int>(pk & 0xffu);
  const int my_first = S.prefix[tid];
  int k = k_begin;
  while (k < k_end) {
    FrameWindow W = tile_win;
    int k1 = k_end;
    LAB_PROF(const unsigned long long prof_p0 = prof_clock();)
    if (!tile_win.valid) {
      // this lane's seed contributes steps [max(k - first, 0), min(kX - first, n) - 1] to the candidate range [k, kX)
      const int j0 = max(k - my_first, 0);
      int u0, y0, u1, y1;
      {
        int bu0, by0, bu1, by1;
        seed_range_band<SIDE>(P, S, tid, my_n > 0 && my_first < k1 && my_first + my_n > k, j0, min(k1 - my_first, my_n) - 1, W.m, W.yref, bu0, by0, bu1, by1);
        block_bbox<SIDE>(S, tid, bu0, by0, bu1, by1, 0);
        __syncthreads();
        block_bbox_read<SIDE>(S, u0, y0, u1, y1, 0);
      }
      if (!window_fits(u0, y0, u1, y1)) {
        const int kb = min(k_end, k + FR_UNIT_ITEMS), kc = min(k_end, k + TILE_PIX);
        int bu0, by0, bu1, by1, cu0, cy0, cu1, cy1;
        // a band over the same items as the one that has just failed is not reduced again: the 4-round band when at most four rounds are left
        // (units of the product pipeline: always), the one-round band when at most one is (most units of a light frame: one reduction pass and
        // one barrier instead of two, 1 us of a unit's 2.3 us of window policy)
        const bool four = kb < k_end, one = kc < kb;
        if (four) {
          seed_range_band<SIDE>(P, S, tid, my_n > 0 && my_first < kb && my_first + my_n > k, j0, min(kb - my_first, my_n) - 1, W.m, W.yref, bu0, by0, bu1, by1);
          block_bbox<SIDE>(S, tid, bu0, by0, bu1, by1, 4);
        }
        if (one) {
          seed_range_band<SIDE>(P, S, tid, my_n > 0 && my_first < kc && my_first + my_n > k, j0, min(kc - my_first, my_n) - 1, W.m, W.yref, cu0, cy0, cu1, cy1);
          block_bbox<SIDE>(S, tid, cu0, cy0, cu1, cy1, 8);
        }
        if (four || one) __syncthreads();
        if (four) block_bbox_read<SIDE>(S, u0, y0, u1, y1, 4);
        k1 = kb;
        if (!window_fits(u0, y0, u1, y1)) {
          if (one) block_bbox_read<SIDE>(S, u0, y0, u1, y1, 8);
          k1 = kc;
        }
      }
      W.set(u0, y0, u1, y1);
      if (!window_fits(u0, y0, u1, y1)) clamp_window(W);  // cut down around the band's centre; the rest reads L2
      LAB_PROF(const unsigned long long prof_p1 = prof_clock();)
      frame_stage_window<SIDE>(P, S, tid, W);
      ++n_windows;
      drain_vmem();  // the LDS-direct transfers of THIS wave have landed before it signals the barrier (see frame_stage_window)
      __syncthreads();
      LAB_PROF(if (tid == 0) { S.prof[4] += prof_p1 - prof_p0; S.prof[5] += prof_clock() - prof_p1; })
    }
    LAB_PROF(const unsigned long long prof_r0 = prof_clock();)
    frame_rounds<SIDE>(P, S, tid, k, k1, W, n_fallback);
    k = k1;
    __syncthreads();  // the window may be re-staged; S.best is complete for [k_begin, k)
    LAB_PROF(if (tid == 0) S.prof[6] += prof_clock() - prof_r0;)
  }
}

// Exclusive prefix of the per-seed step counts of the tile in LDS (S.packed, this thread's own entry) -> S.prefix[0..256]; returns the
// total.  Ends with a barrier.
template <int SIDE>
RMDK_D int frame_prefix(FrameSmem<SIDE>& S, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int n_valid = static_cast<int>(S.packed[tid] & 0xffu);
  const int incl = wave_scan_i32<WaveAdd>(n_valid);
  if (lane == 63) S.red[wave][4] = incl;
  __syncthreads();
  int wave_off = 0, total = 0;
#pragma unroll
  for (int wv = 0; wv < 4; ++wv) {
    const int v = S.red[wv][4];
    wave_off += wv < wave ? v : 0;
    total += v;
  }
  S.prefix[tid] = wave_off + incl - n_valid;
  if (tid == 0) S.prefix[TILE_PIX] = total;
  __syncthreads();
  return total;
}

// Setup.  The previous frame's finalisation, when it has been deferred (Q.fuse_prev: nobody looked at the state in between),
// runs here, in the lane that owns the same seed, on the state values this lane has loaded anyway -- the posterior goes
// straight from registers into the new frame's check.  Everything a lane needs from memory is requested in ONE batch up front
// (the compiler does not hoist loads out of the branches that consume them, and five dependent round trips under a burst of a
// million requests were two thirds of this kernel's time).
// The unit list is built without a planning step: a tile reserves its units in the list of its shard (tile % 16) with one
// returning atomic on the shard's counter; the unit size comes from the PREVIOUS frame's total work (the counters of three
// consecutive frames rotate).  The search kernel reads the sixteen counts and walks the shards' lists as one list.
// A frame in pinned HOST memory converted into a current-image plane by workgroup `part` of `parts`: plain 16-byte loads over the host
// link (pinned host memory is mapped uncached: every load fetches; 8-byte system-scope atomic loads reached 10 GB/s, these reach the
// link's 50).  kind 1: 8-bit rows of `pitch` bytes, x (1/255) like Depthmap::inputImage's convertTo(CV_32F, 1.0f / 255.0f)
// (depthmap.cpp:105); kind 2: w x h floats, unpadded.  The buffer is padded to a multiple of 16 bytes.
RMDK_D void ingest_in_place(int kind, int pitch, const void* src_v, float* __restrict__ dst, int w, int h, int stride, int part, int parts, int tid) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4* src = static_cast<const u32x4*>(src_v);
  if (kind == 1) {
    const int total_bytes = pitch * h, requests = (total_bytes + 15) >> 4;
    for (int d = part * TILE_PIX + tid; d < requests; d += parts * TILE_PIX) {
      const u32x4 v = __builtin_nontemporal_load(src + d);
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // four pixels per dword; rows are a multiple of 4 bytes, so a dword never straddles two rows
        const int byte = d * 16 + q * 4;
        if (byte >= total_bytes) break;
        const int row = byte / pitch, x4 = byte - row * pitch;
        const unsigned int u = v[q];
        float* out = dst + static_cast<size_t>(row) * stride + x4;
        const float f0 = static_cast<float>(u & 0xffu) * (1.0f / 255.0f), f1 = static_cast<float>((u >> 8) & 0xffu) * (1.0f / 255.0f);
        const float f2 = static_cast<float>((u >> 16) & 0xffu) * (1.0f / 255.0f), f3 = static_cast<float>(u >> 24) * (1.0f / 255.0f);
        if (x4 + 3 < w) *reinterpret_cast<float4*>(out) = make_float4(f0, f1, f2, f3);  // plane rows and x4 are multiples of 16 bytes
        else {
          if (x4 < w) out[0] = f0;
          if (x4 + 1 < w) out[1] = f1;
          if (x4 + 2 < w) out[2] = f2;
        }
      }
    }
  } else {
    const int total = w * h, requests = (total + 3) >> 2;
    for (int d = part * TILE_PIX + tid; d < requests; d += parts * TILE_PIX) {
      const u32x4 v = __builtin_nontemporal_load(src + d);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = d * 4 + q;
        if (e >= total) break;
        const int row = e / w;
        dst[static_cast<size_t>(row) * stride + (e - row * w)] = __uint_as_float(v[q]);
      }
    }
  }
}

// Dword `d` of an 8-bit frame (rows of `per_row` dwords, four pixels each) -> its four floats in the current-image plane: x (1/255) like
// Depthmap::inputImage's convertTo(CV_32F, 1.0f / 255.0f) (depthmap.cpp:105), one fp32 multiply per pixel.
RMDK_D void store_u8x4(float* __restrict__ dst, int stride, int w, int per_row, int d, unsigned int u) {
  const int row = d / per_row, x4 = (d - row * per_row) * 4;
  float* out = dst + static_cast<size_t>(row) * stride + x4;
  const float f0 = static_cast<float>(u & 0xffu) * (1.0f / 255.0f), f1 = static_cast<float>((u >> 8) & 0xffu) * (1.0f / 255.0f);
  const float f2 = static_cast<float>((u >> 16) & 0xffu) * (1.0f / 255.0f), f3 = static_cast<float>(u >> 24) * (1.0f / 255.0f);
  if (x4 + 3 < w) *reinterpret_cast<float4*>(out) = make_float4(f0, f1, f2, f3);  // plane rows and x4 are multiples of 16 bytes
  else {
    if (x4 < w) out[0] = f0;
    if (x4 + 1 < w) out[1] = f1;
    if (x4 + 2 < w) out[2] = f2;
  }
}

// A frame in a staging buffer in HBM (MatcherArgs::ingest_kind: 1 = 8-bit rows of `pitch` bytes, 2 = w x h floats) converted into a
// current-image plane by workgroup `part` of `parts`.  INGEST_BATCH dwords per lane are requested before the first is used: a lane's requests
// are `parts` x 256 dwords apart (every instruction fully coalesced), and the whole frame is a round trip or two instead of one per dword --
// at 1920x1080 a lane of 128 workgroups has sixteen dwords to fetch, and fetched one by one (the loads could not be moved across the stores
// of the iteration before) the conversion took longer than the setup kernel it rides in.  LOAD: plain loads for the search kernel's
// bringers (the copy had completed before an EARLIER kernel saw its flag), agent-scope loads for a frame's own setup kernel (the copy
// may complete while the kernel runs).
constexpr int INGEST_BATCH = 4;
template <typename LOAD>
RMDK_D void ingest_staged(int kind, int pitch, const void* src_v, float* __restrict__ dst, int w, int h, int stride, int part, int parts, int tid, LOAD load) {
  const unsigned int* src = static_cast<const unsigned int*>(src_v);
  const int per_row = kind == 1 ? pitch >> 2 : w, total = per_row * h, step = parts * TILE_PIX;
  auto put = [&](int d, unsigned int u) {
    if (d >= total) return;
    if (kind == 1) store_u8x4(dst, stride, w, per_row, d, u);
    else {
      const int row = d / w;
      dst[static_cast<size_t>(row) * stride + (d - row * w)] = __uint_as_float(u);
    }
  };
  static_assert(INGEST_BATCH == 4, "the four requests below");
  for (int d0 = part * TILE_PIX + tid; d0 < total; d0 += INGEST_BATCH * step) {
    const int d1 = d0 + step, d2 = d1 + step, d3 = d2 + step;
    const unsigned int v0 = load(src + d0), v1 = load(src + min(d1, total - 1)), v2 = load(src + min(d2, total - 1)), v3 = load(src + min(d3, total - 1));
    put(d0, v0); put(d1, v1); put(d2, v2); put(d3, v3);
  }
}
// workgroups that convert a staged frame of `dwords` dwords: INGEST_BATCH requests per lane, at most `cap`
inline int ingest_workgroups(long long dwords, int cap) {
  const long long want = (dwords + static_cast<long long>(TILE_PIX) * INGEST_BATCH - 1) / (static_cast<long long>(TILE_PIX) * INGEST_BATCH);
  return static_cast<int>(want < cap ? (want < 1 ? 1 : want) : cap);
}

// Words 2, 3 of a unit entry: the tile's window -- column origin (16 bits, signed: a band may start left of the image) | first row << 16;
// rows (8 bits) | texels per row (9 bits) << 8 | shear (15 bits, signed) << 17.  A unit without UNIT_TILE_BOX carries the shear only.
RMDK_D unsigned int unit_pack_origin(int x0, int y0) { return (static_cast<unsigned int>(x0) & 0xffffu) | (static_cast<unsigned int>(y0) << 16); }
RMDK_D unsigned int unit_pack_shape(int rows, int ww, int m) {
  return static_cast<unsigned int>(rows) | (static_cast<unsigned int>(ww) << 8) | (static_cast<unsigned int>(m) << 17);
}
static_assert(FR_MAX_ROWS < (1 << 8) && FR_MAX_WIDTH < (1 << 9), "window shape fields of a unit entry");
constexpr int unit_tail_shift(int shard) { return shard >= 14 ? 2 : shard >= 12 ? 1 : 0; }  // see seed_setup_compact_kernel
constexpr int UNIT_ROUNDS_SHIFT = 30;  // word 0 of a unit entry: tile | (rounds per unit - 1) << 30
static_assert(MAX_UNIT_ROUNDS <= 4, "two bits of a unit entry");

constexpr int INGEST_WGS = 512;  // at most this many workgroups (per sequence) bring a staged host frame into the current-image plane (the only ones that may wait): ingest_workgroups()
constexpr int INGEST_WGS_REMAP = 512;  // with lens undistortion (two dependent round trips per pixel): a quarter of the chip's wave slots at most
constexpr int INGEST_WGS_IN_PLACE = 256;  // frames read in place from pinned host memory: enough requests in flight to cover the host link's latency

template <int SIDE, int NSEQ>
__global__ __launch_bounds__(TILE_PIX) void seed_setup_compact_kernel(BatchArgs<NSEQ> B, MatcherArgs M, int target_units) {
  __shared__ int red_i[4], red_c[4], red_l[4], red_b[4][6];
  __shared__ unsigned int s_base;
  constexpr int HALF = SIDE / 2;
  const int seq = NSEQ == 1 ? 0 : static_cast<int>(blockIdx.z);
  // one sequence: the named argument, which the compiler fetches with a few wide scalar loads at the top of the kernel; several: the
  // argument segment indexed by the sequence number (see BatchArgs), copied once so that its loads are issued here too and not one
  // by one where the values are used (every one of those was a scalar-cache round trip on this kernel's dependent chain)
  const SeqArgs Q = NSEQ == 1 ? B.seq[0] : seq_table()[seq];  // (a batch's block read by reference instead: +0.5 %, within the noise: profiles/r04_ab_setup_args.txt)
  const SeedParams& P = Q.P;
  if (NSEQ > 1 && !Q.active) return;  // this sequence has no frame in this launch
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tx = tid & (TILE_W - 1), ty = tid >> 4;
  // rows [0, M.ingest_rows) of the grid belong to the frame-ingest workgroups (below), the tile grid follows
  const int tile_by = static_cast<int>(blockIdx.y) - M.ingest_rows;
  const int x = blockIdx.x * TILE_W + tx, y = tile_by * TILE_H + ty;
  const bool in_image = x < P.w && y < P.h && tile_by >= 0;
  const int gi = in_image ? y * P.stride + x : 0;
  const size_t gm = (NSEQ == 1 ? 0 : static_cast<size_t>(seq) * M.seq_plane) + gi;  // the same seed in the workspace planes
  const unsigned long long t_start = P.trace ? wall_clock64() : 0ull;  // timeline probes (diagnostics)
  unsigned long long t_loaded = 0ull;
  // frame ingest (see MatcherArgs): the workgroups IN FRONT of the tile grid (blockIdx.y < ingest_rows, at most INGEST_WGS of them per
  // sequence, launched only when host frames are pending) wait for the staging copy's flag and convert the staged frame into the
  // current-image plane.  They are dispatched first: the conversion runs beside the tiles' latency chains instead of behind the last of
  // them (below the tile grid, as until round 5, a 1920x1080 frame's conversion started when the last of 8 160 tile workgroups had been
  // placed and made the kernel 15 us longer).  Only these few workgroups ever wait: if every tile workgroup did, a device filled with
  // waiting waves could keep a copy that is carried out by a blit kernel from ever running.  The wait is bounded; a copy that never
  // arrives is reported through progress[1].  The flag is read with agent-scope acquire loads and the staged frame with agent-scope
  // loads: the copy may finish after this kernel has started.
  const int wg = tile_by * static_cast<int>(gridDim.x) + static_cast<int>(blockIdx.x);
  if (tile_by < 0 || tile_by >= M.tiles_y) {
    // One step ahead: the verdict for the NEXT frame -- has it been handed over (a pinned host word, frames read in place) / arrived in HBM (the
    // arrival flag of its ring slot)? -- falls to ONE extra row of workgroups behind the tile grid: dispatched last, so that it is taken as late
    // as this kernel can take it (the search kernel's bringers act on it).  (Taken by the last tile workgroup instead, the branch cost the
    // tile path its last free scalar registers: the compiler reserved a scratch segment for every wave of the kernel.)
    if (tile_by >= M.tiles_y) {
      if (NSEQ == 1 && M.ahead && blockIdx.x == 0 && tid == 0) {
        const unsigned int next = M.ingest_number + 1u;
        __hip_atomic_store(M.ahead, static_cast<int>(ld_system(M.submitted) - next) >= 0 ? next : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    const int iw = static_cast<int>(blockIdx.y * gridDim.x + blockIdx.x);
    if (iw >= M.ingest_wgs) return;
    // step numbers are compared modulo 2^32 (a live system never stops counting): "behind" = the signed difference is negative.
    // The flag and the staged frame are read with RELAXED agent-scope loads and NO acquire fence: on this part an agent-scope acquire is a
    // cache invalidation (buffer_inv sc1: the vector L1 AND the lines of this XCD's L2 that other agents may have written), issued by every
    // wave that executes it -- two per ingest wave until round 5, a few hundred of them at the head of every setup kernel with host frames,
    // thrown at the L2 the tile workgroups are loading their seeds' state through (1920x1080, conversion left to the setup kernel: 147 us
    // per update with them, profiles/r06_*).  What the acquire was there for holds without it: the frame's loads are agent-scope loads
    // themselves (they never hit a stale line), and they are issued after the branch that consumed the flag -- the hardware issues a
    // wave's instructions in order and does not speculate; the compiler is kept from moving them by the barrier below.
    auto behind = [&]() { return static_cast<int>(ld_agent(M.ingest_flag) - M.ingest_number) < 0; };
    // No flag: the frame is read IN PLACE from the pinned host buffer the caller's frame was copied into before this kernel was
    // launched (ingest_in_place: 16 bytes per lane and request over the host link; a 640x480 8-bit frame is one round trip of 75
    // workgroups plus 6 us of link time) -- no copy engine, no staging in HBM, nothing to wait for.
    const bool in_place = M.ingest_flag == nullptr;
    // one step ahead (see MatcherArgs): has this frame been brought in already, by the bringers of the previous update's search kernel?
    if (NSEQ == 1 && M.ahead && ld_agent(M.ahead + 2) == M.ingest_number) return;
    unsigned int spins = 0u;
    if (!in_place && behind()) {
      while (behind() && ++spins < (1u << 18)) __builtin_amdgcn_s_sleep(16);
      if (behind() && tid == 0) __hip_atomic_store(M.progress + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (M.ingest_profile && iw == 0 && tid == 0 && seq == M.housekeeper) {  // diagnostics: this frame was not converted one step ahead; did it keep the kernel waiting?
      __hip_atomic_fetch_add(M.progress + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (spins) {
        __hip_atomic_fetch_add(M.progress + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_fetch_add(M.progress + 4, spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    asm volatile("" ::: "memory");  // nothing below is read before the flag has been seen
    if (M.ingest_kind == 1 && Q.ingest_map1) {  // cv::remap through the undistortion maps first (depthmap.cpp:99), destination pixels four at a time per lane
      const unsigned char* src = reinterpret_cast<const unsigned char*>(Q.ingest_u8);
      const int total = P.w * P.h, step = M.ingest_wgs * TILE_PIX;
      for (int d0 = iw * TILE_PIX + tid; d0 < total; d0 += 4 * step) {
        short2 m[4];
        int f[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // all map entries of the batch are requested before the first source pixel
          const int d = min(d0 + q * step, total - 1);
          m[q] = Q.ingest_map1[d];
          f[q] = Q.ingest_map2[d] & 1023;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int d = d0 + q * step;
          if (d >= total) break;
          const int row = d / P.w;
          Q.ingest_dst[static_cast<size_t>(row) * P.stride + (d - row * P.w)] = remap_u8_pixel(src, M.ingest_pitch, m[q], f[q], P.w, P.h, [](const unsigned char* p) {
            return static_cast<int>(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          });
        }
      }
    } else if (in_place) {
      ingest_in_place(M.ingest_kind, M.ingest_pitch, M.ingest_kind == 1 ? static_cast<const void*>(Q.ingest_u8) : static_cast<const void*>(Q.ingest_f32),
                      Q.ingest_dst, P.w, P.h, P.stride, iw, M.ingest_wgs, tid);
    } else {  // the staged frame: x (1/255) for 8-bit frames (store_u8x4), floats as they are
      ingest_staged(M.ingest_kind, M.ingest_pitch, M.ingest_kind == 1 ? static_cast<const void*>(Q.ingest_u8) : static_cast<const void*>(Q.ingest_f32), Q.ingest_dst,
                    P.w, P.h, P.stride, iw, M.ingest_wgs, tid, [](const unsigned int* p) { return ld_agent(p); });
    }
    return;
  }
  const int tile = tile_by * M.tiles_x + static_cast<int>(blockIdx.x);      // within the sequence
  const int tile_g = (NSEQ == 1 ? 0 : seq * M.n_tiles) + tile;               // within the launch
  // A tile in which the previous frame's check left no seed in state UPDATE is DEAD until the next reference frame: BORDER / CONVERGED /
  // DIVERGED are absorbing (nothing but the finalisation of an UPDATE seed ever changes sigma_sq, a, b), so every plane, the tile's CONVERGED
  // count and its empty descriptors already hold what this launch would write.  Its workgroup leaves after one scalar load -- on the ~170
  // light frames of the benchmark sequence that is four tiles in five, which used to fetch 44 bytes per pixel and run the check for nothing.
  // (tile_live: written at the end of this kernel by the tile's own workgroup, read here one launch later; fuse_prev = nobody touched the
  // planes in between.  The launch's housekeeping below must not depend on tile 0 being alive.)
  // (the tile's word: seeds in state UPDATE after the previous frame's check | TILE_WANTS_BAND, see below)
  const unsigned int tile_word = *(const __attribute__((address_space(4))) unsigned int*)(M.tile_live + tile_g);
  const bool dead_tile = Q.fuse_prev && (tile_word & 0xffffu) == 0u;
  // A tile whose samples did not fit a BOX-shaped window one frame ago (a bundle of long diagonal segments) gets a sheared band this frame;
  // everybody else -- nearly every tile of nearly every frame -- pays nothing for the machinery: no slope, no second pair of reductions.  (The
  // first frame on which a tile's box does not fit goes to the search kernel's own window policy, as before round 5.)
  const bool want_band = LAB_WANT_BAND((tile_word & TILE_WANTS_BAND) != 0u);
  // the seed's state: requested before anything else, so that the scalar-load chains below (kernel arguments, the previous frame's
  // counters) run while these are in flight
  if (dead_tile && !(tile == 0 && seq == M.housekeeper)) return;  // (the keeper goes on: its loads are as harmless as they were)
  float mu = P.mu[gi], sigma_sq = P.sigma_sq[gi], a = P.a[gi], b = P.b[gi];
  // ... and, when the previous frame's finalisation runs here, what that needs: the state it left, its arg-max key, its search descriptor
  // (requested whether or not a finalisation is pending -- it nearly always is, and a conditional load costs a register shuffle and a
  // wait at the point where the two paths meet)
  int conv_prev = P.conv[gi];
  unsigned long long key = M.best[gm];
  unsigned int packed_prev = M.packed[gm];
  float lfirst_prev = M.lfirst[gm];
  float2 m_prev = M.mean[gm], d_prev = M.dir[gm];
  if (M.progress && wg == 0 && seq == M.housekeeper && tid == 0) __hip_atomic_store(M.progress, M.ingest_number, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // (computed here, at the top, so that its scalar loads travel with the kernel arguments)
  // unit size: 1..4 rounds of 256 evaluations, from the previous frame's work (a frame differs little from the one before)
  int unit_rounds = MAX_UNIT_ROUNDS;  // (capped at 3 / 2 rounds -- a shorter tail on the heaviest updates, more staging --: update 1 +6 %, batch of 8 -2 % / -8 %: profiles/r04_ab_unit_cap.txt)
  if (M.shards_prev) {
    // the previous frame's counters are not written by anybody while this kernel runs: read them through the scalar path (constant
    // address space), which the compiler schedules with the kernel arguments at the top instead of as a vector-memory round trip
    // between the tile's reduction and its reservation
    typedef const __attribute__((address_space(4))) unsigned long long* const_u64_ptr;
    const const_u64_ptr prev = (const_u64_ptr)(M.shards_prev);
    unsigned long long items = 0;
#pragma unroll
    for (int q = 0; q < UNIT_SHARDS; ++q) items += prev[q] >> 32;
    // (rounding to nearest instead of up, or aiming at 2x / 3x as many units, changes nothing measurable: 48.2 - 49.0 us per update)
    const unsigned long long per_round = static_cast<unsigned long long>(target_units) * TILE_PIX;
    static_assert(MAX_UNIT_ROUNDS == 4, "the ladder below is ceil(items / per_round) clamped to 1..4");
    unit_rounds = items > 3 * per_round ? 4 : items > 2 * per_round ? 3 : items > per_round ? 2 : 1;  // no 64-bit division
  }
  // Graduated unit sizes (one sequence per launch only).  The search kernel's hand-out deals the shards' lists one after the other, so the units
  // of the last shards are the last to be searched, and while the last unit runs every other workgroup waits: up to one unit's time, a fifth of
  // the kernel on the frames that have two to five units per workgroup.  The tiles of shards 12, 13 cut their work into units of half the frame's
  // size, those of shards 14, 15 into quarters (at least one round): 12 % more units, the wait at the end a quarter as long -- update 1 96 -> 92.5 us,
  // a sequence 38.2 -> 37.6 us per update (profiles/r05_ab_unit_tail.txt).  In a batch the other stream groups' kernels fill that wait already
  // and the smaller units only cost their staging: -3 ... -4 % with every graduation tried; there all units of a frame have one size.
  if (NSEQ == 1) unit_rounds = max(1, unit_rounds >> unit_tail_shift(tile_g % UNIT_SHARDS));
  // what the planes hold now (known only when the previous frame's values were loaded for its finalisation): a seed that has
  // converged or diverged keeps writing the same state and an empty descriptor, a third of this kernel's stores -- skipped
  int conv_old = -1;
  unsigned int packed_old = 0xffffffffu;
  if (Q.fuse_prev) {  // uniform over the workgroup
    // Everything requested above is in flight together: left alone the compiler sinks each load into the branch that consumes it (state
    // -> key -> descriptor: three dependent memory round trips on every live lane's chain).  The empty statement below "reads" every
    // value, so the loads stay where they were issued and are waited for once.
    asm volatile("" : "+v"(conv_prev), "+v"(key), "+v"(packed_prev), "+v"(lfirst_prev), "+v"(m_prev.x), "+v"(m_prev.y), "+v"(d_prev.x), "+v"(d_prev.y));
    conv_old = conv_prev; packed_old = packed_prev;
    if (in_image && conv_prev == ST_UPDATE) {
      F2 best_px = F2{0.0f, 0.0f};
      float best_ncc = -1.0f;
      if (key != 0ull) {
        best_ncc = from_orderable_f32(static_cast<unsigned int>(key >> 32));
        const int step = static_cast<int>(0xffffffffu - static_cast<unsigned int>(key & 0xffffffffu));
        const float l = replay_l(lfirst_prev, step - static_cast<int>(packed_prev >> 16));
        best_px = F2{m_prev.x + l * d_prev.x, m_prev.y + l * d_prev.y};
      }
      int state_prev = ST_UPDATE;
      if (best_ncc < 0.5f) state_prev = ST_NO_MATCH;
      else P.match[y * P.stride2 + x] = make_float2(best_px.x, best_px.y);
      const int what = seed_fuse_values(P, Q.T_ref_curr_prev, x, y, state_prev, mu, sigma_sq, a, b, best_px);
      if (what == 1) { P.sigma_sq[gi] = sigma_sq; P.mu[gi] = mu; P.a[gi] = a; P.b[gi] = b; }
      else if (what == 2) P.b[gi] = b;
    }
  }
  // The band along which this tile's samples lie (shear_of): the slope of the epipolar line of the tile's CENTRE pixel.  The ray of pixel p
  // projects to the line through the epipole e = proj(t) (depth 0) and the vanishing point v = proj(R K^-1 p) (depth infinity) whatever the
  // seed's depth estimate is, so the direction needs no state, no reduction and no hint from the previous frame: every lane computes the same
  // number from the kernel arguments (here, where the pose is in scalar registers anyway: the segments below use it).  v - e scaled by t_z: (fx (t_z X / Z - t_x), fy (t_z Y / Z - t_y)) -- finite
  // for a sideways motion (t_z = 0, epipole at infinity).  Approximate reciprocals: the slope steers window shapes, never results.
  int m_tile = 0;
  if (want_band) {  // (uniform over the workgroup)
    const Pose& T = P.T_curr_ref;
    const float fcx = (static_cast<float>(blockIdx.x * TILE_W + TILE_W / 2) - P.cam.cx) * __builtin_amdgcn_rcpf(P.cam.fx);
    const float fcy = (static_cast<float>(tile_by * TILE_H + TILE_H / 2) - P.cam.cy) * __builtin_amdgcn_rcpf(P.cam.fy);
    const float X = T.d[0] * fcx + T.d[1] * fcy + T.d[2], Y = T.d[4] * fcx + T.d[5] * fcy + T.d[6], Z = T.d[8] * fcx + T.d[9] * fcy + T.d[10];
    const float iz = __builtin_amdgcn_rcpf(Z);
    const float dx = P.cam.fx * (T.d[11] * X * iz - T.d[3]), dy = P.cam.fy * (T.d[11] * Y * iz - T.d[7]);
    // flatter than 8 columns per row: a box holds such a bundle (0); a point behind the camera or a degenerate pose: a box as well
    if (Z > 0.0f && fabsf(dx) < 7.9f * fabsf(dy)) m_tile = static_cast<int>(rintf(dx * __builtin_amdgcn_rcpf(dy) * static_cast<float>(1 << FR_SHEAR_BITS)));
  }
  int state = ST_BORDER;
  if (in_image) {
    // A seed that the PREVIOUS frame's check (the same lane, one launch ago: fuse_prev says nobody touched the planes in between) found
    // BORDER / CONVERGED / DIVERGED keeps sigma_sq, a, b -- the finalisation only ever touches UPDATE seeds -- so the check, a pure function
    // of those and of the pixel position, gives the same answer again: it is not re-evaluated (two IEEE divisions per lane; on the ~170
    // light frames of the sequence four waves in five consist of such seeds only).
    const bool settled = conv_old == ST_BORDER || conv_old == ST_CONVERGED || conv_old == ST_DIVERGED;
    state = settled ? conv_old : seed_check(P, x, y, sigma_sq, a, b, SIDE);
    if (state != conv_old) P.conv[gi] = state;  // final for BORDER / CONVERGED / DIVERGED; UPDATE seeds are settled by the finalisation
  }
  if (P.trace) t_loaded = wall_clock64();
  int n_valid = 0, i_first = 0;
  unsigned int n_steps = 0, n_evals = 0;
  int bx0 = BAND_NONE_LO, by0 = BAND_NONE_LO, bx1 = BAND_NONE_HI, by1 = BAND_NONE_HI;  // texel box of this seed's samples ...
  int bu0 = BAND_NONE_LO, bu1 = BAND_NONE_HI;                                          // ... and their band under the tile's shear (band_add_point)
  const bool live = in_image && state == ST_UPDATE;
  if (live) {
    const Segment seg = epipolar_segment(P, x, y, mu, sigma_sq);
    const ValidRun run = find_valid_run(P, seg, SIDE);
    n_valid = run.n_valid; i_first = run.i_first;
    M.best[gm] = 0ull;
    if (n_valid > 0) {
      M.mean[gm] = make_float2(seg.mean.x, seg.mean.y);
      M.dir[gm] = make_float2(seg.dir.x, seg.dir.y);
      M.lfirst[gm] = run.l_first;
      // the positions are monotone along the run: its two ends bound every sample; a sample at p touches texels
      // floor(p) - HALF .. floor(p) + HALF + 1, one more for the replayed roundings (a box that misses a sample costs speed only)
      bx0 = max(static_cast<int>(floorf(fminf(run.px_first.x, run.px_last.x))) - HALF - 1, 0);
      by0 = max(static_cast<int>(floorf(fminf(run.px_first.y, run.px_last.y))) - HALF - 1, 0);
      bx1 = min(static_cast<int>(floorf(fmaxf(run.px_first.x, run.px_last.x))) + HALF + 2, P.w - 1);
      by1 = min(static_cast<int>(floorf(fmaxf(run.px_first.y, run.px_last.y))) + HALF + 2, P.h - 1);
      if (want_band) {  // the same under the tile's shear (rows as above: the band's rows are the box's)
        int ty0 = BAND_NONE_LO, ty1 = BAND_NONE_HI;
        const int yref = tile_by * TILE_H;
        band_add_point(run.px_first.x, run.px_first.y, HALF, m_tile, yref, bu0, ty0, bu1, ty1);
        band_add_point(run.px_last.x, run.px_last.y, HALF, m_tile, yref, bu0, ty0, bu1, ty1);
      }
    }
    if (P.stats) {  // diagnostics only: the full walk, counting what the reference would visit / evaluate
      for (float l = -seg.half_length; l <= seg.half_length; l += 0.7f, ++n_steps) {
        const F2 px = F2{seg.mean.x + l * seg.dir.x, seg.mean.y + l * seg.dir.y};
        if (!px_outside(P, px, SIDE)) ++n_evals;  // NaN positions pass the reference's guard too
      }
    }
  }
  const unsigned int packed_new = (static_cast<unsigned int>(i_first) << 16) | static_cast<unsigned int>(n_valid);
  if (in_image && packed_new != packed_old) M.packed[gm] = packed_new;
  if (P.stats) {
    const unsigned long long s_live = wave_sum_u64(live ? 1ull : 0ull);
    const unsigned long long s_steps = wave_sum_u64(static_cast<unsigned long long>(n_steps));
    const unsigned long long s_evals = wave_sum_u64(static_cast<unsigned long long>(n_evals));
    if (lane == 0 && s_live) { atomicAdd(&P.stats[0], s_live); atomicAdd(&P.stats[1], s_steps); atomicAdd(&P.stats[2], s_evals); }
  }
  // seeds that this frame's check found CONVERGED: what getConvergedCount() reports after this update (seed_matrix.cu:195-198 counts
  // the plane that seed_check has just rewritten; the matcher only ever turns UPDATE into NO_MATCH)
  const int n_conv = __popcll(__ballot(in_image && state == ST_CONVERGED));
  // the wave's work total, and the texel box of ALL samples of the tile: it travels with the tile's work units, so that the search kernel
  // can request the tile's window of the current image together with the tile's descriptors (one memory round trip less per tile).  A
  // wave without work skips the five reductions (sixty-five vector instructions: more than the rest of what such a wave executes).
  int tot = 0;
  if (__any(n_valid > 0)) {  // uniform over the wave: all 64 lanes take part in the DPP reductions
    tot = wave_reduce_i32<WaveAdd>(n_valid);
    bx0 = wave_reduce_i32<WaveMin>(bx0); by0 = wave_reduce_i32<WaveMin>(by0);
    bx1 = wave_reduce_i32<WaveMax>(bx1); by1 = wave_reduce_i32<WaveMax>(by1);
    if (want_band) { bu0 = wave_reduce_i32<WaveMin>(bu0); bu1 = wave_reduce_i32<WaveMax>(bu1); }
  }
  const int n_live = __popcll(__ballot(live));
  if (lane == 0) {
    red_i[wave] = tot; red_c[wave] = n_conv; red_l[wave] = n_live;
    red_b[wave][0] = bx0; red_b[wave][1] = by0; red_b[wave][2] = bx1; red_b[wave][3] = by1;
    red_b[wave][4] = bu0; red_b[wave][5] = bu1;
  }
  __syncthreads();
  const int total = red_i[0] + red_i[1] + red_i[2] + red_i[3];
  const int unit_items = unit_rounds * TILE_PIX;
  // the launch's housekeeping falls to tile 0 of its first sequence that has a frame (M.housekeeper: a sequence that sits the step out
  // leaves at the top of the kernel)
  const bool keeper = tile == 0 && seq == M.housekeeper;
  if (keeper && tid < UNIT_SHARDS) M.shards_next[tid] = 0ull;  // the set the NEXT frame counts in (nobody reads it now)
  if (keeper && tid < UNIT_SHARDS) M.handout[tid * HANDOUT_STRIDE] = 0u;  // the search kernel's hand-out counters
  if (tid == 0) M.tile_conv[tile_g] = static_cast<unsigned int>(red_c[0] + red_c[1] + red_c[2] + red_c[3]);
  // the tile's box of this frame: needed by the lanes that write a unit entry and by lane 0 (the tile's word)
  const int n_units_tile = total > 0 ? units_of(total, unit_rounds) : 0;
  int tx0 = 0, ty0 = 0, tx1 = -1, ty1 = -1;
  bool box = false;
  if (tid == 0 || tid < n_units_tile) {
    tx0 = min(min(red_b[0][0], red_b[1][0]), min(red_b[2][0], red_b[3][0])); ty0 = min(min(red_b[0][1], red_b[1][1]), min(red_b[2][1], red_b[3][1]));
    tx1 = max(max(red_b[0][2], red_b[1][2]), max(red_b[2][2], red_b[3][2])); ty1 = max(max(red_b[0][3], red_b[1][3]), max(red_b[2][3], red_b[3][3]));
    box = window_fits(tx0, ty0, tx1, ty1);
  }
  if (tid == 0)  // seeds in state UPDATE after this frame's check | does the NEXT frame's setup reduce a band for this tile?
    M.tile_live[tile_g] = static_cast<unsigned int>(red_l[0] + red_l[1] + red_l[2] + red_l[3]) | (total > 0 && !box ? TILE_WANTS_BAND : 0u);
  if (P.trace && tid == 0)  // word 2 of the tile's slot: start (low 32 bits of the 10 ns clock), state ready and end relative to it
    P.trace[static_cast<size_t>(tile) * FR_TRACE_WORDS + 2] = (t_start & 0xffffffffull) | (((t_loaded - t_start) & 0xffffull) << 32) |
                                                             (((wall_clock64() - t_start) & 0xffffull) << 48);
  if (total == 0) return;
  const int n_u = n_units_tile;
  if (tid == 0) {
    // (issued right behind the first barrier instead, the compiler still waits for the answer where it is issued -- the branch's end)
    const unsigned long long old = atomicAdd(&M.shards_cur[tile_g % UNIT_SHARDS], (static_cast<unsigned long long>(total) << 32) | static_cast<unsigned long long>(n_u));
    s_base = static_cast<unsigned int>(old);  // units reserved so far in this shard
  }
  __syncthreads();
  if (tid < n_u) {
    const int x0 = tx0, y0 = ty0, x1 = tx1, y1 = ty1;
    const int u0 = min(min(red_b[0][4], red_b[1][4]), min(red_b[2][4], red_b[3][4])), u1 = max(max(red_b[0][5], red_b[1][5]), max(red_b[2][5], red_b[3][5]));
    // the window that holds ALL samples of the tile, if there is one: the box when that fits, else the band along the tile's shear; else the
    // search kernel cuts windows to each unit's own samples (along the same shear, which travels with the unit either way)
    const bool band = !box && want_band && m_tile != 0 && window_fits(u0, y0, u1, y1);
    const int wx0 = box ? x0 : u0, ww = (box ? x1 : u1) - wx0 + 1, rows = y1 - y0 + 1, m = box ? 0 : m_tile;
    const bool whole = box || band;  // (leaving a LARGE tile window that many one-round units share to the units -- each cuts its own -- gains nothing: LAB.md)
    M.units[static_cast<size_t>(tile_g % UNIT_SHARDS) * M.shard_cap + s_base + tid] =
        make_uint4(static_cast<unsigned int>(tile_g) | (static_cast<unsigned int>(unit_rounds - 1) << UNIT_ROUNDS_SHIFT), static_cast<unsigned int>(tid * unit_items) | (whole ? UNIT_TILE_BOX : 0u),
                   whole ? unit_pack_origin(wx0, y0) : 0u, unit_pack_shape(whole ? rows : 0, whole ? ww : 0, m));
  }
}

// Who searches which unit.  The setup kernel appends a tile's units to the list of shard (tile % UNIT_SHARDS); the sixteen counts were written
// by that kernel -- the launch before this one -- and nobody writes them while this kernel runs: they are read through the scalar path
// (constant address space) every time they are needed, sixteen words from the scalar cache, instead of being kept in seventeen scalar
// registers across the whole kernel (which, with the kernel arguments, overflowed the scalar register file: 104 spills to vector-register
// lanes, two hundred v_writelane / v_readlane per workgroup -- executed by all 1 024 workgroups of every launch, most of which have no unit
// on a light frame).  Two numberings (seed_search_compact_kernel chooses per frame):
//  * the shards' lists read as ONE list: entry g lives in the shard s with first[s] <= g < first[s + 1] (unit_entry); workgroup b starts
//    with unit b, the rest is handed out by counters;
//  * light frames -- no shard holds more units than a sixteenth of the grid --: entry i of shard s belongs to workgroup 16 i + s.
typedef const __attribute__((address_space(4))) unsigned long long* const_u64_ptr;
RMDK_D unsigned int unit_count(const_u64_ptr counts) {
  unsigned int n = 0u;
#pragma unroll
  for (int q = 0; q < UNIT_SHARDS; ++q) n += static_cast<unsigned int>(counts[q]);
  return n;
}
RMDK_D unsigned int unit_max(const_u64_ptr counts) {
  unsigned int n = 0u;
#pragma unroll
  for (int q = 0; q < UNIT_SHARDS; ++q) n = max(n, static_cast<unsigned int>(counts[q]));
  return n;
}
RMDK_D const uint4* unit_entry(const MatcherArgs& M, const_u64_ptr counts, unsigned int g) {  // g uniform: scalar arithmetic
  unsigned int acc = 0u, sh = 0u, sh_first = 0u;
#pragma unroll
  for (int q = 0; q < UNIT_SHARDS - 1; ++q) {
    acc += static_cast<unsigned int>(counts[q]);
    if (g >= acc) { sh = static_cast<unsigned int>(q + 1); sh_first = acc; }
  }
  return M.units + static_cast<size_t>(sh) * M.shard_cap + (g - sh_first);
}

template <int SIDE, int NSEQ>
__global__ __launch_bounds__(TILE_PIX, FR_MIN_WAVES) void seed_search_compact_kernel(BatchArgs<NSEQ> B_by_value, MatcherArgs M_by_value) {
  // Both argument blocks are read where they lie, in the kernel-argument segment (scalar loads at the point of use), not through the named
  // parameters: named, every field the kernel uses anywhere is fetched at the top and kept in a scalar register across the whole persistent
  // loop -- with the loop's own state more than the 104 there are, and every spilled one is a v_writelane / v_readlane pair somewhere hot.
  (void)B_by_value; (void)M_by_value;
  const SeqArgs* const Bq = seq_table();
  static_assert(sizeof(BatchArgs<NSEQ>) % alignof(MatcherArgs) == 0 && alignof(SeqArgs) <= 8 && alignof(MatcherArgs) <= 8,
                "the second argument block starts right behind the first in the kernel-argument segment (no padding)");
  const MatcherArgs& M = *reinterpret_cast<const MatcherArgs*>(reinterpret_cast<const char*>(seq_table()) + sizeof(BatchArgs<NSEQ>));
  using Smem = FrameSmem<SIDE>;
  constexpr int HALF = SIDE / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& S = *reinterpret_cast<Smem*>(smem_raw);
  const int tid = threadIdx.x;
  {  // the kernel arguments every workgroup needs first, requested TOGETHER (left alone the compiler fetches ahead_wgs, waits, branches, and
     // only then asks for the pointers behind which the counts and the first unit lie: one more scalar round trip in front of every unit)
    const int a = M.ahead_wgs, cap = M.shard_cap;
    const unsigned int g = gridDim.x;
    const void* p0 = M.units; const void* p1 = M.shards_cur; const void* p2 = M.conv_out;
    // ... and one word from each of the other 64-byte lines of the argument segment that a workgroup with a unit reads on its way to the
    // tile's descriptor loads (image geometry, diagnostics pointer, workspace planes): the segment lies in device memory, and every line
    // touched for the first time in front of those loads was a miss of the scalar cache on that chain
    const int w0 = Bq[0].P.w;
    const void* p3 = Bq[0].P.trace; const void* p4 = M.mean;
    asm volatile("" :: "s"(a), "s"(cap), "s"(g), "s"(p0), "s"(p1), "s"(p2), "s"(w0), "s"(p3), "s"(p4));
  }
  // One step ahead (MatcherArgs::ahead): the first ahead_wgs workgroups -- dispatched first, so they never queue behind the persistent
  // ones -- bring the next host frame into the other current-image plane if the setup kernel found it in the ring, and leave.
  // (As the LAST workgroups of the grid instead they measure the same: profiles/r06_ab_split_frames.txt.)
  if (NSEQ == 1 && blockIdx.x < static_cast<unsigned int>(M.ahead_wgs)) {
    const unsigned int next = M.ingest_number + 1u;
    if (ld_agent(M.ahead) != next) return;
    const SeedParams& P = Bq[0].P;
    const int part = static_cast<int>(blockIdx.x);
    if (M.ingest_flag == nullptr) ingest_in_place(M.ingest_kind, M.ingest_pitch, Bq[0].next_src, Bq[0].next_dst, P.w, P.h, P.stride, part, M.ahead_wgs, tid);
    else ingest_staged(M.ingest_kind, M.ingest_pitch, Bq[0].next_src, Bq[0].next_dst, P.w, P.h, P.stride, part, M.ahead_wgs, tid, [](const unsigned int* p) { return *p; });
    // The last one to finish publishes the frame.  Plane and number are read by the NEXT kernels only, and a kernel's stores are all
    // visible to the kernels behind it on the stream: no fence here (an agent-scope fence writes back and invalidates the L2 the
    // searching workgroups live on -- a hundred of them made every update 20 us longer).
    if (tid == 0 && __hip_atomic_fetch_add(M.ahead + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == static_cast<unsigned int>(M.ahead_wgs) - 1u) {
      __hip_atomic_store(M.ahead + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(M.ahead + 2, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  const unsigned int wg_id = blockIdx.x - static_cast<unsigned int>(M.ahead_wgs), n_wg = gridDim.x - static_cast<unsigned int>(M.ahead_wgs);
  const const_u64_ptr counts = (const_u64_ptr)(M.shards_cur);
  // On a LIGHT frame -- no shard holds more units than a sixteenth of the grid, which is every frame of the benchmark sequence after its first ~75 --
  // the units are not numbered through the shards (unit_entry: the counts first, then the entry: two scalar round trips in front of every
  // workgroup's first descriptor loads) but taken where they lie: entry i of shard s belongs to workgroup 16 i + s, whose address needs no
  // count.  The workgroup requests that entry together with the counts and learns from them whether what came back is a unit.  (Consecutive
  // units of a tile go to workgroups 16 apart: the same XCD -- they share the tile's window in its L2 --, CUs two apart.)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(4))) u32x4* const_entry_ptr;
  const unsigned int my_shard = wg_id & (UNIT_SHARDS - 1), my_index = wg_id / UNIT_SHARDS;
  const u32x4 e_here = *(const_entry_ptr)(M.units + static_cast<size_t>(my_shard) * M.shard_cap + min(my_index, static_cast<unsigned int>(M.shard_cap) - 1u));
  const unsigned int n_units = unit_count(counts);
  const bool light = unit_max(counts) <= n_wg / UNIT_SHARDS;
  // The LAST workgroup (it has no unit of its own on all but the heaviest frames) adds up the per-tile counts of seeds the setup kernel
  // found CONVERGED and mirrors them, stamped with this update's number, to pinned host memory: getConvergedCount() after an update
  // needs no device synchronisation and no kernel of its own (seed_matrix.cu:195-198, depthmap_node.cpp:142-153).
  if (M.conv_out && wg_id == n_wg - 1) {
    for (int q = 0; q < (NSEQ == 1 ? 1 : M.n_seq); ++q) {
      if (NSEQ > 1 && !seq_table()[q].active) continue;
      int c = 0;
      for (int t = tid; t < M.n_tiles; t += TILE_PIX) c += static_cast<int>(M.tile_conv[static_cast<size_t>(q) * M.n_tiles + t]);
      c = wave_reduce_i32<WaveAdd>(c);
      if ((tid & 63) == 0) S.red[tid >> 6][0] = c;
      __syncthreads();
      if (tid == 0) {
        const unsigned long long n = static_cast<unsigned long long>(static_cast<unsigned int>(S.red[0][0] + S.red[1][0] + S.red[2][0] + S.red[3][0]));
        __hip_atomic_store(M.conv_out + q, (static_cast<unsigned long long>(M.update_number) << 32) | n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __syncthreads();
    }
  }
  if (light ? my_index >= static_cast<unsigned int>(counts[my_shard]) : wg_id >= n_units) return;  // no unit for this workgroup: on a light frame most of the grid leaves here, a few dozen scalar instructions in
  unsigned long long* const trace0 = NSEQ == 1 ? Bq[0].P.trace : nullptr;  // diagnostics (single sequences only)
  unsigned long long* const tr = trace0 && static_cast<int>(wg_id) < M.n_tiles ? trace0 + static_cast<size_t>(wg_id) * FR_TRACE_WORDS : nullptr;
  if (tr && tid == 0) tr[0] = wall_clock64();
  LAB_PROF(if (tid < 8) S.prof[tid] = 0ull;)
  const int tx = tid & (TILE_W - 1), ty = tid >> 4;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned int n_fallback = 0, n_windows = 0, n_done = 0, n_items = 0;
  int lds_tile = -1, x0 = 0, y0 = 0, total = 0;
  const SeqArgs* Qp = seq_table();  // the sequence of the tile in LDS (one sequence: the named argument, see the setup kernel)
  size_t so = 0;                    // ... and where its seeds start in the workspace planes
  FrameWindow W;
  W.clear(); W.m = 0; W.yref = 0;
  // Unit wg_id is ours for free.  When there are more units than workgroups the rest is handed out by sixteen counters (workgroup
  // b draws from counter b % 16, which deals the units n_wg + b % 16 + 16 k: one counter word for a thousand workgroups serialises
  // their returning atomics for 12 us).  Claiming the next unit while the current one is searched was measured twice and lost twice: throughout
  // (round 3: a unit claimed one unit-time earlier is a unit the fastest workgroup cannot take, profiles/r03_batch_ab.txt) and only while two
  // more units per workgroup were left, with the claimed entry fetched by a scalar load during the rounds (round 4: batch of 4 -3 %, batch
  // of 8 -1 %, profiles/r04_early_claim_ab.txt) -- the two round trips it hides are not what a unit waits for.
  const bool handout = n_units > n_wg;
  const unsigned int cls = wg_id & (UNIT_SHARDS - 1);
  // the unit entries were written by the setup kernel, the launch before this one: scalar loads (the address is uniform, the words land in
  // scalar registers, nothing waits on the vector memory counter)
  unsigned int u = light ? 0u : wg_id;  // (light: any number below n_units -- the loop below ends after this workgroup's one unit)
  u32x4 e = e_here;
  if (!light) e = *(const_entry_ptr)unit_entry(M, counts, u);
  while (u < n_units) {
    // the unit: (tile, first item | UNIT_TILE_BOX, the texel box of all samples of the tile) -- uniform over the workgroup: scalar registers
    const int tile = static_cast<int>(e.x & ((1u << UNIT_ROUNDS_SHIFT) - 1u));
    const int unit_items = (static_cast<int>(e.x >> UNIT_ROUNDS_SHIFT) + 1) * TILE_PIX;  // the unit size travels with the unit: no word of the setup kernel's to fetch first
    const unsigned int fy = e.y;
    const int first = static_cast<int>(fy & ~UNIT_TILE_BOX);
    const bool boxed = (fy & UNIT_TILE_BOX) != 0u;
    if (tile != lds_tile) {
      const unsigned int box0 = e.z, box1 = e.w;
      if (lds_tile >= 0) {  // hand the previous tile's keys over
        const unsigned long long key = S.best[tid];
        if (key != 0ull) atomicMax(&M.best[so + static_cast<size_t>(y0 + ty) * Qp->P.stride + x0 + tx], key);
      }
      const int seq = NSEQ == 1 ? 0 : tile / M.n_tiles;
      const int tile_s = NSEQ == 1 ? tile : tile - seq * M.n_tiles;  // within its sequence
      if (NSEQ > 1) Qp = seq_table() + seq;
      {  // what stands between the unit's entry and the tile's descriptor loads, requested together (at its point of use each of these
         // scalar loads is waited for on its own: seven scalar-cache round trips in a row on every workgroup's way to its first unit)
        const SeedParams& Pq = Qp->P;
        const int a0 = Pq.w, a1 = Pq.h, a2 = Pq.stride, a3 = M.tiles_x;
        const void* q0 = M.mean; const void* q1 = M.dir; const void* q2 = M.lfirst; const void* q3 = M.packed;
        const void* q4 = Pq.sum_templ; const void* q5 = Pq.denom; const void* q6 = Pq.ref;
        asm volatile("" :: "s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(q0), "s"(q1), "s"(q2), "s"(q3), "s"(q4), "s"(q5), "s"(q6));
      }
      so = NSEQ == 1 ? 0 : static_cast<size_t>(seq) * M.seq_plane;
      const SeedParams& P = Qp->P;
      const int tile_y = tile_s / M.tiles_x, tile_x = tile_s - tile_y * M.tiles_x;
      x0 = tile_x * TILE_W; y0 = tile_y * TILE_H;
      const int x = x0 + tx, y = y0 + ty;
      // ONE batch of loads: the tile's descriptors, its patch halo of the reference image and -- when the setup kernel found that the box
      // of all the tile's samples fits the LDS window and sent it along with the unit -- that window of the current image
      // (a lane outside a ragged tile reads element 0; values of seeds without work are not used)
      const int gi = (x < P.w && y < P.h) ? y * P.stride + x : 0;
      const size_t gm = so + gi;
      const unsigned int packed = M.packed[gm];  // first in-image step << 16 | number of in-image steps
      const float2 m = M.mean[gm], d = M.dir[gm];
      const float lf = M.lfirst[gm], st = P.sum_templ[gi], dn = P.denom[gi];
      // the patch halo, row-wise like the window: wave v takes halo rows v, v + 4, ..., its first REF_W lanes the columns; the clamped
      // column is computed once per lane, the clamped row is scalar
      static_assert(Smem::REF_W <= 64, "one halo row per wave instruction");
      const int ref_col = clampi(x0 - HALF + lane, 0, P.w - 1);
      if (lane < Smem::REF_W) {  // LDS-direct like the window (frame_stage_window): a row of the halo per instruction, nothing held in registers
        typedef const __attribute__((address_space(1))) float* gptr_t;
        typedef __attribute__((address_space(3))) float* lptr_t;
        for (int ry = wave; ry < Smem::REF_H; ry += 4) {
          const float* row = P.ref + static_cast<size_t>(clampi(y0 - HALF + ry, 0, P.h - 1)) * P.stride;
          __builtin_amdgcn_global_load_lds((gptr_t)(row + ref_col), (lptr_t)(S.ref + ry * Smem::REF_W), 4, 0, 0);
        }
      }
      W.m = static_cast<int>(box1) >> 17; W.yref = y0;  // the tile's shear travels with every unit (frame_search cuts its windows along it)
      if (boxed) {
        W.x0 = static_cast<int>(static_cast<short>(box0 & 0xffffu)); W.y0 = static_cast<int>(box0 >> 16);
        W.rows = static_cast<int>(box1 & 0xffu); W.ww = static_cast<int>((box1 >> 8) & 0x1ffu);
        W.ws = window_stride(W.ww);
        W.valid = true;
        frame_stage_window<SIDE>(P, S, tid, W);
      }
      unsigned int pk = 0u;
      if (x < P.w && y < P.h) pk = ((packed >> 16) << 8) | (packed & 0xffu);
      S.mean_x[tid] = m.x; S.mean_y[tid] = m.y; S.dir_x[tid] = d.x; S.dir_y[tid] = d.y;
      S.l_first[tid] = lf;
      S.sum_templ[tid] = st; S.denom[tid] = dn;
      S.packed[tid] = pk;
      S.best[tid] = 0ull;
      // (a unit without the box flag belongs to a tile whose sample box the setup kernel found too large for the LDS window: the box is not
      // computed a second time here -- frame_search cuts windows to the unit's own rounds)
      if (!boxed) W.clear();  // (keeps m and yref)
      drain_vmem();  // halo and window rows are read by OTHER waves after the barriers below: this wave's LDS-direct transfers must have landed
      total = frame_prefix<SIDE>(S, tid);  // barriers inside
      lds_tile = tile;
      if (tr && tid == 0 && n_done == 0) tr[1] = wall_clock64();
    }
    frame_search<SIDE>(Qp->P, S, tid, first, min(first + unit_items, total), W, n_fallback, n_windows);  // ends with a barrier
    ++n_done; n_items += static_cast<unsigned int>(min(first + unit_items, total) - first);
    if (!handout) break;  // light frame: every unit had its own workgroup, nothing to hand out
    if (tid == 0) S.bcast[0] = n_wg + cls + UNIT_SHARDS * atomicAdd(&M.handout[cls * HANDOUT_STRIDE], 1u);
    __syncthreads();
    u = static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(S.bcast[0])));
    __syncthreads();
    if (u < n_units) e = *(const_entry_ptr)unit_entry(M, counts, u);
  }
  if (lds_tile >= 0) {
    const unsigned long long key = S.best[tid];
    if (key != 0ull) atomicMax(&M.best[so + static_cast<size_t>(y0 + ty) * Qp->P.stride + x0 + tx], key);
  }
  if (NSEQ == 1 && Bq[0].P.stats) {  // diagnostics (COLLECT_STATS = 1): evaluations that read their texels from L2 instead of the LDS window, units, windows staged in the search
    const unsigned long long fb = wave_sum_u64(n_fallback);
    if ((tid & 63) == 0 && fb) atomicAdd(&Bq[0].P.stats[3], fb);
    if (tid == 0) { atomicAdd(&Bq[0].P.stats[4], static_cast<unsigned long long>(n_done)); atomicAdd(&Bq[0].P.stats[5], static_cast<unsigned long long>(n_windows)); }
  }
  if (tr && tid < 64) {
    const unsigned long long fb = wave_sum_u64(n_fallback);  // the first wave's lanes only: a hint, not a count
    if (tid == 0) {
      tr[3] = wall_clock64();
      tr[4] = n_items; tr[5] = n_done; tr[6] = static_cast<unsigned long long>(lds_tile >= 0 ? lds_tile : 0);
      LAB_PROF(
      tr[6] = S.prof[0];
      tr[2] = (S.prof[4] & 0xfffffull) | ((S.prof[5] & 0xfffffull) << 20) | ((S.prof[6] & 0xffffffull) << 40);  // window policy, staging, rounds incl. barrier
      )
      tr[7] = fb | (static_cast<unsigned long long>(n_windows) << 32);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// per-seed finalisation (epipolar_match.cu:131-139 + seed_update.cu:39-121), used by seed_finalize_kernel and by the
// fused prologue of the setup kernel
// Decodes the arg-max of one seed whose state is UPDATE, writes the match, runs the Bayesian fusion.  Returns the
// seed's final state of that frame.
RMDK_D int finalize_seed(const SeedParams& P, const MatcherArgs& M, int x, int y, int gi) {
  const unsigned long long key = M.best[gi];
  F2 best_px = F2{0.0f, 0.0f};
  float best_ncc = -1.0f;
  if (key != 0ull) {
    best_ncc = from_orderable_f32(static_cast<unsigned int>(key >> 32));
    const int step = static_cast<int>(0xffffffffu - static_cast<unsigned int>(key & 0xffffffffu));
    const int i_first = static_cast<int>(M.packed[gi] >> 16);
    const float l = replay_l(M.lfirst[gi], step - i_first);
    const float2 m = M.mean[gi], d = M.dir[gi];
    best_px = F2{m.x + l * d.x, m.y + l * d.y};
  }
  int state = ST_UPDATE;
  if (best_ncc < 0.5f) state = ST_NO_MATCH;
  else P.match[y * P.stride2 + x] = make_float2(best_px.x, best_px.y);
  seed_fuse(P, x, y, gi, state, P.mu[gi], P.sigma_sq[gi], P.a[gi], P.b[gi], best_px);
  return state;
}

// ------------------------------------------------------------------------------------------------
// stage 3: the stand-alone finalisation kernel (the per-seed code is finalize_seed above)
static __global__ __launch_bounds__(256) void seed_finalize_kernel(SeedParams P, MatcherArgs M) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= P.w || y >= P.h) return;
  const int gi = y * P.stride + x;
  if (P.conv[gi] != ST_UPDATE) return;  // BORDER / CONVERGED / DIVERGED were settled by seed_setup
  const int state = finalize_seed(P, M, x, y, gi);
  if (state != ST_UPDATE) P.conv[gi] = state;
}

// the stand-alone finalisation of sequence `seq`'s frame whose pipeline was launched last (P must carry that frame's poses)
inline hipError_t launch_seed_finalize(const SeedParams& P, MatcherWorkspace& ws, hipStream_t stream, int seq = 0) {
  const MatcherArgs M = matcher_args_of(ws, seq);
  hipLaunchKernelGGL(seed_finalize_kernel, dim3((P.w + 63) / 64, (P.h + 3) / 4), dim3(64, 4), 0, stream, P, M);
  return hipGetLastError();
}

// setup (+ the deferred finalisation of the previous frame of every sequence with fuse_prev; builds the unit list) -> search, for the
// `n_seq` sequences of B (NSEQ = 1: one sequence; NSEQ = MAX_GROUP_SEQ: 2..MAX_GROUP_SEQ of them).  The caller zeroes ws.d_shards and sets
// ws.frame = 0 whenever a sequence restarts.
template <int SIDE, int NSEQ>
inline hipError_t launch_seed_pipeline_compact(const BatchArgs<NSEQ>& B, int n_seq, MatcherWorkspace& ws, hipStream_t stream, int num_cus,
                                               int target_mult = 1, const IngestArgs* ingest = nullptr) {
  using Smem = FrameSmem<SIDE>;
  ++ws.update_number;
  MatcherArgs M = matcher_args(ws);
  M.n_seq = n_seq;
  M.housekeeper = 0;
  for (int q = n_seq - 1; q >= 0; --q)
    if (B.seq[q].active) M.housekeeper = q;
  if (ingest && ingest->kind) {
    M.ingest_kind = ingest->kind; M.ingest_pitch = ingest->pitch;
    M.ingest_flag = ingest->flag; M.progress = ingest->progress; M.ingest_number = ingest->number;
    if (NSEQ == 1 && ingest->ahead) { M.ahead_wgs = ingest->ahead_wgs; M.submitted = ingest->submitted; M.ahead = ingest->ahead; }
  }
  auto search = seed_search_compact_kernel<SIDE, NSEQ>;
  constexpr int KIND = NSEQ == 1 ? 0 : 1;
  bool& attr = ws.attr_set_compact[SIDE / 2 - 1][KIND];
  if (!attr) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(search), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(sizeof(Smem)));
    if (e != hipSuccess) return e;
    int wg = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg, reinterpret_cast<const void*>(search), TILE_PIX, sizeof(Smem)) != hipSuccess || wg < 1) wg = 1;
    ws.compact_wg_per_cu[SIDE / 2 - 1][KIND] = wg > 4 ? 4 : wg;
    attr = true;
  }
  const int resident = num_cus * ws.compact_wg_per_cu[SIDE / 2 - 1][KIND];
  dim3 tiles(ws.tiles_x, ws.tiles_y, n_seq);
  const SeedParams& P0 = B.seq[0].P;
  if (M.ingest_kind) {  // the ingest workgroups: rows in front of the tile grid
    bool remap = false;
    for (int q = 0; q < n_seq; ++q) remap = remap || (M.ingest_kind == 1 && B.seq[q].ingest_map1);
    const long long dwords = M.ingest_kind == 1 && !remap ? static_cast<long long>(M.ingest_pitch >> 2) * P0.h : static_cast<long long>(P0.w) * P0.h;
    const bool in_place = M.ingest_flag == nullptr;  // 16 bytes per lane and request
    if (remap || in_place) {
      const long long requests = in_place ? (dwords + 3) / 4 : dwords;
      const long long want = (requests + TILE_PIX - 1) / TILE_PIX, cap = remap ? INGEST_WGS_REMAP : INGEST_WGS_IN_PLACE;
      M.ingest_wgs = static_cast<int>(want < cap ? want : cap);
    } else {
      M.ingest_wgs = ingest_workgroups(dwords, INGEST_WGS);
    }
    M.ingest_rows = (M.ingest_wgs + ws.tiles_x - 1) / ws.tiles_x;
    M.ingest_profile = ingest->profile ? 1 : 0;
    tiles.y += static_cast<unsigned int>(M.ingest_rows);
    if (NSEQ == 1 && M.ahead) tiles.y += 1u;  // the row behind the tile grid that takes the verdict for the next frame
  }
  const int target_units = resident * target_mult;
  hipLaunchKernelGGL((seed_setup_compact_kernel<SIDE, NSEQ>), tiles, dim3(TILE_PIX), 0, stream, B, M, target_units);
  hipLaunchKernelGGL(search, dim3(resident + M.ahead_wgs), dim3(TILE_PIX), sizeof(Smem), stream, B, M);
  ++ws.frame;
  return hipGetLastError();
}

}  // namespace rmdk

#endif  // RMD_FRAME_HPP
