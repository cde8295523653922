// The production seed update: seed_check (src/seed_check.cu:28-67) + epipolar NCC search
// (src/epipolar_match.cu:37-140) + triangulation and Bayesian fusion (src/seed_update.cu:39-121,
// src/triangulation.cu) as a load-balanced pipeline of small kernels on one stream:
//
//   seed_setup_kernel     one 256-thread workgroup per 16x16 tile of seeds, one lane per seed.
//                         State check; epipolar segment; ONE walk of the search loop to find the
//                         contiguous run of steps whose patch lies inside the image.  Writes the
//                         per-seed search descriptor, the tile's total work and the image window its
//                         samples fall into.
//   seed_plan_kernel      one workgroup: unit size from the frame's total work, exclusive scan of the tiles' unit counts ->
//                         compact unit list (keeps every atomic out of seed_setup: 1200 tiles x 3 atomics on shared words
//                         cost 40 us per frame).  All scheduling decisions are taken on the device.
//   seed_search_kernel    persistent workgroups pull units from the queue (one returning atomic per
//                         unit), so a tile whose seeds all search 143 steps is spread over many CUs
//                         while converged tiles cost nothing.  Per unit: stage the tile's window of
//                         the current image and its reference-image tile in LDS (batched, coalesced
//                         loads), deal the unit's (seed, step) pairs round-robin to the 256 lanes, one
//                         NCC evaluation per pair with the separable bilinear filter of rmd_math.h
//                         ((SIDE+1)^2 LDS reads feed SIDE*(SIDE+1) + SIDE^2 lerps instead of 4*SIDE^2
//                         texel fetches), segmented wave max, one 64-bit atomic max per seed run on
//                         {orderable(ncc), ~step} (ties -> lowest step, as the reference's strict '>').
//   seed_finalize         one lane per seed: decode the arg-max, match coordinates, triangulation, posterior update.
//                         Deferred: when the next update() arrives before anybody looked at the state, it runs
//                         fused into that frame's seed_setup_kernel (same lane owns the same seed); otherwise as the
//                         stand-alone seed_finalize_kernel.
//
// Why not one lane per pixel (the reference's shape, kept in rmd_kernels.hpp as the A/B baseline):
// the per-seed trip count varies from 0 (converged / diverged / border) to 143, so a wave64 idles at
// the pace of its longest lane and every lane gathers its own texels through the vector memory path.
// Why not one fused kernel per tile (the first version of this file): kernel time was the time of
// the heaviest tile (max workgroup 0.6-2.5 M cycles vs 0.1 M average on the benchmark sequence).
// Results are bit-identical to the reference semantics (tests/test_hip_parity.py).
#ifndef RMD_MATCHER_HPP
#define RMD_MATCHER_HPP

#include "rmd_kernels.hpp"

namespace rmdk {

constexpr int TILE_W = 16, TILE_H = 16, TILE_PIX = TILE_W * TILE_H;
constexpr int MAX_UNIT_ROUNDS = 4;  // a work unit is 1..4 rounds of the 256 lanes (chosen per frame by seed_plan)
constexpr int MIN_UNIT_ITEMS = TILE_PIX;
constexpr int MAX_ITEMS_PER_TILE = TILE_PIX * 144;
constexpr int UNIT_SHARDS = 16;  // plan-free pipeline: unit lists / counters, tile t -> shard t % UNIT_SHARDS
constexpr int TRACE_FRAMES = 256, TRACE_MAX_SEARCH_WGS = 1024;  // diagnostics (trace_record)

// per-tile record written by seed_setup, read by seed_search
struct TileInfo {
  int total;               // NCC evaluations of the tile
  int wx0, wy0, wx1, wy1;  // inclusive texel box of the current image staged for the tile
};

struct MatcherWorkspace {
  int tiles_x = 0, tiles_y = 0, stride = 0;
  float2* d_mean = nullptr;   // per seed: projection of mu into the current frame
  float2* d_dir = nullptr;    // per seed: unit direction of the epipolar segment
  float* d_lfirst = nullptr;  // per seed: accumulated l at the first in-image step
  unsigned int* d_packed = nullptr;      // per seed: first in-image step << 16 | number of in-image steps
  unsigned long long* d_best = nullptr;  // per seed: arg-max key
  TileInfo* d_tiles = nullptr;
  unsigned int* d_tile_plan = nullptr;  // per tile, for seed_plan: work items
  unsigned int* d_tile_pending = nullptr;  // per tile (two-launch pipeline): work units not yet searched
  uint2* d_units = nullptr;         // (tile, first item)
  // counters of the current frame, rewritten by seed_plan every frame: [0] work units, [1] units handed out beyond the
  // static first round, [5] items per unit
  unsigned int* d_queue = nullptr;
  unsigned long long* d_shards = nullptr;  // plan-free pipeline: 3 sets (frame % 3) of UNIT_SHARDS counters {work items << 32 | units}
  long long frame = 0;                     // updates since the last reference (plan-free pipeline: which set is current)
  int shard_cap = 0;                       // unit-list entries per shard
  int lds_bytes = 160 * 1024;              // LDS per CU of the handle's device (gfx950: 160 KB)
  unsigned long long* d_trace = nullptr;  // diagnostics, allocated on demand: TRACE_FRAMES slices of trace_slice_u64() words
  int max_units = 0;
  bool attr_set_small = false, attr_set_large = false;
  int allocate(int w, int h, int stride_elems) {
    tiles_x = (w + TILE_W - 1) / TILE_W;
    tiles_y = (h + TILE_H - 1) / TILE_H;
    stride = stride_elems;
    const size_t n = static_cast<size_t>(stride) * h;
    max_units = (tiles_x * tiles_y + UNIT_SHARDS) * ((MAX_ITEMS_PER_TILE + MIN_UNIT_ITEMS - 1) / MIN_UNIT_ITEMS);
    if (hipMalloc(reinterpret_cast<void**>(&d_mean), n * sizeof(float2)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_dir), n * sizeof(float2)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_lfirst), n * sizeof(float)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_packed), n * sizeof(unsigned int)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_best), n * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_tiles), static_cast<size_t>(tiles_x) * tiles_y * sizeof(TileInfo)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_tile_plan), static_cast<size_t>(tiles_x) * tiles_y * sizeof(unsigned int)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_tile_pending), static_cast<size_t>(tiles_x) * tiles_y * sizeof(unsigned int)) != hipSuccess) return -1;
    (void)hipMemset(d_tile_pending, 0, static_cast<size_t>(tiles_x) * tiles_y * sizeof(unsigned int));
    shard_cap = ((tiles_x * tiles_y + UNIT_SHARDS - 1) / UNIT_SHARDS) * ((MAX_ITEMS_PER_TILE + MIN_UNIT_ITEMS - 1) / MIN_UNIT_ITEMS);  // tiles of a shard x units of a tile
    if (max_units < UNIT_SHARDS * shard_cap) max_units = UNIT_SHARDS * shard_cap;
    if (hipMalloc(reinterpret_cast<void**>(&d_units), static_cast<size_t>(max_units) * sizeof(uint2)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_queue), 8 * sizeof(unsigned int)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_shards), 3 * UNIT_SHARDS * sizeof(unsigned long long)) != hipSuccess) return -1;
    (void)hipMemset(d_shards, 0, 3 * UNIT_SHARDS * sizeof(unsigned long long));
    (void)hipMemset(d_packed, 0, n * sizeof(unsigned int));
    (void)hipMemset(d_best, 0, n * sizeof(unsigned long long));
    (void)hipMemset(d_queue, 0, 8 * sizeof(unsigned int));
    return 0;
  }
  size_t trace_slice_u64() const { return 2 * (static_cast<size_t>(tiles_x) * tiles_y + 1 + TRACE_MAX_SEARCH_WGS); }
  void release() {
    void* all[] = {d_mean, d_dir, d_lfirst, d_packed, d_best, d_tiles, d_tile_plan, d_tile_pending, d_units, d_queue, d_shards, d_trace};
    for (void* p : all)
      if (p) (void)hipFree(p);
    d_mean = d_dir = nullptr; d_lfirst = nullptr; d_packed = nullptr; d_best = nullptr;
    d_tiles = nullptr; d_tile_plan = nullptr; d_tile_pending = nullptr; d_units = nullptr; d_queue = nullptr; d_shards = nullptr; d_trace = nullptr;
  }
};

// device view of the workspace
struct MatcherArgs {
  float2* mean;
  float2* dir;
  float* lfirst;
  unsigned int* packed;
  unsigned long long* best;
  TileInfo* tiles;
  unsigned int* tile_plan;
  unsigned int* tile_pending;
  uint2* units;
  unsigned int* queue;       // this frame's counters (see MatcherWorkspace)
  unsigned long long* shards_cur;         // plan-free pipeline: this frame's shard counters (zero at launch)
  const unsigned long long* shards_prev;  // the previous frame's (null: no previous frame)
  unsigned long long* shards_next;        // cleared by this frame's setup for the next one
  int shard_cap;
  int tiles_x;
  int tiles_y;
  unsigned long long* trace;  // this frame's slice of the timeline buffer, may be null (see trace_record)
  // Frame ingest for frames handed over in host memory (compact pipeline): a copy engine brings the frame as it is into a staging
  // buffer in HBM and then writes the frame's number into `ingest_flag`, both on the handle's copy stream, with NO ordering against
  // the compute stream; a few extra workgroups of the setup kernel wait for the flag themselves (it has normally been set long
  // before) and convert the staged frame into the current-image plane, which only the search kernel -- the next launch -- reads.
  // Null pointers: the frame is resident already.
  const unsigned int* ingest_u8;   // staged 8-bit rows of ingest_pitch bytes (a multiple of 4), four pixels per dword
  const float* ingest_f32;         // or staged float rows of P.w elements, unpadded
  float* ingest_dst;               // the current-image plane, row stride P.stride
  int ingest_pitch;
  const short2* ingest_map1;       // lens undistortion of 8-bit frames (Depthmap::initUndistortionMap): source pixel per destination pixel
  const unsigned short* ingest_map2;  // ... and its 5-bit fractions; null = frames are used as they come
  int ingest_wgs;                  // workgroups below the tile grid that do the conversion
  const unsigned int* ingest_flag; // device word: number of the last frame whose staging copy has completed
  unsigned int* progress;          // pinned host words: [0] <- ingest_number when this setup starts (everything before it has
                                   // completed: the host may reuse that frame's buffers), [1] |= 1 if the flag never came
  unsigned int ingest_number;
};

// what the caller of the compact pipeline hands over when the frame came from host memory
struct IngestArgs {
  const unsigned int* u8 = nullptr;
  const float* f32 = nullptr;
  float* dst = nullptr;
  int pitch = 0;
  const short2* map1 = nullptr;
  const unsigned short* map2 = nullptr;
  const unsigned int* flag = nullptr;
  unsigned int* progress = nullptr;
  unsigned int number = 0;
};

// Timeline probe of one workgroup (diagnostics): record `slot` of the frame's trace slice gets the workgroup's start and
// end in 10 ns ticks of the device-wide wall clock.  Slots: one per setup tile, then seed_plan, then the search workgroups.
RMDK_D void trace_record(unsigned long long* trace, int slot, unsigned long long t0, unsigned long long t1) {
  trace[2 * slot] = t0;
  trace[2 * slot + 1] = t1;
}

RMDK_D unsigned int orderable_f32(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
RMDK_D float from_orderable_f32(unsigned int o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// The reference accumulates the search position, l += 0.7f once per step (epipolar_match.cu:88), and the float sequence has
// no closed form, so whoever needs l at step n replays n additions.  The replays are dependent chains inside divergent
// loops; running them in blocks of 8 additions per loop-control round makes them ~4x cheaper (same additions, same order).
RMDK_D float replay_l(float l, int n) {
  int q = 0;
  for (; q + 8 <= n; q += 8) {
#pragma unroll
    for (int r = 0; r < 8; ++r) l += 0.7f;
  }
  for (; q < n; ++q) l += 0.7f;
  return l;
}
// for (; i < i_target && l <= half; ++i) l += 0.7f;   -- l grows monotonically, so a block of 8 iterations runs to its end
// exactly when the value before its last addition still passes the test
RMDK_D void replay_until(int& i, float& l, int i_target, float half) {
  while (i + 8 <= i_target) {
    float t = l;
#pragma unroll
    for (int r = 0; r < 7; ++r) t += 0.7f;
    if (!(t <= half)) break;
    l = t + 0.7f;
    i += 8;
  }
  for (; i < i_target && l <= half; ++i) l += 0.7f;
}

// number of work units of a tile with `items` work items; a unit is `rounds` (1..MAX_UNIT_ROUNDS) rounds of TILE_PIX items
RMDK_D int units_of(int items, int rounds) {
  const int r = (items + TILE_PIX - 1) / TILE_PIX;  // rounds of work, a shift
  switch (rounds) {
    case 1: return r;
    case 2: return (r + 1) / 2;
    case 3: return (r + 2) / 3;
    default: return (r + 3) / 4;
  }
}

RMDK_D bool px_outside(const SeedParams& P, F2 px, int side) {  // the guard at epipolar_match.cu:91-97
  return px.x >= static_cast<float>(P.w - side) || px.y >= static_cast<float>(P.h - side) ||
         px.x < static_cast<float>(side) || px.y < static_cast<float>(side);
}

// Which steps of the search loop (epipolar_match.cu:88: l = -half; l <= half; l += 0.7f) pass the in-image guard (:91-97)?
// px(l) = fl(mean + fl(l * dir)) is monotone per axis (rounding is monotone) and the guard region is a box, so they form ONE
// contiguous run.  Real-arithmetic bounds [la, lb] of that run tell where to look; the float sequence l_i itself has no
// closed form, so it is replayed (one add per step), and the exact guard is evaluated only around the two ends.
// A non-finite position yields NaN sums in the reference and never becomes a candidate: no work.
struct ValidRun {
  int n_valid, i_first;  // number of in-image steps, index of the first one
  float l_first;         // accumulated l at the first one
  F2 px_first, px_last;  // sample positions of the first and the last one
};
RMDK_D ValidRun find_valid_run(const SeedParams& P, const Segment& seg, int side) {
  ValidRun r;
  r.n_valid = 0; r.i_first = 0; r.l_first = 0.0f;
  r.px_first = F2{0.0f, 0.0f}; r.px_last = F2{0.0f, 0.0f};
  const float half = seg.half_length;
  const bool finite = isfinite(seg.mean.x) && isfinite(seg.mean.y) && isfinite(seg.dir.x) && isfinite(seg.dir.y);
  if (!finite) return r;
  float la = -half, lb = half;
  bool empty = false;
  const float lo_x = static_cast<float>(side), hi_x = static_cast<float>(P.w - side);
  const float lo_y = static_cast<float>(side), hi_y = static_cast<float>(P.h - side);
  // An axis along which the segment barely moves (|dir| * step below the spacing of floats near the image coordinates)
  // gives no usable estimate: the rounding of mean + l * dir decides on which side of a bound the sample falls, and
  // (bound - mean) / dir is off by 0.5 ulp(mean) / |dir| steps.  Along such an axis the whole segment stays within
  // half * TINY_DIR <= 0.1 px of mean: if mean is further than that (and a margin) outside the bounds no step is inside, if
  // it is further inside the axis never matters, and only in the narrow band around a bound the exact guard decides alone.
  constexpr float TINY_DIR = 2e-3f;  // 0.7 * 2e-3 = 12 ulp of 8192.0f: the estimate is good to a fraction of a step above it
  const float band = half * TINY_DIR + 0.125f;
  if (seg.dir.x > TINY_DIR) { la = fmaxf(la, (lo_x - seg.mean.x) / seg.dir.x); lb = fminf(lb, (hi_x - seg.mean.x) / seg.dir.x); }
  else if (seg.dir.x < -TINY_DIR) { la = fmaxf(la, (hi_x - seg.mean.x) / seg.dir.x); lb = fminf(lb, (lo_x - seg.mean.x) / seg.dir.x); }
  else if (seg.dir.x == 0.0f) empty = empty || !(seg.mean.x >= lo_x && seg.mean.x < hi_x);
  else empty = empty || !(seg.mean.x >= lo_x - band && seg.mean.x < hi_x + band);
  if (seg.dir.y > TINY_DIR) { la = fmaxf(la, (lo_y - seg.mean.y) / seg.dir.y); lb = fminf(lb, (hi_y - seg.mean.y) / seg.dir.y); }
  else if (seg.dir.y < -TINY_DIR) { la = fmaxf(la, (hi_y - seg.mean.y) / seg.dir.y); lb = fminf(lb, (lo_y - seg.mean.y) / seg.dir.y); }
  else if (seg.dir.y == 0.0f) empty = empty || !(seg.mean.y >= lo_y && seg.mean.y < hi_y);
  else empty = empty || !(seg.mean.y >= lo_y - band && seg.mean.y < hi_y + band);
  // the estimates are good to ~1e-3 of a step; everything within 2 steps of them is checked exactly
  if (empty || !(la <= lb + 1.5f)) return r;
  const int i_a = max(static_cast<int>(floorf((la + half) * (1.0f / 0.7f))) - 2, 0);
  int i = 0;
  float l = -half;
  replay_until(i, l, i_a, half);  // replay
  for (; l <= half && l <= lb + 1.5f; l += 0.7f, ++i) {  // exact scan for the first in-image step
    const F2 px = F2{seg.mean.x + l * seg.dir.x, seg.mean.y + l * seg.dir.y};
    if (!px_outside(P, px, side)) { r.n_valid = 1; r.i_first = i; r.l_first = l; r.px_first = px; r.px_last = px; break; }
  }
  if (!r.n_valid) return r;
  const int i_b = max(static_cast<int>(floorf((lb + half) * (1.0f / 0.7f))) - 2, r.i_first);
  replay_until(i, l, i_b, half);  // replay across the interior of the run (in-image by convexity)
  int i_last = r.i_first;
  if (l <= half && i > r.i_first) {  // the replayed position: still in the run unless the estimate overshot
    const F2 px = F2{seg.mean.x + l * seg.dir.x, seg.mean.y + l * seg.dir.y};
    if (!px_outside(P, px, side)) { i_last = i; r.px_last = px; }
    else { i = r.i_first; l = r.l_first; }  // overshoot (an axis left to the exact guard): rescan from the first step
  } else if (i > r.i_first) { i = r.i_first; l = r.l_first; }
  for (l += 0.7f, ++i; l <= half; l += 0.7f, ++i) {  // exact scan for the last in-image step
    const F2 px = F2{seg.mean.x + l * seg.dir.x, seg.mean.y + l * seg.dir.y};
    if (px_outside(P, px, side)) break;
    i_last = i; r.px_last = px;
  }
  r.n_valid = i_last - r.i_first + 1;
  return r;
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

// The common case of axis_params: when p - HALF .. p + HALF + 0.5 all lie in p's own binade [2^e, 2^(e+1)), every one of
// the additions above is exact (all operands are multiples of ulp(p) <= 0.5 and no result leaves the binade), so
// b_k = p + (OFFSET + k) exactly, idx[k] = floor(p) + OFFSET + k and every weight equals p - floor(p).  Away from
// powers of two (about 80 % of all positions of a VGA frame) that replaces the replay by three instructions.
template <int SIDE>
RMDK_D bool axis_is_uniform(float p) {
  constexpr float H = static_cast<float>(SIDE / 2);
  const float lo = __uint_as_float(__float_as_uint(p) & 0x7f800000u);  // 2^e
  return p >= 1.0f && p - H >= lo && p + (H + 0.5f) < lo + lo;
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

// Sums of one NCC evaluation over a regular footprint in the LDS window (the hot block of the whole path): the arithmetic of
// ncc_sums_regular, software-pipelined by hand.  The texel row r + 1 and the template row r are requested from the LDS BEFORE
// the filter / accumulate work on row r, and scheduling barriers keep the compiler from sinking the reads back to their first
// use (left alone it issues every read a few instructions before an s_waitcnt: ~60 exposed LDS latencies per evaluation, which
// is most of a round's time when a wave has its SIMD to itself, i.e. on every frame but the first twenty).
template <int SIDE>
RMDK_D void ncc_sums_lds_pipelined(const float* __restrict__ base, int stride, const float (&ax)[SIDE], const float (&ay)[SIDE],
                                   const float* __restrict__ ref_patch, int ref_stride, float& sum_img, float& sum_img_sq,
                                   float& sum_img_templ) {
  float t[2][SIDE + 1], tm[2][SIDE], hprev[SIDE], hcur[SIDE];
#pragma unroll
  for (int c = 0; c <= SIDE; ++c) t[0][c] = base[c];
#pragma unroll
  for (int r = 0; r <= SIDE; ++r) {
    const int cur = r & 1, nxt = cur ^ 1;
    if (r < SIDE) {
      const float* row = base + (r + 1) * stride;
#pragma unroll
      for (int c = 0; c <= SIDE; ++c) t[nxt][c] = row[c];
#pragma unroll
      for (int k = 0; k < SIDE; ++k) tm[nxt][k] = ref_patch[r * ref_stride + k];  // template row r, used with texel rows r, r + 1
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < SIDE; ++k) hcur[k] = rmd_lerp(ax[k], t[cur][k], t[cur][k + 1]);
    if (r > 0) {
      const float by = ay[r - 1];
#pragma unroll
      for (int k = 0; k < SIDE; ++k) {
        const float img = rmd_lerp(by, hprev[k], hcur[k]);
        const float templ = tm[cur][k];
        sum_img += img;
        sum_img_sq += img * img;
        sum_img_templ += img * templ;
      }
    }
#pragma unroll
    for (int k = 0; k < SIDE; ++k) hprev[k] = hcur[k];
    __builtin_amdgcn_sched_barrier(0);
  }
}

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
// per-seed finalisation (epipolar_match.cu:131-139 + seed_update.cu:39-121), used by seed_finalize_kernel and by the
// fused prologue of seed_setup_kernel
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
    for (int u = 0; u < n_u; ++u) M.units[base + u] = make_uint2(static_cast<unsigned int>(t_first + q), static_cast<unsigned int>(u * unit_items));
    base += n_u;
  }
  for (int q = PLAN_TILES_IN_REGS; q < c_tiles; ++q) {
    const int t = t_first + q;
    if (t >= n_tiles) break;
    const int n_u = units_of(static_cast<int>(M.tile_plan[t]), unit_rounds);
    for (int u = 0; u < n_u; ++u) M.units[base + u] = make_uint2(static_cast<unsigned int>(t), static_cast<unsigned int>(u * unit_items));
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
    const uint2 unit = M.units[u];
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

// ------------------------------------------------------------------------------------------------
// stage 3: the stand-alone finalisation kernel (the per-seed code is finalize_seed above)
__global__ __launch_bounds__(256) void seed_finalize_kernel(SeedParams P, MatcherArgs M) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= P.w || y >= P.h) return;
  const int gi = y * P.stride + x;
  if (P.conv[gi] != ST_UPDATE) return;  // BORDER / CONVERGED / DIVERGED were settled by seed_setup
  const int state = finalize_seed(P, M, x, y, gi);
  if (state != ST_UPDATE) P.conv[gi] = state;
}

// ------------------------------------------------------------------------------------------------
inline MatcherArgs matcher_args(const MatcherWorkspace& ws) {
  MatcherArgs M;
  M.mean = ws.d_mean; M.dir = ws.d_dir; M.lfirst = ws.d_lfirst; M.packed = ws.d_packed; M.best = ws.d_best;
  M.tiles = ws.d_tiles; M.tile_plan = ws.d_tile_plan; M.tile_pending = ws.d_tile_pending; M.units = ws.d_units; M.tiles_x = ws.tiles_x; M.tiles_y = ws.tiles_y;
  M.queue = ws.d_queue;
  M.shards_cur = ws.d_shards + (ws.frame % 3) * UNIT_SHARDS;
  M.shards_prev = ws.frame > 0 ? ws.d_shards + ((ws.frame + 2) % 3) * UNIT_SHARDS : nullptr;
  M.shards_next = ws.d_shards + ((ws.frame + 1) % 3) * UNIT_SHARDS;
  M.shard_cap = ws.shard_cap;
  M.trace = nullptr;
  M.ingest_u8 = nullptr; M.ingest_f32 = nullptr; M.ingest_dst = nullptr; M.ingest_pitch = 0; M.ingest_map1 = nullptr; M.ingest_map2 = nullptr; M.ingest_wgs = 0; M.ingest_flag = nullptr; M.progress = nullptr; M.ingest_number = 0u;
  return M;
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

// the stand-alone finalisation of the frame whose pipeline was launched last (P must carry that frame's poses)
inline hipError_t launch_seed_finalize(const SeedParams& P, MatcherWorkspace& ws, hipStream_t stream) {
  const MatcherArgs M = matcher_args(ws);
  hipLaunchKernelGGL(seed_finalize_kernel, dim3((P.w + 63) / 64, (P.h + 3) / 4), dim3(64, 4), 0, stream, P, M);
  return hipGetLastError();
}

}  // namespace rmdk

#endif  // RMD_MATCHER_HPP
