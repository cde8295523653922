// Building blocks of the production seed update -- seed_check (src/seed_check.cu:28-67) + epipolar NCC search
// (src/epipolar_match.cu:37-140) + triangulation and Bayesian fusion (src/seed_update.cu:39-121, src/triangulation.cu) as a
// load-balanced two-launch pipeline (the kernels themselves are in rmd_frame.hpp):
//
//   * the workspace of the pipeline for one or several independent sequences (MatcherWorkspace) and its device view
//     (MatcherArgs); the per-sequence parameter blocks of a launch (SeqArgs / BatchArgs);
//   * the replayed search positions (the reference accumulates l += 0.7f), the contiguous run of in-image steps of a seed
//     (find_valid_run), per-axis sample parameters with the reference's roundings (axis_params / axis_is_uniform);
//   * one NCC evaluation with the separable bilinear filter of rmd_math.h ((SIDE+1)^2 reads feed SIDE*(SIDE+1) + SIDE^2 lerps
//     instead of 4*SIDE^2 texel fetches), from global memory (ncc_sums_regular) or software-pipelined from an LDS window
//     (ncc_sums_lds_pipelined);
//   * the per-seed finalisation (finalize_seed: decode the arg-max {orderable(ncc), ~step} -- ties go to the lowest step like the
//     reference's strict '>' --, match coordinates, triangulation, posterior update) and its stand-alone kernel.  Deferred: when
//     the next update() arrives before anybody looked at the state, it runs fused into that frame's setup kernel (the same lane
//     owns the same seed); otherwise as seed_finalize_kernel.
//
// Why not one lane per pixel (the reference's shape, kept in rmd_kernels.hpp as the A/B baseline): the per-seed trip count
// varies from 0 (converged / diverged / border) to 143, so a wave64 idles at the pace of its longest lane and every lane
// gathers its own texels through the vector memory path.  Why not one fused kernel per tile (the first version of this file):
// kernel time was the time of the heaviest tile (max workgroup 0.6-2.5 M cycles vs 0.1 M average on the benchmark sequence).
// The round-1 form of the pipeline (setup / plan / search with a tile-wide 66 KB window) and the one-launch frame kernel are
// history (LAB.md).  Results are bit-identical to the reference semantics (tests/test_hip_parity.py).
#ifndef RMD_MATCHER_HPP
#define RMD_MATCHER_HPP

#include "rmd_kernels.hpp"

namespace rmdk {

constexpr int TILE_W = 16, TILE_H = 16, TILE_PIX = TILE_W * TILE_H;
constexpr int MAX_UNIT_ROUNDS = 4;  // a work unit is 1..4 rounds of the 256 lanes (chosen per frame from the previous frame's work)
constexpr int MIN_UNIT_ITEMS = TILE_PIX;
// Search steps of one seed: l = -half; l <= half; l += 0.7f with half <= max_extent / 2 (epipolar_match.cu:75,88), i.e. at most
// floor(max_extent / 0.7) + 1.  Step numbers and counts live in 8-bit fields (the search kernel's descriptors, FrameSmem::packed): 255
// steps, i.e. max_extent <= 178 -- the reference's RMD_MAX_EXTENT_EPIPOLAR_SEARCH is an unbounded compile-time constant with default 100
// (CMakeLists.txt:52-53); beyond 178 the fields would have to be widened.
constexpr int MAX_EXTENT_LIMIT = 178;
inline int max_search_steps(int max_extent) { return static_cast<int>(static_cast<float>(max_extent) / 0.7f) + 2; }
static_assert(static_cast<int>(MAX_EXTENT_LIMIT / 0.7f) + 1 <= 255, "step numbers of a seed fit 8 bits");
// unit lists / counters, tile t -> shard t % UNIT_SHARDS; hand-out counters of the search, workgroup b -> b % UNIT_SHARDS
constexpr int UNIT_SHARDS = 16;
constexpr int HANDOUT_STRIDE = 32;  // words between two hand-out counters (128 B: one L2 line each)
// flag in a unit's first-item word: words 2, 3 hold the texel box of ALL samples of the unit's tile, and it fits the LDS window
constexpr unsigned int UNIT_TILE_BOX = 0x80000000u;
// timeline of the tile pipeline, per workgroup: start, setup done, own tile done, exit (10 ns ticks), work items, units searched
constexpr int FR_TRACE_FRAMES = 256, FR_TRACE_WORDS = 8;

// sequences ONE launch pair can carry (the per-sequence parameter blocks travel as kernel arguments: 4 KB)
constexpr int MAX_GROUP_SEQ = 8;
// sequences of a batch: it steps its members in up to three stream groups, one launch pair each (rmd_hip_batch)
constexpr int MAX_BATCH = 3 * MAX_GROUP_SEQ;

// Workspace of the update pipeline for `n_seq` independent sequences of one size that are updated by ONE launch pair (a plain
// SeedMatrix is the case n_seq = 1).  Per-seed planes hold the sequences back to back (`seq_plane` elements each), tiles are
// numbered sequence-major (tile_global = seq * n_tiles + tile), unit lists and shard counters are shared by all sequences.
struct MatcherWorkspace {
  int tiles_x = 0, tiles_y = 0, stride = 0, n_seq = 1;
  size_t seq_plane = 0;       // elements per sequence in the per-seed planes below
  float2* d_mean = nullptr;   // per seed: projection of mu into the current frame
  float2* d_dir = nullptr;    // per seed: unit direction of the epipolar segment
  float* d_lfirst = nullptr;  // per seed: accumulated l at the first in-image step
  unsigned int* d_packed = nullptr;      // per seed: first in-image step << 16 | number of in-image steps
  unsigned long long* d_best = nullptr;  // per seed: arg-max key
  // per tile: seeds in state UPDATE after the last frame's check (0: the tile is dead until the next reference frame)
  unsigned int* d_tile_live = nullptr;
  unsigned int* d_tile_conv = nullptr;   // per tile: seeds that seed_check found CONVERGED in this frame
  uint4* d_units = nullptr;         // work units: (tile, first item | UNIT_TILE_BOX, the tile's sample box x0 | y0 << 16, x1 | y1 << 16)
  unsigned int* d_handout = nullptr;  // UNIT_SHARDS hand-out counters of the search kernel, HANDOUT_STRIDE words apart
  // counters of the current frame: [0] work units (round-1 plan kernel), [1] units handed out beyond the static first round,
  // [5] items per unit
  unsigned long long* d_shards = nullptr;  // 3 sets (frame % 3) of UNIT_SHARDS counters {work items << 32 | units}
  // pinned, one word per sequence: {update number << 32 | CONVERGED seeds at the start of that update}
  unsigned long long* h_conv = nullptr;
  unsigned long long* d_conv = nullptr;    // its device address
  long long frame = 0;                     // updates since the last reference (which set of shard counters is current)
  unsigned int update_number = 0;          // launch pairs so far (modulo 2^32), stamped into h_conv
  int shard_cap = 0;                       // unit-list entries per shard
  int lds_bytes = 160 * 1024;              // LDS per CU of the handle's device (gfx950: 160 KB)
  // diagnostics, allocated on demand: FR_TRACE_FRAMES slices of wg_trace_slice_u64() words (probes of the search workgroups)
  unsigned long long* d_wg_trace = nullptr;
  int max_units = 0;
  bool attr_set_small = false, attr_set_large = false;
  bool attr_set_compact[4][2] = {{false, false}, {false, false}, {false, false}, {false, false}};  // per patch side, {one sequence, batch}
  int compact_wg_per_cu[4][2] = {{1, 1}, {1, 1}, {1, 1}, {1, 1}};
  int allocate(int w, int h, int stride_elems, int sequences = 1, int max_extent = 100) {
    tiles_x = (w + TILE_W - 1) / TILE_W;
    tiles_y = (h + TILE_H - 1) / TILE_H;
    stride = stride_elems;
    n_seq = sequences;
    seq_plane = static_cast<size_t>(stride) * h;
    const size_t n = seq_plane * n_seq;
    const size_t n_tiles_all = static_cast<size_t>(tiles_x) * tiles_y * n_seq;
    const int units_per_tile = max_search_steps(max_extent);  // a tile of 256 seeds with every step in the image, in units of one round
    if (hipMalloc(reinterpret_cast<void**>(&d_mean), n * sizeof(float2)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_dir), n * sizeof(float2)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_lfirst), n * sizeof(float)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_packed), n * sizeof(unsigned int)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_best), n * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_tile_live), n_tiles_all * sizeof(unsigned int)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_tile_conv), n_tiles_all * sizeof(unsigned int)) != hipSuccess) return -1;
    shard_cap = static_cast<int>((n_tiles_all + UNIT_SHARDS - 1) / UNIT_SHARDS) * units_per_tile;  // tiles of a shard x units of a tile
    max_units = static_cast<int>(n_tiles_all + UNIT_SHARDS) * units_per_tile;
    if (max_units < UNIT_SHARDS * shard_cap) max_units = UNIT_SHARDS * shard_cap;
    if (hipMalloc(reinterpret_cast<void**>(&d_units), static_cast<size_t>(max_units) * sizeof(uint4)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_handout), UNIT_SHARDS * HANDOUT_STRIDE * sizeof(unsigned int)) != hipSuccess) return -1;
    if (hipMemset(d_handout, 0, UNIT_SHARDS * HANDOUT_STRIDE * sizeof(unsigned int)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_shards), 3 * UNIT_SHARDS * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (hipHostMalloc(reinterpret_cast<void**>(&h_conv), MAX_GROUP_SEQ * sizeof(unsigned long long),
        hipHostMallocMapped) != hipSuccess) return -1;
    for (int q = 0; q < MAX_GROUP_SEQ; ++q) h_conv[q] = 0ull;
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, h_conv, 0) != hipSuccess) return -1;
    d_conv = static_cast<unsigned long long*>(dev);
    if (hipMemset(d_shards, 0, 3 * UNIT_SHARDS * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (hipMemset(d_packed, 0, n * sizeof(unsigned int)) != hipSuccess) return -1;
    if (hipMemset(d_best, 0, n * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (hipMemset(d_tile_conv, 0, n_tiles_all * sizeof(unsigned int)) != hipSuccess) return -1;
    if (hipMemset(d_tile_live, 0, n_tiles_all * sizeof(unsigned int)) != hipSuccess) return -1;
    return 0;
  }
  int n_tiles() const { return tiles_x * tiles_y; }
  size_t wg_trace_slice_u64() const { return static_cast<size_t>(tiles_x) * tiles_y * 8; }  // FR_TRACE_WORDS per workgroup / tile
  void release() {
    void* all[] = {d_mean, d_dir, d_lfirst, d_packed, d_best, d_tile_live, d_tile_conv, d_units, d_handout, d_shards, d_wg_trace};
    for (void* p : all)
      if (p) (void)hipFree(p);
    if (h_conv) (void)hipHostFree(h_conv);
    d_mean = d_dir = nullptr; d_lfirst = nullptr; d_packed = nullptr; d_best = nullptr;
    d_tile_live = nullptr; d_tile_conv = nullptr; d_units = nullptr; d_handout = nullptr; d_shards = nullptr;
    d_wg_trace = nullptr; h_conv = nullptr; d_conv = nullptr;
  }
};

// device view of the workspace (the same for every sequence of a launch)
struct MatcherArgs {
  float2* mean;
  float2* dir;
  float* lfirst;
  unsigned int* packed;
  unsigned long long* best;
  size_t seq_plane;          // elements per sequence in the five planes above
  unsigned int* tile_live;   // per tile: seeds in state UPDATE after the last frame's check (see seed_setup_compact_kernel)
  unsigned int* tile_conv;
  uint4* units;
  unsigned int* handout;     // UNIT_SHARDS counters, HANDOUT_STRIDE words apart (zero at the search kernel's launch)
  unsigned long long* shards_cur;         // this frame's shard counters (zero at launch)
  const unsigned long long* shards_prev;  // the previous frame's (null: no previous frame)
  unsigned long long* shards_next;        // cleared by this frame's setup for the next one
  unsigned long long* conv_out;           // pinned host words, one per sequence: {update_number << 32 | converged seeds}
  unsigned int update_number;
  int shard_cap;
  int tiles_x;
  int tiles_y;
  int n_tiles;               // tiles of one sequence
  int n_seq;
  int housekeeper;           // the first sequence of the launch that has a frame: tile 0 of it resets the counters of the next launch
  // Frame ingest for frames handed over in host memory, two forms (rmd_capi.hip chooses).  STAGED: a copy engine brings the frames of all
  // sequences of the launch as they are into staging buffers in HBM and then writes the step's number into `ingest_flag`, both on the
  // copy stream, with NO ordering against the compute stream; a few extra workgroups of the setup kernel wait for the flag themselves
  // (it has normally been set long before) and convert the staged frames into the current-image planes, which only the search kernel
  // -- the next launch -- reads.  IN PLACE (ingest_flag == null): those workgroups read the frames straight from the pinned host
  // buffers over the host link (ingest_in_place); the buffers were complete before the kernel was launched, nothing to wait for.
  // The frames and their destinations are per sequence (SeqArgs).  ingest_kind 0: the frames are resident already.
  // 1: 8-bit rows of ingest_pitch bytes (a multiple of 4), four pixels per dword; 2: float rows of P.w elements, unpadded
  int ingest_kind;
  int ingest_pitch;
  int ingest_wgs;                  // workgroups (per sequence) that do the conversion: the first ones of the setup grid ...
  int ingest_rows;                 // ... which occupy this many grid rows IN FRONT of the tile grid (tile row = blockIdx.y - ingest_rows)
  int ingest_profile;              // diagnostics (RMD_HIP_INGEST_PROFILE): count into progress[2..4]
  // device word: number of the last step whose staging copy (into this step's buffers) has completed; null: in place
  const unsigned int* ingest_flag;
  unsigned int* progress;          // pinned host words: [0] <- ingest_number when this setup starts (everything before it has
                                   // completed: the host may reuse that step's buffers), [1] |= 1 if the flag never came;
                                   // diagnostics: [2] frames converted by their own setup kernel (not one step ahead), [3] of those, frames
                                   // the kernel had to wait for, [4] polls of those waits
  unsigned int ingest_number;
  // Host frames ONE STEP AHEAD (single sequences): the caller hands frame n + 1 over while the device is still busy with frame n or
  // n - 1.  One lane of the setup kernel of frame n looks at `submitted` -- has frame n + 1 arrived in its staging buffer (staged) /
  // been put into the pinned ring (in place)? -- and writes its verdict to ahead[0]; if it is there, the first `ahead_wgs` workgroups of
  // frame n's SEARCH kernel, which leaves most of the chip idle on all but the heaviest frames, convert it into the other current-image
  // plane and set ahead[2] = n + 1, and the ingest workgroups of setup n + 1 find nothing left to do.  Nobody waits for a frame on this
  // path: one that was not there in time is handled by its own setup kernel as without it.
  int ahead_wgs;
  // staged: the arrival flag of frame n + 1's staging buffer; in place: a pinned host word, the newest frame of this kind that is complete
  // in the ring
  const unsigned int* submitted;
  // device words: [0] frame to bring in during this update's search kernel or 0, [1] bringers done, [2] newest frame brought in ahead
  unsigned int* ahead;
};

// One sequence of a launch: the reference's mvs::DeviceData of that SeedMatrix for this frame plus what the deferred finalisation
// of ITS previous frame and the ingest of ITS host frame need.
struct SeqArgs {
  SeedParams P;
  Pose T_ref_curr_prev;      // pose of the frame whose finalisation is pending
  int fuse_prev;             // that finalisation is pending (nobody looked at the state since): run it in this launch's setup
  int active;                // 0: this sequence has no frame in this launch (its workgroups leave at once)
  const unsigned int* ingest_u8;   // this sequence's staged frame (see MatcherArgs::ingest_kind)
  const float* ingest_f32;
  float* ingest_dst;               // its current-image plane, row stride P.stride
  const short2* ingest_map1;       // lens undistortion of 8-bit frames (Depthmap::initUndistortionMap): source pixel per destination pixel
  const unsigned short* ingest_map2;  // ... and its 5-bit fractions; null = frames are used as they come
  const void* next_src;            // one step ahead (MatcherArgs::ahead): where the NEXT frame of this kind will be in the pinned ring ...
  float* next_dst;                 // ... and the current-image plane it goes to
};

// The sequences of one launch, passed BY VALUE as the kernels' FIRST argument.  The kernels never name the parameter: indexing a
// by-value aggregate with a run-time sequence number makes the compiler copy all of it into scratch memory; they read the kernel
// argument segment through seq_table() instead (scalar loads with a run-time offset, like any other argument).
template <int NSEQ>
struct BatchArgs {
  SeqArgs seq[NSEQ];
};
static_assert(sizeof(BatchArgs<MAX_GROUP_SEQ>) + sizeof(MatcherArgs) + 64 <= 4096, "kernel arguments are limited to 4 KB");
// (C cast: from the constant address space)
RMDK_D const SeqArgs* seq_table() { return (const SeqArgs*)__builtin_amdgcn_kernarg_segment_ptr(); }

// what the caller of the pipeline hands over when the frames came from host memory
struct IngestArgs {
  int kind = 0;       // 1: 8-bit, 2: float
  int pitch = 0;
  const unsigned int* flag = nullptr;  // null: the frame is read in place from pinned host memory
  unsigned int* progress = nullptr;
  unsigned int number = 0;
  bool no_remap = false;  // 8-bit frames that must not go through the lens-undistortion maps (float frames that travel as bytes)
  bool profile = false;   // count conversions / waits into progress[2..4] (MatcherArgs::ingest_profile)
  int ahead_wgs = 0;  // see MatcherArgs::ahead
  const unsigned int* submitted = nullptr;
  unsigned int* ahead = nullptr;
};

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
// ncc_sums_regular, software-pipelined by hand.  The texel row r + 1 and the template row r are requested from the LDS BEFORE the filter /
// accumulate work on row r, and scheduling barriers keep the compiler from sinking the reads back to their first use (left alone it issues
// every read a few instructions before an s_waitcnt: ~60 exposed LDS latencies per evaluation, which is most of a round's time when a wave
// has its SIMD to itself, i.e. on every frame but the first twenty). With four waves per SIMD the block is bound by VALU issue, not by the
// LDS: builds that leave out 28 % or 72 % of its LDS reads run the heaviest updates in the same time (+-1 %), and the kernel issues one
// VALU instruction per 2.9 cycles and SIMD there, the rate of a pure v_fma_f32 stream on this part (profiles/r03_lds_ceiling.txt).  Keeping
// the two product sums as a register pair for v_pk_add_f32 (4.3 cycles against 2 x 2.7 in isolation) was tried: as a vector-typed
// expression the optimiser sinks the chain of packed adds below the last row with all 81 products alive (168 VGPRs + scratch); pinned by
// volatile inline assembly it costs 125 VGPRs, is bit-identical and gains nothing on the heaviest updates (490 -> 494 us for a batch of 8);
// pairs of columns through the vertical filter cost 114 extra moves.
template <int SIDE>
RMDK_D void ncc_sums_lds_pipelined(const float* __restrict__ win_x, const int* __restrict__ row_start, int off_first, int off_second,
    const float (&ax)[SIDE],
                                   const float (&ay)[SIDE], const float* __restrict__ ref_patch, int ref_stride, float& sum_img,
                                       float& sum_img_sq,
                                   float& sum_img_templ) {
  // win_x = the window + the footprint's first image column; row_start = the window's row table at the footprint's first row: texel row r
  // of the footprint starts at win_x[row_start[r]] (a sheared window has no constant row stride).  Entries 0 and 1 come from the caller;
  // entry r + 2 is requested while row r is worked on -- two rows ahead of the texel reads that need it, so that it costs no wait and two
  // registers.
  float t[2][SIDE + 1], tm[2][SIDE], hprev[SIDE], hcur[SIDE];
  int off_next = off_second, off_after = 0;
#pragma unroll
  for (int c = 0; c <= SIDE; ++c) t[0][c] = (win_x + off_first)[c];
#pragma unroll
  for (int r = 0; r <= SIDE; ++r) {
    const int cur = r & 1, nxt = cur ^ 1;
    if (r < SIDE) {
      const float* row = win_x + off_next;  // texel row r + 1
      if (r + 2 <= SIDE) off_after = row_start[r + 2];
#pragma unroll
      for (int c = 0; c <= SIDE; ++c) t[nxt][c] = row[c];
#pragma unroll
      for (int k = 0; k < SIDE; ++k) tm[nxt][k] = ref_patch[r * ref_stride + k];  // template row r, used with texel rows r, r + 1
      off_next = off_after;
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
        sum_img_sq += img * img;  // (one rounding per operation: the contract is "no contraction"; what contracting these two would gain
        sum_img_templ += img * templ;  //  was measured -- +2.5 % / +5.3 % -- and declined, LAB.md)
      }
    }
#pragma unroll
    for (int k = 0; k < SIDE; ++k) hprev[k] = hcur[k];
    __builtin_amdgcn_sched_barrier(0);
  }
}


// ------------------------------------------------------------------------------------------------
inline MatcherArgs matcher_args(const MatcherWorkspace& ws) {
  MatcherArgs M;
  M.mean = ws.d_mean; M.dir = ws.d_dir; M.lfirst = ws.d_lfirst; M.packed = ws.d_packed; M.best = ws.d_best;
  M.seq_plane = ws.seq_plane;
  M.tile_live = ws.d_tile_live; M.tile_conv = ws.d_tile_conv; M.units = ws.d_units; M.handout = ws.d_handout;
  M.tiles_x = ws.tiles_x; M.tiles_y = ws.tiles_y; M.n_tiles = ws.tiles_x * ws.tiles_y; M.n_seq = ws.n_seq;
  M.shards_cur = ws.d_shards + (ws.frame % 3) * UNIT_SHARDS;
  M.shards_prev = ws.frame > 0 ? ws.d_shards + ((ws.frame + 2) % 3) * UNIT_SHARDS : nullptr;
  M.shards_next = ws.d_shards + ((ws.frame + 1) % 3) * UNIT_SHARDS;
  M.housekeeper = 0;
  M.conv_out = ws.d_conv;
  M.update_number = ws.update_number;
  M.shard_cap = ws.shard_cap;
  M.ingest_kind = 0; M.ingest_pitch = 0; M.ingest_wgs = 0; M.ingest_rows = 0; M.ingest_profile = 0; M.ingest_flag = nullptr; M.progress =
      nullptr; M.ingest_number = 0u;
  M.ahead_wgs = 0; M.submitted = nullptr; M.ahead = nullptr;
  return M;
}

// the planes of ONE sequence of the workspace, for the per-sequence kernels (the stand-alone finalisation)
inline MatcherArgs matcher_args_of(const MatcherWorkspace& ws, int seq) {
  MatcherArgs M = matcher_args(ws);
  const size_t off = ws.seq_plane * static_cast<size_t>(seq);
  M.mean += off; M.dir += off; M.lfirst += off; M.packed += off; M.best += off;
  M.tile_live += static_cast<size_t>(seq) * M.n_tiles; M.tile_conv += static_cast<size_t>(seq) * M.n_tiles;
  return M;
}

}  // namespace rmdk

#endif  // RMD_MATCHER_HPP
