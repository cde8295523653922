// The seed update of one frame as ONE launch (seed_frame_kernel): RETIRED from the product.  Compiled only into A/B builds of
// the library (-DRMD_AB_MATCHERS, tools/ab_make.sh), where RMD_HIP_OPT_MATCHER 2 selects it.  Measured 101-116 us per 640x480
// update against 47 us for the two-launch pipeline of rmd_frame.hpp ("measured and rejected").  Bit-identical results.
//
//   setup      one lane per seed: state check; epipolar segment; the contiguous run of in-image search steps
//              (find_valid_run); per-seed descriptors straight into LDS (they never visit global memory).
//   search     the tile's (seed, step) pairs are dealt round-robin to the 256 lanes in rounds; one NCC evaluation per pair
//              (separable bilinear filter with the reference's roundings, ncc_at_dyn); arg-max per seed with an LDS 64-bit
//              atomic max on {orderable(ncc), ~step}.  The current image is staged in an LDS window of at most
//              FR_WIN_CAP texels whose shape follows the work: the bounding box of the samples of ALL remaining work if that
//              fits, else of the next 4 rounds, else of one round, else a clamped box (samples outside any window read
//              L2 with the same arithmetic).  40 KB of LDS per workgroup: 4 workgroups / 16 waves per CU.
//   finalise   one lane per seed: decode the arg-max, NO_MATCH / match, triangulation, posterior update, final state.
//
// Load balance.  A tile whose seeds search their full range is up to 143 rounds of work, two orders of magnitude above the
// average of a late frame, so a tile with more than `local_max` work items publishes its descriptors (write-through stores)
// and hands its work out as units of 4 rounds through a global queue; every workgroup drains that queue after its own tile,
// the workgroup that completes a tile's last unit finalises the tile.  Tiles that were heavy in the previous frame are
// dispatched first (front of the grid), so that their units are in the queue while most workgroups are still alive.
// Nobody ever waits for a workgroup that is not running: publishers stay until their own tile is finalised and consume
// from the queue meanwhile; everybody else leaves when the queue is empty.  All spins are bounded (error word).
//
// Inter-workgroup visibility follows MI355X_MICROARCH.md (per-XCD L2s are not coherent): published data is written with
// agent-scope (sc1, write-through) stores, drained (s_waitcnt vmcnt(0)) before the queue entry that announces it, and
// read with agent-scope loads; counters are device-scope atomics.  State planes are only ever written by ONE workgroup
// per frame (the tile's owner, or the finaliser of a published tile), and read by the next launch.
#ifndef RMD_FRAME_ONE_LAUNCH_HPP
#define RMD_FRAME_ONE_LAUNCH_HPP

#include "../rmd_frame.hpp"

namespace rmdk {

constexpr int FR_MAX_FRONT = 256;  // grid slots in front of the tile grid for the tiles that were heavy in the previous frame
constexpr int FR_CTL_WORDS = 32;   // per frame parity (3 sets: this frame's, the previous frame's, the one being cleared)
constexpr int FR_STATE_ABSENT = 7; // lanes of a ragged tile that lie outside the image
constexpr unsigned int FR_SPIN_LIMIT = 1u << 22;
#ifndef FR_PUBLISH_AGE_US
#define FR_PUBLISH_AGE_US 16
#endif
constexpr unsigned long long FR_PUBLISH_AGE_TICKS = FR_PUBLISH_AGE_US * 100ull;  // 10 ns ticks: by then the front tiles of a frame have been handed out
enum : int { FC_N_PUB = 0, FC_UNITS = 1, FC_N_HEAVY = 2, FC_ITEMS = 8 /* 16 shards */ };

// One tile handed out for this frame.  `info` = {epoch:32 | tile:20 | rounds per unit:4 | units:8} is written last (the entry
// is valid when its epoch is the frame's); units are claimed by fetch-adds on `next`; `pending` counts units not yet done.
struct PubEntry {
  unsigned long long info;
  unsigned int next;
  unsigned int pending;
};

struct FrameWorkspace {
  unsigned int* d_ctl = nullptr;         // 3 x FR_CTL_WORDS
  unsigned int* d_heavy_list = nullptr;  // 3 x FR_MAX_FRONT tile indices
  unsigned char* d_heavy_flag = nullptr; // 3 x n_tiles
  PubEntry* d_pub = nullptr;             // list of the tiles handed out this frame (at most one entry per tile)
  unsigned long long* d_desc = nullptr;  // 3 planes of n seeds: (mean) (dir) (l_first, packed)
  unsigned long long* d_best = nullptr;  // per seed arg-max key of published tiles; all zero between frames
  unsigned long long* d_trace = nullptr; // diagnostics, allocated on demand: FR_TRACE_FRAMES slices of grid x FR_TRACE_WORDS stamps
  unsigned int* d_error = nullptr;       // sticky error bits of the frame kernels (bounded spins that ran out)
  unsigned int* h_error = nullptr;       // pinned mirror
  int tiles_x = 0, tiles_y = 0, n_tiles = 0, grid = 0;
  size_t n_seeds = 0;
  long long frame = 0;      // frames launched since the last reference (0 = no previous frame's statistics)
  unsigned int epoch = 0;   // tag of this frame's queue entries, never 0
  bool attr_set[4] = {false, false, false, false};
  bool attr_set_compact[4] = {false, false, false, false};
  int compact_wg_per_cu[4] = {1, 1, 1, 1};
  size_t trace_slice_u64() const { return static_cast<size_t>(n_tiles) * FR_TRACE_WORDS; }  // grid <= n_tiles workgroups
  int allocate(int w, int h, int stride_elems) {
    tiles_x = (w + TILE_W - 1) / TILE_W;
    tiles_y = (h + TILE_H - 1) / TILE_H;
    n_tiles = tiles_x * tiles_y;
    n_seeds = static_cast<size_t>(stride_elems) * h;
    if (hipMalloc(reinterpret_cast<void**>(&d_ctl), 3 * FR_CTL_WORDS * sizeof(unsigned int)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_heavy_list), 3 * FR_MAX_FRONT * sizeof(unsigned int)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_heavy_flag), 3 * static_cast<size_t>(n_tiles)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_pub), (static_cast<size_t>(n_tiles) + 64) * sizeof(PubEntry)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_desc), 3 * n_seeds * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_best), n_seeds * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_error), sizeof(unsigned int)) != hipSuccess) return -1;
    if (hipHostMalloc(reinterpret_cast<void**>(&h_error), sizeof(unsigned int)) != hipSuccess) return -1;
    h_error[0] = 0;
    (void)hipMemset(d_error, 0, sizeof(unsigned int));
    (void)hipMemset(d_ctl, 0, 3 * FR_CTL_WORDS * sizeof(unsigned int));
    (void)hipMemset(d_heavy_list, 0, 3 * FR_MAX_FRONT * sizeof(unsigned int));
    (void)hipMemset(d_heavy_flag, 0, 3 * static_cast<size_t>(n_tiles));
    (void)hipMemset(d_pub, 0, (static_cast<size_t>(n_tiles) + 64) * sizeof(PubEntry));
    (void)hipMemset(d_best, 0, n_seeds * sizeof(unsigned long long));
    return 0;
  }
  void release() {
    void* all[] = {d_ctl, d_heavy_list, d_heavy_flag, d_pub, d_desc, d_best, d_error, d_trace};
    for (void* p : all)
      if (p) (void)hipFree(p);
    if (h_error) (void)hipHostFree(h_error);
    d_ctl = nullptr; d_heavy_list = nullptr; d_heavy_flag = nullptr; d_pub = nullptr;
    d_desc = nullptr; d_best = nullptr; d_error = nullptr; h_error = nullptr; d_trace = nullptr;
  }
};

// device view
struct FrameArgs {
  unsigned int* ctl_cur;         // this frame's counters (zero at launch)
  const unsigned int* ctl_prev;  // the previous frame's (null: no previous frame)
  unsigned int* ctl_next;        // cleared by this launch for the next one
  unsigned int* heavy_list_cur;
  const unsigned int* heavy_list_prev;
  unsigned char* heavy_flag_cur;
  const unsigned char* heavy_flag_prev;
  PubEntry* pub;
  unsigned long long* desc;
  unsigned long long* best;
  unsigned int* error;
  unsigned long long* trace;  // this frame's slice of the timeline buffer, or null
  size_t desc_plane;  // elements per descriptor plane
  int tiles_x, n_tiles;
  int n_slots;        // workgroups resident at a time (load estimate)
  unsigned int epoch;
  int local_max_override;  // > 0: fixed threshold (experiments / tests); 0: from the previous frame's load
  int unit_rounds_override;  // 1..4: fixed unit size; 0: from the previous frame's load
};

RMDK_D unsigned long long pack2f(float a, float b) {
  return static_cast<unsigned long long>(__float_as_uint(a)) | (static_cast<unsigned long long>(__float_as_uint(b)) << 32);
}
RMDK_D unsigned long long pack2u(unsigned int a, unsigned int b) { return static_cast<unsigned long long>(a) | (static_cast<unsigned long long>(b) << 32); }

// epipolar_match.cu:131-139 + seed_update.cu:39-121 for one seed whose state after the check is UPDATE.  Writes the seed's
// planes (mu, sigma_sq, a, b, the match) and returns its final state of the frame.
RMDK_D int frame_finalize_seed(const SeedParams& P, int x, int y, int gi, unsigned long long key, int i_first, float l_first,
                               float mean_x, float mean_y, float dir_x, float dir_y) {
  F2 best_px = F2{0.0f, 0.0f};
  float best_ncc = -1.0f;
  if (key != 0ull) {
    best_ncc = from_orderable_f32(static_cast<unsigned int>(key >> 32));
    const int step = static_cast<int>(0xffffffffu - static_cast<unsigned int>(key & 0xffffffffu));
    const float l = replay_l(l_first, step - i_first);
    best_px = F2{mean_x + l * dir_x, mean_y + l * dir_y};
  }
  int state = ST_UPDATE;
  if (best_ncc < 0.5f) state = ST_NO_MATCH;
  else P.match[y * P.stride2 + x] = make_float2(best_px.x, best_px.y);
  seed_fuse(P, x, y, gi, state, P.mu[gi], P.sigma_sq[gi], P.a[gi], P.b[gi], best_px);
  return state;
}

// (frame_prefix: rmd_frame.hpp)

// Stage the tile's patch halo of the reference image (24 x 24 texels at SIDE 9, clamped at the image border like the
// reference's clamp-addressed texture, epipolar_match.cu:107-110).  No barrier.
template <int SIDE>
RMDK_D void frame_stage_ref(const SeedParams& P, FrameSmem<SIDE>& S, int x0, int y0, int tid) {
  using Smem = FrameSmem<SIDE>;
  constexpr int HALF = SIDE / 2;
  for (int i = tid; i < Smem::REF_H * Smem::REF_W; i += TILE_PIX) {
    const int ry = i / Smem::REF_W, rx = i - ry * Smem::REF_W;
    S.ref[i] = P.ref[clampi(y0 - HALF + ry, 0, P.h - 1) * P.stride + clampi(x0 - HALF + rx, 0, P.w - 1)];
  }
}

// Setup of a tile (one lane per seed): state check; for live seeds the epipolar segment and its in-image run; descriptors
// into LDS, S.best = 0.  Returns false (and has written the final states) if no seed of the tile is live.
// Contains one barrier (the vote).
template <int SIDE>
RMDK_D bool frame_setup(const SeedParams& P, FrameSmem<SIDE>& S, int x0, int y0, int tid, unsigned long long& st_live,
                        unsigned long long& st_steps, unsigned long long& st_evals) {
  using Smem = FrameSmem<SIDE>;
  constexpr int HALF = SIDE / 2;
  constexpr int REF_N = Smem::REF_H * Smem::REF_W, REF_PER = (REF_N + TILE_PIX - 1) / TILE_PIX;
  const int tx = tid & (TILE_W - 1), ty = tid >> 4;
  const int x = x0 + tx, y = y0 + ty;
  const bool in_image = x < P.w && y < P.h;
  const int gi = y * P.stride + x;
  int state = FR_STATE_ABSENT;
  float mu = 0.0f, sigma_sq = 0.0f, a = 0.0f, b = 0.0f;
  if (in_image) { mu = P.mu[gi]; sigma_sq = P.sigma_sq[gi]; a = P.a[gi]; b = P.b[gi]; }
  // the tile's patch halo of the reference image (24 x 24 texels at SIDE 9, clamped at the image border like the reference's
  // clamp-addressed texture, epipolar_match.cu:107-110): loads issued now, stored once the tile is known to have live seeds
  float refv[REF_PER];
#pragma unroll
  for (int q = 0; q < REF_PER; ++q) {
    const int i = tid + q * TILE_PIX;
    const int ry = i / Smem::REF_W, rx = i - ry * Smem::REF_W;
    refv[q] = i < REF_N ? P.ref[clampi(y0 - HALF + ry, 0, P.h - 1) * P.stride + clampi(x0 - HALF + rx, 0, P.w - 1)] : 0.0f;
  }
  if (in_image) state = seed_check(P, x, y, sigma_sq, a, b, SIDE);
  if (!__syncthreads_or(state == ST_UPDATE)) {
    if (in_image) P.conv[gi] = state;
    return false;
  }
#pragma unroll
  for (int q = 0; q < REF_PER; ++q) {
    const int i = tid + q * TILE_PIX;
    if (i < REF_N) S.ref[i] = refv[q];
  }
  int n_valid = 0, i_first = 0;
  if (state == ST_UPDATE) {
    const Segment seg = epipolar_segment(P, x, y, mu, sigma_sq);
    const ValidRun run = find_valid_run(P, seg, SIDE);
    n_valid = run.n_valid; i_first = run.i_first;
    if (n_valid > 0) {
      S.mean_x[tid] = seg.mean.x; S.mean_y[tid] = seg.mean.y; S.dir_x[tid] = seg.dir.x; S.dir_y[tid] = seg.dir.y;
      S.l_first[tid] = run.l_first;
      S.sum_templ[tid] = P.sum_templ[gi]; S.denom[tid] = P.denom[gi];
    }
    if (P.stats) {  // diagnostics only: the full walk, counting what the reference would visit / evaluate
      unsigned int n_steps = 0, n_evals = 0;
      for (float l = -seg.half_length; l <= seg.half_length; l += 0.7f, ++n_steps) {
        const F2 px = F2{seg.mean.x + l * seg.dir.x, seg.mean.y + l * seg.dir.y};
        if (!px_outside(P, px, SIDE)) ++n_evals;  // NaN positions pass the reference's guard too
      }
      st_live += 1; st_steps += n_steps; st_evals += n_evals;
    }
  }
  S.packed[tid] = (static_cast<unsigned int>(state) << 16) | (static_cast<unsigned int>(i_first) << 8) | static_cast<unsigned int>(n_valid);
  S.best[tid] = 0ull;
  return true;
}

// Final state of every seed of the tile in LDS: arg-max keys from S.best (all of the tile's search ran in this workgroup)
// or from the global best plane (a tile that several workgroups searched).  One lane per seed.
template <int SIDE>
RMDK_D void frame_finalize(const SeedParams& P, const FrameArgs& A, FrameSmem<SIDE>& S, int x0, int y0, int tid, bool from_global) {
  const int tx = tid & (TILE_W - 1), ty = tid >> 4;
  const int x = x0 + tx, y = y0 + ty;
  const unsigned int pk = S.packed[tid];
  int state = static_cast<int>(pk >> 16);
  if (state == FR_STATE_ABSENT) return;
  const int gi = y * P.stride + x;
  if (state == ST_UPDATE) {
    unsigned long long key;
    if (from_global) {
      key = ld_agent(&A.best[gi]);
      if (key != 0ull) st_agent(&A.best[gi], 0ull);  // the plane is all zero between frames
    } else {
      key = S.best[tid];
    }
    const bool has = (pk & 0xffu) != 0u;  // descriptors exist only for seeds with in-image steps
    state = frame_finalize_seed(P, x, y, gi, has ? key : 0ull, static_cast<int>((pk >> 8) & 0xffu), has ? S.l_first[tid] : 0.0f,
                                has ? S.mean_x[tid] : 0.0f, has ? S.mean_y[tid] : 0.0f, has ? S.dir_x[tid] : 0.0f,
                                has ? S.dir_y[tid] : 0.0f);
  }
  P.conv[gi] = state;
}

// Load a handed-out tile's descriptors into LDS (agent-scope loads), S.best = 0, prefix, reference halo, window.
// Ends with a barrier.
template <int SIDE>
RMDK_D int frame_load_published(const SeedParams& P, const FrameArgs& A, FrameSmem<SIDE>& S, int x0, int y0, int tid, FrameWindow& W) {
  using Smem = FrameSmem<SIDE>;
  constexpr int HALF = SIDE / 2;
  const int tx = tid & (TILE_W - 1), ty = tid >> 4;
  const int x = x0 + tx, y = y0 + ty;
  unsigned int pk = static_cast<unsigned int>(FR_STATE_ABSENT) << 16;
  if (x < P.w && y < P.h) {
    const size_t gi = static_cast<size_t>(y) * P.stride + x;
    const unsigned long long d2 = ld_agent(&A.desc[2 * A.desc_plane + gi]);
    pk = static_cast<unsigned int>(d2 >> 32);
    if ((pk & 0xffu) != 0u) {
      const unsigned long long d0 = ld_agent(&A.desc[gi]), d1 = ld_agent(&A.desc[A.desc_plane + gi]);
      S.mean_x[tid] = __uint_as_float(static_cast<unsigned int>(d0)); S.mean_y[tid] = __uint_as_float(static_cast<unsigned int>(d0 >> 32));
      S.dir_x[tid] = __uint_as_float(static_cast<unsigned int>(d1)); S.dir_y[tid] = __uint_as_float(static_cast<unsigned int>(d1 >> 32));
      S.l_first[tid] = __uint_as_float(static_cast<unsigned int>(d2));
      S.sum_templ[tid] = P.sum_templ[gi]; S.denom[tid] = P.denom[gi];
    }
  }
  S.packed[tid] = pk;
  S.best[tid] = 0ull;
  for (int i = tid; i < Smem::REF_H * Smem::REF_W; i += TILE_PIX) {
    const int ry = i / Smem::REF_W, rx = i - ry * Smem::REF_W;
    S.ref[i] = P.ref[clampi(y0 - HALF + ry, 0, P.h - 1) * P.stride + clampi(x0 - HALF + rx, 0, P.w - 1)];
  }
  return frame_prefix_and_window<SIDE>(P, S, tid, W);
}

// Persistent workgroups (as many as are resident).  Each loops: (A) units of the handed-out tile whose descriptors it holds,
// (B) the next tile of the frame, (C) units of somebody else's handed-out tile; it leaves when all three are exhausted.
// Nobody waits for anybody: a unit, once claimed (fetch-add), is searched by its claimer; an owner hands out its tile and then
// claims from it like everybody else, so when it moves on every unit of the tile has a claimer; the workgroup that counts a
// tile's last unit done finalises the tile.
template <int SIDE>
__global__ __launch_bounds__(TILE_PIX, FR_MIN_WAVES) void seed_frame_kernel(SeedParams P, FrameArgs A) {
  using Smem = FrameSmem<SIDE>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& S = *reinterpret_cast<Smem*>(smem_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  unsigned long long* const tr = A.trace ? A.trace + static_cast<size_t>(blockIdx.x) * FR_TRACE_WORDS : nullptr;
  if (tr && tid == 0) tr[0] = wall_clock64();
  if (blockIdx.x == 0 && tid < FR_CTL_WORDS) A.ctl_next[tid] = 0u;  // the set the NEXT launch counts in (nobody reads it now)

  // order of the frame: the tiles that were handed out in the previous frame first, then the tile grid without them
  const unsigned int n_front = A.ctl_prev ? min(A.ctl_prev[FC_N_HEAVY], static_cast<unsigned int>(FR_MAX_FRONT)) : 0u;
  const unsigned int n_claims = n_front + static_cast<unsigned int>(A.n_tiles);
  // Load estimate from the previous frame (items per resident workgroup) -> how much work a tile keeps to itself, the size of
  // the units a bigger tile is handed out in -- 4 rounds while every workgroup is busy anyway (amortises the per-unit
  // window), 1 round when most of the chip idles and only the latency of the few big tiles counts -- and how many
  // workgroups look for handed-out work at all when they are done with their own tiles (a thousand workgroups polling the
  // same few words cost every one of them tens of microseconds).
  int local_max = 4 * FR_UNIT_ITEMS, unit_items = FR_UNIT_ITEMS;
  unsigned int n_helpers = gridDim.x;
  unsigned int expect_pub = 0u;  // tiles handed out in the previous frame: about as many will be in this one
  if (A.ctl_prev) {
    unsigned int items = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) items += A.ctl_prev[FC_ITEMS + q];
    const unsigned int per_wg = items / static_cast<unsigned int>(A.n_slots);
    unit_items = TILE_PIX * static_cast<int>(min(max(per_wg / (2u * TILE_PIX), 1u), static_cast<unsigned int>(FR_UNIT_ROUNDS)));
    local_max = static_cast<int>(min(max(2u * per_wg, 2u * static_cast<unsigned int>(unit_items)), 16u * FR_UNIT_ITEMS));
    n_helpers = A.ctl_prev[FC_UNITS] + 32u;
    expect_pub = min(A.ctl_prev[FC_N_PUB], static_cast<unsigned int>(A.n_tiles));
  }
  if (A.unit_rounds_override > 0) unit_items = TILE_PIX * A.unit_rounds_override;
  if (A.local_max_override > 0) local_max = A.local_max_override;

  unsigned long long st_live = 0, st_steps = 0, st_evals = 0;
  unsigned int tr_items = 0, tr_units = 0, tr_tiles = 0, tr_pub = 0, tr_fallback = 0, tr_windows = 0;
  // workgroup-uniform state
  int lds_tile = -1, lds_units = 0, lds_total = 0, lds_unit_items = FR_UNIT_ITEMS, x0 = 0, y0 = 0;
  bool lds_open = false;     // the handed-out tile in LDS may still have unclaimed units
  bool lds_mine = false;     // ... and this workgroup handed it out (its S.best has seen every unit it searched since setup)
  int session_units = 0;     // units of lds_tile searched here and not yet counted in tile_pending
  int have_unit = -1;        // a unit of lds_tile claimed while scanning (C), to be searched next
  FrameWindow W;
  W.valid = false; W.x0 = W.y0 = 0; W.x1 = W.y1 = -1; W.ws = 1;
  bool tiles_left = true;
  unsigned int claim = blockIdx.x;  // tiles are dealt statically (block b: claims b, b + grid, ...): no shared counter
  bool first_tile = true;
  int lds_entry = 0;  // the list entry of the handed-out tile in LDS
  bool waited_once = false;
  const unsigned long long t_start = wall_clock64();
  unsigned int prefetched_unit = 0u;  // thread 0: the unit claimed ahead

  // one call site each for the search and the finalisation (the kernel's code must stay small, see ncc_at_dyn)
  int k_begin = 0, k_end = 0;   // pending search: work items of the tile in LDS
  bool search_is_unit = false;  // ... a unit of a handed-out tile (else the whole of a tile this workgroup keeps to itself)
  int finalize = 0;             // pending finalisation of the tile in LDS: 1 keys in S.best, 2 keys in the global plane

  for (;;) {
    if (k_end > k_begin) {
      frame_search<SIDE>(P, S, tid, k_begin, k_end, W, tr_fallback, tr_windows);  // ends with a barrier
      k_begin = k_end = 0;
      if (search_is_unit) {
        ++session_units; ++tr_units;
        // the claim for the next unit of this tile was issued before the search (its latency is hidden behind it)
        if (tid == 0) S.bcast[0] = prefetched_unit;
        __syncthreads();
        have_unit = static_cast<int>(S.bcast[0]);  // >= lds_units: the tile is exhausted
      } else {
        finalize = 1;
      }
    }
    if (finalize) {
      frame_finalize<SIDE>(P, A, S, x0, y0, tid, finalize == 2);
      finalize = 0;
    }
    // ---- (A) units of the handed-out tile in LDS
    if (lds_open) {
      unsigned int u;
      if (have_unit >= 0) { u = static_cast<unsigned int>(have_unit); have_unit = -1; }
      else {
        __syncthreads();  // S.bcast free
        if (tid == 0) S.bcast[0] = atomicAdd(&A.pub[lds_entry].next, 1u);
        __syncthreads();
        u = S.bcast[0];
      }
      if (u < static_cast<unsigned int>(lds_units)) {
        if (tid == 0) prefetched_unit = atomicAdd(&A.pub[lds_entry].next, 1u);  // read after the search
        k_begin = static_cast<int>(u) * lds_unit_items;
        k_end = min(lds_total, k_begin + lds_unit_items);
        search_is_unit = true;
        continue;
      }
      // the tile has no unclaimed units left: close the session
      lds_open = false;
      if (session_units > 0) {
        if (lds_mine && session_units == lds_units) {
          finalize = 1;  // nobody else searched this tile: its keys are all in S.best
        } else {
          const unsigned long long key = S.best[tid];
          if (key != 0ull) {
            const int tx = tid & (TILE_W - 1), ty = tid >> 4;
            atomicMax(&A.best[static_cast<size_t>(y0 + ty) * P.stride + x0 + tx], key);
          }
          drain_vmem();
          __syncthreads();
          if (tid == 0) S.bcast[1] = atomicSub(&A.pub[lds_entry].pending, static_cast<unsigned int>(session_units));
          __syncthreads();
          if (S.bcast[1] == static_cast<unsigned int>(session_units)) finalize = 2;  // these were the tile's last units
        }
        session_units = 0;
      }
      continue;
    }
    // ---- (B) the next tile of the frame
    if (tiles_left) {
      if (claim >= n_claims) {
        tiles_left = false;
        if (tr && tid == 0) tr[2] = wall_clock64();
        if (blockIdx.x >= n_helpers) break;  // enough others look after the handed-out tiles
        continue;
      }
      const unsigned int this_claim = claim;
      claim += gridDim.x;
      int tile;
      if (this_claim < n_front) tile = static_cast<int>(A.heavy_list_prev[this_claim]);
      else {
        tile = static_cast<int>(this_claim - n_front);
        if (n_front && A.heavy_flag_prev[tile]) continue;  // it went first
      }
      const int tile_y = tile / A.tiles_x, tile_x = tile - tile_y * A.tiles_x;
      x0 = tile_x * TILE_W; y0 = tile_y * TILE_H;
      lds_tile = -1;
      ++tr_tiles;
      if (!frame_setup<SIDE>(P, S, x0, y0, tid, st_live, st_steps, st_evals)) {
        if (tid == 0) A.heavy_flag_cur[tile] = 0;
        if (tr && tid == 0 && first_tile) tr[1] = wall_clock64();
        first_tile = false;
        continue;
      }
      const int total = frame_prefix_and_window<SIDE>(P, S, tid, W);  // barriers inside
      if (tr && tid == 0 && first_tile) tr[1] = wall_clock64();
      first_tile = false;
      tr_items += static_cast<unsigned int>(total);
      // hand out what is too much for one workgroup, and tiles whose samples are too scattered for one window (several
      // windows in sequence are slow; several workgroups take them in parallel)
      const bool publish = total > local_max || (!W.valid && total > 2 * TILE_PIX);
      if (tid == 0) {
        if (total > 0) atomicAdd(&A.ctl_cur[FC_ITEMS + (tile & 15)], static_cast<unsigned int>(total));
        unsigned char flag = 0;  // order of the next frame: a tile that is handed out now goes to the front of the grid then
        if (publish) {
          const unsigned int idx = atomicAdd(&A.ctl_cur[FC_N_HEAVY], 1u);
          if (idx < static_cast<unsigned int>(FR_MAX_FRONT)) { A.heavy_list_cur[idx] = static_cast<unsigned int>(tile); flag = 1; }
        }
        A.heavy_flag_cur[tile] = flag;
      }
      if (!publish) {
        if (total > 0) { k_begin = 0; k_end = total; search_is_unit = false; }
        else finalize = 1;  // live seeds without an in-image step: NO_MATCH
        continue;
      }
      // hand the tile out: descriptors (write-through), then the counters, then the entry in the list
      {
        const int tx = tid & (TILE_W - 1), ty = tid >> 4;
        const int x = x0 + tx, y = y0 + ty;
        if (x < P.w && y < P.h) {
          const size_t gi = static_cast<size_t>(y) * P.stride + x;
          const unsigned int pk = S.packed[tid];
          const bool has = (pk & 0xffu) != 0u;
          st_agent(&A.desc[2 * A.desc_plane + gi], pack2f(has ? S.l_first[tid] : 0.0f, __uint_as_float(pk)));
          if (has) {
            st_agent(&A.desc[gi], pack2f(S.mean_x[tid], S.mean_y[tid]));
            st_agent(&A.desc[A.desc_plane + gi], pack2f(S.dir_x[tid], S.dir_y[tid]));
          }
        }
      }
      drain_vmem();
      __syncthreads();
      const int n_units = (total + unit_items - 1) / unit_items;
      if (tid == 0) {
        const unsigned int idx = atomicAdd(&A.ctl_cur[FC_N_PUB], 1u);  // at most one entry per tile: idx < n_tiles
        atomicAdd(&A.ctl_cur[FC_UNITS], static_cast<unsigned int>(n_units));
        st_agent(&A.pub[idx].pending, static_cast<unsigned int>(n_units));
        st_agent(&A.pub[idx].next, 0u);
        drain_vmem();
        st_agent(&A.pub[idx].info, (static_cast<unsigned long long>(A.epoch) << 32) | (static_cast<unsigned long long>(tile) << 12) |
                                       (static_cast<unsigned long long>(unit_items / TILE_PIX) << 8) | static_cast<unsigned long long>(n_units));
        S.bcast[2] = idx;
      }
      __syncthreads();
      lds_entry = static_cast<int>(S.bcast[2]);
      ++tr_pub;
      lds_tile = tile; lds_units = n_units; lds_total = total; lds_unit_items = unit_items; lds_open = true; lds_mine = true; session_units = 0;
      continue;
    }
    // ---- (C) somebody else's handed-out tile: scan the list, 64 entries per memory round trip
    {
      __syncthreads();
      if (tid < 64) {
        unsigned long long got = 0ull;  // bit 63 = claimed; tile << 24 | rounds per unit << 16 | units << 8 | unit
        unsigned int got_entry = 0u, n_valid_seen = 0u;
        // no counter is read: entries are valid by their epoch, the list ends where a block of 64 has an invalid last entry
        for (unsigned int base = 0; base < static_cast<unsigned int>(A.n_tiles) && got == 0ull; base += 64u) {
          const unsigned int i = base + static_cast<unsigned int>(lane);  // the array has 64 entries to spare
          const unsigned long long e = ld_agent(&A.pub[i].info);
          const unsigned int nx = ld_agent(&A.pub[i].next);
          const bool ok = static_cast<unsigned int>(e >> 32) == A.epoch;  // an entry still being written is skipped
          const unsigned int t_i = static_cast<unsigned int>(e >> 12) & 0xfffffu, r_i = static_cast<unsigned int>(e >> 8) & 0xfu, n_i = static_cast<unsigned int>(e) & 0xffu;
          unsigned long long mask = __ballot(ok && nx < n_i);
          n_valid_seen += static_cast<unsigned int>(__popcll(__ballot(ok)));
          const bool more = __shfl(ok ? 1 : 0, 63, 64) != 0;
          const int rot = static_cast<int>(blockIdx.x) & 63;  // workgroups start at different entries
          mask = (mask >> rot) | (rot ? mask << (64 - rot) : 0ull);
          while (mask != 0ull && got == 0ull) {
            const int r = __ffsll(static_cast<long long>(mask)) - 1;
            const int src = (r + rot) & 63;
            const unsigned int t_pick = __shfl(t_i, src, 64), n_pick = __shfl(n_i, src, 64), r_pick = __shfl(r_i, src, 64);
            unsigned int u = 0u;
            if (lane == 0) u = atomicAdd(&A.pub[base + src].next, 1u);
            u = __shfl(u, 0, 64);
            if (u < n_pick) {
              got = (1ull << 63) | (static_cast<unsigned long long>(t_pick) << 24) | (static_cast<unsigned long long>(r_pick) << 16) | (static_cast<unsigned long long>(n_pick) << 8) | u;
              got_entry = base + static_cast<unsigned int>(src);
            }
            mask &= ~(1ull << r);
          }
          if (!more) break;
        }
        if (lane == 0) { S.bcast[2] = got_entry; S.bcast[3] = n_valid_seen; }
        if (lane == 0) { S.bcast[0] = static_cast<unsigned int>(got); S.bcast[1] = static_cast<unsigned int>(got >> 32); }
      }
      __syncthreads();
      const unsigned long long got = static_cast<unsigned long long>(S.bcast[0]) | (static_cast<unsigned long long>(S.bcast[1]) << 32);
      if (got == 0ull) {
        // Nothing to claim right now.  Workgroups whose own tiles were trivial get here a few microseconds before the big
        // tiles of the frame have been set up and handed out (if the previous frame is anything to go by).  They do not
        // poll -- a thousand workgroups re-reading the same few lines slow everybody down -- they sleep until the frame is
        // old enough for the front tiles to have been handed out and look once more.
        if (!waited_once && expect_pub > 0u) {
          waited_once = true;
          while (wall_clock64() - t_start < FR_PUBLISH_AGE_TICKS) __builtin_amdgcn_s_sleep(32);
          continue;
        }
        break;
      }
      const int u_tile = static_cast<int>((got >> 24) & 0xfffffu);
      const int u_entry = static_cast<int>(S.bcast[2]);
      const int tile_y = u_tile / A.tiles_x, tile_x = u_tile - tile_y * A.tiles_x;
      x0 = tile_x * TILE_W; y0 = tile_y * TILE_H;
      __syncthreads();  // S.bcast has been read by everybody before the loads below reuse LDS
      lds_total = frame_load_published<SIDE>(P, A, S, x0, y0, tid, W);  // barrier inside
      lds_tile = u_tile; lds_units = static_cast<int>((got >> 8) & 0xffu); lds_unit_items = TILE_PIX * static_cast<int>((got >> 16) & 0xfu);
      lds_entry = u_entry;
      lds_open = true; lds_mine = false; session_units = 0;
      have_unit = static_cast<int>(got & 0xffu);
    }
  }
  if (P.stats) {
    const unsigned long long s0 = wave_sum_u64(st_live), s1 = wave_sum_u64(st_steps), s2 = wave_sum_u64(st_evals);
    if (lane == 0 && s0) { atomicAdd(&P.stats[0], s0); atomicAdd(&P.stats[1], s1); atomicAdd(&P.stats[2], s2); }
  }
  if (tr && tid < 64) {
    const unsigned long long fb = wave_sum_u64(tr_fallback);  // the first wave's lanes only: a hint, not a count
    if (tid == 0) tr[7] = fb | (static_cast<unsigned long long>(tr_windows) << 32);
  }
  if (tr && tid == 0) {
    tr[3] = wall_clock64();
    tr[4] = static_cast<unsigned long long>(tr_items) | (static_cast<unsigned long long>(tr_pub) << 32);
    tr[5] = tr_units; tr[6] = tr_tiles;
  }
}

inline FrameArgs frame_args(FrameWorkspace& ws, int n_slots, int local_max_override, int unit_rounds_override) {
  FrameArgs A;
  const int cur = static_cast<int>(ws.frame % 3), prev = static_cast<int>((ws.frame + 2) % 3), next = static_cast<int>((ws.frame + 1) % 3);
  A.ctl_cur = ws.d_ctl + cur * FR_CTL_WORDS;
  A.ctl_prev = ws.frame > 0 ? ws.d_ctl + prev * FR_CTL_WORDS : nullptr;
  A.ctl_next = ws.d_ctl + next * FR_CTL_WORDS;
  A.heavy_list_cur = ws.d_heavy_list + cur * FR_MAX_FRONT;
  A.heavy_list_prev = ws.d_heavy_list + prev * FR_MAX_FRONT;
  A.heavy_flag_cur = ws.d_heavy_flag + static_cast<size_t>(cur) * ws.n_tiles;
  A.heavy_flag_prev = ws.d_heavy_flag + static_cast<size_t>(prev) * ws.n_tiles;
  A.pub = ws.d_pub;
  A.desc = ws.d_desc;
  A.best = ws.d_best;
  A.error = ws.d_error;
  A.trace = nullptr;
  A.desc_plane = ws.n_seeds;
  A.tiles_x = ws.tiles_x;
  A.n_tiles = ws.n_tiles;
  A.n_slots = n_slots;
  A.epoch = ws.epoch;
  A.local_max_override = local_max_override;
  A.unit_rounds_override = unit_rounds_override;
  return A;
}

// One frame.  The caller zeroes ws.d_ctl (all three sets) and sets ws.frame = 0 whenever the sequence restarts.
template <int SIDE>
inline hipError_t launch_seed_frame(const SeedParams& P, FrameWorkspace& ws, hipStream_t stream, int num_cus, int local_max_override,
                                    int unit_rounds_override, unsigned long long* trace_slice = nullptr) {
  using Smem = FrameSmem<SIDE>;
  auto kernel = seed_frame_kernel<SIDE>;
  bool& attr = ws.attr_set[SIDE / 2 - 1];
  if (!attr) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(sizeof(Smem)));
    if (e != hipSuccess) return e;
    attr = true;
  }
  ws.epoch = ws.epoch + 1u ? ws.epoch + 1u : 1u;  // never 0
  int wg_per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu, reinterpret_cast<const void*>(kernel), TILE_PIX, sizeof(Smem)) != hipSuccess || wg_per_cu < 1)
    wg_per_cu = 1;
  if (wg_per_cu > 4) wg_per_cu = 4;
  FrameArgs A = frame_args(ws, num_cus * (wg_per_cu > 0 ? wg_per_cu : 1), local_max_override, unit_rounds_override);
  A.trace = trace_slice;
  const int slots = num_cus * (wg_per_cu > 0 ? wg_per_cu : 1);
  ws.grid = ws.n_tiles < slots ? ws.n_tiles : slots;  // persistent: the first tile of a workgroup is its block index, the rest are claimed
  hipLaunchKernelGGL(kernel, dim3(ws.grid), dim3(TILE_PIX), sizeof(Smem), stream, P, A);
  ++ws.frame;
  return hipGetLastError();
}

}  // namespace rmdk

#endif  // RMD_FRAME_ONE_LAUNCH_HPP
