// The seed update of one frame as ONE launch: seed_check (src/seed_check.cu:28-67), the epipolar NCC search
// (src/epipolar_match.cu:37-140) and triangulation + Bayesian fusion (src/seed_update.cu:39-121, src/triangulation.cu)
// for one 16x16 tile of seeds per 256-thread workgroup.
//
// The three-launch pipeline of rmd_matcher.hpp (setup / plan / persistent search, finalisation fused into the next setup)
// spends a third of a VGA frame in launch boundaries, the tail of its setup kernel and a single-workgroup planning step,
// and its search kernel holds a 66 KB window (2 waves per SIMD).  Here a tile's whole frame is local to one workgroup:
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
#ifndef RMD_FRAME_HPP
#define RMD_FRAME_HPP

#include "rmd_matcher.hpp"

namespace rmdk {

#ifndef FR_MIN_WAVES
#define FR_MIN_WAVES 3
#endif
constexpr int FR_WIN_CAP = 5632;  // texels of the LDS window (22 KB); any shape with (width | 1) * height <= FR_WIN_CAP
constexpr int FR_UNIT_ROUNDS = 4, FR_UNIT_ITEMS = FR_UNIT_ROUNDS * TILE_PIX;
constexpr int FR_MAX_FRONT = 256;  // grid slots in front of the tile grid for the tiles that were heavy in the previous frame
constexpr int FR_CTL_WORDS = 32;   // per frame parity (3 sets: this frame's, the previous frame's, the one being cleared)
constexpr int FR_STATE_ABSENT = 7; // lanes of a ragged tile that lie outside the image
constexpr unsigned int FR_SPIN_LIMIT = 1u << 22;
#ifndef FR_PUBLISH_AGE_US
#define FR_PUBLISH_AGE_US 16
#endif
constexpr unsigned long long FR_PUBLISH_AGE_TICKS = FR_PUBLISH_AGE_US * 100ull;  // 10 ns ticks: by then the front tiles of a frame have been handed out
constexpr int FR_TRACE_FRAMES = 256, FR_TRACE_WORDS = 8;  // per workgroup: start, setup done, own tile done, exit (10 ns ticks), work items, units searched
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

// ---- agent-scope accessors (global address space, sc1) ---------------------------------------------------------------
RMDK_D unsigned int ld_agent(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RMDK_D unsigned long long ld_agent(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RMDK_D void st_agent(unsigned int* p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RMDK_D void st_agent(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RMDK_D void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
RMDK_D unsigned long long pack2f(float a, float b) {
  return static_cast<unsigned long long>(__float_as_uint(a)) | (static_cast<unsigned long long>(__float_as_uint(b)) << 32);
}
RMDK_D unsigned long long pack2u(unsigned int a, unsigned int b) { return static_cast<unsigned long long>(a) | (static_cast<unsigned long long>(b) << 32); }

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
  unsigned int bcast[8];
#ifdef RMD_PROFILE_ROUNDS
  unsigned long long prof[8];  // diagnostics build: [4] window policy, [5] staging, [6] rounds + barrier ticks; [0..3] per wave, ticks / count of rounds without (bits 0..23 / 56..63) and with (24..47 / 48..55) a fallback
#endif
};

// One NCC evaluation at px; the LDS window has a run-time row stride.  Two sources for the current-image samples, same
// arithmetic in both: the LDS window when the (regular) footprint lies inside it -- practically always, the window is cut to
// the work -- and per-sample fetches from L2 otherwise (irregular footprints from the replayed roundings, ~1e-6 of the
// steps, and samples outside a clamped window).  The second path is deliberately compact (rolled loops): the kernel's code
// has to stay resident in the instruction cache while workgroups are in all of its phases at once.
template <int SIDE>
RMDK_D float ncc_at_dyn(const SeedParams& P, F2 px, const float* __restrict__ win, int ws, int wx0, int wy0, int wx1, int wy1,
                        const float* __restrict__ ref_patch, int ref_stride, float sum_templ, float denom, unsigned int& n_fallback) {
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
  const bool regular = reg_x && reg_y;
  const bool in_window = regular && ix[0] >= wx0 && iy[0] >= wy0 && ix[0] + SIDE <= wx1 && iy[0] + SIDE <= wy1;
  if (in_window) {
    ncc_sums_lds_pipelined<SIDE>(win + (iy[0] - wy0) * ws + (ix[0] - wx0), ws, ax, ay, ref_patch, ref_stride, sum_img, sum_img_sq,
                                 sum_img_templ);
  } else if (regular) {
    // outside the window (a clamped window, a seed that wandered off): the same separable filter on texel rows from L2, fully
    // unrolled so that the loads overlap -- a rolled loop costs a memory round trip per row and made the few such evaluations
    // the longest thing in a late frame
    ++n_fallback;
    ncc_sums_regular<SIDE, 0>(P.cur + iy[0] * P.cur_stride + ix[0], P.cur_stride, ax, ay, ref_patch, ref_stride, sum_img, sum_img_sq,
                              sum_img_templ);
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

// Texel box of the samples of steps [j0, j1] (relative to the first in-image step) of this lane's seed.  The positions come
// from l = l_first + 0.7 j, which is within 1e-3 px of the replayed value; a sample at p touches texels
// floor(p) - HALF .. floor(p) + HALF + 1 (one more for the replayed roundings), the box keeps another texel to spare.
// A box that misses a sample costs speed, not correctness (ncc_at_dyn reads L2 then).
template <int SIDE>
RMDK_D void seed_range_box(const SeedParams& P, const FrameSmem<SIDE>& S, int tid, bool has, int j0, int j1, int& x0, int& y0, int& x1, int& y1) {
  constexpr int HALF = SIDE / 2;
  x0 = 0x7fffffff; y0 = 0x7fffffff; x1 = -1; y1 = -1;
  if (!has) return;
  const float lf = S.l_first[tid], mx = S.mean_x[tid], my = S.mean_y[tid], dx = S.dir_x[tid], dy = S.dir_y[tid];
  const float la = lf + 0.7f * static_cast<float>(j0), lb = lf + 0.7f * static_cast<float>(j1);
  const float ax = mx + la * dx, bx = mx + lb * dx, ay = my + la * dy, by = my + lb * dy;
  x0 = max(static_cast<int>(floorf(fminf(ax, bx))) - HALF - 1, 0);
  y0 = max(static_cast<int>(floorf(fminf(ay, by))) - HALF - 1, 0);
  x1 = min(static_cast<int>(floorf(fmaxf(ax, bx))) + HALF + 2, P.w - 1);
  y1 = min(static_cast<int>(floorf(fmaxf(ay, by))) + HALF + 2, P.h - 1);
}

// min / max of four ints over the workgroup (all 256 threads call; result uniform)
template <int SIDE>
RMDK_D void block_bbox(FrameSmem<SIDE>& S, int tid, int& x0, int& y0, int& x1, int& y1, int slot) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    x0 = min(x0, __shfl_xor(x0, off, 64)); y0 = min(y0, __shfl_xor(y0, off, 64));
    x1 = max(x1, __shfl_xor(x1, off, 64)); y1 = max(y1, __shfl_xor(y1, off, 64));
  }
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
RMDK_D bool window_fits(int x0, int y0, int x1, int y1) {
  return x1 >= x0 && y1 >= y0 && ((x1 - x0 + 1) | 1) * (y1 - y0 + 1) <= FR_WIN_CAP;
}

// Exclusive prefix of the per-seed step counts of the tile in LDS (S.packed) -> S.prefix[0..256]; returns the total.
// Ends with a barrier.
template <int SIDE>
RMDK_D int frame_prefix(FrameSmem<SIDE>& S, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int n_valid = static_cast<int>(S.packed[tid] & 0xffu);
  int incl = n_valid;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) S.red[wave][0] = incl;
  __syncthreads();
  int wave_off = 0, total = 0;
#pragma unroll
  for (int wv = 0; wv < 4; ++wv) {
    const int v = S.red[wv][0];
    wave_off += wv < wave ? v : 0;
    total += v;
  }
  S.prefix[tid] = wave_off + incl - n_valid;
  if (tid == 0) S.prefix[TILE_PIX] = total;
  __syncthreads();
  return total;
}

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

// The LDS window of the current image, workgroup-uniform.
struct FrameWindow {
  int x0, y0, x1, y1;  // inclusive texel box
  int ws;              // row stride in LDS
  bool valid;          // staged and covering every sample of the tile in LDS
};

// Stage texels [x0, x1] x [y0, y1] of the current image (batches of independent loads: one memory round trip per 2048
// texels).  No barrier.
template <int SIDE>
RMDK_D void frame_stage_window(const SeedParams& P, FrameSmem<SIDE>& S, int tid, const FrameWindow& W) {
  const int ww = W.x1 - W.x0 + 1, wh = W.y1 - W.y0 + 1;
  const int n_el = ww * wh;
  const float inv_ww = 1.0f / static_cast<float>(ww);
  constexpr int BATCH = 8;
  for (int e0 = tid; e0 < n_el; e0 += TILE_PIX * BATCH) {
    float v[BATCH];
    int dst[BATCH];
#pragma unroll
    for (int q = 0; q < BATCH; ++q) {
      const int e = e0 + q * TILE_PIX;
      int r = static_cast<int>(static_cast<float>(e) * inv_ww);  // e / ww, fixed up below (e < 2^14)
      int cc = e - r * ww;
      if (cc < 0) { --r; cc += ww; } else if (cc >= ww) { ++r; cc -= ww; }
      dst[q] = r * W.ws + cc;
      v[q] = e < n_el ? P.cur[(W.y0 + r) * P.cur_stride + W.x0 + cc] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < BATCH; ++q)
      if (e0 + q * TILE_PIX < n_el) S.win[dst[q]] = v[q];
  }
}

#ifdef RMD_PROFILE_ROUNDS
RMDK_D unsigned long long prof_clock() {  // the 100 MHz wall clock, pinned in program order (diagnostics build only)
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return t;
}
#endif
// Rounds of 256 NCC evaluations over work items [k0, k1) of the tile in LDS with window W; arg-max keys accumulate in S.best.
// No barrier.
template <int SIDE>
RMDK_D void frame_rounds(const SeedParams& P, FrameSmem<SIDE>& S, int tid, int k0, int k1, const FrameWindow& W, unsigned int& n_fallback) {
  using Smem = FrameSmem<SIDE>;
  for (int r0 = k0; r0 < k1; r0 += TILE_PIX) {
#ifdef RMD_PROFILE_ROUNDS
    const unsigned long long prof_t0 = prof_clock();
    const unsigned int prof_fb0 = n_fallback;
#endif
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
      const float ncc = ncc_at_dyn<SIDE>(P, px, S.win, W.ws, W.x0, W.y0, W.x1, W.y1, S.ref + pty * Smem::REF_W + ptx, Smem::REF_W,
                                        S.sum_templ[p], S.denom[p], n_fallback);
      if (ncc > -1.0f) {  // NaN and anything the reference's "ncc > best_ncc" would never accept are dropped
        const unsigned int step = ((S.packed[p] >> 8) & 0xffu) + static_cast<unsigned int>(j);
        key = (static_cast<unsigned long long>(orderable_f32(ncc + 0.0f)) << 32) | (0xffffffffu - step);
      }
    }
    // one LDS atomic per evaluation: lanes of one seed hit one address and serialise inside the LDS, which is cheaper than a
    // segmented wave reduction first (12 ds_bpermute round trips)
    if (key != 0ull) atomicMax(&S.best[p], key);
#ifdef RMD_PROFILE_ROUNDS
    {  // diagnostics build only: ticks and count of this wave's rounds with / without an evaluation that left the LDS window
      const bool fb = __any(n_fallback != prof_fb0);
      const unsigned long long dt = prof_clock() - prof_t0;
      if ((tid & 63) == 0) S.prof[tid >> 6] += fb ? (dt << 24) | (1ull << 48) : dt | (1ull << 56);
    }
#endif
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
#ifdef RMD_PROFILE_ROUNDS
    const unsigned long long prof_p0 = prof_clock();
#endif
    if (!tile_win.valid) {
      // this lane's seed contributes steps [max(k - first, 0), min(kX - first, n) - 1] to the candidate range [k, kX)
      const int j0 = max(k - my_first, 0);
      {
        int bx0, by0, bx1, by1;
        seed_range_box<SIDE>(P, S, tid, my_n > 0 && my_first < k1 && my_first + my_n > k, j0, min(k1 - my_first, my_n) - 1, bx0, by0, bx1, by1);
        block_bbox<SIDE>(S, tid, bx0, by0, bx1, by1, 0);
        __syncthreads();
        block_bbox_read<SIDE>(S, W.x0, W.y0, W.x1, W.y1, 0);
      }
      if (!window_fits(W.x0, W.y0, W.x1, W.y1)) {
        const int kb = min(k_end, k + FR_UNIT_ITEMS), kc = min(k_end, k + TILE_PIX);
        int bx0, by0, bx1, by1, cx0, cy0, cx1, cy1;
        const bool four = kb < k_end;  // else the 4-round box IS the box that has just failed (units of <= 4 rounds: always)
        if (four) {
          seed_range_box<SIDE>(P, S, tid, my_n > 0 && my_first < kb && my_first + my_n > k, j0, min(kb - my_first, my_n) - 1, bx0, by0, bx1, by1);
          block_bbox<SIDE>(S, tid, bx0, by0, bx1, by1, 4);
        }
        seed_range_box<SIDE>(P, S, tid, my_n > 0 && my_first < kc && my_first + my_n > k, j0, min(kc - my_first, my_n) - 1, cx0, cy0, cx1, cy1);
        block_bbox<SIDE>(S, tid, cx0, cy0, cx1, cy1, 8);
        __syncthreads();
        if (four) block_bbox_read<SIDE>(S, W.x0, W.y0, W.x1, W.y1, 4);
        k1 = kb;
        if (!window_fits(W.x0, W.y0, W.x1, W.y1)) {
          block_bbox_read<SIDE>(S, W.x0, W.y0, W.x1, W.y1, 8);
          k1 = kc;
          if (!window_fits(W.x0, W.y0, W.x1, W.y1)) {  // clamp around the box's centre; the rest reads L2
            const int ww = W.x1 - W.x0 + 1, wh = W.y1 - W.y0 + 1;
            constexpr int SQ = 73;  // (73 | 1) * 75 <= FR_WIN_CAP
            const int nw = ww > SQ ? (wh > SQ + 2 ? SQ : min(ww, (FR_WIN_CAP / wh - 1) | 1)) : ww;
            const int nh = min(wh, FR_WIN_CAP / (nw | 1));
            W.x0 += (ww - nw) / 2; W.x1 = W.x0 + nw - 1;
            W.y0 += (wh - nh) / 2; W.y1 = W.y0 + nh - 1;
          }
        }
      }
      W.ws = (W.x1 - W.x0 + 1) | 1;
#ifdef RMD_PROFILE_ROUNDS
      const unsigned long long prof_p1 = prof_clock();
#endif
      frame_stage_window<SIDE>(P, S, tid, W);
      ++n_windows;
      __syncthreads();
#ifdef RMD_PROFILE_ROUNDS
      if (tid == 0) { S.prof[4] += prof_p1 - prof_p0; S.prof[5] += prof_clock() - prof_p1; }
#endif
    }
#ifdef RMD_PROFILE_ROUNDS
    const unsigned long long prof_r0 = prof_clock();
#endif
    frame_rounds<SIDE>(P, S, tid, k, k1, W, n_fallback);
    k = k1;
    __syncthreads();  // the window may be re-staged; S.best is complete for [k_begin, k)
#ifdef RMD_PROFILE_ROUNDS
    if (tid == 0) S.prof[6] += prof_clock() - prof_r0;
#endif
  }
}

// Exclusive prefix of the step counts (S.packed) -> S.prefix[0..256], and the texel box of all samples of the tile.  If that
// box fits the LDS window it is staged right away (W.valid).  Returns the tile's number of work items.  Ends with a barrier.
template <int SIDE>
RMDK_D int frame_prefix_and_window(const SeedParams& P, FrameSmem<SIDE>& S, int tid, FrameWindow& W) {
  const int lane = tid & 63, wave = tid >> 6;
  const unsigned int pk = S.packed[tid];
  const int n_valid = static_cast<int>(pk & 0xffu);
  int incl = n_valid;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  int bx0, by0, bx1, by1;
  seed_range_box<SIDE>(P, S, tid, n_valid > 0, 0, n_valid - 1, bx0, by0, bx1, by1);
  block_bbox<SIDE>(S, tid, bx0, by0, bx1, by1, 0);
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
  block_bbox_read<SIDE>(S, W.x0, W.y0, W.x1, W.y1, 0);
  W.valid = total > 0 && window_fits(W.x0, W.y0, W.x1, W.y1);
  W.ws = (W.x1 - W.x0 + 1) | 1;
  if (W.valid) frame_stage_window<SIDE>(P, S, tid, W);
  __syncthreads();
  return total;
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

// ------------------------------------------------------------------------------------------------------------------------
// The two-launch pipeline: the load-balanced structure of rmd_matcher.hpp (every workgroup of the search is present from the
// start and pulls fixed-size work units from one list: one returning atomic per unit, nobody polls) on the compact LDS layout
// and the pipelined NCC block of this file, with the two small launches folded away.
//
//   seed_setup_compact   one workgroup per tile, one lane per seed: state check, epipolar segment, in-image run, descriptors
//                        to global memory; seeds without an in-image step are settled right here (NO_MATCH).  The workgroup
//                        that finishes LAST (ticket counter) turns the per-tile totals into the unit list: what used to be a
//                        launch of its own plus two launch gaps.
//   seed_search_compact  persistent workgroups pull units; per unit the LDS window is cut to the unit's own samples (a few
//                        thousand texels instead of the tile's worst case of 133 x 104: 38 KB of LDS per workgroup instead
//                        of 66 KB, four workgroups per CU).  Arg-max keys are collected per seed in LDS and handed to the
//                        global plane with one atomic per seed and unit; the workgroup that completes a tile's last unit
//                        (per-tile counter) finalises the tile: triangulation + fusion overlap with the other tiles' search
//                        instead of sitting at the head of the next frame's critical path, and the state is final when
//                        update()'s last kernel ends, as in the reference (seed_matrix.cu:139-157).
// Visibility (MI355X_MICROARCH.md): keys travel by device-scope atomics and are read back with agent-scope loads after the
// tile counter (a device-scope atomic, incremented only after the writer drained its atomics) says they are complete;
// everything else crosses a kernel boundary.

// per-tile totals -> unit size, unit list, per-tile unit counts.  One workgroup of 256 threads; reads the totals with
// agent-scope loads (they were written by other workgroups of the same launch).
RMDK_D void plan_units_in_kernel(const MatcherArgs& M, int n_tiles, int target_units, int* s_a, int* s_b) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c_tiles = (n_tiles + TILE_PIX - 1) / TILE_PIX;
  const int t_first = tid * c_tiles;
  int items = 0;
  for (int q = 0; q < c_tiles; ++q) {
    const int t = t_first + q;
    if (t < n_tiles) items += static_cast<int>(ld_agent(&M.tile_plan[t]));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) items += __shfl_xor(items, off, 64);
  if (lane == 0) s_a[wave] = items;
  __syncthreads();
  items = s_a[0] + s_a[1] + s_a[2] + s_a[3];
  int unit_rounds = (items + target_units * TILE_PIX - 1) / (target_units * TILE_PIX);
  unit_rounds = min(max(unit_rounds, 1), MAX_UNIT_ROUNDS);
  const int unit_items = unit_rounds * TILE_PIX;
  int mine = 0;
  for (int q = 0; q < c_tiles; ++q) {
    const int t = t_first + q;
    if (t < n_tiles) mine += units_of(static_cast<int>(ld_agent(&M.tile_plan[t])), unit_rounds);
  }
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) s_b[wave] = incl;
  __syncthreads();
  int wave_off = 0, block_tot = 0;
#pragma unroll
  for (int wv = 0; wv < 4; ++wv) {
    const int v = s_b[wv];
    wave_off += wv < wave ? v : 0;
    block_tot += v;
  }
  if (tid == 0) {
    M.queue[0] = static_cast<unsigned int>(block_tot);
    M.queue[1] = 0u;  // hand-out counter of this frame's search
    M.queue[5] = static_cast<unsigned int>(unit_items);
  }
  int base = wave_off + incl - mine;
  for (int q = 0; q < c_tiles; ++q) {
    const int t = t_first + q;
    if (t >= n_tiles) break;
    const int n_u = units_of(static_cast<int>(ld_agent(&M.tile_plan[t])), unit_rounds);
    M.tile_pending[t] = static_cast<unsigned int>(n_u);
    for (int u = 0; u < n_u; ++u) M.units[base + u] = make_uint2(static_cast<unsigned int>(t), static_cast<unsigned int>(u * unit_items));
    base += n_u;
  }
}

// FUSE_PREV: the previous frame's finalisation has been deferred (nobody looked at the state in between): it runs here, in
// the lane that owns the same seed, on the state values this lane has loaded anyway -- the posterior goes straight from
// registers into the new frame's check.  Everything a lane needs from memory is requested in ONE batch up front (the
// compiler does not hoist loads out of the branches that consume them, and five dependent round trips under a burst of a
// million requests were two thirds of this kernel's time).
// The unit list is built without a planning step: a tile reserves its units in the list of its shard (tile % 16) with one
// returning atomic on the shard's counter; the unit size comes from the PREVIOUS frame's total work (the counters of three
// consecutive frames rotate).  The search kernel reads the sixteen counts and walks the shards' lists as one list.
constexpr int INGEST_WGS = 128;  // workgroups that bring a host frame into the current-image plane (the only ones that may wait)
constexpr int INGEST_WGS_REMAP = 512;  // with lens undistortion (two dependent round trips per pixel): a quarter of the chip's wave slots at most

template <int SIDE, bool FUSE_PREV>
__global__ __launch_bounds__(TILE_PIX) void seed_setup_compact_kernel(SeedParams P, MatcherArgs M, Pose T_ref_curr_prev, int target_units) {
  __shared__ int red_i[4];
  __shared__ unsigned int s_base;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tx = tid & (TILE_W - 1), ty = tid >> 4;
  const int x = blockIdx.x * TILE_W + tx, y = blockIdx.y * TILE_H + ty;
  const bool in_image = x < P.w && y < P.h;
  const int gi = in_image ? y * P.stride + x : 0;
  const unsigned long long t_start = P.trace ? wall_clock64() : 0ull;  // timeline probes (diagnostics)
  unsigned long long t_loaded = 0ull;
  // frame ingest (see MatcherArgs): the workgroups BELOW the tile grid (blockIdx.y >= tiles_y, at most INGEST_WGS of them, launched
  // only when a host frame is pending) wait for the staging copy's flag and convert the staged frame into the current-image plane.
  // Only these few workgroups ever wait: if every tile workgroup did, a device filled with waiting waves could keep a copy that is
  // carried out by a blit kernel from ever running.  The wait is bounded; a copy that never arrives is reported through progress[1].
  // Loads of the flag and of the staged frame are agent-scope: the copy may finish after this kernel has started.
  const int wg = blockIdx.y * gridDim.x + blockIdx.x;
  if (static_cast<int>(blockIdx.y) >= M.tiles_y) {
    const int iw = (static_cast<int>(blockIdx.y) - M.tiles_y) * gridDim.x + blockIdx.x;
    if (iw >= M.ingest_wgs) return;
    // frame numbers are compared modulo 2^32 (a live system never stops counting): "behind" = the signed difference is negative
    auto behind = [&]() { return static_cast<int>(ld_agent(M.ingest_flag) - M.ingest_number) < 0; };
    if (behind()) {
      unsigned int spins = 0u;
      while (behind() && ++spins < (1u << 18)) __builtin_amdgcn_s_sleep(16);
      if (behind() && tid == 0) __hip_atomic_store(M.progress + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (M.ingest_u8 && M.ingest_map1) {  // cv::remap through the undistortion maps first (depthmap.cpp:99), destination pixels four at a time per lane
      const unsigned char* src = reinterpret_cast<const unsigned char*>(M.ingest_u8);
      const int total = P.w * P.h, step = M.ingest_wgs * TILE_PIX;
      for (int d0 = iw * TILE_PIX + tid; d0 < total; d0 += 4 * step) {
        short2 m[4];
        int f[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // all map entries of the batch are requested before the first source pixel
          const int d = min(d0 + q * step, total - 1);
          m[q] = M.ingest_map1[d];
          f[q] = M.ingest_map2[d] & 1023;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int d = d0 + q * step;
          if (d >= total) break;
          const int row = d / P.w;
          M.ingest_dst[static_cast<size_t>(row) * P.stride + (d - row * P.w)] = remap_u8_pixel(src, M.ingest_pitch, m[q], f[q], P.w, P.h, [](const unsigned char* p) {
            return static_cast<int>(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          });
        }
      }
    } else if (M.ingest_u8) {  // x (1/255): Depthmap::inputImage's convertTo(CV_32F, 1.0f / 255.0f), depthmap.cpp:105 -- one fp32 multiply per pixel
      const int per_row = M.ingest_pitch >> 2, total = per_row * P.h;
      for (int d = iw * TILE_PIX + tid; d < total; d += M.ingest_wgs * TILE_PIX) {
        const unsigned int v = ld_agent(M.ingest_u8 + d);
        const int row = d / per_row, x4 = (d - row * per_row) * 4;
        float* out = M.ingest_dst + static_cast<size_t>(row) * P.stride + x4;
        const float f0 = static_cast<float>(v & 0xffu) * (1.0f / 255.0f), f1 = static_cast<float>((v >> 8) & 0xffu) * (1.0f / 255.0f);
        const float f2 = static_cast<float>((v >> 16) & 0xffu) * (1.0f / 255.0f), f3 = static_cast<float>(v >> 24) * (1.0f / 255.0f);
        if (x4 + 3 < P.w) *reinterpret_cast<float4*>(out) = make_float4(f0, f1, f2, f3);  // plane rows and x4 are multiples of 16 bytes
        else {
          if (x4 < P.w) out[0] = f0;
          if (x4 + 1 < P.w) out[1] = f1;
          if (x4 + 2 < P.w) out[2] = f2;
        }
      }
    } else {
      const int total = P.w * P.h;
      for (int d = iw * TILE_PIX + tid; d < total; d += M.ingest_wgs * TILE_PIX) {
        const int row = d / P.w;
        M.ingest_dst[static_cast<size_t>(row) * P.stride + (d - row * P.w)] = __uint_as_float(ld_agent(reinterpret_cast<const unsigned int*>(M.ingest_f32) + d));
      }
    }
    return;
  }
  if (M.progress && wg == 0 && tid == 0) __hip_atomic_store(M.progress, M.ingest_number, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // (computed here, at the top, so that its scalar loads travel with the kernel arguments)
  // unit size: 1..4 rounds of 256 evaluations, from the previous frame's work (a frame differs little from the one before)
  int unit_rounds = MAX_UNIT_ROUNDS;
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
#ifdef RMD_PROFILE_ROUNDS
  unsigned long long prof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // fuse: start, triangulated, uncertainty, normpdf, end; segment done; run done
#endif
  float mu = P.mu[gi], sigma_sq = P.sigma_sq[gi], a = P.a[gi], b = P.b[gi];
  // what the planes hold now (known only when the previous frame's values were loaded for its finalisation): a seed that has
  // converged or diverged keeps writing the same state and an empty descriptor, a third of this kernel's stores -- skipped
  int conv_old = -1;
  unsigned int packed_old = 0xffffffffu;
  if (FUSE_PREV) {
    const int conv_prev = P.conv[gi];
    const unsigned long long key = M.best[gi];
    const unsigned int packed_prev = M.packed[gi];
    conv_old = conv_prev; packed_old = packed_prev;
    const float lfirst_prev = M.lfirst[gi];
    const float2 m_prev = M.mean[gi], d_prev = M.dir[gi];
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
      SeedParams Pprev = P;
      Pprev.T_ref_curr = T_ref_curr_prev;
#ifdef RMD_PROFILE_ROUNDS
      prof_t[0] = prof_clock();
      const int what = seed_fuse_values(Pprev, x, y, state_prev, mu, sigma_sq, a, b, best_px, prof_t + 1);
      prof_t[4] = prof_clock();
#else
      const int what = seed_fuse_values(Pprev, x, y, state_prev, mu, sigma_sq, a, b, best_px);
#endif
      if (what == 1) { P.sigma_sq[gi] = sigma_sq; P.mu[gi] = mu; P.a[gi] = a; P.b[gi] = b; }
      else if (what == 2) P.b[gi] = b;
    }
  }
  int state = ST_BORDER;
  if (in_image) {
    state = seed_check(P, x, y, sigma_sq, a, b, SIDE);
    if (state != conv_old) P.conv[gi] = state;  // final for BORDER / CONVERGED / DIVERGED; UPDATE seeds are settled by the finalisation
  }
  if (P.trace) t_loaded = wall_clock64();
  int n_valid = 0, i_first = 0;
  unsigned int n_steps = 0, n_evals = 0;
  const bool live = in_image && state == ST_UPDATE;
  if (live) {
    const Segment seg = epipolar_segment(P, x, y, mu, sigma_sq);
#ifdef RMD_PROFILE_ROUNDS
    prof_t[5] = prof_clock();
#endif
    const ValidRun run = find_valid_run(P, seg, SIDE);
#ifdef RMD_PROFILE_ROUNDS
    prof_t[6] = prof_clock();
#endif
    n_valid = run.n_valid; i_first = run.i_first;
    M.best[gi] = 0ull;
    if (n_valid > 0) {
      M.mean[gi] = make_float2(seg.mean.x, seg.mean.y);
      M.dir[gi] = make_float2(seg.dir.x, seg.dir.y);
      M.lfirst[gi] = run.l_first;
    }
    if (P.stats) {  // diagnostics only: the full walk, counting what the reference would visit / evaluate
      for (float l = -seg.half_length; l <= seg.half_length; l += 0.7f, ++n_steps) {
        const F2 px = F2{seg.mean.x + l * seg.dir.x, seg.mean.y + l * seg.dir.y};
        if (!px_outside(P, px, SIDE)) ++n_evals;  // NaN positions pass the reference's guard too
      }
    }
  }
  const unsigned int packed_new = (static_cast<unsigned int>(i_first) << 16) | static_cast<unsigned int>(n_valid);
  if (in_image && packed_new != packed_old) M.packed[gi] = packed_new;
  if (P.stats) {
    const unsigned long long s_live = wave_sum_u64(live ? 1ull : 0ull);
    const unsigned long long s_steps = wave_sum_u64(static_cast<unsigned long long>(n_steps));
    const unsigned long long s_evals = wave_sum_u64(static_cast<unsigned long long>(n_evals));
    if (lane == 0 && s_live) { atomicAdd(&P.stats[0], s_live); atomicAdd(&P.stats[1], s_steps); atomicAdd(&P.stats[2], s_evals); }
  }
  int tot = n_valid;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
  if (lane == 0) red_i[wave] = tot;
  __syncthreads();
  const int total = red_i[0] + red_i[1] + red_i[2] + red_i[3];
  const int tile = blockIdx.y * M.tiles_x + blockIdx.x;
  const int unit_items = unit_rounds * TILE_PIX;
  if (tile == 0 && tid < UNIT_SHARDS) M.shards_next[tid] = 0ull;  // the set the NEXT frame counts in (nobody reads it now)
  if (tile == 0 && tid == 0) { M.queue[1] = 0u; M.queue[5] = static_cast<unsigned int>(unit_items); }
#ifdef RMD_PROFILE_ROUNDS
  if (P.trace && prof_t[0] != 0ull && prof_t[6] != 0ull) {  // any live lane that ran both the fusion and the set-up: phases of the setup chain, 10 ns ticks
    auto d = [](unsigned long long a, unsigned long long b) { return static_cast<unsigned long long>(b > a ? (b - a > 511 ? 511 : b - a) : 0); };
    const unsigned long long w = d(t_start, prof_t[0]) | (d(prof_t[0], prof_t[1]) << 9) | (d(prof_t[1], prof_t[2]) << 18) | (d(prof_t[2], prof_t[3]) << 27) |
                                 (d(prof_t[3], prof_t[4]) << 36) | (d(prof_t[4], prof_t[5]) << 45) | (d(prof_t[5], prof_t[6]) << 54);
    P.trace[static_cast<size_t>(tile) * FR_TRACE_WORDS + 2] = w;  // racing lanes: any one of them will do
  }
#endif
#ifndef RMD_PROFILE_ROUNDS
  if (P.trace && tid == 0)  // word 2 of the tile's slot: start (low 32 bits of the 10 ns clock), state ready and end relative to it
    P.trace[static_cast<size_t>(tile) * FR_TRACE_WORDS + 2] = (t_start & 0xffffffffull) | (((t_loaded - t_start) & 0xffffull) << 32) |
                                                             (((wall_clock64() - t_start) & 0xffffull) << 48);
#endif
  if (total == 0) return;
  const int n_u = units_of(total, unit_rounds);
  if (tid == 0) {
    const unsigned long long old = atomicAdd(&M.shards_cur[tile % UNIT_SHARDS], (static_cast<unsigned long long>(total) << 32) | static_cast<unsigned long long>(n_u));
    s_base = static_cast<unsigned int>(old);  // units reserved so far in this shard
  }
  __syncthreads();
  if (tid < n_u) M.units[static_cast<size_t>(tile % UNIT_SHARDS) * M.shard_cap + s_base + tid] =
      make_uint2(static_cast<unsigned int>(tile), static_cast<unsigned int>(tid * unit_items));
  if (tid == 0) M.tile_plan[tile] = static_cast<unsigned int>(total);
}

template <int SIDE>
__global__ __launch_bounds__(TILE_PIX, FR_MIN_WAVES) void seed_search_compact_kernel(SeedParams P, MatcherArgs M) {
  using Smem = FrameSmem<SIDE>;
  constexpr int HALF = SIDE / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& S = *reinterpret_cast<Smem*>(smem_raw);
  const int tid = threadIdx.x;
  const int tx = tid & (TILE_W - 1), ty = tid >> 4;
  // the shards' unit lists as one list: unit g lives in shard s with first[s] <= g < first[s + 1]
  unsigned int shard_first[UNIT_SHARDS + 1];
  shard_first[0] = 0u;
#pragma unroll
  for (int q = 0; q < UNIT_SHARDS; ++q) shard_first[q + 1] = shard_first[q] + static_cast<unsigned int>(M.shards_cur[q]);
  const unsigned int n_units = shard_first[UNIT_SHARDS];
  const int unit_items = static_cast<int>(M.queue[5]);
  unsigned int n_fallback = 0, n_windows = 0, n_done = 0, n_items = 0;
  int lds_tile = -1, x0 = 0, y0 = 0, total = 0;
  unsigned long long* const tr = P.trace && static_cast<int>(blockIdx.x) < M.tiles_x * ((P.h + TILE_H - 1) / TILE_H) ? P.trace + static_cast<size_t>(blockIdx.x) * FR_TRACE_WORDS : nullptr;
  if (tr && tid == 0) tr[0] = wall_clock64();
#ifdef RMD_PROFILE_ROUNDS
  if (tid < 8) S.prof[tid] = 0ull;
#endif
  FrameWindow W;
  W.valid = false; W.x0 = W.y0 = 0; W.x1 = W.y1 = -1; W.ws = 1;
  unsigned int u = blockIdx.x;  // unit blockIdx.x is ours for free; further units come from the shared counter
  while (u < n_units) {
    int sh = 0;
#pragma unroll
    for (int q = 1; q < UNIT_SHARDS; ++q) sh += u >= shard_first[q] ? 1 : 0;
    unsigned int sh_first = 0u;
#pragma unroll
    for (int q = 1; q < UNIT_SHARDS; ++q) sh_first = u >= shard_first[q] ? shard_first[q] : sh_first;
    const uint2 unit = M.units[static_cast<size_t>(sh) * M.shard_cap + (u - sh_first)];
    const int tile = static_cast<int>(unit.x), first = static_cast<int>(unit.y);
    if (tile != lds_tile) {
      if (lds_tile >= 0) {  // hand the previous tile's keys over
        const unsigned long long key = S.best[tid];
        if (key != 0ull) atomicMax(&M.best[static_cast<size_t>(y0 + ty) * P.stride + x0 + tx], key);
      }
      const int tile_y = tile / M.tiles_x, tile_x = tile - tile_y * M.tiles_x;
      x0 = tile_x * TILE_W; y0 = tile_y * TILE_H;
      const int x = x0 + tx, y = y0 + ty;
      unsigned int pk = 0u;
      {  // one batch of loads (a lane outside a ragged tile reads element 0; values of seeds without work are not used)
        const int gi = (x < P.w && y < P.h) ? y * P.stride + x : 0;
        const unsigned int packed = M.packed[gi];  // first in-image step << 16 | number of in-image steps
        const float2 m = M.mean[gi], d = M.dir[gi];
        const float lf = M.lfirst[gi], st = P.sum_templ[gi], dn = P.denom[gi];
        if (x < P.w && y < P.h) pk = ((packed >> 16) << 8) | (packed & 0xffu);
        S.mean_x[tid] = m.x; S.mean_y[tid] = m.y; S.dir_x[tid] = d.x; S.dir_y[tid] = d.y;
        S.l_first[tid] = lf;
        S.sum_templ[tid] = st; S.denom[tid] = dn;
      }
      S.packed[tid] = pk;
      S.best[tid] = 0ull;
      for (int i = tid; i < Smem::REF_H * Smem::REF_W; i += TILE_PIX) {
        const int ry = i / Smem::REF_W, rx = i - ry * Smem::REF_W;
        S.ref[i] = P.ref[clampi(y0 - HALF + ry, 0, P.h - 1) * P.stride + clampi(x0 - HALF + rx, 0, P.w - 1)];
      }
      total = frame_prefix_and_window<SIDE>(P, S, tid, W);  // barriers inside
      lds_tile = tile;
      if (tr && tid == 0 && n_done == 0) tr[1] = wall_clock64();
    }
    frame_search<SIDE>(P, S, tid, first, min(first + unit_items, total), W, n_fallback, n_windows);  // ends with a barrier
    ++n_done; n_items += static_cast<unsigned int>(min(first + unit_items, total) - first);
    if (n_units <= gridDim.x) break;  // light frame: every unit had its own workgroup, nothing to hand out
    if (tid == 0) S.bcast[0] = gridDim.x + atomicAdd(&M.queue[1], 1u);
    __syncthreads();
    u = S.bcast[0];
    __syncthreads();
  }
  if (lds_tile >= 0) {
    const unsigned long long key = S.best[tid];
    if (key != 0ull) atomicMax(&M.best[static_cast<size_t>(y0 + ty) * P.stride + x0 + tx], key);
  }
  if (tr && tid < 64) {
    const unsigned long long fb = wave_sum_u64(n_fallback);  // the first wave's lanes only: a hint, not a count
    if (tid == 0) {
      tr[3] = wall_clock64();
      tr[4] = n_items; tr[5] = n_done; tr[6] = static_cast<unsigned long long>(lds_tile >= 0 ? lds_tile : 0);
#ifdef RMD_PROFILE_ROUNDS
      tr[6] = S.prof[0];
      tr[2] = (S.prof[4] & 0xfffffull) | ((S.prof[5] & 0xfffffull) << 20) | ((S.prof[6] & 0xffffffull) << 40);  // window policy, staging, rounds incl. barrier
#endif
      tr[7] = fb | (static_cast<unsigned long long>(n_windows) << 32);
    }
  }
}

// setup (+ the deferred finalisation of the previous frame when fuse_prev; builds the unit list) -> compact search.
// The caller zeroes ws.d_shards and sets ws.frame = 0 whenever the sequence restarts.
template <int SIDE>
inline hipError_t launch_seed_pipeline_compact(const SeedParams& P, MatcherWorkspace& ws, FrameWorkspace& fws, hipStream_t stream, int num_cus,
                                               bool fuse_prev, const Pose& T_ref_curr_prev, int target_mult = 1, const IngestArgs* ingest = nullptr) {
  using Smem = FrameSmem<SIDE>;
  MatcherArgs M = matcher_args(ws);
  M.trace = nullptr;
  if (ingest) {
    M.ingest_u8 = ingest->u8; M.ingest_f32 = ingest->f32; M.ingest_dst = ingest->dst; M.ingest_pitch = ingest->pitch;
    M.ingest_map1 = ingest->map1; M.ingest_map2 = ingest->map2;
    M.ingest_flag = ingest->flag; M.progress = ingest->progress; M.ingest_number = ingest->number;
  }
  auto search = seed_search_compact_kernel<SIDE>;
  bool& attr = fws.attr_set_compact[SIDE / 2 - 1];
  if (!attr) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(search), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(sizeof(Smem)));
    if (e != hipSuccess) return e;
    int wg = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg, reinterpret_cast<const void*>(search), TILE_PIX, sizeof(Smem)) != hipSuccess || wg < 1) wg = 1;
    fws.compact_wg_per_cu[SIDE / 2 - 1] = wg > 4 ? 4 : wg;
    attr = true;
  }
  const int resident = num_cus * fws.compact_wg_per_cu[SIDE / 2 - 1];
  dim3 tiles(ws.tiles_x, ws.tiles_y);
  M.tiles_y = ws.tiles_y;
  if (M.ingest_u8 || M.ingest_f32) {  // the ingest workgroups: rows below the tile grid
    const long long dwords = M.ingest_u8 && !M.ingest_map1 ? static_cast<long long>(M.ingest_pitch >> 2) * P.h : static_cast<long long>(P.w) * P.h;
    const long long want = (dwords + TILE_PIX - 1) / TILE_PIX, cap = M.ingest_u8 && M.ingest_map1 ? INGEST_WGS_REMAP : INGEST_WGS;
    M.ingest_wgs = static_cast<int>(want < cap ? want : cap);
    tiles.y += static_cast<unsigned int>((M.ingest_wgs + ws.tiles_x - 1) / ws.tiles_x);
  }
  const int target_units = resident * target_mult;
  if (fuse_prev) hipLaunchKernelGGL((seed_setup_compact_kernel<SIDE, true>), tiles, dim3(TILE_PIX), 0, stream, P, M, T_ref_curr_prev, target_units);
  else hipLaunchKernelGGL((seed_setup_compact_kernel<SIDE, false>), tiles, dim3(TILE_PIX), 0, stream, P, M, T_ref_curr_prev, target_units);
  hipLaunchKernelGGL(search, dim3(resident), dim3(TILE_PIX), sizeof(Smem), stream, P, M);
  ++ws.frame;
  return hipGetLastError();
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

#endif  // RMD_FRAME_HPP
