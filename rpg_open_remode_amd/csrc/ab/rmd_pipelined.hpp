// ab/rmd_pipelined.hpp -- EXPERIMENT, A/B builds only (-DRMD_AB_PIPELINE, then RMD_HIP_OPT_PIPELINE / RMD_HIP_PIPELINE=1; one sequence, frames resident
// in HBM): ONE launch per update.  MEASURED AND DROPPED (profiles/r04_ab_one_launch_per_update.txt): bit-identical, 3-10 % slower.
//
// The product pipeline (rmd_frame.hpp) runs two kernels per update, setup(n) -> search(n), and every kernel boundary costs 3-4 us of an
// update's 33-40.  Here a launch is  search(n - 1) + setup(n):  the workgroup that finishes the LAST work unit of a tile in the search of
// frame n - 1 (one returning atomic on a per-tile count) runs that tile's setup for frame n right away -- the fused finalisation of frame
// n - 1 from the arg-max keys that have just been completed, the check, the epipolar segment under frame n's pose, the valid run -- and
// appends the tile's units to the lists the NEXT launch searches.  The dependency is per tile: nothing spins, there is no grid barrier, and
// the setup chains of all tiles but the slowest hide behind other tiles' searches.  The price is one more stage of latency on the host
// side: the search of the newest frame runs with the next update() (or, when somebody looks at the state, as a launch of its own:
// rmdh::seeds_flush), and the caller's frame n - 1 must stay alive during update(n) -- which rmd_hip_seeds_update_device already asks for.
//
// The per-seed arithmetic is that of the product kernels, operation for operation (the setup body below restates seed_setup_compact_kernel
// for one sequence without the diagnostics and the frame ingest; the search part calls the product's device functions), so the results are
// bit-identical (tools/exp_pipeline.py).  What is double-buffered by the parity of the frame number, because a launch consumes one set
// while it produces the other: unit lists, hand-out counters, the unit size word, the per-tile unit / done counts.
#ifndef RMD_PIPELINED_HPP
#define RMD_PIPELINED_HPP

#include "../rmd_frame.hpp"

namespace rmdk {

struct PipeArgs {
  const unsigned int* tile_units_cur;  // per tile: units of the tile in the lists this launch SEARCHES (written by the setup that built them)
  unsigned int* tile_done_cur;         // per tile: units of those finished so far (zero at launch)
  unsigned int* tile_units_new;        // the same for the lists this launch BUILDS: written / zeroed by the tile's setup
  unsigned int* tile_done_new;
  const unsigned int* unit_items_cur;  // items per unit of the lists being searched (one word)
  unsigned int* unit_items_new;        // ... of the lists being built
  unsigned int* handout_new;           // hand-out counters of the NEXT launch's search (zeroed here)
  int target_units;
};

// unit size of the lists being built: 1..4 rounds of 256 evaluations, from the previous frame's work (as in seed_setup_compact_kernel)
RMDK_D int pipe_unit_rounds(const MatcherArgs& M, int target_units) {
  int unit_rounds = MAX_UNIT_ROUNDS;
  if (M.shards_prev) {
    typedef const __attribute__((address_space(4))) unsigned long long* const_u64_ptr_t;
    const const_u64_ptr_t prev = (const_u64_ptr_t)(M.shards_prev);
    unsigned long long items = 0;
#pragma unroll
    for (int q = 0; q < UNIT_SHARDS; ++q) items += prev[q] >> 32;
    const unsigned long long per_round = static_cast<unsigned long long>(target_units) * TILE_PIX;
    unit_rounds = items > 3 * per_round ? 4 : items > 2 * per_round ? 3 : items > per_round ? 2 : 1;
  }
  return unit_rounds;
}

// the launch's housekeeping (one workgroup): the counters of the launches to come
RMDK_D void pipe_housekeeping(const MatcherArgs& M, const PipeArgs& X, int unit_rounds, int tid) {
  if (tid < UNIT_SHARDS) M.shards_next[tid] = 0ull;               // the set the NEXT frame's setup counts in (nobody reads it now)
  if (tid < UNIT_SHARDS) X.handout_new[tid * HANDOUT_STRIDE] = 0u;  // the hand-out counters of the search of the lists being built
  if (tid == 0) *X.unit_items_new = static_cast<unsigned int>(unit_rounds * TILE_PIX);
}

// Setup of ONE tile for the frame of Q / M (seed_setup_compact_kernel's body for one sequence).  Where the previous frame's arg-max keys come
// from: KEYS_PLANE -- the plane, written by an earlier launch; KEYS_COHERENT -- the plane, completed by atomics of THIS launch, possibly on other
// XCDs: read at agent scope; KEYS_HANDED -- `key_in`: the caller has searched the tile's only unit itself and still holds the keys (nothing
// travels through memory).  A tile that has just been searched is alive: only KEYS_PLANE looks at the tile's live count.
// All 256 threads call; uniform returns.
enum : int { KEYS_PLANE = 0, KEYS_COHERENT = 1, KEYS_HANDED = 2 };
template <int SIDE, int KEYS>
RMDK_D void pipe_setup_tile(const SeqArgs& Q, const MatcherArgs& M, const PipeArgs& X, int tile, int unit_rounds, int tid, unsigned long long key_in = 0ull) {
  __shared__ int red_i[4], red_c[4], red_l[4], red_b[4][4];
  __shared__ unsigned int s_base;
  constexpr int HALF = SIDE / 2;
  const SeedParams& P = Q.P;
  __syncthreads();  // the arrays above may still be read by a slower wave of this workgroup's previous tile
  if (KEYS == KEYS_PLANE && Q.fuse_prev && M.tile_live[tile] == 0u) {  // a DEAD tile (see seed_setup_compact_kernel): nothing changes, no units
    if (tid == 0) { X.tile_units_new[tile] = 0u; X.tile_done_new[tile] = 0u; }
    return;
  }
  const int tile_y = tile / M.tiles_x, tile_x = tile - tile_y * M.tiles_x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tx = tid & (TILE_W - 1), ty = tid >> 4;
  const int x = tile_x * TILE_W + tx, y = tile_y * TILE_H + ty;
  const bool in_image = x < P.w && y < P.h;
  const int gi = in_image ? y * P.stride + x : 0;
  const size_t gm = gi;
  float mu = P.mu[gi], sigma_sq = P.sigma_sq[gi], a = P.a[gi], b = P.b[gi];
  int conv_prev = P.conv[gi];
  unsigned long long key = KEYS == KEYS_HANDED ? key_in : KEYS == KEYS_COHERENT ? ld_agent(M.best + gm) : M.best[gm];
  unsigned int packed_prev = M.packed[gm];
  float lfirst_prev = M.lfirst[gm];
  float2 m_prev = M.mean[gm], d_prev = M.dir[gm];
  int conv_old = -1;
  unsigned int packed_old = 0xffffffffu;
  if (Q.fuse_prev) {
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
  int state = ST_BORDER;
  if (in_image) {
    const bool settled = conv_old == ST_BORDER || conv_old == ST_CONVERGED || conv_old == ST_DIVERGED;
    state = settled ? conv_old : seed_check(P, x, y, sigma_sq, a, b, SIDE);
    if (state != conv_old) P.conv[gi] = state;
  }
  int n_valid = 0, i_first = 0;
  int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = -1, by1 = -1;
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
      bx0 = max(static_cast<int>(floorf(fminf(run.px_first.x, run.px_last.x))) - HALF - 1, 0);
      by0 = max(static_cast<int>(floorf(fminf(run.px_first.y, run.px_last.y))) - HALF - 1, 0);
      bx1 = min(static_cast<int>(floorf(fmaxf(run.px_first.x, run.px_last.x))) + HALF + 2, P.w - 1);
      by1 = min(static_cast<int>(floorf(fmaxf(run.px_first.y, run.px_last.y))) + HALF + 2, P.h - 1);
    }
  }
  const unsigned int packed_new = (static_cast<unsigned int>(i_first) << 16) | static_cast<unsigned int>(n_valid);
  if (in_image && packed_new != packed_old) M.packed[gm] = packed_new;
  const int n_conv = __popcll(__ballot(in_image && state == ST_CONVERGED));
  int tot = 0;
  if (__any(n_valid > 0)) {
    tot = wave_reduce_i32<WaveAdd>(n_valid);
    bx0 = wave_reduce_i32<WaveMin>(bx0); by0 = wave_reduce_i32<WaveMin>(by0);
    bx1 = wave_reduce_i32<WaveMax>(bx1); by1 = wave_reduce_i32<WaveMax>(by1);
  }
  const int n_live = __popcll(__ballot(live));
  if (lane == 0) { red_i[wave] = tot; red_c[wave] = n_conv; red_l[wave] = n_live; red_b[wave][0] = bx0; red_b[wave][1] = by0; red_b[wave][2] = bx1; red_b[wave][3] = by1; }
  __syncthreads();
  const int total = red_i[0] + red_i[1] + red_i[2] + red_i[3];
  const int unit_items = unit_rounds * TILE_PIX;
  const int n_u = total > 0 ? units_of(total, unit_rounds) : 0;
  if (tid == 0) {
    M.tile_conv[tile] = static_cast<unsigned int>(red_c[0] + red_c[1] + red_c[2] + red_c[3]);
    M.tile_live[tile] = static_cast<unsigned int>(red_l[0] + red_l[1] + red_l[2] + red_l[3]);
    X.tile_units_new[tile] = static_cast<unsigned int>(n_u);
    X.tile_done_new[tile] = 0u;
  }
  if (total == 0) return;
  if (tid == 0) {
    const unsigned long long old = atomicAdd(&M.shards_cur[tile % UNIT_SHARDS], (static_cast<unsigned long long>(total) << 32) | static_cast<unsigned long long>(n_u));
    s_base = static_cast<unsigned int>(old);
  }
  __syncthreads();
  if (tid < n_u) {
    const int x0 = min(min(red_b[0][0], red_b[1][0]), min(red_b[2][0], red_b[3][0])), y0 = min(min(red_b[0][1], red_b[1][1]), min(red_b[2][1], red_b[3][1]));
    const int x1 = max(max(red_b[0][2], red_b[1][2]), max(red_b[2][2], red_b[3][2])), y1 = max(max(red_b[0][3], red_b[1][3]), max(red_b[2][3], red_b[3][3]));
    const bool boxed = window_fits(x0, y0, x1, y1);
    M.units[static_cast<size_t>(tile % UNIT_SHARDS) * M.shard_cap + s_base + tid] =
        make_uint4(static_cast<unsigned int>(tile), static_cast<unsigned int>(tid * unit_items) | (boxed ? UNIT_TILE_BOX : 0u),
                   boxed ? static_cast<unsigned int>(x0) | (static_cast<unsigned int>(y0) << 16) : 0u,
                   boxed ? static_cast<unsigned int>(x1) | (static_cast<unsigned int>(y1) << 16) : 0u);
  }
}

// the first update after a reference frame (or after somebody looked at the state): no search is pending, every tile is set up by a workgroup of its own
template <int SIDE>
__global__ __launch_bounds__(TILE_PIX) void pipe_setup_kernel(SeqArgs Q, MatcherArgs M, PipeArgs X) {
  const int tid = threadIdx.x;
  const int unit_rounds = pipe_unit_rounds(M, X.target_units);
  const int tile = static_cast<int>(blockIdx.y) * M.tiles_x + static_cast<int>(blockIdx.x);
  if (tile == 0) pipe_housekeeping(M, X, unit_rounds, tid);
  pipe_setup_tile<SIDE, KEYS_PLANE>(Q, M, X, tile, unit_rounds, tid);
}

// search(frame of Qs / Ms) + setup(frame of Qn / Mn).  Persistent, four workgroups per CU like seed_search_compact_kernel, whose unit loop this
// is; the arguments are read by name (the in-place trick of the product kernel is not needed to judge the experiment).
struct PipeKernelArgs { SeqArgs Qs; MatcherArgs Ms; SeqArgs Qn; MatcherArgs Mn; PipeArgs X; };
template <int SIDE>
__global__ __launch_bounds__(TILE_PIX, 4) void pipe_search_setup_kernel(PipeKernelArgs A_by_value) {
  // read where they lie, in the kernel-argument segment (scalar loads at the point of use; see seed_search_compact_kernel)
  (void)A_by_value;
  const PipeKernelArgs& A = *reinterpret_cast<const PipeKernelArgs*>((const char*)__builtin_amdgcn_kernarg_segment_ptr());
  const SeqArgs& Qs = A.Qs; const MatcherArgs& Ms = A.Ms; const SeqArgs& Qn = A.Qn; const MatcherArgs& Mn = A.Mn; const PipeArgs& X = A.X;
  using Smem = FrameSmem<SIDE>;
  constexpr int HALF = SIDE / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& S = *reinterpret_cast<Smem*>(smem_raw);
  const int tid = threadIdx.x;
  const unsigned int wg_id = blockIdx.x, n_wg = gridDim.x;
  const const_u64_ptr counts = (const_u64_ptr)(Ms.shards_cur);
  const unsigned int n_units = unit_count(counts);
  const int unit_rounds_new = pipe_unit_rounds(Mn, X.target_units);
  if (wg_id == n_wg - 1) pipe_housekeeping(Mn, X, unit_rounds_new, tid);
  const int tx = tid & (TILE_W - 1), ty = tid >> 4;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const SeedParams& P = Qs.P;
  // the keys of the tile in LDS go to memory with RETURNING atomics (they have been performed when their values are back), then this
  // workgroup's finished units are added to the tile's count: whoever completes the count owns the tile's next setup
  auto hand_over = [&](int tile, int x0, int y0, unsigned int finished) {
    const unsigned long long key = S.best[tid];
    if (X.tile_units_cur[tile] == 1u) {  // the tile's only unit: the keys go from the LDS straight into the finalisation, no atomics, no round trip
      pipe_setup_tile<SIDE, KEYS_HANDED>(Qn, Mn, X, tile, unit_rounds_new, tid, key);
      return;
    }
    if (key != 0ull) {
      const unsigned long long old = __hip_atomic_fetch_max(&Ms.best[static_cast<size_t>(y0 + ty) * P.stride + x0 + tx], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" : : "v"(old));
    }
    drain_vmem();
    __syncthreads();
    if (tid == 0) {
      const unsigned int before = __hip_atomic_fetch_add(X.tile_done_cur + tile, finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      S.bcast[1] = before + finished == X.tile_units_cur[tile] ? 1u : 0u;
    }
    __syncthreads();
    const bool last = __builtin_amdgcn_readfirstlane(static_cast<int>(S.bcast[1])) != 0;
    __syncthreads();
    if (last) pipe_setup_tile<SIDE, KEYS_COHERENT>(Qn, Mn, X, tile, unit_rounds_new, tid);
  };
  if (wg_id < n_units) {
    const int unit_items = static_cast<int>(*(const __attribute__((address_space(4))) unsigned int*)(X.unit_items_cur));
    unsigned int n_fallback = 0, n_windows = 0, finished = 0;
    int lds_tile = -1, x0 = 0, y0 = 0, total = 0;
    FrameWindow W;
    W.valid = false; W.x0 = W.y0 = 0; W.x1 = W.y1 = -1; W.ws = 1;
    const bool handout = n_units > n_wg;
    const unsigned int cls = wg_id & (UNIT_SHARDS - 1);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) u32x4* const_entry_ptr;
    unsigned int u = wg_id;
    u32x4 e = *(const_entry_ptr)unit_entry(Ms, counts, u);
    while (u < n_units) {
      const int tile = static_cast<int>(e.x);
      const unsigned int fy = e.y;
      const int first = static_cast<int>(fy & ~UNIT_TILE_BOX);
      const bool boxed = (fy & UNIT_TILE_BOX) != 0u;
      if (tile != lds_tile) {
        const unsigned int box0 = e.z, box1 = e.w;
        if (lds_tile >= 0) { hand_over(lds_tile, x0, y0, finished); finished = 0; }
        const int tile_y = tile / Ms.tiles_x, tile_x = tile - tile_y * Ms.tiles_x;
        x0 = tile_x * TILE_W; y0 = tile_y * TILE_H;
        const int x = x0 + tx, y = y0 + ty;
        const int gi = (x < P.w && y < P.h) ? y * P.stride + x : 0;
        const unsigned int packed = Ms.packed[gi];
        const float2 m = Ms.mean[gi], d = Ms.dir[gi];
        const float lf = Ms.lfirst[gi], st = P.sum_templ[gi], dn = P.denom[gi];
        const int ref_col = clampi(x0 - HALF + lane, 0, P.w - 1);
        if (lane < Smem::REF_W) {
          typedef const __attribute__((address_space(1))) float* gptr_t;
          typedef __attribute__((address_space(3))) float* lptr_t;
          for (int ry = wave; ry < Smem::REF_H; ry += 4) {
            const float* row = P.ref + static_cast<size_t>(clampi(y0 - HALF + ry, 0, P.h - 1)) * P.stride;
            __builtin_amdgcn_global_load_lds((gptr_t)(row + ref_col), (lptr_t)(S.ref + ry * Smem::REF_W), 4, 0, 0);
          }
        }
        if (boxed) {
          W.x0 = static_cast<int>(box0 & 0xffffu); W.y0 = static_cast<int>(box0 >> 16);
          W.x1 = static_cast<int>(box1 & 0xffffu); W.y1 = static_cast<int>(box1 >> 16);
          W.ws = (W.x1 - W.x0 + 1) | 1;
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
        if (!boxed) { W.valid = false; W.x0 = W.y0 = 0; W.x1 = W.y1 = -1; W.ws = 1; }
        total = frame_prefix<SIDE>(S, tid);  // barriers inside
        lds_tile = tile;
      }
      frame_search<SIDE>(P, S, tid, first, min(first + unit_items, total), W, n_fallback, n_windows);  // ends with a barrier
      ++finished;
      if (!handout) break;
      if (tid == 0) S.bcast[0] = n_wg + cls + UNIT_SHARDS * atomicAdd(&Ms.handout[cls * HANDOUT_STRIDE], 1u);
      __syncthreads();
      u = static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(S.bcast[0])));
      __syncthreads();
      if (u < n_units) e = *(const_entry_ptr)unit_entry(Ms, counts, u);
    }
    if (lds_tile >= 0) hand_over(lds_tile, x0, y0, finished);
  }
  // tiles without a unit in the lists just searched have no last finisher: they are dealt out by index (most are dead and cost one load)
  for (int t = static_cast<int>(wg_id); t < Ms.n_tiles; t += static_cast<int>(n_wg))
    if (X.tile_units_cur[t] == 0u) pipe_setup_tile<SIDE, KEYS_PLANE>(Qn, Mn, X, t, unit_rounds_new, tid);
}

// per-handle buffers of the experiment (allocated at first use)
struct PipeWorkspace {
  uint4* d_units[2] = {nullptr, nullptr};
  unsigned int* d_handout[2] = {nullptr, nullptr};
  unsigned int* d_unit_items = nullptr;            // two words
  unsigned int* d_tile_units[2] = {nullptr, nullptr};
  unsigned int* d_tile_done[2] = {nullptr, nullptr};
  bool attr_set[4] = {false, false, false, false};
  int allocate(const MatcherWorkspace& ws) {
    const size_t nt = static_cast<size_t>(ws.n_tiles());
    for (int k = 0; k < 2; ++k) {
      if (hipMalloc(reinterpret_cast<void**>(&d_units[k]), static_cast<size_t>(ws.max_units) * sizeof(uint4)) != hipSuccess) return -1;
      if (hipMalloc(reinterpret_cast<void**>(&d_handout[k]), UNIT_SHARDS * HANDOUT_STRIDE * sizeof(unsigned int)) != hipSuccess) return -1;
      if (hipMemset(d_handout[k], 0, UNIT_SHARDS * HANDOUT_STRIDE * sizeof(unsigned int)) != hipSuccess) return -1;
      if (hipMalloc(reinterpret_cast<void**>(&d_tile_units[k]), nt * sizeof(unsigned int)) != hipSuccess) return -1;
      if (hipMalloc(reinterpret_cast<void**>(&d_tile_done[k]), nt * sizeof(unsigned int)) != hipSuccess) return -1;
      if (hipMemset(d_tile_units[k], 0, nt * sizeof(unsigned int)) != hipSuccess) return -1;
      if (hipMemset(d_tile_done[k], 0, nt * sizeof(unsigned int)) != hipSuccess) return -1;
    }
    if (hipMalloc(reinterpret_cast<void**>(&d_unit_items), 2 * sizeof(unsigned int)) != hipSuccess) return -1;
    if (hipMemset(d_unit_items, 0, 2 * sizeof(unsigned int)) != hipSuccess) return -1;
    return 0;
  }
  void release() {
    for (int k = 0; k < 2; ++k) {
      void* all[] = {d_units[k], d_handout[k], d_tile_units[k], d_tile_done[k]};
      for (void* p : all)
        if (p) (void)hipFree(p);
      d_units[k] = nullptr; d_handout[k] = nullptr; d_tile_units[k] = nullptr; d_tile_done[k] = nullptr;
    }
    if (d_unit_items) (void)hipFree(d_unit_items);
    d_unit_items = nullptr;
  }
};

// the workspace as frame `frame` of the sequence sees it: the shard counters of that frame, the double-buffered lists of its parity
inline MatcherArgs pipe_matcher_args(const MatcherWorkspace& ws, const PipeWorkspace& pw, long long frame) {
  MatcherArgs M = matcher_args(ws);
  const int par = static_cast<int>(frame & 1);
  M.units = pw.d_units[par];
  M.handout = pw.d_handout[par];
  M.shards_cur = ws.d_shards + (frame % 3) * UNIT_SHARDS;
  M.shards_prev = frame > 0 ? ws.d_shards + ((frame + 2) % 3) * UNIT_SHARDS : nullptr;
  M.shards_next = ws.d_shards + ((frame + 1) % 3) * UNIT_SHARDS;
  M.conv_out = nullptr;  // the CONVERGED count is not mirrored in this mode (it would be one update late)
  return M;
}
inline PipeArgs pipe_args(const PipeWorkspace& pw, long long frame_new, int target_units) {
  PipeArgs X;
  const int nw = static_cast<int>(frame_new & 1), cur = nw ^ 1;
  X.tile_units_cur = pw.d_tile_units[cur]; X.tile_done_cur = pw.d_tile_done[cur];
  X.tile_units_new = pw.d_tile_units[nw]; X.tile_done_new = pw.d_tile_done[nw];
  X.unit_items_cur = pw.d_unit_items + cur; X.unit_items_new = pw.d_unit_items + nw;
  X.handout_new = pw.d_handout[nw];
  X.target_units = target_units;
  return X;
}

// setup of frame ws.frame alone (nothing to search yet)
template <int SIDE>
inline hipError_t launch_pipe_setup(const SeqArgs& Q, MatcherWorkspace& ws, PipeWorkspace& pw, hipStream_t stream, int num_cus, int target_mult) {
  ++ws.update_number;
  const MatcherArgs M = pipe_matcher_args(ws, pw, ws.frame);
  const PipeArgs X = pipe_args(pw, ws.frame, num_cus * 4 * target_mult);
  hipLaunchKernelGGL((pipe_setup_kernel<SIDE>), dim3(ws.tiles_x, ws.tiles_y), dim3(TILE_PIX), 0, stream, Q, M, X);
  ++ws.frame;
  return hipGetLastError();
}
// search of frame ws.frame - 1 (Qs: that frame's parameters) + setup of frame ws.frame (Qn)
template <int SIDE>
inline hipError_t launch_pipe_search_setup(const SeqArgs& Qs, const SeqArgs& Qn, MatcherWorkspace& ws, PipeWorkspace& pw, hipStream_t stream, int num_cus, int target_mult) {
  using Smem = FrameSmem<SIDE>;
  ++ws.update_number;
  auto kernel = pipe_search_setup_kernel<SIDE>;
  if (!pw.attr_set[SIDE / 2 - 1]) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(sizeof(Smem)));
    if (e != hipSuccess) return e;
    pw.attr_set[SIDE / 2 - 1] = true;
  }
  PipeKernelArgs A;
  A.Qs = Qs; A.Qn = Qn;
  A.Ms = pipe_matcher_args(ws, pw, ws.frame - 1); A.Mn = pipe_matcher_args(ws, pw, ws.frame);
  A.X = pipe_args(pw, ws.frame, num_cus * 4 * target_mult);
  static_assert(sizeof(PipeKernelArgs) <= 4096, "kernel arguments are limited to 4 KB");
  hipLaunchKernelGGL(kernel, dim3(num_cus * 4), dim3(TILE_PIX), sizeof(Smem), stream, A);
  ++ws.frame;
  return hipGetLastError();
}
// the pending search of frame ws.frame - 1 alone (somebody is about to look at the state): the product's search kernel on this mode's lists
template <int SIDE>
inline hipError_t launch_pipe_search_only(const SeqArgs& Qs, MatcherWorkspace& ws, PipeWorkspace& pw, hipStream_t stream, int num_cus) {
  using Smem = FrameSmem<SIDE>;
  auto search = seed_search_compact_kernel<SIDE, 1>;
  bool& attr = ws.attr_set_compact[SIDE / 2 - 1][0];
  if (!attr) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(search), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(sizeof(Smem)));
    if (e != hipSuccess) return e;
    int wg = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg, reinterpret_cast<const void*>(search), TILE_PIX, sizeof(Smem)) != hipSuccess || wg < 1) wg = 1;
    ws.compact_wg_per_cu[SIDE / 2 - 1][0] = wg > 4 ? 4 : wg;
    attr = true;
  }
  MatcherArgs M = pipe_matcher_args(ws, pw, ws.frame - 1);
  // the product kernel reads its unit size from M.queue[5]: this mode keeps that word per parity
  M.queue = pw.d_unit_items + ((ws.frame - 1) & 1) - 5;
  BatchArgs<1> B;
  B.seq[0] = Qs;
  hipLaunchKernelGGL(search, dim3(num_cus * ws.compact_wg_per_cu[SIDE / 2 - 1][0]), dim3(TILE_PIX), sizeof(Smem), stream, B, M);
  return hipGetLastError();
}

}  // namespace rmdk

#endif  // RMD_PIPELINED_HPP
