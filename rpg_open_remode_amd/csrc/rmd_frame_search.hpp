// seed_search_compact_kernel: the second launch of a frame's seed update (rmd_frame.hpp) -- persistent workgroups take work units, stage
// the tile's window of the current image (rmd_frame_window.hpp) and evaluate one NCC per (seed, step) item and lane; arg-max per seed. Who
// writes / reads what:
//   FrameSmem descriptors (mean, dir, l_first, packed, sum_templ, denom, prefix)   written by the tile's 256 lanes when the workgroup
//   changes
//                               tile, published by the barriers inside frame_prefix, read by every round of that tile
//   FrameSmem::best             64-bit LDS atomic max per evaluation; complete for the items searched so far after the barrier that ends
//                               frame_search; handed to MatcherArgs::best with a global atomic max when the workgroup changes tile / exits
//   units, shards_cur           written by the setup kernel (the launch before): read through the scalar path (constant address space)
//   handout                     returning atomics, one per unit taken beyond the static first round
//   conv_out                    pinned host word per sequence, written by the last workgroup
#ifndef RMD_FRAME_SEARCH_HPP
#define RMD_FRAME_SEARCH_HPP

#include "rmd_frame_window.hpp"
#include "rmd_frame_ingest.hpp"

namespace rmdk {

// Rounds of 256 NCC evaluations over work items [k0, k1) of the tile in LDS with window W; arg-max keys accumulate in S.best.
// No barrier.
template <int SIDE>
RMDK_D void frame_rounds(const SeedParams& P, FrameSmem<SIDE>& S, int tid, int k0, int k1, const FrameWindow& W, unsigned int& n_fallback) {
  using Smem = FrameSmem<SIDE>;
  // ncc_at_dyn's in-window test reads the row table at q0, q0 + 1, q0 + SIDE with q0 clamped to [0, rows - 1 - SIDE]: it needs a window of
  // at least SIDE + 1 rows.  Every window cut to in-image samples has SIDE + 3 or more (a sample's rows floor(y) - HALF - 1 .. floor(y) +
  // HALF + 2 lie inside the image: the guard of epipolar_match.cu:91-97 keeps y in [SIDE, h - SIDE)), and clamp_window keeps min(rows, 43);
  // a window that broke that rule is treated as empty -- zero usable columns, every evaluation reads L2 -- instead of trusted (scalar, once
  // per call).
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
      const float ncc = ncc_at_dyn<SIDE>(P, px, S.win, S.row_start, W.ws, W.y0, W.rows, ww_usable, S.ref + pty * Smem::REF_W + ptx,
          Smem::REF_W,
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
        seed_range_band<SIDE>(P, S, tid, my_n > 0 && my_first < k1 && my_first + my_n > k, j0, min(k1 - my_first, my_n) - 1, W.m, W.yref,
            bu0, by0, bu1, by1);
        block_bbox<SIDE>(S, tid, bu0, by0, bu1, by1, 0);
        __syncthreads();
        block_bbox_read<SIDE>(S, u0, y0, u1, y1, 0);
      }
      if (!window_fits(u0, y0, u1, y1)) {
        const int kb = min(k_end, k + FR_UNIT_ITEMS), kc = min(k_end, k + TILE_PIX);
        int bu0, by0, bu1, by1, cu0, cy0, cu1, cy1;
        // a band over the same items as the one that has just failed is not reduced again: the 4-round band when at most four rounds are
        // left (units of the product pipeline: always), the one-round band when at most one is (most units of a light frame: one reduction
        // pass and one barrier instead of two, 1 us of a unit's 2.3 us of window policy)
        const bool four = kb < k_end, one = kc < kb;
        if (four) {
          seed_range_band<SIDE>(P, S, tid, my_n > 0 && my_first < kb && my_first + my_n > k, j0, min(kb - my_first, my_n) - 1, W.m, W.yref,
              bu0, by0, bu1, by1);
          block_bbox<SIDE>(S, tid, bu0, by0, bu1, by1, 4);
        }
        if (one) {
          seed_range_band<SIDE>(P, S, tid, my_n > 0 && my_first < kc && my_first + my_n > k, j0, min(kc - my_first, my_n) - 1, W.m, W.yref,
              cu0, cy0, cu1, cy1);
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

// Who searches which unit.  The setup kernel appends a tile's units to the list of shard (tile % UNIT_SHARDS); the sixteen counts were
// written by that kernel -- the launch before this one -- and nobody writes them while this kernel runs: they are read through the scalar
// path (constant address space) every time they are needed, sixteen words from the scalar cache, instead of being kept in seventeen scalar
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
  // the kernel arguments every workgroup needs first, requested TOGETHER (left alone the compiler fetches ahead_wgs, waits, branches, and
  {
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
  if (NSEQ == 1 && blockIdx.x < static_cast<unsigned int>(M.ahead_wgs)) {
    const unsigned int next = M.ingest_number + 1u;
    if (ld_agent(M.ahead) != next) return;
    const SeedParams& P = Bq[0].P;
    const int part = static_cast<int>(blockIdx.x);
    if (M.ingest_flag == nullptr) ingest_in_place(M.ingest_kind, M.ingest_pitch, Bq[0].next_src, Bq[0].next_dst, P.w, P.h, P.stride, part,
        M.ahead_wgs, tid);
    else ingest_staged(M.ingest_kind, M.ingest_pitch, Bq[0].next_src, Bq[0].next_dst, P.w, P.h, P.stride, part, M.ahead_wgs, tid,
        [](const unsigned int* p) { return *p; });
    // The last one to finish publishes the frame.  Plane and number are read by the NEXT kernels only, and a kernel's stores are all
    // visible to the kernels behind it on the stream: no fence here (an agent-scope fence writes back and invalidates the L2 the
    // searching workgroups live on -- a hundred of them made every update 20 us longer).
    if (tid == 0 && __hip_atomic_fetch_add(M.ahead + 1, 1u, __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_AGENT) == static_cast<unsigned int>(M.ahead_wgs) - 1u) {
      __hip_atomic_store(M.ahead + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(M.ahead + 2, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  const unsigned int wg_id = blockIdx.x - static_cast<unsigned int>(M.ahead_wgs), n_wg = gridDim.x - static_cast<unsigned int>(M.ahead_wgs);
  const const_u64_ptr counts = (const_u64_ptr)(M.shards_cur);
  // On a LIGHT frame -- no shard holds more units than a sixteenth of the grid, which is every frame of the benchmark sequence after its
  // first ~75 -- the units are not numbered through the shards (unit_entry: the counts first, then the entry: two scalar round trips in
  // front of every workgroup's first descriptor loads) but taken where they lie: entry i of shard s belongs to workgroup 16 i + s, whose
  // address needs no count.  The workgroup requests that entry together with the counts and learns from them whether what came back is a
  // unit.  (Consecutive units of a tile go to workgroups 16 apart: the same XCD -- they share the tile's window in its L2 --, CUs two
  // apart.)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(4))) u32x4* const_entry_ptr;
  const unsigned int my_shard = wg_id & (UNIT_SHARDS - 1), my_index = wg_id / UNIT_SHARDS;
  const u32x4 e_here = *(const_entry_ptr)(M.units + static_cast<size_t>(my_shard) * M.shard_cap + min(my_index,
      static_cast<unsigned int>(M.shard_cap) - 1u));
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
        const unsigned long long n = static_cast<unsigned long long>(static_cast<unsigned int>(S.red[0][0] + S.red[1][0] + S.red[2][0]
            + S.red[3][0]));
        __hip_atomic_store(M.conv_out + q, (static_cast<unsigned long long>(M.update_number) << 32) | n, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __syncthreads();
    }
  }
  // no unit for this workgroup: on a light frame most of the grid leaves here, a few dozen scalar instructions in
  if (light ? my_index >= static_cast<unsigned int>(counts[my_shard]) : wg_id >= n_units) return;
  unsigned long long* const trace0 = NSEQ == 1 ? Bq[0].P.trace : nullptr;  // diagnostics (single sequences only)
  unsigned long long* const tr = trace0 && static_cast<int>(wg_id) < M.n_tiles ? trace0 + static_cast<size_t>(wg_id) * FR_TRACE_WORDS
      : nullptr;
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
  // Unit wg_id is ours for free.  When there are more units than workgroups the rest is handed out by sixteen counters (workgroup b draws
  // from counter b % 16, which deals the units n_wg + b % 16 + 16 k: one counter word for a thousand workgroups serialises their returning
  // atomics for 12 us).  A unit is claimed only when the current one is done: a unit claimed earlier is a unit the fastest workgroup cannot
  // take (LAB.md, "early claim").
  const bool handout = n_units > n_wg;
  const unsigned int cls = wg_id & (UNIT_SHARDS - 1);
  // the unit entries were written by the setup kernel, the launch before this one: scalar loads (the address is uniform, the words land in
  // scalar registers, nothing waits on the vector memory counter)
  unsigned int u = light ? 0u : wg_id;  // (light: any number below n_units -- the loop below ends after this workgroup's one unit)
  u32x4 e = e_here;
  if (!light) e = *(const_entry_ptr)unit_entry(M, counts, u);
  while (u < n_units) {
    // the unit: (tile, first item | UNIT_TILE_BOX, the texel box of all samples of the tile) -- uniform over the workgroup: scalar
    // registers
    const int tile = static_cast<int>(e.x & ((1u << UNIT_ROUNDS_SHIFT) - 1u));
    // the unit size travels with the unit: no word of the setup kernel's to fetch first
    const int unit_items = (static_cast<int>(e.x >> UNIT_ROUNDS_SHIFT) + 1) * TILE_PIX;
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
      // LDS-direct like the window (frame_stage_window): a row of the halo per instruction, nothing held in registers
      if (lane < Smem::REF_W) {
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
      // halo and window rows are read by OTHER waves after the barriers below: this wave's LDS-direct transfers must have landed
      drain_vmem();
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
  // diagnostics (COLLECT_STATS = 1): evaluations that read their texels from L2 instead of the LDS window, units, windows staged in the
  // search
  if (NSEQ == 1 && Bq[0].P.stats) {
    const unsigned long long fb = wave_sum_u64(n_fallback);
    if ((tid & 63) == 0 && fb) atomicAdd(&Bq[0].P.stats[3], fb);
    if (tid == 0) { atomicAdd(&Bq[0].P.stats[4], static_cast<unsigned long long>(n_done)); atomicAdd(&Bq[0].P.stats[5],
        static_cast<unsigned long long>(n_windows)); }
  }
  if (tr && tid < 64) {
    const unsigned long long fb = wave_sum_u64(n_fallback);  // the first wave's lanes only: a hint, not a count
    if (tid == 0) {
      tr[3] = wall_clock64();
      tr[4] = n_items; tr[5] = n_done; tr[6] = static_cast<unsigned long long>(lds_tile >= 0 ? lds_tile : 0);
      LAB_PROF(
      tr[6] = S.prof[0];
      // window policy, staging, rounds incl. barrier
      tr[2] = (S.prof[4] & 0xfffffull) | ((S.prof[5] & 0xfffffull) << 20) | ((S.prof[6] & 0xffffffull) << 40);
      )
      tr[7] = fb | (static_cast<unsigned long long>(n_windows) << 32);
    }
  }
}

}  // namespace rmdk

#endif  // RMD_FRAME_SEARCH_HPP
