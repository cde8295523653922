// seed_setup_compact_kernel: the first of the two launches of a frame's seed update (rmd_frame.hpp) -- deferred finalisation of the
// previous frame, seed_check, epipolar segments and in-image runs, the tiles' work units and windows; its leading grid rows bring host
// frames in. Who writes / reads what (device memory):
//   state planes (mu, sigma_sq, a, b, convergence, match)   read and written by the lane that owns the seed, nobody else in this launch
//   MatcherArgs::mean / dir / lfirst / packed / best         written here per live seed, read by the search kernel (next launch, same
//   stream) MatcherArgs::tile_live / tile_conv                       written by tid 0 of the tile's workgroup; read by the same tile one
//   launch later /
//                                                            by the search kernel's last workgroup
//   shards_cur (returning atomics), units                    reserved and written here, read through the scalar path by the search kernel
//   shards_next, handout                                     cleared by the keeper tile for the launches that follow
//   progress[0], ahead[0]                                    pinned / device words for the host and for the search kernel's bringers
#ifndef RMD_FRAME_SETUP_HPP
#define RMD_FRAME_SETUP_HPP

#include "rmd_frame_window.hpp"
#include "rmd_frame_ingest.hpp"

namespace rmdk {

// Setup.  The previous frame's finalisation, when it has been deferred (Q.fuse_prev: nobody looked at the state in between),
// runs here, in the lane that owns the same seed, on the state values this lane has loaded anyway -- the posterior goes
// straight from registers into the new frame's check.  Everything a lane needs from memory is requested in ONE batch up front
// (the compiler does not hoist loads out of the branches that consume them, and five dependent round trips under a burst of a
// million requests were two thirds of this kernel's time).
// The unit list is built without a planning step: a tile reserves its units in the list of its shard (tile % 16) with one
// returning atomic on the shard's counter; the unit size comes from the PREVIOUS frame's total work (the counters of three
// consecutive frames rotate).  The search kernel reads the sixteen counts and walks the shards' lists as one list.
// Words 2, 3 of a unit entry: the tile's window -- column origin (16 bits, signed: a band may start left of the image) | first row << 16;
// rows (8 bits) | texels per row (9 bits) << 8 | shear (15 bits, signed) << 17.  A unit without UNIT_TILE_BOX carries the shear only.
RMDK_D unsigned int unit_pack_origin(int x0, int y0) {
    return (static_cast<unsigned int>(x0) & 0xffffu) | (static_cast<unsigned int>(y0) << 16); }
RMDK_D unsigned int unit_pack_shape(int rows, int ww, int m) {
  return static_cast<unsigned int>(rows) | (static_cast<unsigned int>(ww) << 8) | (static_cast<unsigned int>(m) << 17);
}
static_assert(FR_MAX_ROWS < (1 << 8) && FR_MAX_WIDTH < (1 << 9), "window shape fields of a unit entry");
constexpr int unit_tail_shift(int shard) { return shard >= 14 ? 2 : shard >= 12 ? 1 : 0; }  // see seed_setup_compact_kernel
constexpr int UNIT_ROUNDS_SHIFT = 30;  // word 0 of a unit entry: tile | (rounds per unit - 1) << 30
static_assert(MAX_UNIT_ROUNDS <= 4, "two bits of a unit entry");

// at most this many workgroups (per sequence) bring a staged host frame into the current-image plane (the only ones that may wait):
// ingest_workgroups()
constexpr int INGEST_WGS = 512;
// with lens undistortion (two dependent round trips per pixel): a quarter of the chip's wave slots at most
constexpr int INGEST_WGS_REMAP = 512;
// frames read in place from pinned host memory: enough requests in flight to cover the host link's latency
constexpr int INGEST_WGS_IN_PLACE = 256;

template <int SIDE, int NSEQ>
__global__ __launch_bounds__(TILE_PIX) void seed_setup_compact_kernel(BatchArgs<NSEQ> B, MatcherArgs M, int target_units) {
  __shared__ int red_i[4], red_c[4], red_l[4], red_b[4][6];
  __shared__ unsigned int s_base;
  constexpr int HALF = SIDE / 2;
  const int seq = NSEQ == 1 ? 0 : static_cast<int>(blockIdx.z);
  // one sequence: the named argument, which the compiler fetches with a few wide scalar loads at the top of the kernel; several: the
  // argument segment indexed by the sequence number (see BatchArgs), copied once so that its loads are issued here too and not one
  // by one where the values are used (every one of those is a scalar-cache round trip on this kernel's dependent chain)
  const SeqArgs Q = NSEQ == 1 ? B.seq[0] : seq_table()[seq];
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
  // them.  Only these few workgroups ever wait: if every tile workgroup did, a device filled with
  // waiting waves could keep a copy that is carried out by a blit kernel from ever running.  The wait is bounded; a copy that never
  // arrives is reported through progress[1].  Flag and staged frame are read with agent-scope loads: the copy may finish after this kernel
  // has started.
  const int wg = tile_by * static_cast<int>(gridDim.x) + static_cast<int>(blockIdx.x);
  if (tile_by < 0 || tile_by >= M.tiles_y) {
    // One step ahead: the verdict for the NEXT frame -- has it been handed over (a pinned host word, frames read in place) / arrived in HBM
    // (the arrival flag of its ring slot)? -- falls to ONE extra row of workgroups behind the tile grid: dispatched last, so that it is
    // taken as late as this kernel can take it (the search kernel's bringers act on it).  (Taken by the last tile workgroup instead, the
    // branch cost the tile path its last free scalar registers: the compiler reserved a scratch segment for every wave of the kernel.)
    if (tile_by >= M.tiles_y) {
      if (NSEQ == 1 && M.ahead && blockIdx.x == 0 && tid == 0) {
        const unsigned int next = M.ingest_number + 1u;
        __hip_atomic_store(M.ahead, static_cast<int>(ld_system(M.submitted) - next) >= 0 ? next : 0u, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    const int iw = static_cast<int>(blockIdx.y * gridDim.x + blockIdx.x);
    if (iw >= M.ingest_wgs) return;
    // step numbers are compared modulo 2^32 (a live system never stops counting): "behind" = the signed difference is negative.
    // The flag and the staged frame are read with RELAXED agent-scope loads and NO acquire fence: on this part an agent-scope acquire is a
    // cache invalidation (buffer_inv sc1: the vector L1 AND the lines of this XCD's L2 that other agents may have written), issued by every
    // wave that executes it, thrown at the L2 the tile workgroups are loading their seeds' state through.  What an acquire would be there
    // for holds without it: the frame's loads are agent-scope loads
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
    // diagnostics: this frame was not converted one step ahead; did it keep the kernel waiting?
    if (M.ingest_profile && iw == 0 && tid == 0 && seq == M.housekeeper) {
      __hip_atomic_fetch_add(M.progress + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (spins) {
        __hip_atomic_fetch_add(M.progress + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_fetch_add(M.progress + 4, spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    asm volatile("" ::: "memory");  // nothing below is read before the flag has been seen
    // cv::remap through the undistortion maps first (depthmap.cpp:99), destination pixels four at a time per lane
    if (M.ingest_kind == 1 && Q.ingest_map1) {
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
          Q.ingest_dst[static_cast<size_t>(row) * P.stride + (d - row * P.w)] = remap_u8_pixel(src, M.ingest_pitch, m[q], f[q], P.w, P.h,
              [](const unsigned char* p) {
            return static_cast<int>(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          });
        }
      }
    } else if (in_place) {
      ingest_in_place(M.ingest_kind, M.ingest_pitch, M.ingest_kind == 1 ? static_cast<const void*>(Q.ingest_u8)
          : static_cast<const void*>(Q.ingest_f32),
                      Q.ingest_dst, P.w, P.h, P.stride, iw, M.ingest_wgs, tid);
    } else {  // the staged frame: x (1/255) for 8-bit frames (store_u8x4), floats as they are
      ingest_staged(M.ingest_kind, M.ingest_pitch, M.ingest_kind == 1 ? static_cast<const void*>(Q.ingest_u8)
          : static_cast<const void*>(Q.ingest_f32), Q.ingest_dst,
                    P.w, P.h, P.stride, iw, M.ingest_wgs, tid, [](const unsigned int* p) { return ld_agent(p); });
    }
    return;
  }
  const int tile = tile_by * M.tiles_x + static_cast<int>(blockIdx.x);      // within the sequence
  const int tile_g = (NSEQ == 1 ? 0 : seq * M.n_tiles) + tile;               // within the launch
  // A tile in which the previous frame's check left no seed in state UPDATE is DEAD until the next reference frame: BORDER / CONVERGED /
  // DIVERGED are absorbing (nothing but the finalisation of an UPDATE seed ever changes sigma_sq, a, b), so every plane, the tile's
  // CONVERGED count and its empty descriptors already hold what this launch would write.  Its workgroup leaves after one scalar load (on
  // the ~170 light frames of the benchmark sequence: four tiles in five). (tile_live: written at the end of this kernel by the tile's own
  // workgroup, read here one launch later; fuse_prev = nobody touched the planes in between.  The launch's housekeeping below must not
  // depend on tile 0 being alive.) (the tile's word: seeds in state UPDATE after the previous frame's check | TILE_WANTS_BAND, see below)
  const unsigned int tile_word = *(const __attribute__((address_space(4))) unsigned int*)(M.tile_live + tile_g);
  const bool dead_tile = Q.fuse_prev && (tile_word & 0xffffu) == 0u;
  // A tile whose samples did not fit a BOX-shaped window one frame ago (a bundle of long diagonal segments) gets a sheared band this frame;
  // everybody else -- nearly every tile of nearly every frame -- pays nothing for the machinery: no slope, no second pair of reductions.
  // (The first frame on which a tile's box does not fit goes to the search kernel's own window policy.)
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
  if (M.progress && wg == 0 && seq == M.housekeeper && tid == 0) __hip_atomic_store(M.progress, M.ingest_number, __ATOMIC_RELAXED,
      __HIP_MEMORY_SCOPE_SYSTEM);
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
    const unsigned long long per_round = static_cast<unsigned long long>(target_units) * TILE_PIX;
    static_assert(MAX_UNIT_ROUNDS == 4, "the ladder below is ceil(items / per_round) clamped to 1..4");
    unit_rounds = items > 3 * per_round ? 4 : items > 2 * per_round ? 3 : items > per_round ? 2 : 1;  // no 64-bit division
  }
  // Graduated unit sizes (one sequence per launch only).  The search kernel's hand-out deals the shards' lists one after the other, so the
  // units of the last shards are the last to be searched, and while the last unit runs every other workgroup waits: up to one unit's time,
  // a fifth of the kernel on the frames that have two to five units per workgroup.  The tiles of shards 12, 13 cut their work into units of
  // half the frame's size, those of shards 14, 15 into quarters (at least one round): 12 % more units, the wait at the end a quarter as
  // long.  In a batch the other stream groups' kernels fill that wait already and the smaller units only cost their staging: there all
  // units of a frame have one size.
  if (NSEQ == 1) unit_rounds = max(1, unit_rounds >> unit_tail_shift(tile_g % UNIT_SHARDS));
  // what the planes hold now (known only when the previous frame's values were loaded for its finalisation): a seed that has
  // converged or diverged keeps writing the same state and an empty descriptor, a third of this kernel's stores -- skipped
  int conv_old = -1;
  unsigned int packed_old = 0xffffffffu;
  if (Q.fuse_prev) {  // uniform over the workgroup
    // Everything requested above is in flight together: left alone the compiler sinks each load into the branch that consumes it (state
    // -> key -> descriptor: three dependent memory round trips on every live lane's chain).  The empty statement below "reads" every
    // value, so the loads stay where they were issued and are waited for once.
    asm volatile("" : "+v"(conv_prev), "+v"(key), "+v"(packed_prev), "+v"(lfirst_prev), "+v"(m_prev.x), "+v"(m_prev.y), "+v"(d_prev.x),
        "+v"(d_prev.y));
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
  // seed's depth estimate is, so the direction needs no state, no reduction and no hint from the previous frame: every lane computes the
  // same number from the kernel arguments (here, where the pose is in scalar registers anyway: the segments below use it).  v - e scaled by
  // t_z: (fx (t_z X / Z - t_x), fy (t_z Y / Z - t_y)) -- finite for a sideways motion (t_z = 0, epipole at infinity).  Approximate
  // reciprocals: the slope steers window shapes, never results.
  int m_tile = 0;
  if (want_band) {  // (uniform over the workgroup)
    const Pose& T = P.T_curr_ref;
    const float fcx = (static_cast<float>(blockIdx.x * TILE_W + TILE_W / 2) - P.cam.cx) * __builtin_amdgcn_rcpf(P.cam.fx);
    const float fcy = (static_cast<float>(tile_by * TILE_H + TILE_H / 2) - P.cam.cy) * __builtin_amdgcn_rcpf(P.cam.fy);
    const float X = T.d[0] * fcx + T.d[1] * fcy + T.d[2], Y = T.d[4] * fcx + T.d[5] * fcy + T.d[6],
        Z = T.d[8] * fcx + T.d[9] * fcy + T.d[10];
    const float iz = __builtin_amdgcn_rcpf(Z);
    const float dx = P.cam.fx * (T.d[11] * X * iz - T.d[3]), dy = P.cam.fy * (T.d[11] * Y * iz - T.d[7]);
    // flatter than 8 columns per row: a box holds such a bundle (0); a point behind the camera or a degenerate pose: a box as well
    if (Z > 0.0f && fabsf(dx) < 7.9f * fabsf(dy)) m_tile = static_cast<int>(rintf(dx * __builtin_amdgcn_rcpf(dy) * static_cast<float>(1
        << FR_SHEAR_BITS)));
  }
  int state = ST_BORDER;
  if (in_image) {
    // A seed that the PREVIOUS frame's check (the same lane, one launch ago: fuse_prev says nobody touched the planes in between) found
    // BORDER / CONVERGED / DIVERGED keeps sigma_sq, a, b -- the finalisation only ever touches UPDATE seeds -- so the check, a pure
    // function of those and of the pixel position, gives the same answer again: it is not re-evaluated (two IEEE divisions per lane; on the
    // ~170 light frames of the sequence four waves in five consist of such seeds only).
    const bool settled = conv_old == ST_BORDER || conv_old == ST_CONVERGED || conv_old == ST_DIVERGED;
    state = settled ? conv_old : seed_check(P, x, y, sigma_sq, a, b, SIDE);
    if (state != conv_old) P.conv[gi] = state;  // final for BORDER / CONVERGED / DIVERGED; UPDATE seeds are settled by the finalisation
  }
  if (P.trace) t_loaded = wall_clock64();
  int n_valid = 0, i_first = 0;
  unsigned int n_steps = 0, n_evals = 0;
  int bx0 = BAND_NONE_LO, by0 = BAND_NONE_LO, bx1 = BAND_NONE_HI, by1 = BAND_NONE_HI;  // texel box of this seed's samples ...
  // ... and their band under the tile's shear (band_add_point)
  int bu0 = BAND_NONE_LO, bu1 = BAND_NONE_HI;
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
    tx0 = min(min(red_b[0][0], red_b[1][0]), min(red_b[2][0], red_b[3][0])); ty0 = min(min(red_b[0][1], red_b[1][1]), min(red_b[2][1],
        red_b[3][1]));
    tx1 = max(max(red_b[0][2], red_b[1][2]), max(red_b[2][2], red_b[3][2])); ty1 = max(max(red_b[0][3], red_b[1][3]), max(red_b[2][3],
        red_b[3][3]));
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
    const unsigned long long old = atomicAdd(&M.shards_cur[tile_g % UNIT_SHARDS],
        (static_cast<unsigned long long>(total) << 32) | static_cast<unsigned long long>(n_u));
    s_base = static_cast<unsigned int>(old);  // units reserved so far in this shard
  }
  __syncthreads();
  if (tid < n_u) {
    const int x0 = tx0, y0 = ty0, x1 = tx1, y1 = ty1;
    const int u0 = min(min(red_b[0][4], red_b[1][4]), min(red_b[2][4], red_b[3][4])), u1 = max(max(red_b[0][5], red_b[1][5]),
        max(red_b[2][5], red_b[3][5]));
    // the window that holds ALL samples of the tile, if there is one: the box when that fits, else the band along the tile's shear; else
    // the search kernel cuts windows to each unit's own samples (along the same shear, which travels with the unit either way)
    const bool band = !box && want_band && m_tile != 0 && window_fits(u0, y0, u1, y1);
    const int wx0 = box ? x0 : u0, ww = (box ? x1 : u1) - wx0 + 1, rows = y1 - y0 + 1, m = box ? 0 : m_tile;
    // (leaving a LARGE tile window that many one-round units share to the units -- each cuts its own -- gains nothing: LAB.md)
    const bool whole = box || band;
    M.units[static_cast<size_t>(tile_g % UNIT_SHARDS) * M.shard_cap + s_base + tid] =
        make_uint4(static_cast<unsigned int>(tile_g) | (static_cast<unsigned int>(unit_rounds - 1) << UNIT_ROUNDS_SHIFT),
            static_cast<unsigned int>(tid * unit_items) | (whole ? UNIT_TILE_BOX : 0u),
                   whole ? unit_pack_origin(wx0, y0) : 0u, unit_pack_shape(whole ? rows : 0, whole ? ww : 0, m));
  }
}

}  // namespace rmdk

#endif  // RMD_FRAME_SETUP_HPP
