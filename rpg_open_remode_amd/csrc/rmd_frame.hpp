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
//
// Files: rmd_frame_window.hpp (the LDS window and one NCC evaluation at it), rmd_frame_ingest.hpp (host frames, device side),
// rmd_frame_setup.hpp, rmd_frame_search.hpp (the two kernels); here: the stand-alone finalisation and the launch of the pair.
#ifndef RMD_FRAME_HPP
#define RMD_FRAME_HPP

#include "rmd_frame_setup.hpp"
#include "rmd_frame_search.hpp"

namespace rmdk {

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
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg, reinterpret_cast<const void*>(search), TILE_PIX,
        sizeof(Smem)) != hipSuccess || wg < 1) wg = 1;
    ws.compact_wg_per_cu[SIDE / 2 - 1][KIND] = wg > 4 ? 4 : wg;
    attr = true;
  }
  const int resident = num_cus * ws.compact_wg_per_cu[SIDE / 2 - 1][KIND];
  dim3 tiles(ws.tiles_x, ws.tiles_y, n_seq);
  const SeedParams& P0 = B.seq[0].P;
  if (M.ingest_kind) {  // the ingest workgroups: rows in front of the tile grid
    bool remap = false;
    for (int q = 0; q < n_seq; ++q) remap = remap || (M.ingest_kind == 1 && B.seq[q].ingest_map1);
    const long long dwords = M.ingest_kind == 1 && !remap ? static_cast<long long>(M.ingest_pitch >> 2) * P0.h
        : static_cast<long long>(P0.w) * P0.h;
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
