// librmd_hip.so -- the one translation unit that instantiates the seed kernels (seed_init.cu, seed_check.cu, epipolar_match.cu,
// seed_update.cu of the reference as seed_init_kernel + the two-launch pipeline of rmd_frame.hpp): launches for one SeedMatrix and for the
// stream groups of a batch.
#include "rmd_host.hpp"
#include "rmd_frame.hpp"

namespace rmdh {

namespace {
template <typename F>
int dispatch_side(int side, F&& f) {
  switch (side) {
    case 3: return f(std::integral_constant<int, 3>());
    case 5: return f(std::integral_constant<int, 5>());
    case 7: return f(std::integral_constant<int, 7>());
    case 9: return f(std::integral_constant<int, 9>());
    default: return fail(RMD_HIP_ERR_INVALID_ARG, "unsupported patch side %d (3, 5, 7, 9)", side);
  }
}

// this handle's block of a launch of the update pipeline: the frame's parameters, the pending finalisation of its previous frame
rmdk::SeqArgs seq_args_of(const rmd_hip_seeds* s, const rmdk::SeedParams& P) {
  rmdk::SeqArgs Q;
  memset(&Q, 0, sizeof(Q));
  Q.P = P;
  Q.fuse_prev = s->finalize_pending ? 1 : 0;
  Q.T_ref_curr_prev = s->finalize_pending ? s->P_pending.T_ref_curr : P.T_ref_curr;
  Q.active = 1;
  return Q;
}
}  // namespace


// the deferred finalisation of this handle's last update, as a kernel of its own (an observer is about to look at the state)
int seeds_flush(rmd_hip_seeds* s) {
  if (s->finalize_pending) {
    s->finalize_pending = false;
    HIP_TRY(rmdk::launch_seed_finalize(s->P_pending, *s->mws, s->stream, s->seq));
  }
  return RMD_HIP_OK;
}

int seeds_launch_init(rmd_hip_seeds* s) {
  const dim3 block(64, 4), grid((s->width + 63) / 64, (s->height + 3) / 4);
  ScopedStage st(s->opt_timing ? &s->timers[RMD_HIP_STAGE_SEED_INIT] : nullptr, s->stream);
  return dispatch_side(s->patch_side, [&](auto side) {
    hipLaunchKernelGGL((rmdk::seed_init_kernel<decltype(side)::value>), grid, block, 0, s->stream, s->P);
    HIP_TRY(hipGetLastError());
    return RMD_HIP_OK;
  });
}

int seeds_launch_update(rmd_hip_seeds* s, const PendingIngest* ingest) {
  if (s->batch) return fail(RMD_HIP_ERR_INVALID_ARG,
      "update: this SeedMatrix is a member of a batch; its updates are issued with rmd_hip_batch_update*");
  rmdk::SeedParams P = s->P;
  P.stats = nullptr;
  P.trace = nullptr;
  if (s->opt_stats == 1) {
    HIP_TRY(hipMemsetAsync(s->d_scalars + 1, 0, 16 * sizeof(unsigned long long), s->stream));
    P.stats = s->d_scalars + 1;
  }
  int rc;
  if (s->opt_timing == 2) ++s->region_updates;
  s->async_count_valid = false;
  {
    ScopedStage st(s->opt_timing == 1 ? &s->timers[RMD_HIP_STAGE_UPDATE] : nullptr, s->stream);
    rc = dispatch_side(s->patch_side, [&](auto side) {
      constexpr int SIDE = decltype(side)::value;
      if (s->opt_matcher == 0) {
        TRY(seeds_flush(s));
        const dim3 block(64, 4), grid((s->width + 63) / 64, (s->height + 3) / 4);
        hipLaunchKernelGGL((rmdk::seed_update_pixel_kernel<SIDE>), grid, block, 0, s->stream, P);
      } else {
        rmdk::SeedParams Pt = P;
        if (s->opt_stats == 2 && s->matcher_ws.d_wg_trace) {  // timeline probes of the setup tiles and the search workgroups
          Pt.trace = s->matcher_ws.d_wg_trace + static_cast<size_t>(s->trace_frame % rmdk::FR_TRACE_FRAMES)
              * s->matcher_ws.wg_trace_slice_u64();
          ++s->trace_frame;
        }
        rmdk::BatchArgs<1> B;
        B.seq[0] = seq_args_of(s, Pt);
        if (ingest) {
          B.seq[0].ingest_u8 = ingest->u8; B.seq[0].ingest_f32 = ingest->f32;
          B.seq[0].ingest_dst = const_cast<float*>(P.cur);
          B.seq[0].ingest_map1 = ingest->u8 && !ingest->no_remap ? s->d_undist_map1 : nullptr;  // null without lens undistortion
          B.seq[0].ingest_map2 = ingest->u8 && !ingest->no_remap ? s->d_undist_map2 : nullptr;
          B.seq[0].next_src = ingest->next_src; B.seq[0].next_dst = ingest->next_dst;
        }
        HIP_TRY((rmdk::launch_seed_pipeline_compact<SIDE, 1>(B, 1, s->matcher_ws, s->stream, s->num_cus, s->opt_unit_target,
            ingest ? &ingest->common : nullptr)));
        s->P_pending = P;
        s->P_pending.stats = nullptr;
        s->P_pending.trace = nullptr;
        s->finalize_pending = true;
        s->async_count_valid = true;  // the search kernel mirrors this frame's CONVERGED count to pinned memory
        s->async_number = s->matcher_ws.update_number;
        if (!s->opt_lazy || s->opt_stats == 1) TRY(seeds_flush(s));
      }
      HIP_TRY(hipGetLastError());
      return RMD_HIP_OK;
    });
  }
  TRY(rc);
  if (s->opt_stats == 1) {
    HIP_TRY(hipMemcpyAsync(s->h_scalars + 1, s->d_scalars + 1, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                           s->stream));
    s->stats_pending = true;
  }
  return RMD_HIP_OK;
}

// one step: the update pipeline for every member whose bit is set in `active` (their frames are in place: P.cur / the staged host
// frames), one launch pair per group
int batch_launch(rmd_hip_batch* b, unsigned int active, const rmdk::IngestArgs* ingest, const unsigned char* d_stage, size_t frame_bytes) {
  if (b->opt_timing == 2) ++b->region_updates;
  for (int g = 0; g < b->n_groups; ++g) {
    rmd_hip_batch::Group& G = b->groups[g];
    const unsigned int g_active = (active >> G.first) & ((1u << G.n) - 1u);
    if (!g_active) continue;
    rmdk::BatchArgs<rmdk::MAX_GROUP_SEQ> B;
    memset(&B, 0, sizeof(B));
    for (int j = 0; j < G.n; ++j) {
      rmd_hip_seeds* m = b->members[G.first + j];
      rmdk::SeedParams P = m->P;
      P.stats = nullptr; P.trace = nullptr;
      B.seq[j] = seq_args_of(m, P);
      B.seq[j].active = (g_active >> j) & 1u;
      if (!B.seq[j].active) B.seq[j].fuse_prev = 0;
      if (ingest && B.seq[j].active) {
        const unsigned char* src = d_stage + static_cast<size_t>(G.first + j) * frame_bytes;
        if (ingest->kind == 1) {
          B.seq[j].ingest_u8 = reinterpret_cast<const unsigned int*>(src);
          B.seq[j].ingest_map1 = ingest->no_remap ? nullptr : m->d_undist_map1;
          B.seq[j].ingest_map2 = m->d_undist_map2;
        } else {
          B.seq[j].ingest_f32 = reinterpret_cast<const float*>(src);
        }
        B.seq[j].ingest_dst = const_cast<float*>(P.cur);
      }
    }
    rmdk::IngestArgs in;
    if (ingest) {
      in = *ingest;
      void* dev_progress = nullptr;
      HIP_TRY(hipHostGetDevicePointer(&dev_progress, G.h_progress, 0));
      in.progress = static_cast<unsigned int*>(dev_progress);
    }
    const int rc = dispatch_side(b->members[0]->patch_side, [&](auto side) {
      constexpr int SIDE = decltype(side)::value;
      if (G.n == 1) {
        rmdk::BatchArgs<1> B1;
        B1.seq[0] = B.seq[0];
        HIP_TRY((rmdk::launch_seed_pipeline_compact<SIDE, 1>(B1, 1, G.ws, G.stream, b->num_cus, b->opt_unit_target,
            ingest ? &in : nullptr)));
      } else {
        HIP_TRY((rmdk::launch_seed_pipeline_compact<SIDE, rmdk::MAX_GROUP_SEQ>(B, G.n, G.ws, G.stream, b->num_cus, b->opt_unit_target,
            ingest ? &in : nullptr)));
      }
      return RMD_HIP_OK;
    });
    TRY(rc);
    for (int j = 0; j < G.n; ++j) {
      if (!((g_active >> j) & 1u)) continue;
      rmd_hip_seeds* m = b->members[G.first + j];
      m->P_pending = m->P;
      m->P_pending.stats = nullptr; m->P_pending.trace = nullptr;
      m->finalize_pending = true;
      m->async_count_valid = true;
      m->async_number = G.ws.update_number;
    }
  }
  return RMD_HIP_OK;
}

}  // namespace rmdh
