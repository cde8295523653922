// librmd_hip.so -- rmd_hip_batch_*: several SeedMatrix objects of one size stepped by ONE launch pair per stream group, and
// TV-L1 for all of them in one launch sequence.
#include "rmd_host.hpp"
#include "rmd_copy_pool.hpp"
#include "rmd_engines.hpp"

using namespace rmdh;

// ---- batches of SeedMatrix objects ----------------------------------------------------------
// BASELINE configs[3] / SURVEY 8(e): independent sequences.  One MI355X is far from full with one 640x480 sequence (a frame is ~2
// rounds of work per workgroup and a third of it is a latency chain), so up to MAX_BATCH sequences of one size are stepped TOGETHER:
// one setup launch + one search launch per step for all of them, unit lists and the persistent search workgroups shared.
namespace rmdh {

int batch_bind_device(const rmd_hip_batch* b) {
  int cur = -1;
  HIP_TRY(hipGetDevice(&cur));
  if (cur != b->device) HIP_TRY(hipSetDevice(b->device));
  return RMD_HIP_OK;
}

}  // namespace rmdh

namespace {

// frames in host memory (8-bit gray if `gray`, else float): the protocol of ingest_current_fused with ONE sequence number per step
int batch_update_host(rmd_hip_batch* b, const unsigned char* const* gray, const float* const* f32, const float* T_curr_world) {
  TRY(batch_bind_device(b));
  const rmd_hip_seeds* m0 = b->members[0];
  const int u8_pitch = (m0->width + 3) / 4 * 4;
  const size_t bytes_u8 = static_cast<size_t>(u8_pitch) * m0->height;
  size_t frame_bytes = gray ? bytes_u8 : static_cast<size_t>(m0->width) * m0->height * sizeof(float);
  unsigned int active = 0;
  for (int i = 0; i < b->n; ++i) {
    if (!(gray ? static_cast<const void*>(gray[i]) : static_cast<const void*>(f32[i]))) continue;
    if (!b->members[i]->has_reference) return fail(RMD_HIP_ERR_NOT_READY, "batch update: member %d has no reference image", i);
    active |= 1u << i;
  }
  if (!active) return RMD_HIP_OK;
  const double t_a = b->ingest_profile ? host_now_us() : 0.0;
  const unsigned long long n64 = ++b->step_number;
  const unsigned int n = static_cast<unsigned int>(n64);
  // Staged on a copy engine addressed directly (b->engines): the pinned block of a step is free as soon as the engine has read it, and what
  // the kernels read is a staging buffer in HBM -- of which there are SLOTS_MAX, so that the caller may hand steps over that far ahead of
  // the slowest stream group (the groups drift apart; with one ring for both, the fastest group ran into the newest step's frames while
  // the host was held back by the slowest: 70-97 % of its setup kernels waited for their copy).  kp: pinned slot, k: staging slot.
  bool any_maps = false;
  int n_active = 0;
  for (int i = 0; i < b->n; ++i) {
    if (!((active >> i) & 1u)) continue;
    ++n_active;
    any_maps = any_maps || (gray && b->members[i]->d_undist_map1);
  }
  // In place or staged?  What RMD_HIP_TUNE_HOST_FRAMES says; left alone (-1): staged on the engine while a step is at most 5 MB (sixteen
  // 640x480 frames; Mpix/s staged against in place: 2 frames 12 140-12 200 / 11 040-11 070, 4: 14 890-15 080 / 13 370-13 440, 8: 16 220-
  // 16 630 / 15 510-15 540, 12: 16 400 / 14 770-16 080, 16: 15 920-16 290 / 15 610-16 150), else in place.  Frames that go through the
  // lens-undistortion maps are always staged (the remap gathers single bytes).
  const bool in_place = [&] {
    if (any_maps) return false;
    const int forced = tunables().v[RMD_HIP_TUNE_HOST_FRAMES];
    if (forced != HOST_FRAMES_DEFAULT) return forced == HOST_FRAMES_INPLACE || forced == HOST_FRAMES_INPLACE_AHEAD;
    return !(b->engines && static_cast<size_t>(n_active) * (gray ? bytes_u8 : frame_bytes) <= (size_t(5) << 20));
  }();
  const bool deep = b->engines != nullptr && !in_place;
  const int kp = static_cast<int>(n64 % static_cast<unsigned long long>(b->slots));
  const int k = deep ? static_cast<int>(n64 % static_cast<unsigned long long>(rmd_hip_batch::SLOTS_MAX)) : kp;
  // The pinned block kp: an engine may still be reading it (its last step was staged on the engine), or kernels (it was read in place).
  if (b->engines_used && (!b->engines_used->wait_idle(b->sig_frame[kp], 2e6) || !b->engines_used->wait_idle(b->sig_flag[kp], 2e6)))
    return fail(RMD_HIP_ERR_RUNTIME, "a copy engine has not completed pinned slot %d's previous step after 2 s", kp);
  if (deep && b->pinned_in_place[kp]) {  // (a change of mode: the only time a staged step waits for kernels on behalf of its PINNED block)
    const unsigned long long used = b->pinned_in_place[kp];
    for (int g = 0; g < b->n_groups; ++g) {
      rmd_hip_batch::Group& G = b->groups[g];
      if (G.last_step > used) TRY(wait_for_progress(G.h_progress, static_cast<unsigned int>(used) + 1u, G.stream));
      else HIP_TRY(hipStreamSynchronize(G.stream));
    }
  }
  b->pinned_in_place[kp] = in_place ? n64 : 0ull;
  // Slot k was last read by the setup kernels of the step recorded in slot_step[k]; such a kernel is done once a LATER setup kernel of
  // the same group has started (the progress word), or, if the group has not been launched since, once its stream is idle.
  static_assert(rmd_hip_batch::SLOTS_MAX <= 8, "Group::slot_step");
  for (int g = 0; g < b->n_groups; ++g) {
    rmd_hip_batch::Group& G = b->groups[g];
    const unsigned long long used = G.slot_step[k];
    if (!used) continue;
    if (G.last_step > used) TRY(wait_for_progress(G.h_progress, static_cast<unsigned int>(used) + 1u, G.stream));
    else HIP_TRY(hipStreamSynchronize(G.stream));
  }
  const double t_b = b->ingest_profile ? host_now_us() : 0.0;
  if (b->ingest_profile) {
    const int lead = static_cast<int>(n - *static_cast<volatile unsigned int*>(b->groups[0].h_progress));
    ++b->ingest_lead[lead < 0 ? 0 : lead > 4 ? 4 : lead];
  }
  // float frames: the larger kind
  const size_t need = static_cast<size_t>(b->n) * static_cast<size_t>(m0->width) * m0->height * sizeof(float);
  if (b->stage_bytes < need) {
    for (int g = 0; g < b->n_groups; ++g) HIP_TRY(hipStreamSynchronize(b->groups[g].stream));
    HIP_TRY(hipStreamSynchronize(b->copy_stream));
    for (int q = 0; b->engines_used && q < rmd_hip_batch::SLOTS_MAX; ++q)
      if (!b->engines_used->wait_idle(b->sig_frame[q], 2e6) || !b->engines_used->wait_idle(b->sig_flag[q], 2e6))
        return fail(RMD_HIP_ERR_RUNTIME, "a copy engine has not completed slot %d's step after 2 s", q);
    for (int q = 0; q < rmd_hip_batch::SLOTS_MAX; ++q) {
      if (b->h_stage[q]) (void)hipHostFree(b->h_stage[q]);
      if (b->d_stage[q]) (void)hipFree(b->d_stage[q]);
      b->h_stage[q] = nullptr; b->d_stage[q] = nullptr;
      if (q < b->slots) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&b->h_stage[q]), need + 16, hipHostMallocDefault));
      if (q < b->slots || b->engines_used) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_stage[q]), need));
    }
    b->stage_bytes = need;
  }
  // float frames of 8-bit levels travel as bytes (pack_float_rows_u8), if every frame of the step qualifies
  bool packed = false;
  if (!gray && float_frames_as_bytes() && b->pack_backoff > 0) --b->pack_backoff;
  else if (!gray && float_frames_as_bytes()) {
    packed = true;
    for (int i = 0; i < b->n && packed; ++i)
      if ((active >> i) & 1u) packed = CopyPool::instance().pack(f32[i], b->h_stage[kp] + static_cast<size_t>(i) * bytes_u8, m0->width,
          m0->height, u8_pitch);
    if (packed) frame_bytes = bytes_u8;
    else b->pack_backoff = 15;
  }
  const bool as_u8 = gray != nullptr || packed;
  int first = -1, last = -1, n_segs = 0;
  CopyPool::Segment segs[rmdk::MAX_BATCH];
  for (int i = 0; i < b->n; ++i) {
    if (!((active >> i) & 1u)) continue;
    if (first < 0) first = i;
    last = i;
    if (packed) continue;
    unsigned char* dst = b->h_stage[kp] + static_cast<size_t>(i) * frame_bytes;
    if (gray && u8_pitch != m0->width) {
      for (int y = 0; y < m0->height; ++y) memcpy(dst + static_cast<size_t>(y) * u8_pitch, gray[i] + static_cast<size_t>(y) * m0->width,
          m0->width);
    } else {
      segs[n_segs].dst = dst;
      segs[n_segs].src = gray ? static_cast<const void*>(gray[i]) : static_cast<const void*>(f32[i]);
      ++n_segs;
    }
  }
  // The frames of the step go into the pinned block, spread over the copy threads.
  if (n_segs) CopyPool::instance().copy_many(segs, n_segs, frame_bytes);
  const double t_c = b->ingest_profile ? host_now_us() : 0.0;
  const unsigned char* frames_dev = b->d_stage[k];
  rmdk::IngestArgs in;
  if (in_place) {  // the setup kernels read the pinned block themselves
    void* dev = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dev, b->h_stage[kp], 0));
    frames_dev = static_cast<const unsigned char*>(dev);
    in.flag = nullptr;
  } else {
    in.flag = b->d_flag;
  }
  in.kind = as_u8 ? 1 : 2;
  in.pitch = u8_pitch;
  in.number = n;
  in.no_remap = packed;
  in.profile = b->ingest_profile;
  for (int i = 0; i < b->n; ++i) {
    if (!((active >> i) & 1u)) continue;
    rmd_hip_seeds* m = b->members[i];
    // setup k writes it after search k - 1 has run (same stream)
    m->P.cur = static_cast<const float*>(m->planes[RMD_HIP_PLANE_CURR_IMG].data);
    m->P.cur_stride = m->P.stride;
    seeds_frame_pose(m, T_curr_world + 12 * i);
  }
  // The arrival flag of a batch always travels as the whole block, i.e. through the copy engine: the 4-byte form is a blit KERNEL, and a
  // step's ingest workgroups -- up to 75 per member, dispatched ahead of the tiles -- may all be waiting for it while the other groups'
  // persistent search workgroups hold the rest of the wave slots (batch of 16, staged: the blit never ran, the kernels' bounded wait
  // expired after 0.13 s).
  const size_t fw = FLAG_WORDS;
  for (int g = 0; g < b->n_groups; ++g) {
    rmd_hip_batch::Group& G = b->groups[g];
    if ((active >> G.first) & ((1u << G.n) - 1u)) { G.slot_step[k] = n64; G.last_step = n64; }
  }
  // A staged step's copy is handed to the engine BEFORE its kernels are launched, never after: a setup kernel that is running waits
  // (bounded: 0.13 s) for the arrival flag, and a host thread between "launched" and "copy submitted" can be held up for as long as ANOTHER
  // thread of the process sits in a runtime call that waits for the device to drain (hipFree, hipMalloc ...) -- which the spinning kernel
  // keeps from draining.  (Until the end of round 6 the launches were queued while the helper threads copied, to save the host 25 us per
  // step: four handles on four threads, one of them a batch, timed out in every second run.)
  if (!in_place) {
    const size_t off = static_cast<size_t>(first) * frame_bytes, len = static_cast<size_t>(last - first + 1) * frame_bytes;
    bool sent = false;
    if (deep) {
      // ONE engine for the batch, block and flag in its order: the batch has one flag word, and the steps' numbers must reach it in order
      b->h_seq[kp * FLAG_SLOT_WORDS] = n;
      sent = b->engines->submit(0u, 0u, b->d_stage[k] + off, b->h_stage[kp] + off, len, b->sig_frame[kp], b->d_flag,
                                b->h_seq + kp * FLAG_SLOT_WORDS, sizeof(unsigned int), b->sig_flag[kp]);
      if (!sent) b->engines = nullptr;  // refused: the copy stream from here on (this step's block may be in flight: sent again below)
    }
    if (!sent) {
      HIP_TRY(hipMemcpyAsync(b->d_stage[k] + off, b->h_stage[kp] + off, len, hipMemcpyHostToDevice, b->copy_stream));
      // behind the frames on the same stream: when the kernel sees n, they are in HBM
      fill_flag_block(b->h_seq + kp * FLAG_SLOT_WORDS, n, fw);
      HIP_TRY(hipMemcpyAsync(b->d_flag, b->h_seq + kp * FLAG_SLOT_WORDS, fw * sizeof(unsigned int), hipMemcpyHostToDevice, b->copy_stream));
      // (the step in which the engines were given up: its pinned block is not in the books the copy-stream path keeps)
      if (deep) HIP_TRY(hipStreamSynchronize(b->copy_stream));
    }
  }
  const int rc = batch_launch(b, active, &in, frames_dev, frame_bytes);
  if (b->ingest_profile) {
    const double t_d = host_now_us();
    b->ingest_us[0] += t_b - t_a; b->ingest_us[1] += t_c - t_b; b->ingest_us[2] += t_d - t_c; b->ingest_us[3] += 1.0;
  }
  return rc;
}

}  // namespace

extern "C" {

int rmd_hip_batch_destroy(rmd_hip_batch_t* b) {
  if (!b) return RMD_HIP_OK;
  (void)hipSetDevice(b->device);
  for (auto& G : b->groups)
    if (G.stream) (void)hipStreamSynchronize(G.stream);
  if (b->copy_stream) (void)hipStreamSynchronize(b->copy_stream);
  {
    bool any = false;
    for (int q = 0; q < rmd_hip_batch::SLOTS_MAX; ++q) any = any || b->sig_frame[q] || b->sig_flag[q];
    if (CopyEngines* e = any ? b->engines_used : nullptr) {  // (not b->engines: it may have been given up)
      for (int q = 0; q < rmd_hip_batch::SLOTS_MAX; ++q)
        for (uint64_t* sig : {&b->sig_frame[q], &b->sig_flag[q]}) {
          if (*sig && e->wait_idle(*sig, 2e6)) e->destroy_signal(*sig);  // (an engine that still owes a signal keeps it)
          *sig = 0;
        }
    }
  }
  if (b->ingest_profile && b->ingest_us[3] > 0) {
    fprintf(stderr, "[rmd_hip ingest] batch of %d, %.0f steps: wait for slot %.2f us, host copy %.2f us, submit %.2f us per step; "
                    "longest wait %.0f us, %lu waits gave up (stream idle, word not reached); "
                    "steps handed over <=0 / 1 / 2 / 3 / >=4 ahead of group 0's newest started setup kernel: %lu / %lu / %lu / %lu / %lu\n",
            b->n, b->ingest_us[3], b->ingest_us[0] / b->ingest_us[3], b->ingest_us[1] / b->ingest_us[3], b->ingest_us[2] / b->ingest_us[3],
                g_progress_max_wait_us,
            g_progress_timeouts, b->ingest_lead[0], b->ingest_lead[1], b->ingest_lead[2], b->ingest_lead[3], b->ingest_lead[4]);
    for (int g = 0; g < b->n_groups; ++g)
      if (b->groups[g].h_progress)
        fprintf(stderr,
            "[rmd_hip ingest]   group %d: steps whose setup kernel converted its frames %u, of which it waited for %u (%u polls)\n", g,
                b->groups[g].h_progress[2], b->groups[g].h_progress[3], b->groups[g].h_progress[4]);
  }
  for (int i = 0; i < rmdk::MAX_BATCH; ++i)
    if (b->members[i]) (void)seeds_destroy_impl(b->members[i]);
  for (int q = 0; q < rmd_hip_batch::SLOTS_MAX; ++q) {
    if (b->h_stage[q]) (void)hipHostFree(b->h_stage[q]);
    if (b->d_stage[q]) (void)hipFree(b->d_stage[q]);
  }
  if (b->h_seq) (void)hipHostFree(b->h_seq);
  if (b->d_flag) (void)hipFree(b->d_flag);
  {
    rmd_hip_batch::Denoise& dn = b->dn;
    if (dn.stream) (void)hipStreamSynchronize(dn.stream);
    rmd_hip_image* all[] = {&dn.u[0], &dn.u[1], &dn.u_head[0], &dn.u_head[1], &dn.p[0], &dn.p[1], &dn.g};
    for (auto* im : all)
      if (im->owns && im->data) (void)hipFree(im->data);
    if (dn.ev0) (void)hipEventDestroy(dn.ev0);
    if (dn.ev1) (void)hipEventDestroy(dn.ev1);
    if (dn.d_table) (void)hipFree(dn.d_table);
    if (dn.h_staging) (void)hipHostFree(dn.h_staging);
    if (dn.stream) (void)hipStreamDestroy(dn.stream);
  }
  if (b->region_start) (void)hipEventDestroy(b->region_start);
  if (b->region_stop) (void)hipEventDestroy(b->region_stop);
  for (auto& G : b->groups) {
    G.ws.release();
    if (G.h_progress) (void)hipHostFree(G.h_progress);
    if (G.ev) (void)hipEventDestroy(G.ev);
  }
  if (b->copy_stream) (void)hipStreamDestroy(b->copy_stream);
  for (auto& G : b->groups)
    if (G.stream) (void)hipStreamDestroy(G.stream);
  delete b;
  return RMD_HIP_OK;
}

int rmd_hip_batch_create(int n, int width, int height, float fx, float fy, float cx, float cy, int patch_side, int max_extent,
    rmd_hip_batch_t** out) {
  if (!out) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_create: null output");
  *out = nullptr;
  if (n < 1 || n > rmdk::MAX_BATCH) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_create: %d sequences (1..%d)", n, rmdk::MAX_BATCH);
  if (width <= 0 || height <= 0) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_create: bad size %dx%d", width, height);
  int ndev = 0;
  TRY(rmd_hip_device_count(&ndev));
  rmd_hip_batch* b = new (std::nothrow) rmd_hip_batch();
  if (!b) return fail(RMD_HIP_ERR_RUNTIME, "batch_create: out of host memory");
  (void)hipGetDevice(&b->device);
  auto bail = [&](int rc) { rmd_hip_batch_destroy(b); return rc; };
  // measured (profiles/r03_batch_ab.txt): three groups beat two by 3-9 %, a fourth shares a hardware-queue pool and loses 25 %
  int want_groups = n >= 3 ? 3 : n;
  if (tunables().v[RMD_HIP_TUNE_BATCH_GROUPS] > 0) want_groups = tunables().v[RMD_HIP_TUNE_BATCH_GROUPS];  // (A/B)
  if (want_groups < 1) want_groups = 1;
  if (want_groups > rmd_hip_batch::MAX_GROUPS) want_groups = rmd_hip_batch::MAX_GROUPS;
  if (want_groups > n) want_groups = n;
  // a group is ONE launch pair, and a launch pair carries at most MAX_GROUP_SEQ sequences (their parameter blocks are kernel arguments)
  while (want_groups * rmdk::MAX_GROUP_SEQ < n) ++want_groups;
  b->n_groups = want_groups;
  b->ingest_profile = tunables().v[RMD_HIP_TUNE_INGEST_PROFILE] != 0;
  if (tunables().v[RMD_HIP_TUNE_RING_DEPTH] > 0) b->slots = tunables().v[RMD_HIP_TUNE_RING_DEPTH] < 3 ? 3
      : tunables().v[RMD_HIP_TUNE_RING_DEPTH];
  b->opt_unit_target = 1;  // (2x / 3x as many, smaller units: +4 % with one group of 4, nothing with two groups)
  const size_t pitch = (static_cast<size_t>(width) * 4 + 255) / 256 * 256;  // as image_alloc lays the members' planes out
  for (int g = 0; g < b->n_groups; ++g) {
    rmd_hip_batch::Group& G = b->groups[g];
    G.first = g == 0 ? 0 : b->groups[g - 1].first + b->groups[g - 1].n;
    G.n = n / b->n_groups + (g < n % b->n_groups ? 1 : 0);  // the larger groups first
    // (one priority level each, see create_stream; a fourth group shares the first one's pool)
    if (create_stream(&G.stream, g % 3) != hipSuccess) return bail(fail(RMD_HIP_ERR_RUNTIME, "batch_create: hipStreamCreate failed"));
    if (G.ws.allocate(width, height, static_cast<int>(pitch / 4), G.n, max_extent) != 0) return bail(fail(RMD_HIP_ERR_RUNTIME,
        "batch_create: update workspace"));
    if (hipHostMalloc(reinterpret_cast<void**>(&G.h_progress), 64, hipHostMallocMapped) != hipSuccess) return bail(fail(RMD_HIP_ERR_RUNTIME,
        "batch_create: progress words"));
    for (int q = 0; q < 16; ++q) G.h_progress[q] = 0u;
    if (hipEventCreate(&G.ev) != hipSuccess) return bail(fail(RMD_HIP_ERR_RUNTIME, "batch_create: event"));
  }
  if (create_stream(&b->copy_stream, 2) != hipSuccess) return bail(fail(RMD_HIP_ERR_RUNTIME, "batch_create: hipStreamCreate failed"));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, b->device) == hipSuccess && prop.multiProcessorCount > 0) b->num_cus = prop.multiProcessorCount;
  if (hipHostMalloc(reinterpret_cast<void**>(&b->h_seq), rmd_hip_batch::SLOTS_MAX * FLAG_SLOT_WORDS * sizeof(unsigned int),
      hipHostMallocDefault) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&b->d_flag), FLAG_ALLOC_BYTES) != hipSuccess || hipMemset(b->d_flag, 0,
          FLAG_ALLOC_BYTES) != hipSuccess)
    return bail(fail(RMD_HIP_ERR_RUNTIME, "batch_create: ingest words"));
  if (tunables().v[RMD_HIP_TUNE_COPY_ENGINES] != 0 && (b->engines = CopyEngines::for_device(b->device, 1)) != nullptr) {
    b->engines_used = b->engines;
    for (int q = 0; q < rmd_hip_batch::SLOTS_MAX && b->engines; ++q) {
      b->sig_frame[q] = b->engines->create_signal();
      b->sig_flag[q] = b->engines->create_signal();
      if (!b->sig_frame[q] || !b->sig_flag[q]) b->engines = nullptr;  // (what was created is destroyed with the batch)
    }
  }
  b->n = n;  // (group_of needs it while the members are created)
  for (int i = 0; i < n; ++i) {
    const int rc = seeds_create_impl(width, height, fx, fy, cx, cy, patch_side, max_extent, b, i, &b->members[i]);
    if (rc != RMD_HIP_OK) return bail(rc);
  }
  if (hipDeviceSynchronize() != hipSuccess) return bail(fail(RMD_HIP_ERR_RUNTIME, "batch_create: device synchronisation failed"));
  *out = b;
  return RMD_HIP_OK;
}

int rmd_hip_batch_size(const rmd_hip_batch_t* b, int* n) {
  if (!b || !n) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_size: null argument");
  *n = b->n;
  return RMD_HIP_OK;
}

int rmd_hip_batch_member(rmd_hip_batch_t* b, int index, rmd_hip_seeds_t** member) {
  if (!b || !member) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_member: null argument");
  if (index < 0 || index >= b->n) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_member: index %d outside [0, %d)", index, b->n);
  *member = b->members[index];
  return RMD_HIP_OK;
}

int rmd_hip_batch_update_device(rmd_hip_batch_t* b, const float* const* dev_imgs, const size_t* stride_elems, const float* T_curr_world) {
  if (!b || !dev_imgs || !stride_elems || !T_curr_world) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_update_device: null argument");
  TRY(batch_bind_device(b));
  unsigned int active = 0;
  for (int i = 0; i < b->n; ++i) {
    if (!dev_imgs[i]) continue;
    rmd_hip_seeds* m = b->members[i];
    if (!m->has_reference) return fail(RMD_HIP_ERR_NOT_READY, "batch_update_device: member %d has no reference image", i);
    if (stride_elems[i] < static_cast<size_t>(m->width)) return fail(RMD_HIP_ERR_INVALID_ARG,
        "batch_update_device: stride < width (member %d)", i);
    active |= 1u << i;
  }
  if (!active) return RMD_HIP_OK;
  for (int i = 0; i < b->n; ++i) {
    if (!((active >> i) & 1u)) continue;
    rmd_hip_seeds* m = b->members[i];
    m->P.cur = dev_imgs[i];  // zero copy, like rmd_hip_seeds_update_device
    m->P.cur_stride = static_cast<int>(stride_elems[i]);
    seeds_frame_pose(m, T_curr_world + 12 * i);
  }
  return batch_launch(b, active, nullptr, nullptr, 0);
}

int rmd_hip_batch_update_u8(rmd_hip_batch_t* b, const unsigned char* const* host_gray, const float* T_curr_world) {
  if (!b || !host_gray || !T_curr_world) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_update_u8: null argument");
  return batch_update_host(b, host_gray, nullptr, T_curr_world);
}

int rmd_hip_batch_update(rmd_hip_batch_t* b, const float* const* host_imgs, const float* T_curr_world) {
  if (!b || !host_imgs || !T_curr_world) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_update: null argument");
  return batch_update_host(b, nullptr, host_imgs, T_curr_world);
}

int rmd_hip_batch_sync(rmd_hip_batch_t* b) {
  if (!b) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_sync: null handle");
  TRY(batch_bind_device(b));
  for (int i = 0; i < b->n; ++i) TRY(seeds_flush(b->members[i]));
  for (int g = 0; g < b->n_groups; ++g) HIP_TRY(hipStreamSynchronize(b->groups[g].stream));
  for (int g = 0; g < b->n_groups; ++g) TRY(ingest_error_check(b->groups[g].h_progress));
  return RMD_HIP_OK;
}

int rmd_hip_batch_set_option(rmd_hip_batch_t* b, int option, int value) {
  if (!b) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_set_option: null handle");
  switch (option) {
    case RMD_HIP_OPT_TIMING:
      if (value != 0 && value != 2) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_set_option: timing mode %d (0 or 2)", value);
      b->opt_timing = value;
      return RMD_HIP_OK;
    case RMD_HIP_OPT_UNIT_TARGET:
      if (value < 1 || value > 4) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_set_option: unit target %d outside 1..4", value);
      b->opt_unit_target = value;
      return RMD_HIP_OK;
    default: return fail(RMD_HIP_ERR_INVALID_ARG, "batch_set_option: unknown option %d", option);
  }
}

// the region timer spans both groups' streams: it starts when stream 0 reaches the start event (stream 1 waits for that event) and stops
// when both streams have run everything queued before the query
int rmd_hip_batch_timing_reset(rmd_hip_batch_t* b) {
  if (!b) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_timing_reset: null handle");
  TRY(rmd_hip_batch_sync(b));
  if (!b->region_start) HIP_TRY(hipEventCreate(&b->region_start));
  HIP_TRY(hipEventRecord(b->region_start, b->groups[0].stream));
  for (int g = 1; g < b->n_groups; ++g) HIP_TRY(hipStreamWaitEvent(b->groups[g].stream, b->region_start, 0));
  b->region_updates = 0;
  return RMD_HIP_OK;
}

int rmd_hip_batch_timing(rmd_hip_batch_t* b, double* total_ms, long* steps) {
  if (!b) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_timing: null handle");
  if (!b->region_start) return fail(RMD_HIP_ERR_NOT_READY, "batch_timing: call timing_reset first");
  TRY(batch_bind_device(b));
  if (!b->region_stop) HIP_TRY(hipEventCreate(&b->region_stop));
  for (int i = 0; i < b->n; ++i) TRY(seeds_flush(b->members[i]));  // the deferred finalisations belong to the region
  for (int g = 1; g < b->n_groups; ++g) {
    HIP_TRY(hipEventRecord(b->groups[g].ev, b->groups[g].stream));
    HIP_TRY(hipStreamWaitEvent(b->groups[0].stream, b->groups[g].ev, 0));
  }
  HIP_TRY(hipEventRecord(b->region_stop, b->groups[0].stream));
  HIP_TRY(hipEventSynchronize(b->region_stop));
  float ms = 0.0f;
  HIP_TRY(hipEventElapsedTime(&ms, b->region_start, b->region_stop));
  if (total_ms) *total_ms = ms;
  if (steps) *steps = b->region_updates;
  return RMD_HIP_OK;
}

}  // extern "C"

// ---- TV-L1 for every member of a batch in one launch sequence ---------------------------------
// DepthmapDenoiser::denoise (depthmap_denoiser.cu:179-224) per member; what changes is the launch shape: at 640x480 one depth map is 50
// launches of 1 200 workgroups, each bound by dispatch and load latency (14 us for 4 iterations), and the B maps of a batch cost B times
// that when they are denoised one after the other.  With grid z = member the same 50 launches carry B x 1 200 workgroups.
extern "C" {

int rmd_hip_batch_denoise(rmd_hip_batch_t* b, const float* depth_range, float lambda, int iterations, float* const* host_denoised) {
  if (!b || !depth_range) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_denoise: null argument");
  if (iterations < 0) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_denoise: negative iteration count");
  TRY(batch_bind_device(b));
  const rmd_hip_seeds* m0 = b->members[0];
  const int w = m0->width, h = m0->height, n = b->n;
  rmd_hip_batch::Denoise& dn = b->dn;
  // a member that never got a reference frame holds no depth map (its planes are the zeros of its creation): an error like denoise() before
  // setLargeSigmaSq(), not a map of the prior
  for (int i = 0; i < n; ++i)
    if (!b->members[i]->has_reference) return fail(RMD_HIP_ERR_NOT_READY, "batch_denoise: member %d has no reference image", i);
  // (a call that fails half-way allocates only what is still missing when it is repeated; batch_destroy releases whatever exists)
  if (!dn.ready) {
    rmd_hip_image* f32[] = {&dn.u[0], &dn.u[1], &dn.u_head[0], &dn.u_head[1], &dn.g};
    for (auto* im : f32)
      if (!im->data) TRY(image_alloc(im, RMD_HIP_KIND_F32, w, h * n));
    for (int k = 0; k < 2; ++k)
      if (!dn.p[k].data) TRY(image_alloc(&dn.p[k], RMD_HIP_KIND_F32X2, w, h * n));
    if (!dn.d_table) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dn.d_table), static_cast<size_t>(rmdk::MAX_BATCH) * rmdk::TV_MEMBER_WORDS
        * sizeof(unsigned long long)));
    if (!dn.h_staging) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&dn.h_staging), static_cast<size_t>(n) * w * h * sizeof(float)));
    if (!dn.stream) HIP_TRY(hipStreamCreateWithFlags(&dn.stream, hipStreamNonBlocking));
    HIP_TRY(hipDeviceSynchronize());
    dn.ready = true;
  }
  // the members' state must be final and at rest: deferred finalisations, then every group's stream (the members' last kernels are left in
  // flight)
  for (int i = 0; i < n; ++i) TRY(seeds_flush(b->members[i]));
  for (int g = 0; g < b->n_groups; ++g) HIP_TRY(hipStreamSynchronize(b->groups[g].stream));
  for (int g = 0; g < b->n_groups; ++g) TRY(ingest_error_check(b->groups[g].h_progress));
  HIP_TRY(hipStreamSynchronize(dn.stream));
  unsigned long long table[rmdk::MAX_BATCH * rmdk::TV_MEMBER_WORDS] = {};
  for (int i = 0; i < n; ++i) {
    const rmd_hip_seeds* m = b->members[i];
    // DepthmapDenoiser::setLargeSigmaSq, depthmap_denoiser.cu:226-229
    const float large_sigma_sq = depth_range[i] * depth_range[i] / 72.0f;
    unsigned int bits;
    memcpy(&bits, &large_sigma_sq, 4);
    unsigned long long* t = table + static_cast<size_t>(i) * rmdk::TV_MEMBER_WORDS;
    t[0] = reinterpret_cast<unsigned long long>(m->P.mu); t[1] = reinterpret_cast<unsigned long long>(m->P.sigma_sq);
    t[2] = reinterpret_cast<unsigned long long>(m->P.a); t[3] = reinterpret_cast<unsigned long long>(m->P.b);
    t[4] = bits;
  }
  HIP_TRY(hipMemcpyAsync(dn.d_table, table, sizeof(table), hipMemcpyHostToDevice, dn.stream));
  HIP_TRY(hipStreamSynchronize(dn.stream));  // (the source is on this function's stack)
  rmdk::TvParams P;
  P.w = w; P.h = h;
  P.stride = static_cast<int>(dn.g.stride);
  P.stride2 = static_cast<int>(dn.p[0].stride);
  P.mu = nullptr; P.sigma_sq = nullptr; P.a = nullptr; P.b = nullptr;  // per member, from the table
  P.in_stride = m0->P.stride;
  P.g = static_cast<float*>(dn.g.data);
  P.large_sigma_sq = 0.0f;
  // denoise::DeviceData constructor, depthmap_denoiser.cu:124-141 (the constants of rmd_hip_denoiser_create)
  const float L = sqrtf(8.0f);
  P.tau = 0.02f; P.sigma = (1 / (L * L)) / P.tau; P.theta = 0.5f; P.lambda = lambda;
  P.members = dn.d_table;
  P.member_stride = dn.g.stride * static_cast<size_t>(h);
  P.member_stride2 = dn.p[0].stride * static_cast<size_t>(h);
  if (!dn.ev0) HIP_TRY(hipEventCreate(&dn.ev0));  // (kept with the workspace: an error return below must not leak them)
  if (!dn.ev1) HIP_TRY(hipEventCreate(&dn.ev1));
  hipEvent_t ev0 = dn.ev0, ev1 = dn.ev1;
  float* us[2] = {static_cast<float*>(dn.u[0].data), static_cast<float*>(dn.u[1].data)};
  float* uhs[2] = {static_cast<float*>(dn.u_head[0].data), static_cast<float*>(dn.u_head[1].data)};
  float2* ps[2] = {static_cast<float2*>(dn.p[0].data), static_cast<float2*>(dn.p[1].data)};
  int cur_buf = 0;
  long n_launches = 0;
  const int rc = tv_run(P, us, uhs, ps, n, iterations, 0, 0, dn.stream, ev0, &cur_buf, &n_launches);
  if (rc != RMD_HIP_OK) return rc;
  HIP_TRY(hipEventRecord(ev1, dn.stream));
  dn.result_index = cur_buf;
  const rmd_hip_image& r = dn.u[cur_buf];
  const size_t row = static_cast<size_t>(w) * 4, plane = static_cast<size_t>(w) * h;
  for (int i = 0; i < n; ++i) {
    rmd_hip_image& v = dn.result[i];
    v = rmd_hip_image();
    v.kind = RMD_HIP_KIND_F32; v.width = w; v.height = h; v.device = b->device; v.pitch = r.pitch; v.stride = r.stride;
    v.data = static_cast<char*>(r.data) + static_cast<size_t>(i) * h * r.pitch;
    v.owns = false; v.owner_stream = dn.stream;
    if (host_denoised && host_denoised[i])
      HIP_TRY(hipMemcpy2DAsync(dn.h_staging + i * plane, row, v.data, v.pitch, row, h, hipMemcpyDeviceToHost, dn.stream));
  }
  HIP_TRY(hipStreamSynchronize(dn.stream));
  for (int i = 0; i < n; ++i)
    if (host_denoised && host_denoised[i]) memcpy(host_denoised[i], dn.h_staging + i * plane, plane * sizeof(float));
  float ms = 0.0f;
  if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) { dn.total_ms = ms; dn.launches = n_launches; }
  return RMD_HIP_OK;
}

int rmd_hip_batch_denoise_result(const rmd_hip_batch_t* b, int index, const rmd_hip_image_t** view) {
  if (!b || !view) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_denoise_result: null argument");
  if (index < 0 || index >= b->n) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_denoise_result: index %d outside [0, %d)", index, b->n);
  if (!b->dn.ready || !b->dn.result[index].data) return fail(RMD_HIP_ERR_NOT_READY,
      "batch_denoise_result: rmd_hip_batch_denoise has not run");
  *view = &b->dn.result[index];
  return RMD_HIP_OK;
}

int rmd_hip_batch_denoise_timing(const rmd_hip_batch_t* b, double* total_ms, long* launches) {
  if (!b) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_denoise_timing: null handle");
  if (total_ms) *total_ms = b->dn.total_ms;
  if (launches) *launches = b->dn.launches;
  return RMD_HIP_OK;
}

}  // extern "C"

