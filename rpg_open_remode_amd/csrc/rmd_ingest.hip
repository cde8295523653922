// librmd_hip.so -- frames handed over in host memory (SeedMatrix::setReferenceImage / update with host pointers, seed_matrix.cu:87-158;
// Depthmap::inputImage, depthmap.cpp:95-106): pinned ring, staging copies and arrival flags, conversion one step ahead, lens-undistortion
// maps.
#include "rmd_host.hpp"
#include "rmd_copy_pool.hpp"
#include "rmd_engines.hpp"

#include <sys/prctl.h>

using namespace rmdh;

namespace rmdk {  // (kernels of this unit only: each kernel of the library has ONE home translation unit)

// Frame ingest (reference: Depthmap::inputImage, src/depthmap.cpp:95-106 -- cv::Mat::convertTo(CV_32F, 1.0f/255.0f) on the
// host): 8-bit gray -> f32 plane on the device.  One fp32 multiply per pixel, identical bits to the host conversion.
// 4 pixels per lane: one 32-bit load, one 128-bit store.
static __global__ __launch_bounds__(256) void ingest_u8_kernel(const unsigned char* __restrict__ src, int src_pitch,
    float* __restrict__ dst,
                                                        int dst_stride, int w, int h) {
  const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x4 >= w || y >= h) return;
  const unsigned char* row = src + static_cast<size_t>(y) * src_pitch;
  float* out = dst + static_cast<size_t>(y) * dst_stride;
  if (x4 + 3 < w) {
    const unsigned int v = *reinterpret_cast<const unsigned int*>(row + x4);  // src_pitch and x4 are multiples of 4
    float4 f;
    f.x = static_cast<float>(v & 0xffu) * (1.0f / 255.0f);
    f.y = static_cast<float>((v >> 8) & 0xffu) * (1.0f / 255.0f);
    f.z = static_cast<float>((v >> 16) & 0xffu) * (1.0f / 255.0f);
    f.w = static_cast<float>(v >> 24) * (1.0f / 255.0f);
    *reinterpret_cast<float4*>(out + x4) = f;
  } else {
    for (int x = x4; x < w; ++x) out[x] = static_cast<float>(row[x]) * (1.0f / 255.0f);
  }
}

// The same with lens undistortion in front (Depthmap::inputImage with is_distorted_, depthmap.cpp:95-106): cv::remap of the
static __global__ __launch_bounds__(256) void ingest_u8_remap_kernel(const unsigned char* __restrict__ src, int src_pitch,
                                                              const short2* __restrict__ map1, const unsigned short* __restrict__ map2,
                                                              float* __restrict__ dst, int dst_stride, int w, int h) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const short2 m = map1[static_cast<size_t>(y) * w + x];
  const int f = map2[static_cast<size_t>(y) * w + x] & 1023;
  dst[static_cast<size_t>(y) * dst_stride + x] = remap_u8_pixel(src, src_pitch, m, f, w, h, [](const unsigned char* p) {
      return static_cast<int>(*p); });
}

}  // namespace rmdk

// diagnostics (RMD_HIP_INGEST_PROFILE): waits for a staging slot that ended because the stream was idle before the word was reached
unsigned long g_progress_timeouts = 0;
double g_progress_max_wait_us = 0.0;    // ... and the longest such wait

namespace rmdh {

// Frames that arrive in host memory.  Two pipelines, both with SLOTS frames in flight and no call that waits for the device:
//
// (a) default, tile pipeline (ingest_current_fused): host copies the frame into pinned buffer n % SLOTS (the caller's buffer is free
//     on return, as with the reference's synchronous cudaMemcpy, seed_matrix.cu:128); the copy stream moves it to a staging buffer in
//     HBM and writes the frame's number behind it; the setup kernel of that frame waits for the number ITSELF and converts the frame
//     into the current-image plane (x(1/255) for 8-bit frames).  No event, no cross-stream wait: a barrier packet on the compute
//     queue cost 6 us per frame, an event record a little less, and a kernel that reads the pinned buffer across PCIe slows every
//     load around it down (+5 us).  The host learns from a word the setup kernel writes into pinned memory which frames have been
//     consumed.  640x480, 8-bit frames: 51.7 us per update against 48.6 us with resident frames; pipeline (b): 60.9 us.
// (b) the other matchers, the reference frame, and 8-bit frames with lens undistortion (ingest_frame): upload and conversion /
//     remap kernel on the copy stream, events between the two streams:
//   host      wait until slot's staging buffer has been read (SLOTS frames ago), copy the caller's frame into it
//   copy      wait until the update that read the slot's plane (SLOTS frames ago) has run -> H2D -> [u8: x(1/255) / remap kernel]
//   compute   wait until the slot's plane is staged -> this frame's kernels -> mark the slot's plane free
//     The current image rotates through SLOTS planes; planes[CURR_IMG] always names the one of the latest frame.
void ingest_release_engines(rmd_hip_seeds* s) {
  bool any = false;
  for (int q = 0; q < rmd_hip_seeds::RING_MAX; ++q) any = any || s->sig_frame[q] || s->sig_flag[q];
  s->engines = nullptr;
  if (!any) return;
  CopyEngines* e = CopyEngines::for_device(s->device, 0);  // (not s->engines: it may have been given up while copies were in flight)
  for (int q = 0; q < rmd_hip_seeds::RING_MAX; ++q) {
    for (uint64_t* sig : {&s->sig_frame[q], &s->sig_flag[q]}) {
      if (!*sig || !e) continue;
      // an engine that still owes this signal its completion writes it -- and the staging buffer -- when it gets there: neither is freed
      // under it (a withheld flag leaves no copy behind: its signal was never raised)
      if (e->wait_idle(*sig, 2e6)) e->destroy_signal(*sig);
      *sig = 0;
    }
  }
  s->engines = nullptr;
}

int ingest_init(rmd_hip_seeds* s) {
  if (s->ingest_ready) return RMD_HIP_OK;
  if (!s->copy_stream) HIP_TRY(create_stream(&s->copy_stream, 2));  // (a batch member uses the batch's)
  if (!s->batch && !s->copy_stream2 && tunables().v[RMD_HIP_TUNE_COPY_STREAMS] > 1) HIP_TRY(create_stream(&s->copy_stream2, 2));
  const Tunables& T = tunables();
  s->ingest_profile = T.v[RMD_HIP_TUNE_INGEST_PROFILE] != 0;
  s->opt_fused_ingest = T.v[RMD_HIP_TUNE_FUSED_INGEST] != 0;
  s->pack_backoff_len = T.v[RMD_HIP_TUNE_PACK_BACKOFF];  // (tests: 0 examines every float frame)
  s->ingest_ready = true;
  // engines in rotation; -1: the library's choice -- two, three for frames of 1.5 Mpixel or more (1920x1080: 41 -> 37 us of engine time per
  // frame against a 45-us light update, +10 %; smaller frames gain nothing from a third engine and lose 6 % with it when eight processes
  // share a device: profiles/r06_ab_copy_engines.txt)
  s->engine_route = s->batch ? 0 : T.v[RMD_HIP_TUNE_COPY_ENGINES];
  if (s->engine_route < 0) s->engine_route = static_cast<long long>(s->width) * s->height >= 1500000ll ? 3 : 2;
  // (an engine beyond the first that refuses its first copy -- a node whose runtime keeps it for something else -- leaves the handle with
  // the engines that work: one engine is still a frame and its flag without a second command's fixed cost on hipMemcpyAsync's engine)
  while (s->engine_route > 1 && CopyEngines::for_device(s->device, s->engine_route) == nullptr) --s->engine_route;
  if (s->engine_route > 0 && (s->engines = CopyEngines::for_device(s->device, s->engine_route)) != nullptr) {
    s->engines_used = s->engines;
    for (int q = 0; q < rmd_hip_seeds::RING_MAX; ++q) {
      s->sig_frame[q] = s->engines->create_signal();
      s->sig_flag[q] = s->engines->create_signal();
      if (!s->sig_frame[q] || !s->sig_flag[q]) { ingest_release_engines(s); break; }
    }
  }
  {
    const int depth = T.v[RMD_HIP_TUNE_RING_DEPTH];
    if (depth > 0) s->ring = depth < 3 ? 3 : depth;  // (three: the caller's slot, the copy engine's, the one the kernels read)
  }
  // a member only ever stages REFERENCE frames (ingest_frame: pinned slot -> plane on the copy stream): events, nothing else
  if (s->batch) {
    s->cur_planes[0] = s->planes[RMD_HIP_PLANE_CURR_IMG].data;
    s->u8_pitch = (s->width + 3) / 4 * 4;
    for (int k = 0; k < rmd_hip_seeds::SLOTS; ++k) {
      HIP_TRY(hipEventCreateWithFlags(&s->staged[k], hipEventDisableTiming | hipEventReleaseToDevice));
      HIP_TRY(hipEventCreateWithFlags(&s->frame_done[k], hipEventDisableTiming | hipEventReleaseToDevice));
      HIP_TRY(hipEventRecord(s->staged[k], s->copy_stream));
      HIP_TRY(hipEventRecord(s->frame_done[k], s->stream));
    }
    return RMD_HIP_OK;
  }
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_progress), 64, hipHostMallocMapped));
  for (int q = 0; q < 16; ++q) s->h_progress[q] = 0u;
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_seq), rmd_hip_seeds::RING_MAX * FLAG_SLOT_WORDS * sizeof(unsigned int),
      hipHostMallocDefault));
  // one flag block per ring slot and kind of frame
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_zc_flag), 2 * rmd_hip_seeds::RING_MAX * FLAG_ALLOC_BYTES));
  HIP_TRY(hipMemset(s->d_zc_flag, 0, 2 * rmd_hip_seeds::RING_MAX * FLAG_ALLOC_BYTES));
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_submitted), 64, hipHostMallocMapped));
  static_assert(2 * rmd_hip_seeds::RING_MAX * sizeof(unsigned int) <= 64, "h_submitted");
  for (int q = 0; q < 2 * rmd_hip_seeds::RING_MAX; ++q) s->h_submitted[q] = 0u;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_ahead), 64));
  HIP_TRY(hipMemset(s->d_ahead, 0, 64));
  HIP_TRY(hipStreamSynchronize(nullptr));
  const rmd_hip_image& im = s->planes[RMD_HIP_PLANE_CURR_IMG];
  s->cur_planes[0] = im.data;
  for (int k = 1; k < rmd_hip_seeds::SLOTS; ++k) {
    HIP_TRY(hipMalloc(&s->cur_planes[k], im.pitch * im.height));
    HIP_TRY(hipMemset(s->cur_planes[k], 0, im.pitch * im.height));
  }
  HIP_TRY(hipStreamSynchronize(nullptr));
  s->u8_pitch = (s->width + 3) / 4 * 4;  // rows start on a dword (the conversion kernel reads 4 pixels at a time)
  // The events only order work of this device's two streams (and tell the host that a staging buffer has been read): a
  // device-scope release is enough.  The default -- a system-scope fence with cache write-back and invalidation at every
  // record -- cost more per frame than the upload it was ordering.
  for (int k = 0; k < rmd_hip_seeds::SLOTS; ++k) {
    HIP_TRY(hipEventCreateWithFlags(&s->staged[k], hipEventDisableTiming | hipEventReleaseToDevice));
    HIP_TRY(hipEventCreateWithFlags(&s->frame_done[k], hipEventDisableTiming | hipEventReleaseToDevice));
    HIP_TRY(hipEventRecord(s->staged[k], s->copy_stream));
    HIP_TRY(hipEventRecord(s->frame_done[k], s->stream));
  }
  return RMD_HIP_OK;
}

static int ingest_frame(rmd_hip_seeds* s, const unsigned char* host_gray, const float* host_f32, void* dst, size_t dst_pitch,
    bool dst_is_ref, int k) {
  const double t_a = s->ingest_profile ? host_now_us() : 0.0;
  HIP_TRY(hipEventSynchronize(s->staged[k]));  // the upload that last used this slot's staging buffers has run
  const double t_b = s->ingest_profile ? host_now_us() : 0.0;
  const size_t row_f32 = static_cast<size_t>(s->width) * 4;
  if (host_gray) {
    const size_t bytes = static_cast<size_t>(s->u8_pitch) * s->height;
    if (!s->h_u8[k]) {
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_u8[k]), bytes));
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_u8[k]), bytes));
    }
    if (s->u8_pitch == s->width) memcpy(s->h_u8[k], host_gray, bytes);
    else
      for (int y = 0; y < s->height; ++y)
        memcpy(s->h_u8[k] + static_cast<size_t>(y) * s->u8_pitch, host_gray + static_cast<size_t>(y) * s->width, s->width);
  } else {
    if (!s->h_f32[k]) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_f32[k]), row_f32 * s->height));
    host_copy(s->h_f32[k], host_f32, row_f32 * s->height);
  }
  const double t_c = s->ingest_profile ? host_now_us() : 0.0;
  if (dst_is_ref) {  // everything issued so far may read the reference plane
    HIP_TRY(hipEventRecord(s->frame_done[k], s->stream));
  }
  HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->frame_done[k], 0));
  if (host_gray) {
    HIP_TRY(hipMemcpyAsync(s->d_u8[k], s->h_u8[k], static_cast<size_t>(s->u8_pitch) * s->height, hipMemcpyHostToDevice, s->copy_stream));
    const int dst_stride = static_cast<int>(dst_pitch / 4);
    if (s->d_undist_map1) {
      const dim3 block(64, 4), grid((s->width + 63) / 64, (s->height + 3) / 4);
      hipLaunchKernelGGL(rmdk::ingest_u8_remap_kernel, grid, block, 0, s->copy_stream, s->d_u8[k], s->u8_pitch, s->d_undist_map1,
          s->d_undist_map2,
                         static_cast<float*>(dst), dst_stride, s->width, s->height);
    } else {
      const dim3 block(64, 4), grid((s->width + 255) / 256, (s->height + 3) / 4);
      hipLaunchKernelGGL(rmdk::ingest_u8_kernel, grid, block, 0, s->copy_stream, s->d_u8[k], s->u8_pitch, static_cast<float*>(dst),
          dst_stride,
                         s->width, s->height);
    }
    HIP_TRY(hipGetLastError());
  } else {
    HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch, s->h_f32[k], row_f32, row_f32, s->height, hipMemcpyHostToDevice, s->copy_stream));
  }
  HIP_TRY(hipEventRecord(s->staged[k], s->copy_stream));
  // The compute stream has to run behind the staging.  A stream-side wait (barrier packet) costs the compute queue ~6 us per
  // frame even when the event has long fired; the staging of a frame finishes while the PREVIOUS frame's kernels still run,
  // so the host can simply wait for it before it queues this frame's kernels behind them (no bubble, no packet).
  HIP_TRY(hipStreamWaitEvent(s->stream, s->staged[k], 0));
  if (s->ingest_profile) {
    const double t_d = host_now_us();
    s->ingest_us[0] += t_b - t_a; s->ingest_us[1] += t_c - t_b; s->ingest_us[2] += t_d - t_c; s->ingest_us[3] += 1.0;
  }
  return RMD_HIP_OK;
}

// a host frame becomes the current image: stage it into the plane that is NOT being read by the update in flight The fused path: frame n
// goes through pinned buffer and staging buffer n % SLOTS, last read by the copy engine / the setup kernel of frame n - SLOTS.  That kernel
// has completed once the setup kernel of frame n - SLOTS + 1 has STARTED (same stream), which is what h_progress reports. Wait (on the
// host, without touching the device) until the setup kernel of step `need` has started, as reported through the pinned word `progress`: the
// staging buffers of SLOTS steps ago are free then.  Numbers are compared modulo 2^32 like the kernel's test. The word is written by the
// DEVICE (a store of the setup kernel into pinned memory): there is nothing a futex or a condition variable could be woken by.  So: a
// polite spin for the first 2 us (the usual wait when the host is barely ahead), then short sleeps -- the caller is up to three frames (>=
// 100 us of device work) ahead of the kernel it waits for, an overslept wake-up of a few microseconds stalls nothing, and a rank's host
// thread no longer burns a whole core while the device works (eight ranks share the 16 CPUs of the measurement box's quota:
// profiles/r05_nranks/).  Linux rounds a sleep up by the thread's timer slack (50 us by default): the slack is 2 us WHILE THIS FUNCTION
// SLEEPS and the caller's own value again when it returns (a library does not leave a scheduling attribute of the application's thread
// changed -- threads created later would inherit it); if the slack cannot be read or set the wait goes on with the caller's slack, in
// coarser steps.  RMD_HIP_TUNE_HOST_WAIT = 0 restores the pure spin (A/B).
int wait_for_progress(volatile unsigned int* progress, unsigned int need, hipStream_t stream) {
  auto behind = [&]() { return static_cast<int>(*progress - need) < 0; };
  if (behind()) {
    const double t0 = host_now_us();
    const bool may_sleep = tunables().v[RMD_HIP_TUNE_HOST_WAIT] != 0;
    long saved_slack = -1;  // >= 0: the caller's timer slack, to be restored
    int rc = RMD_HIP_OK;
    double next_query = 2000.0;
    while (behind()) {
      const double waited = host_now_us() - t0;
      // A long wait is either a long kernel (the first update of a 1920x1080 sequence takes 7-9 ms: keep waiting, in longer sleeps) or a
      // word that will never move because nothing this wait is for is queued any more (another matcher ran the frames in between): then the
      // stream is idle and has read everything.  Looked at twice per millisecond; counted as "gave up" only in the second case.
      if (waited > next_query) {
        next_query = waited + 500.0;
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) {
          if (behind()) ++g_progress_timeouts;
          break;
        }
        if (q != hipErrorNotReady) {
          rc = fail(RMD_HIP_ERR_RUNTIME, "waiting for a free slot of the frame ring: hipStreamQuery failed: %s", hipGetErrorString(q));
          break;
        }
      }
      if (may_sleep && waited > 2.0) {
        if (saved_slack < 0) {
          const long cur = prctl(PR_GET_TIMERSLACK, 0UL, 0UL, 0UL, 0UL);
          if (cur > 2000 && prctl(PR_SET_TIMERSLACK, 2000UL, 0UL, 0UL, 0UL) == 0) saved_slack = cur;
          else saved_slack = 0;  // (already fine, or not ours to change: nothing to restore)
        }
        timespec ts = {0, waited > 2000.0 ? 50000 : 8000};  // 8 us + the slack + the scheduler's wake-up: 15-20 us in practice
        (void)nanosleep(&ts, nullptr);
      } else {
        cpu_relax();
      }
    }
    if (saved_slack > 0) (void)prctl(PR_SET_TIMERSLACK, static_cast<unsigned long>(saved_slack), 0UL, 0UL, 0UL);
    const double w = host_now_us() - t0;
    if (w > g_progress_max_wait_us) g_progress_max_wait_us = w;
    return rc;
  }
  return RMD_HIP_OK;
}

static int ingest_current_fused(rmd_hip_seeds* s, const unsigned char* host_gray, const float* host_f32, const float* T_curr_world) {
  const double t_a = s->ingest_profile ? host_now_us() : 0.0;
  const unsigned long long n64 = ++s->zc_number;
  const unsigned int n = static_cast<unsigned int>(n64);
  const unsigned int ring = static_cast<unsigned int>(s->ring);
  const int k = static_cast<int>(n64 % ring);
  if (n64 > static_cast<unsigned long long>(ring)) TRY(wait_for_progress(s->h_progress, n - ring + 1u, s->stream));
  const double t_b = s->ingest_profile ? host_now_us() : 0.0;
  // how far ahead of the device is the caller?  frame n is handed over while setup `started` is the newest that has begun
  if (s->ingest_profile) {
    const int lead = static_cast<int>(n - *static_cast<volatile unsigned int*>(s->h_progress));
    ++s->ingest_lead[lead < 0 ? 0 : lead > 4 ? 4 : lead];
  }
  PendingIngest in;
  bool in_place = false;
  // every ring slot has its own arrival flag, one per kind of frame (8-bit / float: they use different staging buffers): the setup kernel
  // of frame n asks for frame n's; its verdict for frame n + 1 reads the flag of that slot for ITS kind, which a frame of the other kind
  // never sets
  auto flag_of = [&](int kind, int slot) {
      return s->d_zc_flag + (static_cast<size_t>(kind) * rmd_hip_seeds::RING_MAX + slot) * (FLAG_ALLOC_BYTES / sizeof(unsigned int)); };
  void* stage_src = nullptr; void* stage_dst = nullptr; size_t stage_bytes = 0;
  auto ensure_u8_ring = [&]() -> int {
    const size_t bytes = static_cast<size_t>(s->u8_pitch) * s->height;
    for (int q = 0; q < s->ring; ++q) {  // (all slots at once: the search kernel is told where the NEXT frame will be)
      if (s->h_zc_u8[q]) continue;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_zc_u8[q]), bytes + 16, hipHostMallocDefault));
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_zc_u8[q]), bytes));
    }
    return RMD_HIP_OK;
  };
  // a float frame of 8-bit levels goes the 8-bit way from here on, never through the undistortion maps (pack_float_rows_u8)
  bool packed = false;
  if (!host_gray && float_frames_as_bytes()) {
    if (s->pack_backoff > 0) --s->pack_backoff;
    else {
      TRY(ensure_u8_ring());
      packed = CopyPool::instance().pack(host_f32, s->h_zc_u8[k], s->width, s->height, s->u8_pitch);
      if (!packed) s->pack_backoff = s->pack_backoff_len;
    }
  }
  const bool as_u8 = host_gray != nullptr || packed;
  const bool remap = host_gray != nullptr && s->d_undist_map1 != nullptr;
  // a frame the caller keeps in pinned memory: the engine reads it where it lies (rmd_hip_seeds_update_u8_pinned)
  const bool from_caller = s->next_frame_pinned && host_gray && !remap && s->engines && s->u8_pitch == s->width && !frame_in_place(false,
      remap);
  s->next_frame_pinned = false;
  in.no_remap = packed;
  if (as_u8) {
    const size_t bytes = static_cast<size_t>(s->u8_pitch) * s->height;
    TRY(ensure_u8_ring());
    if (packed || from_caller) {
    } else if (s->u8_pitch == s->width) host_copy(s->h_zc_u8[k], host_gray, bytes);
    else
      for (int y = 0; y < s->height; ++y)
        memcpy(s->h_zc_u8[k] + static_cast<size_t>(y) * s->u8_pitch, host_gray + static_cast<size_t>(y) * s->width, s->width);
    in_place = frame_in_place(false, remap);
    if (in_place) {
      void* dev = nullptr;
      HIP_TRY(hipHostGetDevicePointer(&dev, s->h_zc_u8[k], 0));
      in.u8 = static_cast<const unsigned int*>(dev);
    } else {
      stage_src = from_caller ? const_cast<unsigned char*>(host_gray) : s->h_zc_u8[k];
      stage_dst = s->d_zc_u8[k]; stage_bytes = bytes;
      in.u8 = reinterpret_cast<const unsigned int*>(s->d_zc_u8[k]);
    }
    in.common.kind = 1;
    in.common.pitch = s->u8_pitch;
  } else {
    const size_t bytes = static_cast<size_t>(s->width) * s->height * sizeof(float);
    for (int q = 0; q < s->ring; ++q) {
      if (s->h_zc_f32[q]) continue;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_zc_f32[q]), bytes + 16, hipHostMallocDefault));
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_zc_f32[q]), bytes));
    }
    host_copy(s->h_zc_f32[k], host_f32, bytes);
    in_place = frame_in_place(false, false);
    if (in_place) {
      void* dev = nullptr;
      HIP_TRY(hipHostGetDevicePointer(&dev, s->h_zc_f32[k], 0));
      in.f32 = static_cast<const float*>(dev);
    } else {
      stage_src = s->h_zc_f32[k]; stage_dst = s->d_zc_f32[k]; stage_bytes = bytes;
      in.f32 = s->d_zc_f32[k];
    }
    in.common.kind = 2;
  }
  const bool ahead = frame_ahead(remap);
  const double t_copied = s->ingest_profile ? host_now_us() : 0.0;
  if (in_place) {
    in.common.flag = nullptr;  // the setup kernel reads the pinned buffer itself: it is complete before the kernel is launched
  } else {
    unsigned int* slot_flag = flag_of(as_u8 ? 0 : 1, k);
    // fault injection: the frame arrives, its flag never does -> the kernel's bounded wait runs out
    const bool withhold = s->inject_withhold_flag;
    s->inject_withhold_flag = false;
    bool sent = false;
    if (s->engines) {
      // The frame on a data engine, its flag -- one word -- on a flag engine that waits for the frame's completion in hardware
      // (rmd_engines.hpp).  The slot's signals are idle: the slot was consumed, so its flag had landed; what may be a few microseconds
      // behind is the flag engine's own completion signal.
      if (!s->engines->wait_idle(s->sig_frame[k], 2e6) || !s->engines->wait_idle(s->sig_flag[k], 2e6))
        return fail(RMD_HIP_ERR_RUNTIME, "a copy engine has not completed ring slot %d's previous frame after 2 s", k);
      s->h_seq[k * FLAG_SLOT_WORDS] = n;
      if (!from_caller) s->slot_ticket[k] = 0ull;  // (whatever frame of the caller's the slot stood for has been read: its signal is idle)
      // frames rotate over engine_route engines (1..4); a frame's flag is the next command of its own engine
      const unsigned de = static_cast<unsigned>(n64 % static_cast<unsigned long long>(s->engine_route)), fe = de;
      sent = s->engines->submit(de, fe, stage_dst, stage_src, stage_bytes,
                                s->sig_frame[k], withhold ? nullptr : slot_flag, s->h_seq + k * FLAG_SLOT_WORDS, sizeof(unsigned int),
                                s->sig_flag[k]);
      if (sent && from_caller) s->slot_ticket[k] = s->pinned_issued;  // (the entry point has counted this frame already)
      if (!sent) {  // refused: this frame and every later one take the copy stream
        if (s->ingest_profile) fprintf(stderr, "[rmd_hip ingest] copy engines given up: %s\n", s->engines->last_error());
        s->engines = nullptr;
      }
    }
    ++(sent ? s->staged_by_engines : s->staged_by_stream);
    if (!sent) {
      // (frame and flag on ONE stream, consecutive frames on alternating streams: see copy_stream2)
      hipStream_t cs = (s->copy_stream2 && (n64 & 1ull)) ? s->copy_stream2 : s->copy_stream;
      HIP_TRY(hipMemcpyAsync(stage_dst, stage_src, stage_bytes, hipMemcpyHostToDevice, cs));
      const size_t fw = flag_words(s->h_progress, n);
      // behind the frame on the same stream: when the kernel sees n, the frame is in HBM
      fill_flag_block(s->h_seq + k * FLAG_SLOT_WORDS, n, fw);
      if (!withhold)
        HIP_TRY(hipMemcpyAsync(slot_flag, s->h_seq + k * FLAG_SLOT_WORDS, fw * sizeof(unsigned int), hipMemcpyHostToDevice, cs));
      if (from_caller) HIP_TRY(hipStreamSynchronize(cs));  // (the engines were given up on this very frame: its ticket is done on return)
    }
    in.common.flag = slot_flag;
  }
  int plane = 0;  // one plane is enough: setup k writes it after search k - 1 has run (same stream)
  if (ahead) {  // ... unless the previous update's search kernel brings the frame in: frame n lives in plane n % 2
    const int kind = as_u8 ? 0 : 1, k_next = static_cast<int>((n64 + 1) % ring);
    void* dev = nullptr;
    if (in_place) {
      // frame n is complete in ITS slot, for ITS kind: the verdict of setup n - 1 read this very word, and setup n's verdict for frame n +
      // 1 reads the word of slot k_next for this kind -- which a frame of the other kind, or a frame two steps ahead, never sets (one word
      // per kind for the whole ring let setup n take "frame n + 2 of this kind is there" for "frame n + 1 is", and convert stale bytes)
      __atomic_store_n(&s->h_submitted[kind * rmd_hip_seeds::RING_MAX + k], n, __ATOMIC_RELEASE);
      HIP_TRY(hipHostGetDevicePointer(&dev, s->h_submitted, 0));
      in.common.submitted = static_cast<const unsigned int*>(dev) + kind * rmd_hip_seeds::RING_MAX + k_next;
      HIP_TRY(hipHostGetDevicePointer(&dev, as_u8 ? static_cast<void*>(s->h_zc_u8[k_next]) : static_cast<void*>(s->h_zc_f32[k_next]), 0));
      in.next_src = dev;
    } else {
      in.common.submitted = flag_of(kind, k_next);  // the arrival flag of the next frame's slot, for this kind
      in.next_src = as_u8 ? static_cast<const void*>(s->d_zc_u8[k_next]) : static_cast<const void*>(s->d_zc_f32[k_next]);
    }
    in.common.ahead = s->d_ahead;
    {  // no more bringers than the frame has work for (rmdk::INGEST_BATCH requests per lane)
      const long long dwords = as_u8 ? static_cast<long long>(s->u8_pitch >> 2) * s->height : static_cast<long long>(s->width) * s->height;
      const long long want = (dwords + 1023) / 1024;  // (staged: INGEST_BATCH dwords per lane; in place: one 16-byte request)
      const int cap = tunables().v[RMD_HIP_TUNE_AHEAD_WGS];
      in.common.ahead_wgs = static_cast<int>(want < cap ? (want < 1 ? 1 : want) : cap);
    }
    plane = static_cast<int>(n64 & 1ull);
    in.next_dst = static_cast<float*>(s->cur_planes[plane ^ 1]);
  }
  const double t_c = s->ingest_profile ? host_now_us() : 0.0;
  rmd_hip_image& im = s->planes[RMD_HIP_PLANE_CURR_IMG];
  im.data = s->cur_planes[plane];
  void* dev_progress = nullptr;
  HIP_TRY(hipHostGetDevicePointer(&dev_progress, s->h_progress, 0));
  in.common.progress = static_cast<unsigned int*>(dev_progress);
  in.common.number = n;
  in.common.profile = s->ingest_profile;
  s->P.cur = static_cast<const float*>(im.data);
  s->P.cur_stride = s->P.stride;
  const int rc = seeds_after_frame(s, T_curr_world, &in);
  if (s->ingest_profile) {
    const double t_d = host_now_us();
    s->ingest_us[0] += t_b - t_a; s->ingest_us[1] += t_c - t_b; s->ingest_us[2] += t_d - t_c; s->ingest_us[3] += 1.0;
    // ring wait, copy into the slot, copy-engine commands, launches
    const double phase[4] = {t_b - t_a, t_copied - t_b, t_c - t_copied, t_d - t_c};
    for (int q = 0; q < 4; ++q)
      if (phase[q] > s->ingest_max_us[q]) { s->ingest_max_us[q] = phase[q]; s->ingest_max_at[q] = n64; }
  }
  return rc;
}

int ingest_current(rmd_hip_seeds* s, const unsigned char* host_gray, const float* host_f32, const float* T_curr_world) {
  TRY(ingest_init(s));
  if (s->opt_fused_ingest && s->opt_matcher == 3) return ingest_current_fused(s, host_gray, host_f32, T_curr_world);
  const int k = s->ingest_slot;
  s->ingest_slot = (k + 1) % rmd_hip_seeds::SLOTS;
  rmd_hip_image& im = s->planes[RMD_HIP_PLANE_CURR_IMG];
  im.data = s->cur_planes[k];  // the plane last used SLOTS frames ago (its frame_done event is this slot's)
  TRY(ingest_frame(s, host_gray, host_f32, im.data, im.pitch, false, k));
  s->P.cur = static_cast<const float*>(im.data);
  s->P.cur_stride = s->P.stride;
  const int rc = seeds_after_frame(s, T_curr_world);
  HIP_TRY(hipEventRecord(s->frame_done[k], s->stream));
  return rc;
}

int ingest_reference(rmd_hip_seeds* s, const unsigned char* host_gray, const float* host_f32, const float* T_curr_world, float min_depth,
                            float max_depth) {
  TRY(ingest_init(s));
  TRY(seeds_flush(s));
  const int k = s->ingest_slot;
  s->ingest_slot = (k + 1) % rmd_hip_seeds::SLOTS;
  const rmd_hip_image& im = s->planes[RMD_HIP_PLANE_REF_IMG];
  TRY(ingest_frame(s, host_gray, host_f32, im.data, im.pitch, true, k));
  return seeds_after_reference(s, T_curr_world, min_depth, max_depth);
}

}  // namespace rmdh

extern "C" {

int rmd_hip_seeds_set_reference(rmd_hip_seeds_t* s, const float* host_img, const float* T_curr_world, float min_depth,
                                float max_depth) {
  if (!s || !host_img || !T_curr_world) return fail(RMD_HIP_ERR_INVALID_ARG, "set_reference: null argument");
  TRY(seeds_bind_device(s));
  return ingest_reference(s, nullptr, host_img, T_curr_world, min_depth, max_depth);
}

int rmd_hip_seeds_update(rmd_hip_seeds_t* s, const float* host_img, const float* T_curr_world) {
  if (!s || !host_img || !T_curr_world) return fail(RMD_HIP_ERR_INVALID_ARG, "update: null argument");
  if (s->batch) return fail(RMD_HIP_ERR_INVALID_ARG,
      "update: this SeedMatrix is a member of a batch; its updates are issued with rmd_hip_batch_update*");
  if (!s->has_reference) return fail(RMD_HIP_ERR_NOT_READY, "update: setReferenceImage has not been called");
  TRY(seeds_bind_device(s));
  // the frame is copied into pinned memory here (the caller's buffer is free on return, as after the reference's blocking
  // cudaMemcpy, seed_matrix.cu:128) and uploaded beside the previous frame's kernels; nothing waits for the device
  return ingest_current(s, nullptr, host_img, T_curr_world);
}

int rmd_hip_seeds_set_reference_u8(rmd_hip_seeds_t* s, const unsigned char* host_gray, const float* T_curr_world, float min_depth,
                                   float max_depth) {
  if (!s || !host_gray || !T_curr_world) return fail(RMD_HIP_ERR_INVALID_ARG, "set_reference_u8: null argument");
  TRY(seeds_bind_device(s));
  return ingest_reference(s, host_gray, nullptr, T_curr_world, min_depth, max_depth);
}

int rmd_hip_host_alloc(void** ptr, size_t bytes) {
  if (!ptr || !bytes) return fail(RMD_HIP_ERR_INVALID_ARG, "host_alloc: null argument");
  HIP_TRY(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
  return RMD_HIP_OK;
}

int rmd_hip_host_free(void* ptr) {
  if (ptr) HIP_TRY(hipHostFree(ptr));
  return RMD_HIP_OK;
}

int rmd_hip_seeds_pinned_frames_done(rmd_hip_seeds_t* s, unsigned long long* ticket_done) {
  if (!s || !ticket_done) return fail(RMD_HIP_ERR_INVALID_ARG, "pinned_frames_done: null argument");
  unsigned long long done = s->pinned_issued;
  for (int k = 0; k < rmd_hip_seeds::RING_MAX; ++k) {
    const unsigned long long t = s->slot_ticket[k];
    if (!t) continue;
    if (s->engines_used && s->engines_used->idle(s->sig_frame[k])) s->slot_ticket[k] = 0ull;  // read: the caller may write the frame again
    else if (t - 1ull < done) done = t - 1ull;
  }
  *ticket_done = done;
  return RMD_HIP_OK;
}

int rmd_hip_seeds_update_u8_pinned(rmd_hip_seeds_t* s, const unsigned char* pinned_gray, const float* T_curr_world,
                                   unsigned long long* ticket) {
  if (!s || !pinned_gray || !T_curr_world) return fail(RMD_HIP_ERR_INVALID_ARG, "update_u8_pinned: null argument");
  if (s->batch) return fail(RMD_HIP_ERR_INVALID_ARG,
      "update_u8_pinned: this SeedMatrix is a member of a batch; its updates are issued with rmd_hip_batch_update*");
  if (!s->has_reference) return fail(RMD_HIP_ERR_NOT_READY, "update_u8_pinned: setReferenceImage has not been called");
  TRY(seeds_bind_device(s));
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, pinned_gray) != hipSuccess || attr.type != hipMemoryTypeHost) {
    (void)hipGetLastError();
    return fail(RMD_HIP_ERR_INVALID_ARG, "update_u8_pinned: the frame is not in pinned host memory (rmd_hip_host_alloc, hipHostMalloc)");
  }
  ++s->pinned_issued;
  if (ticket) *ticket = s->pinned_issued;
  s->next_frame_pinned = true;
  const int rc = ingest_current(s, pinned_gray, nullptr, T_curr_world);
  s->next_frame_pinned = false;  // (a path that copies the frame -- another matcher, lens undistortion -- never looks at it)
  return rc;
}

int rmd_hip_seeds_update_u8(rmd_hip_seeds_t* s, const unsigned char* host_gray, const float* T_curr_world) {
  if (!s || !host_gray || !T_curr_world) return fail(RMD_HIP_ERR_INVALID_ARG, "update_u8: null argument");
  if (s->batch) return fail(RMD_HIP_ERR_INVALID_ARG,
      "update_u8: this SeedMatrix is a member of a batch; its updates are issued with rmd_hip_batch_update*");
  if (!s->has_reference) return fail(RMD_HIP_ERR_NOT_READY, "update_u8: setReferenceImage has not been called");
  TRY(seeds_bind_device(s));
  return ingest_current(s, host_gray, nullptr, T_curr_world);
}

// Depthmap::initUndistortionMap (depthmap.cpp:45-61) = cv::initUndistortRectifyMap(K, (k1, k2, r1, r2), I, K, size, CV_16SC2).
// OpenCV is not part of this build (and the reference does not pin its version); the map is computed here the way OpenCV's
// scalar code does (imgproc/undistort.cpp, core cv::invert): K and the coefficients are float values widened to double;
// ir = K^-1 by the closed-form 3x3 inverse cv::invert uses for n <= 3 (cofactors times 1/det, every product written out, zeros
// included); the normalised coordinates are ACCUMULATED along a row (_x += ir[0] per column); u, v are rounded to 1/32 pixel
// (cvRound: ties to even) and split into the integer position (map1) and the two 5-bit fractions (map2 = fy * 32 + fx).
// Host code, IEEE double, no contraction.  oracle/host_steps.py restates the same in numpy; neither can be pinned against OpenCV
// here ("parity unpinned" for this step).
int rmd_hip_compute_undistortion_map(int w, int h, float cam_fx, float cam_fy, float cam_cx, float cam_cy, float k1, float k2, float r1,
                                     float r2, short* map1_xy, unsigned short* map2) {
  if (w <= 0 || h <= 0 || !map1_xy || !map2) return fail(RMD_HIP_ERR_INVALID_ARG, "compute_undistortion_map: bad argument");
  const double fx = cam_fx, fy = cam_fy, u0 = cam_cx, v0 = cam_cy;
  const double dk1 = k1, dk2 = k2, p1 = r1, p2 = r2, k3 = 0.0, k4 = 0.0, k5 = 0.0, k6 = 0.0;
  const double S[3][3] = {{fx, 0.0, u0}, {0.0, fy, v0}, {0.0, 0.0, 1.0}};
  double d = S[0][0] * (S[1][1] * S[2][2] - S[1][2] * S[2][1]) - S[0][1] * (S[1][0] * S[2][2] - S[1][2] * S[2][0]) +
             S[0][2] * (S[1][0] * S[2][1] - S[1][1] * S[2][0]);
  d = 1.0 / d;
  const double ir[9] = {(S[1][1] * S[2][2] - S[1][2] * S[2][1]) * d, (S[0][2] * S[2][1] - S[0][1] * S[2][2]) * d,
                        (S[0][1] * S[1][2] - S[0][2] * S[1][1]) * d, (S[1][2] * S[2][0] - S[1][0] * S[2][2]) * d,
                        (S[0][0] * S[2][2] - S[0][2] * S[2][0]) * d, (S[0][2] * S[1][0] - S[0][0] * S[1][2]) * d,
                        (S[1][0] * S[2][1] - S[1][1] * S[2][0]) * d, (S[0][1] * S[2][0] - S[0][0] * S[2][1]) * d,
                        (S[0][0] * S[1][1] - S[0][1] * S[1][0]) * d};
  for (int i = 0; i < h; ++i) {
    double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
    for (int j = 0; j < w; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
      const double wi = 1. / _w, x = _x * wi, y = _y * wi;
      const double x2 = x * x, y2 = y * y;
      const double r2_ = x2 + y2, _2xy = 2 * x * y;
      const double kr = (1 + ((k3 * r2_ + dk2) * r2_ + dk1) * r2_) / (1 + ((k6 * r2_ + k5) * r2_ + k4) * r2_);
      const double u = fx * (x * kr + p1 * _2xy + p2 * (r2_ + 2 * x2)) + u0;
      const double v = fy * (y * kr + p1 * (r2_ + 2 * y2) + p2 * _2xy) + v0;
      const int iu = static_cast<int>(lrint(u * 32)), iv = static_cast<int>(lrint(v * 32));  // saturate_cast<int>(double) = cvRound
      const size_t k = static_cast<size_t>(i) * w + j;
      map1_xy[2 * k] = static_cast<short>(iu >> 5);
      map1_xy[2 * k + 1] = static_cast<short>(iv >> 5);
      map2[k] = static_cast<unsigned short>((iv & 31) * 32 + (iu & 31));
    }
  }
  return RMD_HIP_OK;
}

int rmd_hip_seeds_init_undistortion_map(rmd_hip_seeds_t* s, float k1, float k2, float r1, float r2) {
  if (!s) return fail(RMD_HIP_ERR_INVALID_ARG, "init_undistortion_map: null handle");
  TRY(seeds_bind_device(s));
  const int w = s->width, h = s->height;
  s->h_undist_map1.assign(static_cast<size_t>(w) * h * 2, 0);
  s->h_undist_map2.assign(static_cast<size_t>(w) * h, 0);
  TRY(rmd_hip_compute_undistortion_map(w, h, s->P.cam.fx, s->P.cam.fy, s->P.cam.cx, s->P.cam.cy, k1, k2, r1, r2, s->h_undist_map1.data(),
                                       s->h_undist_map2.data()));
  TRY(seeds_sync(s));
  if (!s->d_undist_map1) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_undist_map1), static_cast<size_t>(w) * h * sizeof(short2)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_undist_map2), static_cast<size_t>(w) * h * sizeof(unsigned short)));
  }
  HIP_TRY(hipMemcpy(s->d_undist_map1, s->h_undist_map1.data(), static_cast<size_t>(w) * h * sizeof(short2), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(s->d_undist_map2, s->h_undist_map2.data(), static_cast<size_t>(w) * h * sizeof(unsigned short), hipMemcpyHostToDevice));
  return RMD_HIP_OK;
}

int rmd_hip_seeds_undistortion_map(const rmd_hip_seeds_t* s, short* map1_xy, unsigned short* map2) {
  if (!s || !map1_xy || !map2) return fail(RMD_HIP_ERR_INVALID_ARG, "undistortion_map: null argument");
  if (s->h_undist_map1.empty()) return fail(RMD_HIP_ERR_NOT_READY, "undistortion_map: initUndistortionMap has not been called");
  memcpy(map1_xy, s->h_undist_map1.data(), s->h_undist_map1.size() * sizeof(short));
  memcpy(map2, s->h_undist_map2.data(), s->h_undist_map2.size() * sizeof(unsigned short));
  return RMD_HIP_OK;
}

}  // extern "C"
