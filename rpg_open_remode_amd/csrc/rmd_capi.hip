// librmd_hip.so: C ABI (include/rmd_hip.h) over the HIP kernels.  Host orchestration of
// rmd::SeedMatrix (seed_matrix.cu:28-230), rmd::DepthmapDenoiser (depthmap_denoiser.cu:124-229),
// rmd::ImageReducer (reduction.cu) and rmd::DeviceImage (device_image.cuh), redesigned around
// per-handle HIP streams, kernarg parameter blocks and pinned staging instead of the reference's
// default stream, device-resident descriptor structs and global texture references.
#include "rmd_hip.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#include "rmd_kernels.hpp"
#include "rmd_matcher.hpp"
#include "rmd_frame.hpp"
#ifdef RMD_AB_MATCHERS  // retired variants of the update, A/B builds only (tools/ab_make.sh)
#include "ab/rmd_matcher_r01.hpp"
#include "ab/rmd_frame_one_launch.hpp"
#endif

#define RMD_HIP_VERSION_NUMBER 310


namespace {

thread_local char g_last_error[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                                      \
  do {                                                                                                     \
    const hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess)                                                                                  \
      return fail(RMD_HIP_ERR_RUNTIME, "%s failed: %s (%d) at %s:%d", #expr, hipGetErrorString(e_),       \
                  static_cast<int>(e_), __FILE__, __LINE__);                                               \
  } while (0)

#define TRY(expr)             \
  do {                        \
    const int rc_ = (expr);   \
    if (rc_ != RMD_HIP_OK) return rc_; \
  } while (0)

struct ScopedDevice {  // run on `device`, restore the caller's current device afterwards
  int prev = -1;
  bool switched = false;
  explicit ScopedDevice(int device) {
    if (hipGetDevice(&prev) == hipSuccess && device >= 0 && prev != device) switched = hipSetDevice(device) == hipSuccess;
  }
  ~ScopedDevice() {
    if (switched) (void)hipSetDevice(prev);
  }
};

size_t kind_size(int kind) { return kind == RMD_HIP_KIND_F32X2 ? 8 : 4; }

// Poses on the host: se3.cuh:78-95 (inverse), :144-162 (compose), operation order preserved.
rmdk::Pose pose_inverse(const rmdk::Pose& p) {
  rmdk::Pose r;
  const float* d = p.d;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.d[4 * i + j] = d[4 * j + i];
  for (int i = 0; i < 3; ++i) r.d[4 * i + 3] = -d[i] * d[3] - d[4 + i] * d[7] - d[8 + i] * d[11];
  return r;
}
rmdk::Pose pose_compose(const rmdk::Pose& l, const rmdk::Pose& r) {
  rmdk::Pose o;
  for (int row = 0; row < 3; ++row) {
    const float l0 = l.d[4 * row], l1 = l.d[4 * row + 1], l2 = l.d[4 * row + 2], lt = l.d[4 * row + 3];
    for (int col = 0; col < 3; ++col) o.d[4 * row + col] = l0 * r.d[col] + l1 * r.d[4 + col] + l2 * r.d[8 + col];
    o.d[4 * row + 3] = lt + l0 * r.d[3] + l1 * r.d[7] + l2 * r.d[11];
  }
  return o;
}

struct StageTimer {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  std::vector<hipEvent_t> pool;
  double total_ms = 0.0;
  long launches = 0;
  hipEvent_t get() {
    if (!pool.empty()) {
      hipEvent_t e = pool.back();
      pool.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  void drain() {  // caller has synchronised the stream
    for (auto& pr : pending) {
      float ms = 0.0f;
      if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) total_ms += ms;
      ++launches;
      pool.push_back(pr.first);
      pool.push_back(pr.second);
    }
    pending.clear();
  }
  void reset() { total_ms = 0.0; launches = 0; }
  void destroy() {
    for (auto& pr : pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto e : pool) (void)hipEventDestroy(e);
    pending.clear();
    pool.clear();
  }
};

struct ScopedStage {
  StageTimer* t;
  hipStream_t stream;
  hipEvent_t start = nullptr;
  ScopedStage(StageTimer* timer, hipStream_t s) : t(timer), stream(s) {
    if (t) {
      start = t->get();
      (void)hipEventRecord(start, stream);
    }
  }
  ~ScopedStage() {
    if (t) {
      hipEvent_t stop = t->get();
      (void)hipEventRecord(stop, stream);
      t->pending.emplace_back(start, stop);
    }
  }
};

}  // namespace

// ---- rmd::DeviceImage<T> -------------------------------------------------------------------
struct rmd_hip_image {
  int kind = 0, width = 0, height = 0, device = 0;
  size_t pitch = 0;   // bytes
  size_t stride = 0;  // elements
  void* data = nullptr;
  bool owns = false;
  hipStream_t owner_stream = nullptr;  // stream of the handle that writes this image (views), else null
  struct rmd_hip_seeds* owner_seeds = nullptr;  // SeedMatrix whose plane this is: observers must let it settle first
};

namespace {

int seeds_sync(const rmd_hip_seeds* s);

// wait until the owner of an image (if any) has settled it
int image_settle(const rmd_hip_image* img) {
  ScopedDevice dev(img->device);  // the owner's stream belongs to the image's device, whatever the caller's current one is
  if (img->owner_seeds) return seeds_sync(img->owner_seeds);
  if (img->owner_stream) HIP_TRY(hipStreamSynchronize(img->owner_stream));
  return RMD_HIP_OK;
}

int image_alloc(rmd_hip_image* img, int kind, int width, int height) {
  if (width <= 0 || height <= 0 || kind < 0 || kind > RMD_HIP_KIND_F32X2)
    return fail(RMD_HIP_ERR_INVALID_ARG, "image: bad kind/size (%d, %dx%d)", kind, width, height);
  const size_t es = kind_size(kind);
  // rows padded to 256 B so that every row starts on a full HBM burst / 64-lane dword access
  const size_t pitch = (static_cast<size_t>(width) * es + 255) / 256 * 256;
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, pitch * height));
  // hipMemset runs on the null stream; the handles' kernels run on non-blocking streams that do NOT order against it,
  // so the fill must have completed before the buffer is handed out
  HIP_TRY(hipMemset(p, 0, pitch * height));
  HIP_TRY(hipStreamSynchronize(nullptr));
  img->kind = kind; img->width = width; img->height = height;
  img->pitch = pitch; img->stride = pitch / es; img->data = p; img->owns = true;
  (void)hipGetDevice(&img->device);
  return RMD_HIP_OK;
}

}  // namespace

// The "frame n has arrived" word that follows every staged host frame on the copy stream.  A pinned -> device copy of up to 16 KB
// is carried out by a shader kernel of the runtime (__amd_rocclr_copyBuffer, tools/ubench/copy_path.hip), which needs wave slots of
// its own: behind a thousand persistent search workgroups it ran 7 us on average and up to 98 us, and with two stream groups of a
// batch overlapping there is hardly ever a gap for it (8-bit host frames, batch of 8: 8 700 instead of 12 000 Mpix/s).  From 64 KB on
// a copy goes to the SDMA engine like the frame itself.  So the flag is 64 KB of the same number: whichever of its words the engine
// writes first or last, a reader of word 0 sees either the old number or the new one, and the new one only after the frame copy in
// front of it (same stream) has completed.  The 64 KB form costs the copy engine 7 us more per frame, all of it latency when the
// device is waiting for the frame (live use: the node asks for the converged count after every update, so the host never runs ahead):
// there the 4-byte form is used -- the device has caught up, so the shader copy finds an empty chip (flag_words()).
// How a frame that was handed over in host memory reaches the device.  The caller's buffer is always copied into a pinned ring first (the
// caller may reuse it when update() returns); from there
//   staged    the copy engine brings it into a staging buffer in HBM, followed by its arrival flag, with no ordering against the compute
//             stream; the ingest workgroups of the frame's setup kernel wait for the flag and convert the frame into the current-image
//             plane (rmdk::MatcherArgs);
//   in place  the ingest workgroups read the pinned ring themselves over the host link: no copy engine, no staging, no flag.
// "+ ahead" (single sequences): frame n + 1 is converted during the SEARCH kernel of frame n whenever it has arrived (staged) or been
// handed over (in place) by the time setup n runs (rmdk::MatcherArgs::ahead); setup n + 1 then finds nothing left to do.
// Defaults (measured, profiles/r03_h2d.txt): a single sequence uses staged + ahead -- the copy engine does not touch the CUs, whereas
// link reads issued by a CU delay the loads of the workgroups it shares its memory pipeline with (in place + ahead: search +4.5 us per
// update); a batch uses in place -- its setup kernels are long enough to hide the link time and the copy engine's 25-30 us of fixed
// cost per copy is what bounds a step of 4..8 frames (8 sequences: 10 470 -> 12 740 Mpix/s).  Frames that go through the lens-
// undistortion maps are always staged, without ahead (the remap gathers single bytes).  A/B: RMD_HIP_HOST_FRAMES =
// staged | staged_ahead | inplace | inplace_ahead.
enum { HOST_FRAMES_DEFAULT = -1, HOST_FRAMES_STAGED = 0, HOST_FRAMES_STAGED_AHEAD = 1, HOST_FRAMES_INPLACE = 2, HOST_FRAMES_INPLACE_AHEAD = 3 };
static int host_frames_mode(bool batch) {
  static const int forced = [] {
    const char* e = getenv("RMD_HIP_HOST_FRAMES");
    if (e && !strcmp(e, "staged")) return static_cast<int>(HOST_FRAMES_STAGED);
    if (e && !strcmp(e, "staged_ahead")) return static_cast<int>(HOST_FRAMES_STAGED_AHEAD);
    if (e && !strcmp(e, "inplace")) return static_cast<int>(HOST_FRAMES_INPLACE);
    if (e && !strcmp(e, "inplace_ahead")) return static_cast<int>(HOST_FRAMES_INPLACE_AHEAD);
    return static_cast<int>(HOST_FRAMES_DEFAULT);
  }();
  if (forced != HOST_FRAMES_DEFAULT) return forced;
  return batch ? HOST_FRAMES_INPLACE : HOST_FRAMES_STAGED_AHEAD;
}
static bool frame_in_place(bool batch, bool remap) {
  const int m = host_frames_mode(batch);
  return !remap && (m == HOST_FRAMES_INPLACE || m == HOST_FRAMES_INPLACE_AHEAD);
}
static bool frame_ahead(bool remap) {
  const int m = host_frames_mode(false);
  return !remap && (m == HOST_FRAMES_STAGED_AHEAD || m == HOST_FRAMES_INPLACE_AHEAD);
}
constexpr int AHEAD_WGS = 128;  // workgroups of the search kernel that bring the next frame in (MatcherArgs::ahead)

constexpr size_t FLAG_WORDS = 16384;
static void fill_flag_block(unsigned int* block, unsigned int n, size_t words) {
  for (size_t i = 0; i < words; ++i) block[i] = n;
}
// how many words of the flag block to send behind frame `n`: 1 when the device has already started the previous frame (it is waiting for
// this one, or about to), the whole block while the host runs ahead of it
static size_t flag_words(const unsigned int* h_progress, unsigned int n) {
  const unsigned int started = *static_cast<const volatile unsigned int*>(h_progress);
  return static_cast<int>(started - (n - 1u)) >= 0 ? 1 : FLAG_WORDS;
}
constexpr size_t FLAG_ALLOC_BYTES = FLAG_WORDS * sizeof(unsigned int) < 64 ? 64 : FLAG_WORDS * sizeof(unsigned int);
constexpr size_t FLAG_SLOT_WORDS = FLAG_WORDS < 16 ? 16 : FLAG_WORDS;  // pinned source blocks, one per slot, a cache line apart

// Streams that must run CONCURRENTLY are created on different priority levels.  The runtime keeps one pool of (at most four) hardware
// queues per priority level and maps a new stream to the least-used queue of its level's pool; once a process holds more than four
// streams of one level, two of them share a hardware queue and their kernels serialise -- a handle's copy stream behind its compute
// stream (measured: 3300 instead of 4630 Mpix/s with host frames), or the two stream groups of a batch behind each other (93 instead of
// 59 us per step for two sequences).  Streams of different levels never share a queue.  level 0: compute (normal priority), 1: a
// batch's second stream group (high), 2: copy streams and a batch's third stream group (low; a batch reads its host frames in place and
// leaves its copy stream idle).
static hipError_t create_stream(hipStream_t* out, int level) {
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest)
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
  if (level == 2)
    if (const char* e = getenv("RMD_HIP_COPY_STREAM_LEVEL")) level = atoi(e);  // (A/B)
  const int prio = level == 0 ? (least + greatest) / 2 : level == 1 ? greatest : least;  // numerically smaller = higher priority
  return hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio);
}

static unsigned long g_progress_timeouts = 0;  // diagnostics (RMD_HIP_INGEST_PROFILE): waits for a staging slot that gave up after 2 ms
static double g_progress_max_wait_us = 0.0;    // ... and the longest such wait

// ---- rmd::SeedMatrix ------------------------------------------------------------------------
struct rmd_hip_seeds {
  int width = 0, height = 0, patch_side = 5, device = 0, num_cus = 256;
  rmd_hip_image planes[RMD_HIP_NUM_PLANES];
  rmdk::SeedParams P;
  rmdk::Pose T_world_ref;
  float dist_from_ref = 0.0f;
  bool has_reference = false;
  hipStream_t stream = nullptr;
  unsigned long long* d_scalars = nullptr;  // [0] count result, [1..16] diagnostics
  unsigned long long* h_scalars = nullptr;  // pinned mirror
  // (unit target 2: a single sequence's search is a latency chain with a tail; units of half the size shorten the tail now that a unit's staging is cheap:
  // 45.4 -> 44.3 us per update, profiles/r04_unit_target.txt; a batch keeps 1, its tails are filled by the other stream groups)
  int opt_matcher = 3, opt_timing = 0, opt_stats = 0, opt_window = 0, opt_local_max = 0, opt_unit_rounds = 0, opt_unit_target = 2;
  // a SeedMatrix that is a member of a batch (rmd_hip_batch_*) shares the batch's streams and update workspace: its update
  // kernels are launched by the batch, for all members at once; everything else (reference frames, observers) works per member
  struct rmd_hip_batch* batch = nullptr;
  int seq = 0;                              // index in its group's workspace (0 for a plain SeedMatrix)
  int batch_index = 0;                      // index in the batch
  rmdk::MatcherWorkspace* mws = nullptr;    // the update workspace: &matcher_ws, or the batch's
  bool async_count_valid = false;           // the pinned CONVERGED count of the workspace belongs to this handle's latest update ...
  unsigned int async_number = 0;            // ... which carried this number
  hipEvent_t region_start = nullptr, region_stop = nullptr;
  long region_updates = 0;
  // deferred finalisation of the last tile-pipeline update (see rmd_matcher.hpp): pending until the next update()
  // fuses it or an observer forces it
  bool finalize_pending = false;
  rmdk::SeedParams P_pending;
  int opt_lazy = 1;
  // 8-bit ingest: two pinned staging buffers + two device byte planes, used alternately so that the host-side copy of
  // frame k+1 overlaps the device work of frame k; an event per slot says when its H2D copy has been consumed
  static constexpr int SLOTS = 3;           // frames in flight between the host and the update kernels
  unsigned char* h_u8[SLOTS] = {};
  unsigned char* d_u8[SLOTS] = {};
  float* h_f32[SLOTS] = {};                 // pinned staging of float frames (update / set_reference with host pointers)
  hipEvent_t staged[SLOTS] = {};            // copy stream: the slot's frame is in its f32 plane
  hipEvent_t frame_done[SLOTS] = {};        // compute stream: the update that read the slot's plane has run
  hipStream_t copy_stream = nullptr;        // frame uploads and conversions run here, beside the previous frames' kernels
  void* cur_planes[SLOTS] = {};             // current-image planes (pitch of planes[CURR_IMG]), used in rotation; [0] is the handle's own
  int u8_pitch = 0, ingest_slot = 0;
  // fused ingest (tile pipeline): the caller's frame is copied into one of SLOTS pinned buffers, a copy engine moves it to a staging
  // buffer in HBM and then writes the frame's number next to it (copy stream); the setup kernel waits for that number itself,
  // converts the frame into the current-image plane and tells the host through `h_progress` which frames it has consumed.
  // No events, no cross-stream waits: neither queue ever holds a barrier packet for the other.
  // RING slots: the caller may be RING - 1 frames ahead of the setup kernel that has started last.  With three, frame n was handed over
  // when setup n - 2 started and reached HBM 55-60 us later (host copy, submission, 35-45 us of copy engine) -- after setup n - 1 had
  // looked for it, so it was rarely converted one step ahead (rmdk::MatcherArgs::ahead); with four it always is.
  static constexpr int RING = 4;
  unsigned char* h_zc_u8[RING] = {};
  float* h_zc_f32[RING] = {};
  unsigned char* d_zc_u8[RING] = {};
  float* d_zc_f32[RING] = {};
  unsigned int* h_seq = nullptr;            // pinned, one block per slot: the frame number the copy stream writes into d_zc_flag
  unsigned int* d_zc_flag = nullptr;        // device, one block per ring slot: number of the last frame whose copy into that slot's staging buffer has completed
  unsigned int* h_submitted = nullptr;      // pinned, [kind * RING + slot]: number of the newest 8-bit (kind 0) / float (kind 1) frame that is complete in that ring slot (frames read in place, one step ahead)
  unsigned int* d_ahead = nullptr;          // device: the words of rmdk::MatcherArgs::ahead
  int pack_backoff = 0, pack_backoff_len = 15;  // float frames that are not 8-bit levels: the next pack_backoff_len frames are not examined (pack_float_rows_u8)
  unsigned int* h_progress = nullptr;       // pinned: [0] number of the ingested frame whose setup kernel has started, [1] error bits
  unsigned long long zc_number = 0;         // ingested frames so far (the device sees the low 32 bits and compares modulo 2^32)
  int opt_fused_ingest = 1;                 // RMD_HIP_FUSED_INGEST=0 switches back to the copy-stream pipeline (A/B)
  bool ingest_ready = false;                // ingest_init has run
  bool inject_withhold_flag = false;        // test hook (RMD_HIP_OPT_INJECT_FAULT): the arrival flag of the next staged host frame is not sent
  double ingest_us[4] = {0, 0, 0, 0};  // diagnostics (RMD_HIP_INGEST_PROFILE): host time waiting for a slot, copying, submitting; frames
  bool ingest_profile = false, ingest_host_wait = false;
  StageTimer timers[RMD_HIP_NUM_SEED_STAGES];
  long long last_stats[16] = {0};
  bool stats_pending = false;
  long long trace_frame = 0;  // updates launched since timeline tracing was switched on
  short2* d_undist_map1 = nullptr;          // lens undistortion (initUndistortionMap): source pixel per destination pixel
  unsigned short* d_undist_map2 = nullptr;  // ... and its 5-bit fractions; null = frames are used as they come
  std::vector<short> h_undist_map1;
  std::vector<unsigned short> h_undist_map2;
  unsigned char* d_bgr = nullptr;       // coloured convergence map (allocated at the first request): W x H x 3 bytes on the device ...
  unsigned char* h_bgr = nullptr;       // ... and their pinned landing buffer
  unsigned int* d_pc_counts = nullptr;  // point cloud (allocated at the first request): per-block counts / offsets, [n_blocks] = total
  float4* d_pc_points = nullptr;        // W x H points
  rmdk::MatcherWorkspace matcher_ws;
#ifdef RMD_AB_MATCHERS
  rmdk::FrameWorkspace frame_ws;
#endif
};

// rmd_hip_batch_*: up to rmdk::MAX_BATCH SeedMatrix objects of one size whose update() calls are issued together, as ONE launch pair
struct rmd_hip_batch {
  int n = 0, device = 0, num_cus = 256;
  rmd_hip_seeds* members[rmdk::MAX_BATCH] = {};
  // The members are stepped in up to three GROUPS, each with its own stream (one per priority level, see create_stream) and update
  // workspace: one launch pair per group and step, the pairs issued back to back on their streams.  While one group's setup kernel runs
  // its latency chain, or its search is down to its last units, the other groups' kernels fill the chip (four sequences: one group
  // 10 800 Mpix/s, two 11 940, three 12 460; a fourth group would share a hardware-queue pool with the first: 8 860).
  struct Group {
    hipStream_t stream = nullptr;
    rmdk::MatcherWorkspace ws;
    int first = 0, n = 0;                   // members [first, first + n)
    unsigned int* h_progress = nullptr;     // pinned: [0] step whose setup kernel has started, [1] error bits (see ingest_current_fused)
    unsigned long long slot_step[3] = {0, 0, 0};  // host frames: the step of this group's last launch that read staging slot k (0: none)
    unsigned long long last_step = 0;       // ... and of its last launch altogether
    hipEvent_t ev = nullptr;                // fork / join of the region timer
  };
  static constexpr int MAX_GROUPS = 4;
  int n_groups = 0;
  Group groups[MAX_GROUPS];
  hipStream_t copy_stream = nullptr;
  // frames handed over in host memory: the frames of one step, back to back, through SLOTS pinned buffers and SLOTS staging buffers
  // (see ingest_current_fused: the same protocol, one sequence number per step)
  static constexpr int SLOTS = 3;
  unsigned char* h_stage[SLOTS] = {};
  unsigned char* d_stage[SLOTS] = {};
  size_t stage_bytes = 0;                   // capacity of each of the buffers above
  unsigned int* h_seq = nullptr;
  unsigned int* d_flag = nullptr;
  unsigned long long step_number = 0;
  int opt_timing = 0, opt_unit_target = 1;
  int pack_backoff = 0;
  hipEvent_t region_start = nullptr, region_stop = nullptr;
  long region_updates = 0;
  // TV-L1 for all members in one launch sequence (rmd_hip_batch_denoise), allocated at the first request: the denoiser's planes hold the
  // members back to back (one tall image each: member z = rows [z * height, (z + 1) * height)), the members' input planes come from `table`
  struct Denoise {
    rmd_hip_image u[2], u_head[2], p[2], g;
    unsigned long long* d_table = nullptr;   // rmdk::TV_MEMBER_WORDS words per member (device)
    float* h_staging = nullptr;              // pinned, n x W x H
    hipStream_t stream = nullptr;
    rmd_hip_image result[rmdk::MAX_BATCH];   // views of the members' results of the last run
    int result_index = 0;
    double total_ms = 0.0;
    long launches = 0;
    bool ready = false;
  } dn;
  Group& group_of(int member) {
    int g = 0;
    while (g + 1 < n_groups && member >= groups[g + 1].first) ++g;
    return groups[g];
  }
};

namespace {

bool side_supported(int s) { return s == 3 || s == 5 || s == 7 || s == 9; }

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

// the deferred finalisation of this handle's last update, as a kernel of its own (an observer is about to look at the state)
int seeds_flush(rmd_hip_seeds* s) {
  if (s->finalize_pending) {
    s->finalize_pending = false;
    HIP_TRY(rmdk::launch_seed_finalize(s->P_pending, *s->mws, s->stream, s->seq));
  }
  return RMD_HIP_OK;
}

// the error word the setup kernel's ingest workgroups raise when a staging copy never arrived: reported ONCE (the frames of that update are
// invalid), then cleared, so that the handle is usable again from the next reference frame on
int ingest_error_check(unsigned int* h_progress) {
  if (h_progress && h_progress[1] != 0u) {
    h_progress[1] = 0u;
    return fail(RMD_HIP_ERR_RUNTIME, "seed update: the staging copy of a host frame did not complete within the kernel's bounded wait (about "
                                     "0.1 s); the seed state is invalid until the next setReferenceImage");
  }
  return RMD_HIP_OK;
}

// every observer of the seed state goes through here: settle deferred work, then wait for the stream
int seeds_sync(const rmd_hip_seeds* s) {
  rmd_hip_seeds* m = const_cast<rmd_hip_seeds*>(s);
  TRY(seeds_flush(m));
#ifdef RMD_AB_MATCHERS
  if (m->frame_ws.frame > 0) HIP_TRY(hipMemcpyAsync(m->frame_ws.h_error, m->frame_ws.d_error, sizeof(unsigned int), hipMemcpyDeviceToHost, s->stream));
#endif
  HIP_TRY(hipStreamSynchronize(s->stream));
  TRY(ingest_error_check(m->batch ? m->batch->group_of(m->batch_index).h_progress : m->h_progress));
#ifdef RMD_AB_MATCHERS
  if (m->frame_ws.h_error && m->frame_ws.h_error[0] != 0u) {
    const unsigned int bits = m->frame_ws.h_error[0];
    m->frame_ws.h_error[0] = 0u;
    (void)hipMemsetAsync(m->frame_ws.d_error, 0, sizeof(unsigned int), s->stream);
    return fail(RMD_HIP_ERR_RUNTIME, "seed update: a bounded wait inside the frame kernel ran out (error bits 0x%x); results are invalid", bits);
  }
#endif
  for (auto& t : m->timers) t.drain();
  if (m->stats_pending) {
    for (int k = 0; k < 16; ++k) m->last_stats[k] = static_cast<long long>(m->h_scalars[1 + k]);
    m->stats_pending = false;
  }
  return RMD_HIP_OK;
}

int seeds_bind_device(const rmd_hip_seeds* s) {
  int cur = -1;
  HIP_TRY(hipGetDevice(&cur));
  if (cur != s->device) HIP_TRY(hipSetDevice(s->device));
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

// per-handle part of a host frame that the setup kernel's ingest workgroups bring in (see ingest_current_fused / batch_update_host)
struct PendingIngest {
  rmdk::IngestArgs common;
  const unsigned int* u8 = nullptr;
  const float* f32 = nullptr;
  const void* next_src = nullptr;  // one step ahead: the next frame's place in the ring and its plane
  bool no_remap = false;           // a float frame that travels as bytes: never through the undistortion maps
  float* next_dst = nullptr;
};

int seeds_launch_update(rmd_hip_seeds* s, const PendingIngest* ingest = nullptr) {
  if (s->batch) return fail(RMD_HIP_ERR_INVALID_ARG, "update: this SeedMatrix is a member of a batch; its updates are issued with rmd_hip_batch_update*");
  rmdk::SeedParams P = s->P;
  P.stats = nullptr;
  P.trace = nullptr;
  if (s->opt_stats == 1) {
    HIP_TRY(hipMemsetAsync(s->d_scalars + 1, 0, 16 * sizeof(unsigned long long), s->stream));
    P.stats = s->d_scalars + 1;
  } else if (s->opt_stats == 2 && s->opt_matcher == 1 && s->matcher_ws.d_trace) {  // timeline probes only: the pipeline runs as in production
    P.trace = s->matcher_ws.d_trace + static_cast<size_t>(s->trace_frame % rmdk::TRACE_FRAMES) * s->matcher_ws.trace_slice_u64();
    ++s->trace_frame;
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
#ifdef RMD_AB_MATCHERS
      } else if (s->opt_matcher == 2) {
        TRY(seeds_flush(s));
        unsigned long long* slice = nullptr;
        if (s->opt_stats == 2 && s->frame_ws.d_trace) {  // timeline probes
          slice = s->frame_ws.d_trace + static_cast<size_t>(s->trace_frame % rmdk::FR_TRACE_FRAMES) * s->frame_ws.trace_slice_u64();
          ++s->trace_frame;
        }
        HIP_TRY(rmdk::launch_seed_frame<SIDE>(P, s->frame_ws, s->stream, s->num_cus, s->opt_local_max, s->opt_unit_rounds, slice));
      } else if (s->opt_matcher == 1) {
        const bool fuse = s->finalize_pending;
        const rmdk::Pose T_prev = fuse ? s->P_pending.T_ref_curr : P.T_ref_curr;
        HIP_TRY(rmdk::launch_seed_update_tile<SIDE>(P, s->matcher_ws, s->stream, s->num_cus, s->opt_window, fuse, T_prev));
        s->P_pending = P;
        s->P_pending.stats = nullptr;
        s->P_pending.trace = nullptr;
        s->finalize_pending = true;
        if (!s->opt_lazy || s->opt_stats == 1) TRY(seeds_flush(s));
#endif
      } else {
        rmdk::SeedParams Pt = P;
        if (s->opt_stats == 2 && s->matcher_ws.d_wg_trace) {  // timeline probes of the setup tiles and the search workgroups
          Pt.trace = s->matcher_ws.d_wg_trace + static_cast<size_t>(s->trace_frame % rmdk::FR_TRACE_FRAMES) * s->matcher_ws.wg_trace_slice_u64();
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
        HIP_TRY((rmdk::launch_seed_pipeline_compact<SIDE, 1>(B, 1, s->matcher_ws, s->stream, s->num_cus, s->opt_unit_target, ingest ? &ingest->common : nullptr)));
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

// common tail of setReferenceImage (seed_matrix.cu:95-113) once the frame is in planes[REF_IMG]
int seeds_after_reference(rmd_hip_seeds* s, const float* T_curr_world, float min_depth, float max_depth) {
  TRY(seeds_flush(s));  // a deferred finalisation of the old reference must not run after the re-initialisation
  rmdk::SeedParams& P = s->P;
  P.avg_depth = (min_depth + max_depth) / 2.0f;
  P.depth_range = max_depth - min_depth;
  P.sigma_sq_max = P.depth_range * P.depth_range / 36.0f;
  P.eta_inlier = 0.7f;
  P.eta_outlier = 0.05f;
  P.epsilon = P.depth_range / 1000.0f;
  rmdk::Pose T;
  memcpy(T.d, T_curr_world, sizeof(T.d));
  s->T_world_ref = pose_inverse(T);
  TRY(seeds_launch_init(s));
  s->async_count_valid = false;
  // the pipeline's load statistics (unit size from the previous frame's work) belong to the old sequence; in a batch the other
  // members' next frame simply starts from the largest unit size again
#ifdef RMD_AB_MATCHERS
  s->frame_ws.frame = 0;
  HIP_TRY(hipMemsetAsync(s->frame_ws.d_ctl, 0, 3 * rmdk::FR_CTL_WORDS * sizeof(unsigned int), s->stream));
#endif
  s->mws->frame = 0;
  HIP_TRY(hipMemsetAsync(s->mws->d_shards, 0, 3 * rmdk::UNIT_SHARDS * sizeof(unsigned long long), s->stream));
  s->has_reference = true;
  // the reference synchronises here (seed_matrix.cu:113); so do we: the host image is borrowed
  return seeds_sync(s);
}

// common tail of update (seed_matrix.cu:124-157) once the frame is in planes[CURR_IMG]
void seeds_frame_pose(rmd_hip_seeds* s, const float* T_curr_world) {
  rmdk::Pose T;
  memcpy(T.d, T_curr_world, sizeof(T.d));
  const rmdk::Pose T_curr_ref = pose_compose(T, s->T_world_ref);
  const float tx = T_curr_ref.d[3], ty = T_curr_ref.d[7], tz = T_curr_ref.d[11];
  s->dist_from_ref = sqrtf(tx * tx + ty * ty + tz * tz);
  s->P.T_curr_ref = T_curr_ref;
  s->P.T_ref_curr = pose_inverse(T_curr_ref);
}
int seeds_after_frame(rmd_hip_seeds* s, const float* T_curr_world, const PendingIngest* ingest = nullptr) {
  seeds_frame_pose(s, T_curr_world);
  return seeds_launch_update(s, ingest);
}

}  // namespace

// ---- rmd::DepthmapDenoiser ------------------------------------------------------------------
struct rmd_hip_denoiser {
  int width = 0, height = 0, device = 0;
  rmd_hip_image u[2], u_head[2], p[2], g;
  float L, tau, sigma, theta;
  float large_sigma_sq = -1.0f;
  hipStream_t stream = nullptr;
  int result_index = 0;
  float* h_staging = nullptr;  // pinned, W x H: device -> pinned (async DMA) -> caller's pageable buffer
  int opt_timing = 0, opt_iters_per_launch = 0, opt_geometry = 0;
  StageTimer timer;
};

extern "C" {

const char* rmd_hip_last_error(void) { return g_last_error; }
int rmd_hip_version(void) { return RMD_HIP_VERSION_NUMBER; }

// ---- device selection (check_cuda_device.cu:23-117) -------------------------------------------
int rmd_hip_device_count(int* count) {
  if (!count) return fail(RMD_HIP_ERR_INVALID_ARG, "device_count: null output");
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    *count = 0;
    return fail(RMD_HIP_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(e));
  }
  *count = n;
  return RMD_HIP_OK;
}
int rmd_hip_set_device(int device_id) {
  int n = 0;
  TRY(rmd_hip_device_count(&n));
  if (device_id < 0 || device_id >= n)
    return fail(RMD_HIP_ERR_INVALID_ARG, "invalid device id %d: specify a value in [0, %d]", device_id, n - 1);
  HIP_TRY(hipSetDevice(device_id));
  return RMD_HIP_OK;
}
int rmd_hip_device_name(int device_id, char* buf, size_t buf_len) {
  if (!buf || buf_len == 0) return fail(RMD_HIP_ERR_INVALID_ARG, "device_name: null buffer");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  snprintf(buf, buf_len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return RMD_HIP_OK;
}

// ---- DeviceImage ----------------------------------------------------------------------------
int rmd_hip_image_create(int kind, int width, int height, rmd_hip_image_t** out) {
  if (!out) return fail(RMD_HIP_ERR_INVALID_ARG, "image_create: null output");
  rmd_hip_image* img = new (std::nothrow) rmd_hip_image();
  if (!img) return fail(RMD_HIP_ERR_RUNTIME, "image_create: out of host memory");
  const int rc = image_alloc(img, kind, width, height);
  if (rc != RMD_HIP_OK) {
    delete img;
    return rc;
  }
  *out = img;
  return RMD_HIP_OK;
}
int rmd_hip_image_destroy(rmd_hip_image_t* img) {
  if (!img) return RMD_HIP_OK;
  ScopedDevice dev(img->device);
  if (img->owns && img->data) (void)hipFree(img->data);  // destructors must not throw (device_image.cuh:124-132 does)
  delete img;
  return RMD_HIP_OK;
}
int rmd_hip_image_upload(rmd_hip_image_t* img, const void* host) {
  if (!img || !host) return fail(RMD_HIP_ERR_INVALID_ARG, "image_upload: null argument");
  TRY(image_settle(img));
  ScopedDevice dev(img->device);
  const size_t row = static_cast<size_t>(img->width) * kind_size(img->kind);
  HIP_TRY(hipMemcpy2D(img->data, img->pitch, host, row, row, img->height, hipMemcpyHostToDevice));
  return RMD_HIP_OK;
}
int rmd_hip_image_download(const rmd_hip_image_t* img, void* host) {
  if (!img || !host) return fail(RMD_HIP_ERR_INVALID_ARG, "image_download: null argument");
  TRY(image_settle(img));
  ScopedDevice dev(img->device);
  const size_t row = static_cast<size_t>(img->width) * kind_size(img->kind);
  HIP_TRY(hipMemcpy2D(host, row, img->data, img->pitch, row, img->height, hipMemcpyDeviceToHost));
  return RMD_HIP_OK;
}
int rmd_hip_image_zero(rmd_hip_image_t* img) {
  if (!img) return fail(RMD_HIP_ERR_INVALID_ARG, "image_zero: null image");
  TRY(image_settle(img));
  ScopedDevice dev(img->device);  // the null stream that is waited on below is the image's device's
  HIP_TRY(hipMemset(img->data, 0, img->pitch * img->height));
  HIP_TRY(hipStreamSynchronize(nullptr));
  return RMD_HIP_OK;
}
int rmd_hip_image_copy(rmd_hip_image_t* dst, const rmd_hip_image_t* src) {
  if (!dst || !src) return fail(RMD_HIP_ERR_INVALID_ARG, "image_copy: null argument");
  if (dst == src) return RMD_HIP_OK;
  if (dst->kind != src->kind || dst->width != src->width || dst->height != src->height)
    return fail(RMD_HIP_ERR_INVALID_ARG, "image_copy: shape mismatch");
  TRY(image_settle(src));
  TRY(image_settle(dst));
  ScopedDevice dev(dst->device);
  const size_t row = static_cast<size_t>(src->width) * kind_size(src->kind);
  HIP_TRY(hipMemcpy2D(dst->data, dst->pitch, src->data, src->pitch, row, src->height, hipMemcpyDeviceToDevice));
  return RMD_HIP_OK;
}
int rmd_hip_image_info(const rmd_hip_image_t* img, int* kind, int* width, int* height, size_t* pitch_bytes,
                       size_t* stride_elems, void** device_data) {
  if (!img) return fail(RMD_HIP_ERR_INVALID_ARG, "image_info: null image");
  if (kind) *kind = img->kind;
  if (width) *width = img->width;
  if (height) *height = img->height;
  if (pitch_bytes) *pitch_bytes = img->pitch;
  if (stride_elems) *stride_elems = img->stride;
  if (device_data) *device_data = img->data;
  return RMD_HIP_OK;
}

// ---- SeedMatrix -----------------------------------------------------------------------------
static int seeds_destroy_impl(rmd_hip_seeds* s);
int rmd_hip_seeds_destroy(rmd_hip_seeds_t* s) {
  if (s && s->batch) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_destroy: this SeedMatrix belongs to a batch (rmd_hip_batch_destroy releases it)");
  return seeds_destroy_impl(s);
}
static int seeds_destroy_impl(rmd_hip_seeds* s) {
  if (!s) return RMD_HIP_OK;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (auto& t : s->timers) t.destroy();
  if (s->copy_stream) (void)hipStreamSynchronize(s->copy_stream);
  if (s->ingest_profile && s->ingest_us[3] > 0)
    fprintf(stderr, "[rmd_hip ingest] %.0f frames: wait for slot %.2f us, host copy %.2f us, submit %.2f us per frame; longest wait %.0f us, %lu waits gave up after 2 ms\n",
            s->ingest_us[3], s->ingest_us[0] / s->ingest_us[3], s->ingest_us[1] / s->ingest_us[3], s->ingest_us[2] / s->ingest_us[3], g_progress_max_wait_us,
            g_progress_timeouts);
  for (int k = 0; k < rmd_hip_seeds::RING; ++k) {
    if (s->h_zc_u8[k]) (void)hipHostFree(s->h_zc_u8[k]);
    if (s->h_zc_f32[k]) (void)hipHostFree(s->h_zc_f32[k]);
    if (s->d_zc_u8[k]) (void)hipFree(s->d_zc_u8[k]);
    if (s->d_zc_f32[k]) (void)hipFree(s->d_zc_f32[k]);
  }
  for (int k = 0; k < rmd_hip_seeds::SLOTS; ++k) {
    if (s->h_u8[k]) (void)hipHostFree(s->h_u8[k]);
    if (s->h_f32[k]) (void)hipHostFree(s->h_f32[k]);
    if (s->d_u8[k]) (void)hipFree(s->d_u8[k]);
    if (s->staged[k]) (void)hipEventDestroy(s->staged[k]);
    if (s->frame_done[k]) (void)hipEventDestroy(s->frame_done[k]);
    if (k > 0 && s->cur_planes[k]) (void)hipFree(s->cur_planes[k]);  // [0] belongs to planes[]
  }
  if (s->cur_planes[0]) s->planes[RMD_HIP_PLANE_CURR_IMG].data = s->cur_planes[0];
  if (s->copy_stream && !s->batch) (void)hipStreamDestroy(s->copy_stream);
  if (s->region_start) (void)hipEventDestroy(s->region_start);
  if (s->region_stop) (void)hipEventDestroy(s->region_stop);
  for (auto& pl : s->planes)
    if (pl.owns && pl.data) (void)hipFree(pl.data);
  s->matcher_ws.release();
#ifdef RMD_AB_MATCHERS
  s->frame_ws.release();
#endif
  if (s->d_undist_map1) (void)hipFree(s->d_undist_map1);
  if (s->d_undist_map2) (void)hipFree(s->d_undist_map2);
  if (s->d_bgr) (void)hipFree(s->d_bgr);
  if (s->h_bgr) (void)hipHostFree(s->h_bgr);
  if (s->d_pc_counts) (void)hipFree(s->d_pc_counts);
  if (s->d_pc_points) (void)hipFree(s->d_pc_points);
  if (s->d_scalars) (void)hipFree(s->d_scalars);
  if (s->h_scalars) (void)hipHostFree(s->h_scalars);
  if (s->h_progress) (void)hipHostFree(s->h_progress);
  if (s->h_seq) (void)hipHostFree(s->h_seq);
  if (s->d_zc_flag) (void)hipFree(s->d_zc_flag);
  if (s->h_submitted) (void)hipHostFree(s->h_submitted);
  if (s->d_ahead) (void)hipFree(s->d_ahead);
  if (s->stream && !s->batch) (void)hipStreamDestroy(s->stream);
  delete s;
  return RMD_HIP_OK;
}

static int ingest_init(rmd_hip_seeds* s);
static int seeds_create_impl(int width, int height, float fx, float fy, float cx, float cy, int patch_side, int max_extent, rmd_hip_batch* batch, int seq,
                             rmd_hip_seeds_t** out);

int rmd_hip_seeds_create(int width, int height, float fx, float fy, float cx, float cy, int patch_side, int max_extent,
                         rmd_hip_seeds_t** out) {
  return seeds_create_impl(width, height, fx, fy, cx, cy, patch_side, max_extent, nullptr, 0, out);
}

// batch != null: member `seq` of that batch -- the batch's streams and update workspace instead of its own
static int seeds_create_impl(int width, int height, float fx, float fy, float cx, float cy, int patch_side, int max_extent, rmd_hip_batch* batch, int seq,
                             rmd_hip_seeds_t** out) {
  if (!out) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_create: null output");
  *out = nullptr;
  if (width <= 0 || height <= 0) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_create: bad size %dx%d", width, height);
  if (!side_supported(patch_side))
    return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_create: unsupported patch side %d (3, 5, 7, 9)", patch_side);
  if (max_extent <= 0 || max_extent > 100)
    return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_create: max_extent %d outside (0, 100]", max_extent);
  int ndev = 0;
  TRY(rmd_hip_device_count(&ndev));
  rmd_hip_seeds* s = new (std::nothrow) rmd_hip_seeds();
  if (!s) return fail(RMD_HIP_ERR_RUNTIME, "seeds_create: out of host memory");
  s->width = width; s->height = height; s->patch_side = patch_side;
  (void)hipGetDevice(&s->device);
  s->batch = batch; s->batch_index = seq;
  rmd_hip_batch::Group* grp = batch ? &batch->group_of(seq) : nullptr;
  s->seq = batch ? seq - grp->first : 0;
  s->mws = batch ? &grp->ws : &s->matcher_ws;
  auto bail = [&](int rc) { seeds_destroy_impl(s); return rc; };
  if (batch) { s->stream = grp->stream; s->copy_stream = batch->copy_stream; }
  else if (create_stream(&s->stream, 0) != hipSuccess)
    return bail(fail(RMD_HIP_ERR_RUNTIME, "seeds_create: hipStreamCreate failed"));
  for (int p = 0; p < RMD_HIP_NUM_PLANES; ++p) {
    const int kind = p == RMD_HIP_PLANE_CONVERGENCE ? RMD_HIP_KIND_I32
                     : p == RMD_HIP_PLANE_EPIPOLAR_MATCHES ? RMD_HIP_KIND_F32X2 : RMD_HIP_KIND_F32;
    const int rc = image_alloc(&s->planes[p], kind, width, height);
    if (rc != RMD_HIP_OK) return bail(rc);
    s->planes[p].owner_stream = s->stream;
    s->planes[p].owner_seeds = s;
  }
  if (hipMalloc(reinterpret_cast<void**>(&s->d_scalars), 17 * sizeof(unsigned long long)) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&s->h_scalars), 17 * sizeof(unsigned long long)) != hipSuccess)
    return bail(fail(RMD_HIP_ERR_RUNTIME, "seeds_create: scalar buffers"));
  if (hipMemset(s->d_scalars, 0, 17 * sizeof(unsigned long long)) != hipSuccess) return bail(fail(RMD_HIP_ERR_RUNTIME, "seeds_create: scalar buffers"));
  memset(s->h_scalars, 0, 17 * sizeof(unsigned long long));
  rmdk::SeedParams& P = s->P;
  memset(&P, 0, sizeof(P));
  P.w = width; P.h = height;
  P.stride = static_cast<int>(s->planes[RMD_HIP_PLANE_MU].stride);
  P.stride2 = static_cast<int>(s->planes[RMD_HIP_PLANE_EPIPOLAR_MATCHES].stride);
  P.ref = static_cast<const float*>(s->planes[RMD_HIP_PLANE_REF_IMG].data);
  P.cur = static_cast<const float*>(s->planes[RMD_HIP_PLANE_CURR_IMG].data);
  P.cur_stride = P.stride;
  P.sum_templ = static_cast<float*>(s->planes[RMD_HIP_PLANE_SUM_TEMPL].data);
  P.denom = static_cast<float*>(s->planes[RMD_HIP_PLANE_CONST_TEMPL_DENOM].data);
  P.mu = static_cast<float*>(s->planes[RMD_HIP_PLANE_MU].data);
  P.sigma_sq = static_cast<float*>(s->planes[RMD_HIP_PLANE_SIGMA_SQ].data);
  P.a = static_cast<float*>(s->planes[RMD_HIP_PLANE_A].data);
  P.b = static_cast<float*>(s->planes[RMD_HIP_PLANE_B].data);
  P.conv = static_cast<int*>(s->planes[RMD_HIP_PLANE_CONVERGENCE].data);
  P.match = static_cast<float2*>(s->planes[RMD_HIP_PLANE_EPIPOLAR_MATCHES].data);
  P.cam = rmdk::Cam{fx, fy, cx, cy};
  P.one_pix_angle = atan2f(1.0f, 2.0f * fx) * 2.0f;  // pinhole_camera.cuh:56-59
  P.max_extent = static_cast<float>(max_extent);
  if (!batch && s->matcher_ws.allocate(width, height, P.stride) != 0) return bail(fail(RMD_HIP_ERR_RUNTIME, "seeds_create: update workspace"));
  if (batch && (grp->ws.stride != P.stride || grp->ws.tiles_x != (width + rmdk::TILE_W - 1) / rmdk::TILE_W || grp->ws.tiles_y != (height + rmdk::TILE_H - 1) / rmdk::TILE_H))
    return bail(fail(RMD_HIP_ERR_RUNTIME, "seeds_create: the batch's workspace has another geometry"));
#ifdef RMD_AB_MATCHERS
  if (s->frame_ws.allocate(width, height, P.stride) != 0) return bail(fail(RMD_HIP_ERR_RUNTIME, "seeds_create: frame workspace"));
#endif
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, s->device) == hipSuccess && prop.multiProcessorCount > 0) s->num_cus = prop.multiProcessorCount;
  int lds = 0;
  if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, s->device) == hipSuccess && lds > 0) s->mws->lds_bytes = lds;
  if (hipDeviceSynchronize() != hipSuccess) return bail(fail(RMD_HIP_ERR_RUNTIME, "seeds_create: device synchronisation failed"));  // all fills done
  // the copy stream is created right next to the compute stream.  A batch member's updates go through the BATCH's staging buffers and
  // progress words: it gets its own (the staging slots of set_reference*) lazily, at its first host reference frame (ingest_reference)
  if (!batch && ingest_init(s) != RMD_HIP_OK) return bail(RMD_HIP_ERR_RUNTIME);
  *out = s;
  return RMD_HIP_OK;
}

static int ingest_current(rmd_hip_seeds* s, const unsigned char* host_gray, const float* host_f32, const float* T_curr_world);
static int ingest_reference(rmd_hip_seeds* s, const unsigned char* host_gray, const float* host_f32, const float* T_curr_world, float min_depth,
                            float max_depth);

int rmd_hip_seeds_set_reference(rmd_hip_seeds_t* s, const float* host_img, const float* T_curr_world, float min_depth,
                                float max_depth) {
  if (!s || !host_img || !T_curr_world) return fail(RMD_HIP_ERR_INVALID_ARG, "set_reference: null argument");
  TRY(seeds_bind_device(s));
  return ingest_reference(s, nullptr, host_img, T_curr_world, min_depth, max_depth);
}

int rmd_hip_seeds_set_reference_device(rmd_hip_seeds_t* s, const float* dev_img, size_t stride_elems,
                                       const float* T_curr_world, float min_depth, float max_depth) {
  if (!s || !dev_img || !T_curr_world) return fail(RMD_HIP_ERR_INVALID_ARG, "set_reference_device: null argument");
  if (stride_elems < static_cast<size_t>(s->width)) return fail(RMD_HIP_ERR_INVALID_ARG, "set_reference_device: stride < width");
  TRY(seeds_bind_device(s));
  const rmd_hip_image& im = s->planes[RMD_HIP_PLANE_REF_IMG];
  const size_t row = static_cast<size_t>(s->width) * 4;
  HIP_TRY(hipMemcpy2DAsync(im.data, im.pitch, dev_img, stride_elems * 4, row, s->height, hipMemcpyDeviceToDevice, s->stream));
  return seeds_after_reference(s, T_curr_world, min_depth, max_depth);
}

int rmd_hip_seeds_update(rmd_hip_seeds_t* s, const float* host_img, const float* T_curr_world) {
  if (!s || !host_img || !T_curr_world) return fail(RMD_HIP_ERR_INVALID_ARG, "update: null argument");
  if (s->batch) return fail(RMD_HIP_ERR_INVALID_ARG, "update: this SeedMatrix is a member of a batch; its updates are issued with rmd_hip_batch_update*");
  if (!s->has_reference) return fail(RMD_HIP_ERR_NOT_READY, "update: setReferenceImage has not been called");
  TRY(seeds_bind_device(s));
  // the frame is copied into pinned memory here (the caller's buffer is free on return, as after the reference's blocking
  // cudaMemcpy, seed_matrix.cu:128) and uploaded beside the previous frame's kernels; nothing waits for the device
  return ingest_current(s, nullptr, host_img, T_curr_world);
}

int rmd_hip_seeds_update_device(rmd_hip_seeds_t* s, const float* dev_img, size_t stride_elems, const float* T_curr_world) {
  if (!s || !dev_img || !T_curr_world) return fail(RMD_HIP_ERR_INVALID_ARG, "update_device: null argument");
  if (s->batch) return fail(RMD_HIP_ERR_INVALID_ARG, "update_device: this SeedMatrix is a member of a batch; its updates are issued with rmd_hip_batch_update*");
  if (!s->has_reference) return fail(RMD_HIP_ERR_NOT_READY, "update_device: setReferenceImage has not been called");
  if (stride_elems < static_cast<size_t>(s->width)) return fail(RMD_HIP_ERR_INVALID_ARG, "update_device: stride < width");
  TRY(seeds_bind_device(s));
  // zero copy: the kernels read the caller's buffer in place (see the header for the lifetime rule)
  s->P.cur = dev_img;
  s->P.cur_stride = static_cast<int>(stride_elems);
  return seeds_after_frame(s, T_curr_world);
}

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
static int ingest_init(rmd_hip_seeds* s) {
  if (s->ingest_ready) return RMD_HIP_OK;
  if (!s->copy_stream) HIP_TRY(create_stream(&s->copy_stream, 2));  // (a batch member uses the batch's)
  s->ingest_profile = getenv("RMD_HIP_INGEST_PROFILE") != nullptr;
  if (const char* e = getenv("RMD_HIP_INGEST_HOST_WAIT")) s->ingest_host_wait = e[0] == '1';
  if (const char* e = getenv("RMD_HIP_FUSED_INGEST")) s->opt_fused_ingest = e[0] != '0';
  if (const char* e = getenv("RMD_HIP_PACK_BACKOFF")) s->pack_backoff_len = atoi(e);  // (tests: 0 examines every float frame)
  s->ingest_ready = true;
  if (s->batch) {  // a member only ever stages REFERENCE frames (ingest_frame: pinned slot -> plane on the copy stream): events, nothing else
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
  s->h_progress[0] = s->h_progress[1] = 0u;
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_seq), rmd_hip_seeds::RING * FLAG_SLOT_WORDS * sizeof(unsigned int), hipHostMallocDefault));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_zc_flag), 2 * rmd_hip_seeds::RING * FLAG_ALLOC_BYTES));  // one flag block per ring slot and kind of frame
  HIP_TRY(hipMemset(s->d_zc_flag, 0, 2 * rmd_hip_seeds::RING * FLAG_ALLOC_BYTES));
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_submitted), 64, hipHostMallocMapped));
  static_assert(2 * rmd_hip_seeds::RING * sizeof(unsigned int) <= 64, "h_submitted");
  for (int q = 0; q < 2 * rmd_hip_seeds::RING; ++q) s->h_submitted[q] = 0u;
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

// stage a host frame (8-bit gray if host_gray, else float) into `dst` (an f32 plane with the pitch of the SeedMatrix planes);
// on return the compute stream is ordered behind the staging.  dst_is_ref: every earlier update reads that plane.
static double host_now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

static inline void cpu_relax() {  // a polite spin, whatever the host architecture
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#else
  std::this_thread::yield();
#endif
}

// Copy of a caller's frame into a pinned staging buffer.  A float VGA frame is 1.2 MB: one core moves it in ~50 us, which made the host,
// not the GPU, the bound of update(float*) -- the reference's own signature (seed_matrix.cu:120-128).  Frames of 256 KB and more are
// split across a few persistent helper threads (created at the first such copy, parked on a condition variable in between).
namespace {
// Float frames whose every pixel is an 8-bit level -- what the reference's own host path produces: Depthmap::inputImage converts the
// 8-bit camera image with convertTo(CV_32F, 1.0f / 255.0f) and hands the floats to SeedMatrix::update (depthmap.cpp:105, 75-77) -- travel to
// the device as bytes: a quarter of the copy-engine time, which is what bounds float frames (1.2 MB at the engine's 23 GB/s take longer than
// the update).  The device multiplies by the same 1.0f / 255.0f, and a row is only accepted if that product has the caller's BIT PATTERN in
// every pixel, so the current image is the caller's image bit for bit.  Rows [y0, y1) of a w-wide image; false at the first other pixel.
#if defined(__HIP_DEVICE_COMPILE__) || !defined(__x86_64__)
#define RMD_HOST_SIMD_CLONES
#else
#define RMD_HOST_SIMD_CLONES __attribute__((target_clones("avx2", "default")))  // (host pass only: an AVX2 body where the CPU has it)
#endif
RMD_HOST_SIMD_CLONES static bool pack_float_rows_u8(const float* src, unsigned char* dst, int w, int pitch, int y0, int y1) {
  for (int y = y0; y < y1; ++y) {
    const float* in = src + static_cast<size_t>(y) * w;
    unsigned char* out = dst + static_cast<size_t>(y) * pitch;
    unsigned int bad = 0u;
    for (int x = 0; x < w; ++x) {
      const float f = in[x];
      const float c = f >= 0.0f && f <= 1.0f ? f : 2.0f;  // (NaN, negative and large values fail the comparison below)
      const int u = static_cast<int>(c * 255.0f + 0.5f);
      const float back = static_cast<float>(u) * (1.0f / 255.0f);
      unsigned int fb, bb;
      memcpy(&fb, &f, 4);
      memcpy(&bb, &back, 4);
      bad |= (fb ^ bb) | static_cast<unsigned int>(u >> 8);  // (u > 255: the stand-in for values outside [0, 1])
      out[x] = static_cast<unsigned char>(u);
    }
    if (bad) return false;
  }
  return true;
}

class CopyPool {
 public:
  static CopyPool& instance() {
    static CopyPool pool;
    return pool;
  }
  struct Segment { void* dst; const void* src; };
  // several buffers of `bytes` each (the frames of one batch step): the participants take whole buffers in turn
  void copy_many(const Segment* segs, int n, size_t bytes) {
    if (n == 1) { copy(segs[0].dst, segs[0].src, bytes); return; }
    if (n_workers_ == 0 || bytes * static_cast<size_t>(n) < kMinBytes) {
      for (int i = 0; i < n; ++i) memcpy(segs[i].dst, segs[i].src, bytes);
      return;
    }
    std::lock_guard<std::mutex> one_copy_at_a_time(call_mutex_);
    segs_ = segs; n_segs_ = n; bytes_ = bytes;
    post();
    for (int i = n_workers_; i < n; i += n_workers_ + 1) memcpy(segs[i].dst, segs[i].src, bytes);  // the caller's share
    wait();
    segs_ = nullptr; n_segs_ = 0;
  }
  // pack_float_rows_u8 over the rows of one frame, split over the participants; true if every row was accepted
  bool pack(const float* src, unsigned char* dst, int w, int h, int pitch) {
    if (!pack_float_rows_u8(src, dst, w, pitch, h / 2, h / 2 + 1)) return false;  // an image of other floats is turned down before anybody is woken
    if (n_workers_ == 0 || static_cast<size_t>(w) * h * sizeof(float) < kMinBytes) return pack_float_rows_u8(src, dst, w, pitch, 0, h);
    std::lock_guard<std::mutex> one_copy_at_a_time(call_mutex_);
    const int parts = n_workers_ + 1, rows = (h + parts - 1) / parts;
    pack_src_ = src; pack_dst_ = dst; pack_w_ = w; pack_h_ = h; pack_pitch_ = pitch; pack_rows_ = rows;
    __atomic_store_n(&pack_ok_, 1, __ATOMIC_RELAXED);
    post();
    const int y0 = n_workers_ * rows;  // the caller takes the last part
    if (y0 < h && !pack_float_rows_u8(src, dst, w, pitch, y0, h)) __atomic_store_n(&pack_ok_, 0, __ATOMIC_RELAXED);
    wait();
    pack_src_ = nullptr;
    return __atomic_load_n(&pack_ok_, __ATOMIC_ACQUIRE) != 0;
  }
  void copy(void* dst, const void* src, size_t bytes) {
    const int parts = n_workers_ + 1;
    if (n_workers_ == 0 || bytes < kMinBytes) {
      memcpy(dst, src, bytes);
      return;
    }
    std::lock_guard<std::mutex> one_copy_at_a_time(call_mutex_);
    const size_t chunk = ((bytes + parts - 1) / parts + 4095) & ~static_cast<size_t>(4095);
    dst_ = static_cast<char*>(dst); src_ = static_cast<const char*>(src); bytes_ = bytes; chunk_ = chunk;
    post();
    const size_t mine = static_cast<size_t>(n_workers_) * chunk;  // the caller takes the last part
    if (mine < bytes) memcpy(dst_ + mine, src_ + mine, bytes - mine);
    wait();
  }

 private:
  void post() {
    __atomic_store_n(&pending_, n_workers_, __ATOMIC_RELAXED);
    // generation_ / parked_ form a store-buffering (Dekker) handshake -- the poster bumps generation_ then reads parked_, a helper bumps
    // parked_ then reads generation_ --: both sides must be sequentially consistent or each may miss the other (helper asleep, caller
    // spinning in wait() for ever); release / acquire alone only happens to work where locked RMWs are full fences
    __atomic_fetch_add(&generation_, 1ull, __ATOMIC_SEQ_CST);  // publishes the job to the helpers that are polling
    if (__atomic_load_n(&parked_, __ATOMIC_SEQ_CST) != 0) {    // ... and wakes those that went to sleep
      std::lock_guard<std::mutex> lk(m_);
      cv_.notify_all();
    }
  }
  void wait() {
    while (__atomic_load_n(&pending_, __ATOMIC_ACQUIRE) != 0) cpu_relax();  // the helpers' parts take a few microseconds
  }
  static constexpr size_t kMinBytes = 1024 * 1024;  // 8-bit VGA frames (300 KB) are copied by the caller alone
  // A helper does NOT poll for its next job: it sleeps on the condition variable.  Polling helpers (300 us after their last job) made
  // the copy of a float frame three times faster, but four busy threads per stream ran the process into its container's CPU quota on
  // the measurement box: one run in four lost 50 - 70 ms to a throttled thread (RMD_HIP_INGEST_PROFILE: "longest wait 58945 us").
  static constexpr double kPollUs = 0.0;
  CopyPool() {
    int n = 3;
    if (const char* e = getenv("RMD_HIP_COPY_THREADS")) n = atoi(e) - 1;
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw != 0 && static_cast<unsigned>(n + 1) > hw) n = static_cast<int>(hw) - 1;
    if (n < 0) n = 0;
    if (n > 15) n = 15;
    n_workers_ = n;
    for (int i = 0; i < n; ++i) workers_.emplace_back([this, i] { run(i); });
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      __atomic_store_n(&stop_, true, __ATOMIC_RELEASE);
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  void run(int index) {
    unsigned long long seen = 0;
    for (;;) {
      // poll for a new job, then sleep
      const double t0 = host_now_us();
      unsigned int spins = 0;
      while (__atomic_load_n(&generation_, __ATOMIC_ACQUIRE) == seen && !__atomic_load_n(&stop_, __ATOMIC_ACQUIRE)) {
        cpu_relax();
        if ((++spins & 63u) == 0u && host_now_us() - t0 > kPollUs) {
          std::unique_lock<std::mutex> lk(m_);
          __atomic_fetch_add(&parked_, 1, __ATOMIC_SEQ_CST);
          cv_.wait(lk, [&] { return __atomic_load_n(&stop_, __ATOMIC_ACQUIRE) || __atomic_load_n(&generation_, __ATOMIC_SEQ_CST) != seen; });
          __atomic_fetch_sub(&parked_, 1, __ATOMIC_SEQ_CST);
        }
      }
      if (__atomic_load_n(&stop_, __ATOMIC_ACQUIRE)) return;
      seen = __atomic_load_n(&generation_, __ATOMIC_ACQUIRE);
      if (pack_src_) {
        const int y0 = index * pack_rows_, y1 = y0 + pack_rows_ < pack_h_ ? y0 + pack_rows_ : pack_h_;
        if (y0 < y1 && !pack_float_rows_u8(pack_src_, pack_dst_, pack_w_, pack_pitch_, y0, y1)) __atomic_store_n(&pack_ok_, 0, __ATOMIC_RELAXED);
      } else if (segs_) {
        for (int i = index; i < n_segs_; i += n_workers_ + 1) memcpy(segs_[i].dst, segs_[i].src, bytes_);
      } else {
        const size_t off = static_cast<size_t>(index) * chunk_;
        if (off < bytes_) memcpy(dst_ + off, src_ + off, bytes_ - off < chunk_ ? bytes_ - off : chunk_);
      }
      __atomic_fetch_sub(&pending_, 1, __ATOMIC_RELEASE);
    }
  }
  std::vector<std::thread> workers_;
  int n_workers_ = 0;
  std::mutex m_, call_mutex_;
  std::condition_variable cv_;
  bool stop_ = false;
  unsigned long long generation_ = 0;
  int parked_ = 0;
  char* dst_ = nullptr; const char* src_ = nullptr;
  const Segment* segs_ = nullptr;
  int n_segs_ = 0;
  size_t bytes_ = 0, chunk_ = 0;
  const float* pack_src_ = nullptr; unsigned char* pack_dst_ = nullptr;
  int pack_w_ = 0, pack_h_ = 0, pack_pitch_ = 0, pack_rows_ = 0, pack_ok_ = 1;
  int pending_ = 0;
};
// (A/B: RMD_HIP_FLOAT_AS_BYTES=0 sends every float frame as floats)
static bool float_frames_as_bytes() {
  static const bool on = [] { const char* e = getenv("RMD_HIP_FLOAT_AS_BYTES"); return !(e && e[0] == '0'); }();
  return on;
}
}  // namespace
static inline void host_copy(void* dst, const void* src, size_t bytes) { CopyPool::instance().copy(dst, src, bytes); }

static int ingest_frame(rmd_hip_seeds* s, const unsigned char* host_gray, const float* host_f32, void* dst, size_t dst_pitch, bool dst_is_ref, int k) {
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
      hipLaunchKernelGGL(rmdk::ingest_u8_remap_kernel, grid, block, 0, s->copy_stream, s->d_u8[k], s->u8_pitch, s->d_undist_map1, s->d_undist_map2,
                         static_cast<float*>(dst), dst_stride, s->width, s->height);
    } else {
      const dim3 block(64, 4), grid((s->width + 255) / 256, (s->height + 3) / 4);
      hipLaunchKernelGGL(rmdk::ingest_u8_kernel, grid, block, 0, s->copy_stream, s->d_u8[k], s->u8_pitch, static_cast<float*>(dst), dst_stride,
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
  if (s->ingest_host_wait) HIP_TRY(hipEventSynchronize(s->staged[k]));
  else HIP_TRY(hipStreamWaitEvent(s->stream, s->staged[k], 0));
  if (s->ingest_profile) {
    const double t_d = host_now_us();
    s->ingest_us[0] += t_b - t_a; s->ingest_us[1] += t_c - t_b; s->ingest_us[2] += t_d - t_c; s->ingest_us[3] += 1.0;
  }
  return RMD_HIP_OK;
}

// a host frame becomes the current image: stage it into the plane that is NOT being read by the update in flight
// The fused path: frame n goes through pinned buffer and staging buffer n % SLOTS, last read by the copy engine / the setup kernel of
// frame n - SLOTS.  That kernel has completed once the setup kernel of frame n - SLOTS + 1 has STARTED (same stream), which is
// what h_progress reports.
// Wait (on the host, without touching the device) until the setup kernel of step `need` has started, as reported through the pinned word
// `progress`: the staging buffers of SLOTS steps ago are free then.  Numbers are compared modulo 2^32 like the kernel's test.
static int wait_for_progress(volatile unsigned int* progress, unsigned int need, hipStream_t stream) {
  auto behind = [&]() { return static_cast<int>(*progress - need) < 0; };
  if (behind()) {
    const double t0 = host_now_us();
    while (behind()) {
      if (host_now_us() - t0 > 2000.0) {  // far behind (or another matcher ran the frames in between): an idle stream has read everything
        ++g_progress_timeouts;
        HIP_TRY(hipStreamSynchronize(stream));
        break;
      }
      cpu_relax();
    }
    const double w = host_now_us() - t0;
    if (w > g_progress_max_wait_us) g_progress_max_wait_us = w;
  }
  return RMD_HIP_OK;
}

static int ingest_current_fused(rmd_hip_seeds* s, const unsigned char* host_gray, const float* host_f32, const float* T_curr_world) {
  const double t_a = s->ingest_profile ? host_now_us() : 0.0;
  const unsigned long long n64 = ++s->zc_number;
  const unsigned int n = static_cast<unsigned int>(n64);
  const int k = static_cast<int>(n64 % rmd_hip_seeds::RING);
  if (n64 > static_cast<unsigned long long>(rmd_hip_seeds::RING)) TRY(wait_for_progress(s->h_progress, n - rmd_hip_seeds::RING + 1u, s->stream));
  const double t_b = s->ingest_profile ? host_now_us() : 0.0;
  PendingIngest in;
  bool in_place = false;
  // every ring slot has its own arrival flag, one per kind of frame (8-bit / float: they use different staging buffers): the setup kernel
  // of frame n asks for frame n's; its verdict for frame n + 1 reads the flag of that slot for ITS kind, which a frame of the other kind never sets
  auto flag_of = [&](int kind, int slot) { return s->d_zc_flag + (static_cast<size_t>(kind) * rmd_hip_seeds::RING + slot) * (FLAG_ALLOC_BYTES / sizeof(unsigned int)); };
  void* stage_src = nullptr; void* stage_dst = nullptr; size_t stage_bytes = 0;
  auto ensure_u8_ring = [&]() -> int {
    const size_t bytes = static_cast<size_t>(s->u8_pitch) * s->height;
    for (int q = 0; q < rmd_hip_seeds::RING; ++q) {  // (all slots at once: the search kernel is told where the NEXT frame will be)
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
  in.no_remap = packed;
  if (as_u8) {
    const size_t bytes = static_cast<size_t>(s->u8_pitch) * s->height;
    TRY(ensure_u8_ring());
    if (packed) {
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
      stage_src = s->h_zc_u8[k]; stage_dst = s->d_zc_u8[k]; stage_bytes = bytes;
      in.u8 = reinterpret_cast<const unsigned int*>(s->d_zc_u8[k]);
    }
    in.common.kind = 1;
    in.common.pitch = s->u8_pitch;
  } else {
    const size_t bytes = static_cast<size_t>(s->width) * s->height * sizeof(float);
    for (int q = 0; q < rmd_hip_seeds::RING; ++q) {
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
  if (in_place) {
    in.common.flag = nullptr;  // the setup kernel reads the pinned buffer itself: it is complete before the kernel is launched
  } else {
    HIP_TRY(hipMemcpyAsync(stage_dst, stage_src, stage_bytes, hipMemcpyHostToDevice, s->copy_stream));
    const size_t fw = flag_words(s->h_progress, n);
    fill_flag_block(s->h_seq + k * FLAG_SLOT_WORDS, n, fw);  // behind the frame on the same stream: when the kernel sees n, the frame is in HBM
    unsigned int* slot_flag = flag_of(as_u8 ? 0 : 1, k);
    if (s->inject_withhold_flag) s->inject_withhold_flag = false;  // fault injection: the frame arrives, its flag never does -> the kernel's bounded wait runs out
    else HIP_TRY(hipMemcpyAsync(slot_flag, s->h_seq + k * FLAG_SLOT_WORDS, fw * sizeof(unsigned int), hipMemcpyHostToDevice, s->copy_stream));
    in.common.flag = slot_flag;
  }
  int plane = 0;  // one plane is enough: setup k writes it after search k - 1 has run (same stream)
  if (ahead) {  // ... unless the previous update's search kernel brings the frame in: frame n lives in plane n % 2
    const int kind = as_u8 ? 0 : 1, k_next = static_cast<int>((n64 + 1) % rmd_hip_seeds::RING);
    void* dev = nullptr;
    if (in_place) {
      // frame n is complete in ITS slot, for ITS kind: the verdict of setup n - 1 read this very word, and setup n's verdict for frame n + 1
      // reads the word of slot k_next for this kind -- which a frame of the other kind, or a frame two steps ahead, never sets (one word
      // per kind for the whole ring let setup n take "frame n + 2 of this kind is there" for "frame n + 1 is", and convert stale bytes)
      __atomic_store_n(&s->h_submitted[kind * rmd_hip_seeds::RING + k], n, __ATOMIC_RELEASE);
      HIP_TRY(hipHostGetDevicePointer(&dev, s->h_submitted, 0));
      in.common.submitted = static_cast<const unsigned int*>(dev) + kind * rmd_hip_seeds::RING + k_next;
      HIP_TRY(hipHostGetDevicePointer(&dev, as_u8 ? static_cast<void*>(s->h_zc_u8[k_next]) : static_cast<void*>(s->h_zc_f32[k_next]), 0));
      in.next_src = dev;
    } else {
      in.common.submitted = flag_of(kind, k_next);  // the arrival flag of the next frame's slot, for this kind
      in.next_src = as_u8 ? static_cast<const void*>(s->d_zc_u8[k_next]) : static_cast<const void*>(s->d_zc_f32[k_next]);
    }
    in.common.ahead = s->d_ahead;
    static const int ahead_wgs = [] { const char* e = getenv("RMD_HIP_AHEAD_WGS"); return e ? atoi(e) : AHEAD_WGS; }();  // (A/B)
    in.common.ahead_wgs = ahead_wgs;
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
  s->P.cur = static_cast<const float*>(im.data);
  s->P.cur_stride = s->P.stride;
  const int rc = seeds_after_frame(s, T_curr_world, &in);
  if (s->ingest_profile) {
    const double t_d = host_now_us();
    s->ingest_us[0] += t_b - t_a; s->ingest_us[1] += t_c - t_b; s->ingest_us[2] += t_d - t_c; s->ingest_us[3] += 1.0;
  }
  return rc;
}

static int ingest_current(rmd_hip_seeds* s, const unsigned char* host_gray, const float* host_f32, const float* T_curr_world) {
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

static int ingest_reference(rmd_hip_seeds* s, const unsigned char* host_gray, const float* host_f32, const float* T_curr_world, float min_depth,
                            float max_depth) {
  TRY(ingest_init(s));
  TRY(seeds_flush(s));
  const int k = s->ingest_slot;
  s->ingest_slot = (k + 1) % rmd_hip_seeds::SLOTS;
  const rmd_hip_image& im = s->planes[RMD_HIP_PLANE_REF_IMG];
  TRY(ingest_frame(s, host_gray, host_f32, im.data, im.pitch, true, k));
  return seeds_after_reference(s, T_curr_world, min_depth, max_depth);
}

int rmd_hip_seeds_set_reference_u8(rmd_hip_seeds_t* s, const unsigned char* host_gray, const float* T_curr_world, float min_depth,
                                   float max_depth) {
  if (!s || !host_gray || !T_curr_world) return fail(RMD_HIP_ERR_INVALID_ARG, "set_reference_u8: null argument");
  TRY(seeds_bind_device(s));
  return ingest_reference(s, host_gray, nullptr, T_curr_world, min_depth, max_depth);
}

int rmd_hip_seeds_update_u8(rmd_hip_seeds_t* s, const unsigned char* host_gray, const float* T_curr_world) {
  if (!s || !host_gray || !T_curr_world) return fail(RMD_HIP_ERR_INVALID_ARG, "update_u8: null argument");
  if (s->batch) return fail(RMD_HIP_ERR_INVALID_ARG, "update_u8: this SeedMatrix is a member of a batch; its updates are issued with rmd_hip_batch_update*");
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

int rmd_hip_seeds_download(const rmd_hip_seeds_t* s, int plane, void* host_dst) {
  if (!s || !host_dst) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_download: null argument");
  if (plane < 0 || plane >= RMD_HIP_NUM_PLANES) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_download: bad plane %d", plane);
  TRY(seeds_bind_device(s));
  TRY(seeds_sync(s));
  const rmd_hip_image& im = s->planes[plane];
  const size_t row = static_cast<size_t>(im.width) * kind_size(im.kind);
  HIP_TRY(hipMemcpy2D(host_dst, row, im.data, im.pitch, row, im.height, hipMemcpyDeviceToHost));
  return RMD_HIP_OK;
}

int rmd_hip_seeds_upload(rmd_hip_seeds_t* s, int plane, const float* host_src) {
  if (!s || !host_src) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_upload: null argument");
  if (plane < RMD_HIP_PLANE_MU || plane > RMD_HIP_PLANE_B) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_upload: plane %d", plane);
  TRY(seeds_bind_device(s));
  TRY(seeds_sync(s));
  s->async_count_valid = false;
  const rmd_hip_image& im = s->planes[plane];
  const size_t row = static_cast<size_t>(im.width) * 4;
  HIP_TRY(hipMemcpy2D(im.data, im.pitch, host_src, row, row, im.height, hipMemcpyHostToDevice));
  return RMD_HIP_OK;
}

int rmd_hip_seeds_plane(const rmd_hip_seeds_t* s, int plane, const rmd_hip_image_t** view) {
  if (!s || !view) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_plane: null argument");
  if (plane < 0 || plane >= RMD_HIP_NUM_PLANES) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_plane: bad plane %d", plane);
  *view = &s->planes[plane];
  return RMD_HIP_OK;
}

int rmd_hip_seeds_converged_count(const rmd_hip_seeds_t* s, size_t* count) {
  if (!s || !count) return fail(RMD_HIP_ERR_INVALID_ARG, "converged_count: null argument");
  TRY(seeds_bind_device(s));
  rmd_hip_seeds* m = const_cast<rmd_hip_seeds*>(s);
  if (m->async_count_valid) {
    // The latest thing that happened to the seed state is an update of the tile pipeline: its setup kernel -- which IS seed_check
    // (seed_matrix.cu:139-142) -- counted the seeds it found CONVERGED and the search kernel's last workgroup mirrored the sum,
    // stamped with the update's number, to pinned memory.  Nothing later in the update changes that count (the matcher only turns
    // UPDATE into NO_MATCH), so it is what countEqual(convergence, CONVERGED) returns after the whole update -- without waiting
    // for the search, without a kernel and without flushing the deferred finalisation.
    volatile unsigned long long* word = m->mws->h_conv + m->seq;
    const unsigned int want = m->async_number;
    const double t0 = host_now_us();
    bool synced = false;
    for (;;) {
      const unsigned long long v = *word;
      if (static_cast<unsigned int>(v >> 32) == want) {
        *count = static_cast<size_t>(v & 0xffffffffull);
        return ingest_error_check(m->batch ? m->batch->group_of(m->batch_index).h_progress : m->h_progress);
      }
      if (synced) break;  // cannot happen; fall through to the counting kernel
      if (host_now_us() - t0 > 2000.0) {
        HIP_TRY(hipStreamSynchronize(m->stream));
        synced = true;
        continue;
      }
      cpu_relax();
    }
  }
  TRY(seeds_flush(m));
  HIP_TRY(hipMemsetAsync(m->d_scalars, 0, sizeof(unsigned long long), m->stream));
  {
    ScopedStage st(m->opt_timing ? &m->timers[RMD_HIP_STAGE_COUNT] : nullptr, m->stream);
    const dim3 block(256), grid((s->width + 255) / 256, s->height < 64 ? s->height : 64);
    hipLaunchKernelGGL(rmdk::count_eq_kernel, grid, block, 0, m->stream,
                       static_cast<const int*>(s->planes[RMD_HIP_PLANE_CONVERGENCE].data), s->width, s->height,
                       s->P.stride, static_cast<int>(RMD_HIP_STATE_CONVERGED), m->d_scalars);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipMemcpyAsync(m->h_scalars, m->d_scalars, sizeof(unsigned long long), hipMemcpyDeviceToHost, m->stream));
  TRY(seeds_sync(s));
  *count = static_cast<size_t>(m->h_scalars[0]);
  return RMD_HIP_OK;
}

int rmd_hip_seeds_point_cloud(rmd_hip_seeds_t* s, const rmd_hip_image_t* depth, float* out_xyzi, size_t capacity, size_t* n_points) {
  if (!s || !n_points || (!out_xyzi && capacity)) return fail(RMD_HIP_ERR_INVALID_ARG, "point_cloud: null argument");
  if (!s->has_reference) return fail(RMD_HIP_ERR_NOT_READY, "point_cloud: no reference image set");
  if (depth && (depth->kind != RMD_HIP_KIND_F32 || depth->width != s->width || depth->height != s->height))
    return fail(RMD_HIP_ERR_INVALID_ARG, "point_cloud: depth must be an f32 %dx%d image", s->width, s->height);
  TRY(seeds_bind_device(s));
  TRY(seeds_flush(s));
  if (depth && !(depth->owner_seeds == s) && (depth->owner_seeds || (depth->owner_stream && depth->owner_stream != s->stream)))
    TRY(image_settle(depth));
  const int n_pix = s->width * s->height;
  const int n_blocks = (n_pix + rmdk::PC_BLOCK - 1) / rmdk::PC_BLOCK;
  if (!s->d_pc_counts) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_pc_counts), (static_cast<size_t>(n_blocks) + 1) * sizeof(unsigned int)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_pc_points), static_cast<size_t>(n_pix) * sizeof(float4)));
  }
  rmdk::PointCloudParams P;
  P.w = s->width; P.h = s->height;
  P.stride = s->P.stride;
  P.depth = depth ? static_cast<const float*>(depth->data) : s->P.mu;
  P.depth_stride = depth ? static_cast<int>(depth->stride) : s->P.stride;
  P.conv = s->P.conv;
  P.ref = s->P.ref;
  P.cam = s->P.cam;
  P.T_world_ref = s->T_world_ref;
  hipLaunchKernelGGL(rmdk::pc_count_kernel, dim3(n_blocks), dim3(rmdk::PC_BLOCK), 0, s->stream, P, s->d_pc_counts);
  hipLaunchKernelGGL(rmdk::pc_scan_kernel, dim3(1), dim3(1024), 0, s->stream, s->d_pc_counts, n_blocks, s->d_pc_counts + n_blocks);
  hipLaunchKernelGGL(rmdk::pc_write_kernel, dim3(n_blocks), dim3(rmdk::PC_BLOCK), 0, s->stream, P, s->d_pc_counts, s->d_pc_points,
                     static_cast<unsigned int>(n_pix));
  HIP_TRY(hipGetLastError());
  unsigned int total = 0;
  HIP_TRY(hipMemcpyAsync(&total, s->d_pc_counts + n_blocks, sizeof(total), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  *n_points = total;
  const size_t n_copy = total < capacity ? total : capacity;
  if (n_copy) HIP_TRY(hipMemcpy(out_xyzi, s->d_pc_points, n_copy * sizeof(float4), hipMemcpyDeviceToHost));
  return RMD_HIP_OK;
}

int rmd_hip_seeds_convergence_bgr8(rmd_hip_seeds_t* s, unsigned char* host_bgr) {
  if (!s || !host_bgr) return fail(RMD_HIP_ERR_INVALID_ARG, "convergence_bgr8: null argument");
  if (!s->has_reference) return fail(RMD_HIP_ERR_NOT_READY, "convergence_bgr8: no reference image set");
  TRY(seeds_bind_device(s));
  // No flush of the deferred finalisation: it only ever turns UPDATE into NO_MATCH, and neither has a colour (publisher.cpp:124-134);
  // CONVERGED / DIVERGED were settled by the update's seed_check.  Stream order puts the kernel behind the update.
  const size_t bytes = static_cast<size_t>(s->width) * s->height * 3;
  if (!s->d_bgr) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_bgr), (bytes + 15) & ~static_cast<size_t>(15)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_bgr), bytes));
  }
  const long long groups = (static_cast<long long>(s->width) * s->height + 3) / 4;
  hipLaunchKernelGGL(rmdk::convergence_bgr8_kernel, dim3(static_cast<unsigned int>((groups + 255) / 256)), dim3(256), 0, s->stream, s->P.ref, s->P.conv, s->width,
                     s->height, s->P.stride, s->d_bgr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(s->h_bgr, s->d_bgr, bytes, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  TRY(ingest_error_check(s->batch ? s->batch->group_of(s->batch_index).h_progress : s->h_progress));
  memcpy(host_bgr, s->h_bgr, bytes);
  return RMD_HIP_OK;
}

int rmd_hip_seeds_dist_from_ref(const rmd_hip_seeds_t* s, float* dist) {
  if (!s || !dist) return fail(RMD_HIP_ERR_INVALID_ARG, "dist_from_ref: null argument");
  *dist = s->dist_from_ref;
  return RMD_HIP_OK;
}

int rmd_hip_seeds_sync(const rmd_hip_seeds_t* s) {
  if (!s) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_sync: null handle");
  TRY(seeds_bind_device(s));
  return seeds_sync(s);
}

int rmd_hip_seeds_set_option(rmd_hip_seeds_t* s, int option, int value) {
  if (!s) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: null handle");
  // a batch member's updates are launched by the batch (batch_launch), which knows nothing of per-member statistics, per-update event
  // pairs, eager finalisation or unit targets: accepting such a setting and then ignoring it would leave last_stats / timing stale
  if (s->batch && ((option == RMD_HIP_OPT_COLLECT_STATS && value != 0) || (option == RMD_HIP_OPT_TIMING && value != 0) ||
                   (option == RMD_HIP_OPT_LAZY_FINALIZE && value == 0) || option == RMD_HIP_OPT_UNIT_TARGET || option == RMD_HIP_OPT_SEARCH_FLAGS))
    return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: option %d has no effect on a member of a batch (rmd_hip_batch_set_option sets the batch's timing, unit "
                                         "target and search flags)", option);
  switch (option) {
    case RMD_HIP_OPT_MATCHER:
#ifdef RMD_AB_MATCHERS
      if (value < 0 || value > 3) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: matcher %d", value);
#else
      if (value != 0 && value != 3)
        return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: matcher %d is not part of this build (0 = per-pixel kernel, 3 = tile pipeline; the retired "
                                             "variants 1 and 2 exist in A/B builds only, tools/ab_make.sh)", value);
#endif
      if (s->batch && value != 3) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: members of a batch use the tile pipeline");
      if (value != s->opt_matcher) {  // each matcher keeps its own per-frame state: settle the old one first
        TRY(seeds_bind_device(s));
        TRY(seeds_sync(s));
#ifdef RMD_AB_MATCHERS
        s->frame_ws.frame = 0;
        HIP_TRY(hipMemsetAsync(s->frame_ws.d_ctl, 0, 3 * rmdk::FR_CTL_WORDS * sizeof(unsigned int), s->stream));
#endif
        s->mws->frame = 0;
        HIP_TRY(hipMemsetAsync(s->mws->d_shards, 0, 3 * rmdk::UNIT_SHARDS * sizeof(unsigned long long), s->stream));
        s->async_count_valid = false;
      }
      s->opt_matcher = value;
      return RMD_HIP_OK;
    case RMD_HIP_OPT_UNIT_TARGET:
      if (value < 1 || value > 4) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: unit target %d outside 1..4", value);
      s->opt_unit_target = value;
      return RMD_HIP_OK;
    case RMD_HIP_OPT_SEARCH_FLAGS:
      if (value < 0 || value > 7) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: search flags %d outside 0..7", value);
      s->mws->search_flags = value;
      return RMD_HIP_OK;
    case RMD_HIP_OPT_LOCAL_MAX:
      if (value < 0) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: local_max %d", value);
      s->opt_local_max = value;
      return RMD_HIP_OK;
    case RMD_HIP_OPT_UNIT_ROUNDS:
      if (value < 0 || value > 4) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: unit_rounds %d", value);
      s->opt_unit_rounds = value;
      return RMD_HIP_OK;
    case RMD_HIP_OPT_TIMING:
      if (value < 0 || value > 2) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: timing mode %d", value);
      s->opt_timing = value;
      return RMD_HIP_OK;
    case RMD_HIP_OPT_LAZY_FINALIZE:
      s->opt_lazy = value != 0;
      return RMD_HIP_OK;
    case RMD_HIP_OPT_INJECT_FAULT:
      if (value != 0 && value != 1) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: fault %d (1 = withhold the arrival flag of the next staged host frame)", value);
      if (s->batch) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: fault injection is for plain SeedMatrix handles");
      s->inject_withhold_flag = value == 1;
      return RMD_HIP_OK;
    case RMD_HIP_OPT_WINDOW:
      if (value < 0 || value > 2) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: window %d", value);
      s->opt_window = value;
      return RMD_HIP_OK;
    case RMD_HIP_OPT_COLLECT_STATS:
      s->opt_stats = value == 2 ? 2 : (value != 0);
      if (s->opt_stats == 2) {  // (re)start a timeline: zeroed buffer, frame counter 0
        if (s->batch) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: timeline probes are for plain SeedMatrix handles");
        TRY(seeds_bind_device(s));
        rmdk::MatcherWorkspace& ws = s->matcher_ws;
        const size_t wbytes = ws.wg_trace_slice_u64() * rmdk::FR_TRACE_FRAMES * sizeof(unsigned long long);
        if (!ws.d_wg_trace) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ws.d_wg_trace), wbytes));
        HIP_TRY(hipStreamSynchronize(s->stream));
        HIP_TRY(hipMemset(ws.d_wg_trace, 0, wbytes));
#ifdef RMD_AB_MATCHERS
        const size_t bytes = ws.trace_slice_u64() * rmdk::TRACE_FRAMES * sizeof(unsigned long long);
        if (!ws.d_trace) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ws.d_trace), bytes));
        rmdk::FrameWorkspace& fw = s->frame_ws;
        const size_t fbytes = fw.trace_slice_u64() * rmdk::FR_TRACE_FRAMES * sizeof(unsigned long long);
        if (!fw.d_trace) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&fw.d_trace), fbytes));
        HIP_TRY(hipMemset(ws.d_trace, 0, bytes));
        HIP_TRY(hipMemset(fw.d_trace, 0, fbytes));
#endif
        HIP_TRY(hipDeviceSynchronize());
        s->trace_frame = 0;
      }
      return RMD_HIP_OK;
    default: return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: unknown option %d", option);
  }
}

int rmd_hip_seeds_timing(const rmd_hip_seeds_t* s, int stage, double* total_ms, long* launches) {
  if (!s || stage < 0 || stage >= RMD_HIP_NUM_SEED_STAGES) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_timing: bad argument");
  TRY(seeds_bind_device(s));
  if (s->opt_timing == 2 && stage == RMD_HIP_STAGE_UPDATE) {
    // region mode: ONE event pair around everything queued since timing_reset (no per-launch markers in the stream)
    rmd_hip_seeds* m = const_cast<rmd_hip_seeds*>(s);
    if (!m->region_start) return fail(RMD_HIP_ERR_NOT_READY, "seeds_timing: call timing_reset first");
    if (!m->region_stop) HIP_TRY(hipEventCreate(&m->region_stop));
    TRY(seeds_flush(m));  // the deferred finalisation of the last update belongs to the region
    HIP_TRY(hipEventRecord(m->region_stop, m->stream));
    HIP_TRY(hipEventSynchronize(m->region_stop));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, m->region_start, m->region_stop));
    if (total_ms) *total_ms = ms;
    if (launches) *launches = m->region_updates;
    return RMD_HIP_OK;
  }
  TRY(seeds_sync(s));
  if (total_ms) *total_ms = s->timers[stage].total_ms;
  if (launches) *launches = s->timers[stage].launches;
  return RMD_HIP_OK;
}
int rmd_hip_seeds_timing_reset(rmd_hip_seeds_t* s) {
  if (!s) return fail(RMD_HIP_ERR_INVALID_ARG, "timing_reset: null handle");
  TRY(seeds_bind_device(s));
  TRY(seeds_sync(s));
  for (auto& t : s->timers) t.reset();
  if (s->opt_timing == 2) {
    if (!s->region_start) HIP_TRY(hipEventCreate(&s->region_start));
    HIP_TRY(hipEventRecord(s->region_start, s->stream));
    s->region_updates = 0;
  }
  return RMD_HIP_OK;
}
int rmd_hip_seeds_last_stats(const rmd_hip_seeds_t* s, long long* out3) {
  if (!s || !out3) return fail(RMD_HIP_ERR_INVALID_ARG, "last_stats: null argument");
  TRY(seeds_bind_device(s));
  TRY(seeds_sync(s));
  for (int k = 0; k < 3; ++k) out3[k] = s->last_stats[k];
  return RMD_HIP_OK;
}
int rmd_hip_seeds_last_diagnostics(const rmd_hip_seeds_t* s, long long* out16) {
  if (!s || !out16) return fail(RMD_HIP_ERR_INVALID_ARG, "last_diagnostics: null argument");
  TRY(seeds_bind_device(s));
  TRY(seeds_sync(s));
  for (int k = 0; k < 16; ++k) out16[k] = s->last_stats[k];
  return RMD_HIP_OK;
}

int rmd_hip_seeds_trace_download(rmd_hip_seeds_t* s, int frame, unsigned long long* out, size_t capacity, size_t* written) {
  if (!s || !out || !written) return fail(RMD_HIP_ERR_INVALID_ARG, "trace_download: null argument");
  const rmdk::MatcherWorkspace& ws = s->matcher_ws;
  if (!ws.d_wg_trace) return fail(RMD_HIP_ERR_NOT_READY, "trace_download: set RMD_HIP_OPT_COLLECT_STATS to 2 first");
  if (frame < 0 || frame >= s->trace_frame || frame < s->trace_frame - rmdk::TRACE_FRAMES)
    return fail(RMD_HIP_ERR_INVALID_ARG, "trace_download: frame not in the buffer");
  if (s->opt_matcher == 3) {  // tile pipeline: FR_TRACE_WORDS words per tile / search workgroup
    const size_t fn = ws.wg_trace_slice_u64();
    if (capacity < fn) return fail(RMD_HIP_ERR_INVALID_ARG, "trace_download: buffer too small (%zu words needed)", fn);
    TRY(seeds_bind_device(s));
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(out, ws.d_wg_trace + static_cast<size_t>(frame % rmdk::FR_TRACE_FRAMES) * fn, fn * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    *written = fn;
    return RMD_HIP_OK;
  }
#ifdef RMD_AB_MATCHERS
  if (s->opt_matcher == 2) {  // one-launch frame kernel: FR_TRACE_WORDS words per workgroup (256 front slots, then the tile grid)
    const rmdk::FrameWorkspace& fw = s->frame_ws;
    const size_t fn = fw.trace_slice_u64();
    if (capacity < fn) return fail(RMD_HIP_ERR_INVALID_ARG, "trace_download: buffer too small (%zu words needed)", fn);
    TRY(seeds_bind_device(s));
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(out, fw.d_trace + static_cast<size_t>(frame % rmdk::FR_TRACE_FRAMES) * fn, fn * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    *written = fn;
    return RMD_HIP_OK;
  }
  const size_t n = ws.trace_slice_u64();
  if (capacity < n) return fail(RMD_HIP_ERR_INVALID_ARG, "trace_download: buffer too small");
  TRY(seeds_bind_device(s));
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipMemcpy(out, ws.d_trace + static_cast<size_t>(frame % rmdk::TRACE_FRAMES) * n, n * sizeof(unsigned long long),
                    hipMemcpyDeviceToHost));
  *written = n;
  return RMD_HIP_OK;
#else
  return fail(RMD_HIP_ERR_INVALID_ARG, "trace_download: timeline probes exist for the tile pipeline (matcher 3) only in this build");
#endif
}

}  // extern "C"

// ---- batches of SeedMatrix objects ----------------------------------------------------------
// BASELINE configs[3] / SURVEY 8(e): independent sequences.  One MI355X is far from full with one 640x480 sequence (a frame is ~2
// rounds of work per workgroup and a third of it is a latency chain), so up to MAX_BATCH sequences of one size are stepped TOGETHER:
// one setup launch + one search launch per step for all of them, unit lists and the persistent search workgroups shared.
namespace {

int batch_bind_device(const rmd_hip_batch* b) {
  int cur = -1;
  HIP_TRY(hipGetDevice(&cur));
  if (cur != b->device) HIP_TRY(hipSetDevice(b->device));
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
    rmdk::BatchArgs<rmdk::MAX_BATCH> B;
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
        HIP_TRY((rmdk::launch_seed_pipeline_compact<SIDE, 1>(B1, 1, G.ws, G.stream, b->num_cus, b->opt_unit_target, ingest ? &in : nullptr)));
      } else {
        HIP_TRY((rmdk::launch_seed_pipeline_compact<SIDE, rmdk::MAX_BATCH>(B, G.n, G.ws, G.stream, b->num_cus, b->opt_unit_target, ingest ? &in : nullptr)));
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
  const unsigned long long n64 = ++b->step_number;
  const unsigned int n = static_cast<unsigned int>(n64);
  const int k = static_cast<int>(n64 % rmd_hip_batch::SLOTS);
  // Slot k was last read by the setup kernels of the step recorded in slot_step[k]; such a kernel is done once a LATER setup kernel of
  // the same group has started (the progress word), or, if the group has not been launched since, once its stream is idle.
  static_assert(rmd_hip_batch::SLOTS == 3, "Group::slot_step");
  for (int g = 0; g < b->n_groups; ++g) {
    rmd_hip_batch::Group& G = b->groups[g];
    const unsigned long long used = G.slot_step[k];
    if (!used) continue;
    if (G.last_step > used) TRY(wait_for_progress(G.h_progress, static_cast<unsigned int>(used) + 1u, G.stream));
    else HIP_TRY(hipStreamSynchronize(G.stream));
  }
  const size_t need = static_cast<size_t>(b->n) * static_cast<size_t>(m0->width) * m0->height * sizeof(float);  // float frames: the larger kind
  if (b->stage_bytes < need) {
    for (int g = 0; g < b->n_groups; ++g) HIP_TRY(hipStreamSynchronize(b->groups[g].stream));
    HIP_TRY(hipStreamSynchronize(b->copy_stream));
    for (int q = 0; q < rmd_hip_batch::SLOTS; ++q) {
      if (b->h_stage[q]) (void)hipHostFree(b->h_stage[q]);
      if (b->d_stage[q]) (void)hipFree(b->d_stage[q]);
      b->h_stage[q] = nullptr; b->d_stage[q] = nullptr;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&b->h_stage[q]), need + 16, hipHostMallocDefault));
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_stage[q]), need));
    }
    b->stage_bytes = need;
  }
  // float frames of 8-bit levels travel as bytes (pack_float_rows_u8), if every frame of the step qualifies
  bool packed = false;
  if (!gray && float_frames_as_bytes() && b->pack_backoff > 0) --b->pack_backoff;
  else if (!gray && float_frames_as_bytes()) {
    packed = true;
    for (int i = 0; i < b->n && packed; ++i)
      if ((active >> i) & 1u) packed = CopyPool::instance().pack(f32[i], b->h_stage[k] + static_cast<size_t>(i) * bytes_u8, m0->width, m0->height, u8_pitch);
    if (packed) frame_bytes = bytes_u8;
    else b->pack_backoff = 15;
  }
  const bool as_u8 = gray != nullptr || packed;
  bool any_maps = false;
  for (int i = 0; i < b->n; ++i) any_maps = any_maps || (gray && ((active >> i) & 1u) && b->members[i]->d_undist_map1);
  int first = -1, last = -1, n_segs = 0;
  CopyPool::Segment segs[rmdk::MAX_BATCH];
  for (int i = 0; i < b->n; ++i) {
    if (!((active >> i) & 1u)) continue;
    if (first < 0) first = i;
    last = i;
    if (packed) continue;
    unsigned char* dst = b->h_stage[k] + static_cast<size_t>(i) * frame_bytes;
    if (gray && u8_pitch != m0->width) {
      for (int y = 0; y < m0->height; ++y) memcpy(dst + static_cast<size_t>(y) * u8_pitch, gray[i] + static_cast<size_t>(y) * m0->width, m0->width);
    } else {
      segs[n_segs].dst = dst;
      segs[n_segs].src = gray ? static_cast<const void*>(gray[i]) : static_cast<const void*>(f32[i]);
      ++n_segs;
    }
  }
  if (n_segs) CopyPool::instance().copy_many(segs, n_segs, frame_bytes);  // the frames of the step, spread over the copy threads
  const bool in_place = frame_in_place(true, any_maps);  // (the remap gathers single bytes: staged)
  const unsigned char* frames_dev = b->d_stage[k];
  rmdk::IngestArgs in;
  if (in_place) {  // the setup kernels read the pinned block themselves
    void* dev = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dev, b->h_stage[k], 0));
    frames_dev = static_cast<const unsigned char*>(dev);
    in.flag = nullptr;
  } else {
    const size_t off = static_cast<size_t>(first) * frame_bytes, len = static_cast<size_t>(last - first + 1) * frame_bytes;
    HIP_TRY(hipMemcpyAsync(b->d_stage[k] + off, b->h_stage[k] + off, len, hipMemcpyHostToDevice, b->copy_stream));
    size_t fw = 1;
    for (int g = 0; g < b->n_groups; ++g)
      if (b->groups[g].last_step != 0 && flag_words(b->groups[g].h_progress, static_cast<unsigned int>(b->groups[g].last_step) + 1u) != 1) fw = FLAG_WORDS;
    fill_flag_block(b->h_seq + k * FLAG_SLOT_WORDS, n, fw);  // behind the frames on the same stream: when the kernel sees n, they are in HBM
    HIP_TRY(hipMemcpyAsync(b->d_flag, b->h_seq + k * FLAG_SLOT_WORDS, fw * sizeof(unsigned int), hipMemcpyHostToDevice, b->copy_stream));
    in.flag = b->d_flag;
  }
  in.kind = as_u8 ? 1 : 2;
  in.pitch = u8_pitch;
  in.number = n;
  in.no_remap = packed;
  for (int i = 0; i < b->n; ++i) {
    if (!((active >> i) & 1u)) continue;
    rmd_hip_seeds* m = b->members[i];
    m->P.cur = static_cast<const float*>(m->planes[RMD_HIP_PLANE_CURR_IMG].data);  // setup k writes it after search k - 1 has run (same stream)
    m->P.cur_stride = m->P.stride;
    seeds_frame_pose(m, T_curr_world + 12 * i);
  }
  for (int g = 0; g < b->n_groups; ++g) {
    rmd_hip_batch::Group& G = b->groups[g];
    if ((active >> G.first) & ((1u << G.n) - 1u)) { G.slot_step[k] = n64; G.last_step = n64; }
  }
  return batch_launch(b, active, &in, frames_dev, frame_bytes);
}

}  // namespace

extern "C" {

int rmd_hip_batch_destroy(rmd_hip_batch_t* b) {
  if (!b) return RMD_HIP_OK;
  (void)hipSetDevice(b->device);
  for (auto& G : b->groups)
    if (G.stream) (void)hipStreamSynchronize(G.stream);
  if (b->copy_stream) (void)hipStreamSynchronize(b->copy_stream);
  for (int i = 0; i < rmdk::MAX_BATCH; ++i)
    if (b->members[i]) (void)seeds_destroy_impl(b->members[i]);
  for (int q = 0; q < rmd_hip_batch::SLOTS; ++q) {
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

int rmd_hip_batch_create(int n, int width, int height, float fx, float fy, float cx, float cy, int patch_side, int max_extent, rmd_hip_batch_t** out) {
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
  int want_groups = n >= 3 ? 3 : n;  // measured (profiles/r03_batch_ab.txt): three groups beat two by 3-9 %, a fourth shares a hardware-queue pool and loses 25 %
  if (const char* e = getenv("RMD_HIP_BATCH_GROUPS")) want_groups = atoi(e);  // (A/B)
  if (want_groups < 1) want_groups = 1;
  if (want_groups > rmd_hip_batch::MAX_GROUPS) want_groups = rmd_hip_batch::MAX_GROUPS;
  if (want_groups > n) want_groups = n;
  b->n_groups = want_groups;
  b->opt_unit_target = 1;  // (2x / 3x as many, smaller units: +4 % with one group of 4, nothing with two groups)
  const size_t pitch = (static_cast<size_t>(width) * 4 + 255) / 256 * 256;  // as image_alloc lays the members' planes out
  for (int g = 0; g < b->n_groups; ++g) {
    rmd_hip_batch::Group& G = b->groups[g];
    G.first = g == 0 ? 0 : b->groups[g - 1].first + b->groups[g - 1].n;
    G.n = n / b->n_groups + (g < n % b->n_groups ? 1 : 0);  // the larger groups first
    // (one priority level each, see create_stream; a fourth group shares the first one's pool)
    if (create_stream(&G.stream, g % 3) != hipSuccess) return bail(fail(RMD_HIP_ERR_RUNTIME, "batch_create: hipStreamCreate failed"));
    if (G.ws.allocate(width, height, static_cast<int>(pitch / 4), G.n) != 0) return bail(fail(RMD_HIP_ERR_RUNTIME, "batch_create: update workspace"));
    if (hipHostMalloc(reinterpret_cast<void**>(&G.h_progress), 64, hipHostMallocMapped) != hipSuccess) return bail(fail(RMD_HIP_ERR_RUNTIME, "batch_create: progress words"));
    G.h_progress[0] = G.h_progress[1] = 0u;
    if (hipEventCreate(&G.ev) != hipSuccess) return bail(fail(RMD_HIP_ERR_RUNTIME, "batch_create: event"));
  }
  if (create_stream(&b->copy_stream, 2) != hipSuccess) return bail(fail(RMD_HIP_ERR_RUNTIME, "batch_create: hipStreamCreate failed"));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, b->device) == hipSuccess && prop.multiProcessorCount > 0) b->num_cus = prop.multiProcessorCount;
  if (hipHostMalloc(reinterpret_cast<void**>(&b->h_seq), rmd_hip_batch::SLOTS * FLAG_SLOT_WORDS * sizeof(unsigned int), hipHostMallocDefault) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&b->d_flag), FLAG_ALLOC_BYTES) != hipSuccess || hipMemset(b->d_flag, 0, FLAG_ALLOC_BYTES) != hipSuccess)
    return bail(fail(RMD_HIP_ERR_RUNTIME, "batch_create: ingest words"));
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
    if (stride_elems[i] < static_cast<size_t>(m->width)) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_update_device: stride < width (member %d)", i);
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
    case RMD_HIP_OPT_SEARCH_FLAGS:
      if (value < 0 || value > 7) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_set_option: search flags %d outside 0..7", value);
      for (int g = 0; g < b->n_groups; ++g) b->groups[g].ws.search_flags = value;
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


namespace {

// tv_prepare + `iterations` primal-dual iterations of TvParams P for `n_z` depth maps (grid z; 1 = the single denoiser, P.members null) on
// `stream`, ping-ponging between the two sets of iterate planes; *result_index = the set that holds the result.  ev0 (may be null) is
// recorded between the preparation and the first iteration.
int tv_run(const rmdk::TvParams& P, float* const u[2], float* const uh[2], float2* const p[2], int n_z, int iterations, int opt_iters_per_launch, int opt_geometry,
           hipStream_t stream, hipEvent_t ev0, int* result_index, long* launches) {
  const unsigned int nz = static_cast<unsigned int>(n_z);
  {
    const dim3 block(64, 4), grid((P.w + 63) / 64, (P.h + 3) / 4, nz);
    hipLaunchKernelGGL(rmdk::tv_prepare_kernel, grid, block, 0, stream, P, u[0], uh[0], p[0]);
    HIP_TRY(hipGetLastError());
  }
  int cur_buf = 0;
  if (ev0) HIP_TRY(hipEventRecord(ev0, stream));
  long n_launches = 0;
  if (opt_iters_per_launch == 1) {
    const dim3 block(rmdk::TV_TX, rmdk::TV_TY);
    const dim3 grid((P.w + rmdk::TV_TX - 1) / rmdk::TV_TX, (P.h + rmdk::TV_TY - 1) / rmdk::TV_TY, nz);
    for (int it = 0; it < iterations; ++it) {
      const int nxt = cur_buf ^ 1;
      hipLaunchKernelGGL(rmdk::tv_iterate_kernel, grid, block, 0, stream, P, u[cur_buf], uh[cur_buf], p[cur_buf], u[nxt], uh[nxt], p[nxt]);
      cur_buf = nxt;
      ++n_launches;
    }
  } else {
    // Temporally blocked kernel: tile geometry and blocking depth K by image size (K iterations per launch; the halo
    // grows with K, so small tiles pay more redundant work per iteration, but a VGA launch is latency-bound: fewer, fatter
    // launches win there).  opt_geometry (experiments): 0 = by size, 1..n = a fixed entry of the table below.
    auto run = [&](auto geom, int kmax, auto kernel) {
      using G = decltype(geom);
      const int k = opt_iters_per_launch == 0 ? kmax : (opt_iters_per_launch < kmax ? opt_iters_per_launch : kmax);
      const dim3 block(G::THREADS), grid((P.w + G::BX - 1) / G::BX, (P.h + G::BY - 1) / G::BY, nz);
      for (int done = 0; done < iterations; done += k) {
        const int now = iterations - done < k ? iterations - done : k;
        const int nxt = cur_buf ^ 1;
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, P, u[cur_buf], uh[cur_buf], p[cur_buf], u[nxt], uh[nxt], p[nxt], now);
        cur_buf = nxt;
        ++n_launches;
      }
    };
    // Measured (tools/denoise_sweep.py, MI355X): 16x16 tiles with K = 4 are the fastest at 640x480 (3.4 us per iteration, 50
    // launches for 200 iterations, latency-bound) AND at 1920x1080 (11.5 us per iteration = 7.2 TB/s of algorithmic traffic);
    // deeper blocking (K = 8) loses more to the redundant halo work than it saves in launches.
    int geometry = opt_geometry;
    if (geometry == 0) geometry = opt_iters_per_launch == 2 ? 1 : 4;
    switch (geometry) {
      case 1: run(rmdk::TvBlocked<32, 8, 2>(), 2, rmdk::tv_iterate_blocked_kernel<32, 8, 2>); break;
      case 2: run(rmdk::TvBlocked<64, 16, 4>(), 4, rmdk::tv_iterate_blocked_kernel<64, 16, 4>); break;
      case 3: run(rmdk::TvBlocked<32, 16, 4>(), 4, rmdk::tv_iterate_blocked_kernel<32, 16, 4>); break;
      case 4: run(rmdk::TvBlocked<16, 16, 4>(), 4, rmdk::tv_iterate_blocked_kernel<16, 16, 4>); break;
      case 5: run(rmdk::TvBlocked<16, 16, 8>(), 8, rmdk::tv_iterate_blocked_kernel<16, 16, 8>); break;
      default: return fail(RMD_HIP_ERR_INVALID_ARG, "denoise: unknown geometry %d", geometry);
    }
  }
  HIP_TRY(hipGetLastError());
  *result_index = cur_buf;
  *launches = n_launches;
  return RMD_HIP_OK;
}

}  // namespace

extern "C" {

// ---- DepthmapDenoiser -----------------------------------------------------------------------
int rmd_hip_denoiser_destroy(rmd_hip_denoiser_t* d) {
  if (!d) return RMD_HIP_OK;
  (void)hipSetDevice(d->device);
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  d->timer.destroy();
  if (d->h_staging) (void)hipHostFree(d->h_staging);
  rmd_hip_image* all[] = {&d->u[0], &d->u[1], &d->u_head[0], &d->u_head[1], &d->p[0], &d->p[1], &d->g};
  for (auto* im : all)
    if (im->owns && im->data) (void)hipFree(im->data);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
  return RMD_HIP_OK;
}

int rmd_hip_denoiser_create(int width, int height, rmd_hip_denoiser_t** out) {
  if (!out) return fail(RMD_HIP_ERR_INVALID_ARG, "denoiser_create: null output");
  *out = nullptr;
  if (width <= 0 || height <= 0) return fail(RMD_HIP_ERR_INVALID_ARG, "denoiser_create: bad size %dx%d", width, height);
  int ndev = 0;
  TRY(rmd_hip_device_count(&ndev));
  rmd_hip_denoiser* d = new (std::nothrow) rmd_hip_denoiser();
  if (!d) return fail(RMD_HIP_ERR_RUNTIME, "denoiser_create: out of host memory");
  d->width = width; d->height = height;
  (void)hipGetDevice(&d->device);
  // denoise::DeviceData constructor, depthmap_denoiser.cu:124-141
  d->L = sqrtf(8.0f);
  d->tau = 0.02f;
  d->sigma = (1 / (d->L * d->L)) / d->tau;
  d->theta = 0.5f;
  auto bail = [&](int rc) { rmd_hip_denoiser_destroy(d); return rc; };
  if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(RMD_HIP_ERR_RUNTIME, "denoiser_create: hipStreamCreate failed"));
  rmd_hip_image* f32[] = {&d->u[0], &d->u[1], &d->u_head[0], &d->u_head[1], &d->g};
  for (auto* im : f32) {
    const int rc = image_alloc(im, RMD_HIP_KIND_F32, width, height);
    if (rc != RMD_HIP_OK) return bail(rc);
    im->owner_stream = d->stream;
  }
  for (int k = 0; k < 2; ++k) {
    const int rc = image_alloc(&d->p[k], RMD_HIP_KIND_F32X2, width, height);
    if (rc != RMD_HIP_OK) return bail(rc);
    d->p[k].owner_stream = d->stream;
  }
  if (hipHostMalloc(reinterpret_cast<void**>(&d->h_staging), static_cast<size_t>(width) * height * 4) != hipSuccess)
    return bail(fail(RMD_HIP_ERR_RUNTIME, "denoiser_create: pinned staging buffer"));
  if (hipDeviceSynchronize() != hipSuccess) return bail(fail(RMD_HIP_ERR_RUNTIME, "denoiser_create: device synchronisation failed"));
  *out = d;
  return RMD_HIP_OK;
}

int rmd_hip_denoiser_set_large_sigma_sq(rmd_hip_denoiser_t* d, float depth_range) {
  if (!d) return fail(RMD_HIP_ERR_INVALID_ARG, "set_large_sigma_sq: null handle");
  d->large_sigma_sq = depth_range * depth_range / 72.0f;
  return RMD_HIP_OK;
}

int rmd_hip_denoiser_constants(const rmd_hip_denoiser_t* d, float* out4) {
  if (!d || !out4) return fail(RMD_HIP_ERR_INVALID_ARG, "denoiser_constants: null argument");
  out4[0] = d->L; out4[1] = d->tau; out4[2] = d->sigma; out4[3] = d->theta;
  return RMD_HIP_OK;
}

int rmd_hip_denoiser_set_option(rmd_hip_denoiser_t* d, int option, int value) {
  if (!d) return fail(RMD_HIP_ERR_INVALID_ARG, "denoiser_set_option: null handle");
  switch (option) {
    case RMD_HIP_DENOISE_OPT_TIMING: d->opt_timing = value != 0; return RMD_HIP_OK;
    case RMD_HIP_DENOISE_OPT_ITERS_PER_LAUNCH:
      if (value < 0 || value > 8) return fail(RMD_HIP_ERR_INVALID_ARG, "iters_per_launch %d outside 0..8", value);
      d->opt_iters_per_launch = value;
      return RMD_HIP_OK;
    case RMD_HIP_DENOISE_OPT_GEOMETRY:
      if (value < 0 || value > 5) return fail(RMD_HIP_ERR_INVALID_ARG, "geometry %d outside 0..5", value);
      d->opt_geometry = value;
      return RMD_HIP_OK;
    default: return fail(RMD_HIP_ERR_INVALID_ARG, "denoiser_set_option: unknown option %d", option);
  }
}

int rmd_hip_denoiser_denoise(rmd_hip_denoiser_t* d, const rmd_hip_image_t* mu, const rmd_hip_image_t* sigma_sq,
                             const rmd_hip_image_t* a, const rmd_hip_image_t* b, float* host_denoised, float lambda,
                             int iterations) {
  if (!d || !mu || !sigma_sq || !a || !b) return fail(RMD_HIP_ERR_INVALID_ARG, "denoise: null argument");
  if (iterations < 0) return fail(RMD_HIP_ERR_INVALID_ARG, "denoise: negative iteration count");
  if (d->large_sigma_sq < 0.0f) return fail(RMD_HIP_ERR_NOT_READY, "denoise: setLargeSigmaSq must be called before this method");
  const rmd_hip_image_t* ins[4] = {mu, sigma_sq, a, b};
  for (auto* im : ins) {
    if (im->kind != RMD_HIP_KIND_F32 || im->width != d->width || im->height != d->height || im->stride != mu->stride)
      return fail(RMD_HIP_ERR_INVALID_ARG, "denoise: input planes must be f32 %dx%d with one common stride", d->width, d->height);
  }
  int cur = -1;
  HIP_TRY(hipGetDevice(&cur));
  if (cur != d->device) HIP_TRY(hipSetDevice(d->device));
  // inputs may still be written by their owner (the SeedMatrix's last kernel is left in flight)
  for (auto* im : ins)
    if (im->owner_seeds || (im->owner_stream && im->owner_stream != d->stream)) TRY(image_settle(im));
  HIP_TRY(hipStreamSynchronize(d->stream));
  d->timer.drain();
  d->timer.reset();

  rmdk::TvParams P;
  P.w = d->width; P.h = d->height;
  P.stride = static_cast<int>(d->g.stride);
  P.stride2 = static_cast<int>(d->p[0].stride);
  P.mu = static_cast<const float*>(mu->data);
  P.sigma_sq = static_cast<const float*>(sigma_sq->data);
  P.a = static_cast<const float*>(a->data);
  P.b = static_cast<const float*>(b->data);
  P.in_stride = static_cast<int>(mu->stride);
  P.g = static_cast<float*>(d->g.data);
  P.large_sigma_sq = d->large_sigma_sq;
  P.tau = d->tau; P.sigma = d->sigma; P.theta = d->theta; P.lambda = lambda;
  P.members = nullptr; P.member_stride = 0; P.member_stride2 = 0;

  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (d->opt_timing) {  // one event pair around the whole iteration loop (per-launch markers would serialise it)
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
  }
  int cur_buf = 0;
  long n_launches = 0;
  float* us[2] = {static_cast<float*>(d->u[0].data), static_cast<float*>(d->u[1].data)};
  float* uhs[2] = {static_cast<float*>(d->u_head[0].data), static_cast<float*>(d->u_head[1].data)};
  float2* ps[2] = {static_cast<float2*>(d->p[0].data), static_cast<float2*>(d->p[1].data)};
  TRY(tv_run(P, us, uhs, ps, 1, iterations, d->opt_iters_per_launch, d->opt_geometry, d->stream, ev0, &cur_buf, &n_launches));
  HIP_TRY(hipGetLastError());
  if (d->opt_timing) HIP_TRY(hipEventRecord(ev1, d->stream));
  d->result_index = cur_buf;
  const size_t out_bytes = static_cast<size_t>(d->width) * d->height * 4;
  if (host_denoised) {
    const rmd_hip_image& r = d->u[cur_buf];
    const size_t row = static_cast<size_t>(r.width) * 4;
    HIP_TRY(hipMemcpy2DAsync(d->h_staging, row, r.data, r.pitch, row, r.height, hipMemcpyDeviceToHost, d->stream));
  }
  HIP_TRY(hipStreamSynchronize(d->stream));
  if (host_denoised) memcpy(host_denoised, d->h_staging, out_bytes);
  if (d->opt_timing) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) { d->timer.total_ms = ms; d->timer.launches = n_launches; }
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
  }
  return RMD_HIP_OK;
}

int rmd_hip_denoiser_result(const rmd_hip_denoiser_t* d, const rmd_hip_image_t** view) {
  if (!d || !view) return fail(RMD_HIP_ERR_INVALID_ARG, "denoiser_result: null argument");
  *view = &d->u[d->result_index];
  return RMD_HIP_OK;
}

int rmd_hip_denoiser_timing(const rmd_hip_denoiser_t* d, double* total_ms, long* launches) {
  if (!d) return fail(RMD_HIP_ERR_INVALID_ARG, "denoiser_timing: null handle");
  if (total_ms) *total_ms = d->timer.total_ms;
  if (launches) *launches = d->timer.launches;
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
  if (!dn.ready) {
    rmd_hip_image* f32[] = {&dn.u[0], &dn.u[1], &dn.u_head[0], &dn.u_head[1], &dn.g};
    for (auto* im : f32) TRY(image_alloc(im, RMD_HIP_KIND_F32, w, h * n));
    for (int k = 0; k < 2; ++k) TRY(image_alloc(&dn.p[k], RMD_HIP_KIND_F32X2, w, h * n));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dn.d_table), static_cast<size_t>(rmdk::MAX_BATCH) * rmdk::TV_MEMBER_WORDS * sizeof(unsigned long long)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&dn.h_staging), static_cast<size_t>(n) * w * h * sizeof(float)));
    HIP_TRY(hipStreamCreateWithFlags(&dn.stream, hipStreamNonBlocking));
    HIP_TRY(hipDeviceSynchronize());
    dn.ready = true;
  }
  // the members' state must be final and at rest: deferred finalisations, then every group's stream (the members' last kernels are left in flight)
  for (int i = 0; i < n; ++i) TRY(seeds_flush(b->members[i]));
  for (int g = 0; g < b->n_groups; ++g) HIP_TRY(hipStreamSynchronize(b->groups[g].stream));
  for (int g = 0; g < b->n_groups; ++g) TRY(ingest_error_check(b->groups[g].h_progress));
  HIP_TRY(hipStreamSynchronize(dn.stream));
  unsigned long long table[rmdk::MAX_BATCH * rmdk::TV_MEMBER_WORDS] = {};
  for (int i = 0; i < n; ++i) {
    const rmd_hip_seeds* m = b->members[i];
    const float large_sigma_sq = depth_range[i] * depth_range[i] / 72.0f;  // DepthmapDenoiser::setLargeSigmaSq, depthmap_denoiser.cu:226-229
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
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  HIP_TRY(hipEventCreate(&ev0));
  HIP_TRY(hipEventCreate(&ev1));
  float* us[2] = {static_cast<float*>(dn.u[0].data), static_cast<float*>(dn.u[1].data)};
  float* uhs[2] = {static_cast<float*>(dn.u_head[0].data), static_cast<float*>(dn.u_head[1].data)};
  float2* ps[2] = {static_cast<float2*>(dn.p[0].data), static_cast<float2*>(dn.p[1].data)};
  int cur_buf = 0;
  long n_launches = 0;
  const int rc = tv_run(P, us, uhs, ps, n, iterations, 0, 0, dn.stream, ev0, &cur_buf, &n_launches);
  if (rc != RMD_HIP_OK) { (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1); return rc; }
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
  (void)hipEventDestroy(ev0);
  (void)hipEventDestroy(ev1);
  return RMD_HIP_OK;
}

int rmd_hip_batch_denoise_result(const rmd_hip_batch_t* b, int index, const rmd_hip_image_t** view) {
  if (!b || !view) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_denoise_result: null argument");
  if (index < 0 || index >= b->n) return fail(RMD_HIP_ERR_INVALID_ARG, "batch_denoise_result: index %d outside [0, %d)", index, b->n);
  if (!b->dn.ready || !b->dn.result[index].data) return fail(RMD_HIP_ERR_NOT_READY, "batch_denoise_result: rmd_hip_batch_denoise has not run");
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

// ---- ImageReducer ---------------------------------------------------------------------------
namespace {

// scratch of the reducers: allocated once per device (the reference keeps dev_partial_ / dev_final_ in the object,
// reduction.cu:29-73), used on the null stream under a lock
struct ReduceScratch {
  double* parts = nullptr;            // 8 x 64 fp64 partials
  float* out_f32 = nullptr;
  unsigned long long* out_u64 = nullptr;
};
constexpr int MAX_DEVICES = 64;
ReduceScratch g_reduce_scratch[MAX_DEVICES];
std::mutex g_reduce_mutex;

int reduce_scratch(ReduceScratch** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= MAX_DEVICES) return fail(RMD_HIP_ERR_RUNTIME, "reduce: device index %d", dev);
  ReduceScratch& r = g_reduce_scratch[dev];
  if (!r.parts) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&r.parts), 8 * 64 * sizeof(double)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&r.out_f32), sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&r.out_u64), sizeof(unsigned long long)));
  }
  *out = &r;
  return RMD_HIP_OK;
}

int reduce_sum_f32_dev(const float* data, size_t stride, size_t width, size_t height, float* sum) {
  if (width == 0 || height == 0 || width > 0x7fffffff || height > 0x7fffffff || stride > 0x7fffffff) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_sum: bad shape");
  std::lock_guard<std::mutex> lock(g_reduce_mutex);
  ReduceScratch* r = nullptr;
  TRY(reduce_scratch(&r));
  const int w = static_cast<int>(width), h = static_cast<int>(height);
  const dim3 block(256), grid((w + 255) / 256 < 8 ? (w + 255) / 256 : 8, h < 64 ? h : 64);
  hipLaunchKernelGGL(rmdk::sum_partial_kernel, grid, block, 0, nullptr, data, w, h, static_cast<int>(stride), r->parts);
  hipLaunchKernelGGL(rmdk::sum_final_kernel, dim3(1), dim3(64), 0, nullptr, r->parts, static_cast<int>(grid.x * grid.y), r->out_f32);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(sum, r->out_f32, sizeof(float), hipMemcpyDeviceToHost));
  return RMD_HIP_OK;
}

int reduce_u64_dev(bool count_eq, const int* data, size_t stride, size_t width, size_t height, int value, unsigned long long* result) {
  if (width == 0 || height == 0 || width > 0x7fffffff || height > 0x7fffffff || stride > 0x7fffffff) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce: bad shape");
  std::lock_guard<std::mutex> lock(g_reduce_mutex);
  ReduceScratch* r = nullptr;
  TRY(reduce_scratch(&r));
  const int w = static_cast<int>(width), h = static_cast<int>(height);
  HIP_TRY(hipMemsetAsync(r->out_u64, 0, sizeof(unsigned long long), nullptr));
  const dim3 block(256), grid((w + 255) / 256, h < 64 ? h : 64);
  if (count_eq) hipLaunchKernelGGL(rmdk::count_eq_kernel, grid, block, 0, nullptr, data, w, h, static_cast<int>(stride), value, r->out_u64);
  else hipLaunchKernelGGL(rmdk::sum_i32_kernel, grid, block, 0, nullptr, data, w, h, static_cast<int>(stride), r->out_u64);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(result, r->out_u64, sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return RMD_HIP_OK;
}

}  // namespace

extern "C" {

int rmd_hip_reduce_sum_f32(const rmd_hip_image_t* img, float* sum) {
  if (!img || !sum) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_sum: null argument");
  if (img->kind != RMD_HIP_KIND_F32) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_sum: image is not f32");
  ScopedDevice dev(img->device);
  TRY(image_settle(img));
  return reduce_sum_f32_dev(static_cast<const float*>(img->data), img->stride, img->width, img->height, sum);
}

int rmd_hip_reduce_sum_i32(const rmd_hip_image_t* img, int* sum) {
  if (!img || !sum) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_sum_i32: null argument");
  if (img->kind != RMD_HIP_KIND_I32) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_sum_i32: image is not i32");
  ScopedDevice dev(img->device);
  TRY(image_settle(img));
  unsigned long long r = 0;
  TRY(reduce_u64_dev(false, static_cast<const int*>(img->data), img->stride, img->width, img->height, 0, &r));
  *sum = static_cast<int>(static_cast<unsigned int>(r));
  return RMD_HIP_OK;
}

int rmd_hip_reduce_count_eq_i32(const rmd_hip_image_t* img, int value, size_t* count) {
  if (!img || !count) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_count_eq: null argument");
  if (img->kind != RMD_HIP_KIND_I32) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_count_eq: image is not i32");
  ScopedDevice dev(img->device);
  TRY(image_settle(img));
  unsigned long long r = 0;
  TRY(reduce_u64_dev(true, static_cast<const int*>(img->data), img->stride, img->width, img->height, value, &r));
  *count = static_cast<size_t>(r);
  return RMD_HIP_OK;
}

// the raw-pointer forms of the reference (reduction.cuh:33-47): device pointers on the current device; the caller is
// responsible for the data being complete (as with the reference, which launches on the default stream)
int rmd_hip_reduce_sum_f32_raw(const float* dev_data, size_t stride_elems, size_t width, size_t height, float* sum) {
  if (!dev_data || !sum) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_sum_raw: null argument");
  HIP_TRY(hipDeviceSynchronize());
  return reduce_sum_f32_dev(dev_data, stride_elems, width, height, sum);
}
int rmd_hip_reduce_sum_i32_raw(const int* dev_data, size_t stride_elems, size_t width, size_t height, int* sum) {
  if (!dev_data || !sum) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_sum_i32_raw: null argument");
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long r = 0;
  TRY(reduce_u64_dev(false, dev_data, stride_elems, width, height, 0, &r));
  *sum = static_cast<int>(static_cast<unsigned int>(r));
  return RMD_HIP_OK;
}
int rmd_hip_reduce_count_eq_i32_raw(const int* dev_data, size_t stride_elems, size_t width, size_t height, int value, size_t* count) {
  if (!dev_data || !count) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_count_eq_raw: null argument");
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long r = 0;
  TRY(reduce_u64_dev(true, dev_data, stride_elems, width, height, value, &r));
  *count = static_cast<size_t>(r);
  return RMD_HIP_OK;
}

// ---- self test of the float-frame examination (host code, the same entry the update path uses) ----
int rmd_hip_selftest_pack_float_frame(const float* host_img, int width, int height, int pitch, unsigned char* bytes, int* all_levels) {
  if (!host_img || !bytes || !all_levels || width <= 0 || height <= 0 || pitch < width)
    return fail(RMD_HIP_ERR_INVALID_ARG, "selftest_pack_float_frame: bad argument");
  *all_levels = CopyPool::instance().pack(host_img, bytes, width, height, pitch) ? 1 : 0;
  return RMD_HIP_OK;
}

// ---- self test of the wave primitives the kernels rely on -----------------------------------
int rmd_hip_selftest_wave_primitives(int* mismatching_lanes) {
  if (!mismatching_lanes) return fail(RMD_HIP_ERR_INVALID_ARG, "selftest: null output");
  unsigned int* d = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), sizeof(unsigned int)));
  hipError_t e = hipMemset(d, 0, sizeof(unsigned int));
  if (e == hipSuccess) {
    for (unsigned int seed = 1; seed <= 8; ++seed) hipLaunchKernelGGL(rmdk::wave_primitives_selftest_kernel, dim3(64), dim3(64), 0, nullptr, seed, d);
    e = hipGetLastError();
  }
  unsigned int bad = 0;
  if (e == hipSuccess) e = hipMemcpy(&bad, d, sizeof(bad), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(RMD_HIP_ERR_RUNTIME, "selftest: %s", hipGetErrorString(e));
  *mismatching_lanes = static_cast<int>(bad);
  return RMD_HIP_OK;
}

// ---- arithmetic-contract self test ----------------------------------------------------------
int rmd_hip_math_eval(int op, const float* x, const float* y, const float* z, float* out, size_t n) {
  if (!x || !out || n == 0 || op < 0 || op > 6) return fail(RMD_HIP_ERR_INVALID_ARG, "math_eval: bad argument");
  if ((op == 5 && !y) || (op == 6 && (!y || !z))) return fail(RMD_HIP_ERR_INVALID_ARG, "math_eval: missing operand");
  float *dx = nullptr, *dy = nullptr, *dz = nullptr, *dout = nullptr;
  const size_t bytes = n * sizeof(float);
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dx), bytes));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dy), bytes));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dz), bytes));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dout), bytes));
  const hipError_t e_in = [&] {
    hipError_t r = hipMemcpy(dx, x, bytes, hipMemcpyHostToDevice);
    if (r == hipSuccess) r = hipMemcpy(dy, y ? y : x, bytes, hipMemcpyHostToDevice);
    if (r == hipSuccess) r = hipMemcpy(dz, z ? z : x, bytes, hipMemcpyHostToDevice);
    return r;
  }();
  if (e_in != hipSuccess) {
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dz); (void)hipFree(dout);
    return fail(RMD_HIP_ERR_RUNTIME, "math_eval: upload failed: %s", hipGetErrorString(e_in));
  }
  hipLaunchKernelGGL(rmdk::math_eval_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, op, dx, dy,
                     dz, dout, n);
  const hipError_t e = hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
  (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dz); (void)hipFree(dout);
  if (e != hipSuccess) return fail(RMD_HIP_ERR_RUNTIME, "math_eval: %s", hipGetErrorString(e));
  return RMD_HIP_OK;
}

}  // extern "C"
