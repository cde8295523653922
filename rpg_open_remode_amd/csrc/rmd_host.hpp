// Host side of librmd_hip.so shared by its translation units: the handle structs behind the opaque types of include/rmd_hip.h, error
// plumbing, and the functions one unit needs from another.  Who owns what:
//   rmd_capi.hip     library / device / rmd::DeviceImage / rmd::SeedMatrix entry points (everything but the frame sources), observers
//   rmd_update.hip   the ONLY unit that instantiates the seed kernels: seed_init, the two-launch update pipeline for one sequence or a
//   batch,
//                    the stand-alone finalisation, the per-pixel A/B baseline
//   rmd_ingest.hip   frames handed over in host memory: pinned ring, staging copies and arrival flags, conversion one step ahead,
//                    float frames that travel as bytes, lens-undistortion maps
//   rmd_batch.hip    rmd_hip_batch_*: several SeedMatrix objects stepped by one launch pair, TV-L1 for all of them in one launch sequence
//   rmd_denoise.hip  rmd::DepthmapDenoiser
//   rmd_reduce.hip   rmd::ImageReducer, the self tests
//   rmd_publish.hip  point cloud and coloured convergence map (the steps after the path)
// Every kernel has ONE home unit (the one that launches it): kernels are defined in their unit's .hip file or in a header only that unit
// includes (rmd_frame.hpp -> rmd_update.hip, rmd_tv_kernels.hpp -> rmd_denoise.hip); rmd_kernels.hpp / rmd_matcher.hpp hold what the
// host units share (parameter blocks, device functions, kernel TEMPLATES, which are emitted only where they are instantiated).
#ifndef RMD_HOST_HPP
#define RMD_HOST_HPP

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

#define RMD_HIP_VERSION_NUMBER 400

namespace rmdh {

// description of the last failure on the calling thread (rmd_hip_last_error); returns `code`
int fail(int code, const char* fmt, ...);
const char* last_error();

#define HIP_TRY(expr)                                                                                      \
  do {                                                                                                     \
    const hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess)                                                                                  \
      return ::rmdh::fail(RMD_HIP_ERR_RUNTIME, "%s failed: %s (%d) at %s:%d", #expr, hipGetErrorString(e_),       \
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

inline size_t kind_size(int kind) { return kind == RMD_HIP_KIND_F32X2 ? 8 : 4; }

// Poses on the host: se3.cuh:78-95 (inverse), :144-162 (compose), operation order preserved.
inline rmdk::Pose pose_inverse(const rmdk::Pose& p) {
  rmdk::Pose r;
  const float* d = p.d;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.d[4 * i + j] = d[4 * j + i];
  for (int i = 0; i < 3; ++i) r.d[4 * i + 3] = -d[i] * d[3] - d[4 + i] * d[7] - d[8 + i] * d[11];
  return r;
}
inline rmdk::Pose pose_compose(const rmdk::Pose& l, const rmdk::Pose& r) {
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

}  // namespace rmdh

namespace rmdh { class CopyEngines; }
using rmdh::CopyEngines;

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


// The "frame n has arrived" word that follows every staged host frame on the copy stream.  A pinned -> device copy of up to 16 KB
// is carried out by a shader kernel of the runtime (__amd_rocclr_copyBuffer, tools/ubench/copy_path.hip), which needs wave slots of
// its own: behind a thousand persistent search workgroups it ran 7 us on average and up to 98 us, and with two stream groups of a
// batch overlapping there is hardly ever a gap for it (8-bit host frames, batch of 8: 8 700 instead of 12 000 Mpix/s).  From 64 KB on
// a copy goes to the SDMA engine like the frame itself.  So the flag is 64 KB of the same number: whichever of its words the engine
// writes first or last, a reader of word 0 sees either the old number or the new one, and the new one only after the frame copy in
// front of it (same stream) has completed.  The 64 KB form costs the copy engine 7 us more per frame, all of it latency when the
// device is waiting for the frame (live use: the node asks for the converged count after every update, so the host never runs ahead):
// there the 4-byte form is used -- the device has caught up, so the shader copy finds an empty chip (flag_words()).  (On copy engines
// addressed directly -- rmd_engines.hpp, the default -- the flag is one word, copied by the engine itself behind the frame.)
//
// How a frame that was handed over in host memory reaches the device.  The caller's buffer is always copied into a pinned ring first (the
// caller may reuse it when update() returns); from there
//
//   staged:   a copy engine brings it into a staging buffer in HBM, followed by its arrival flag, with no ordering against the compute
//   stream; the ingest workgroups of the frame's setup kernel wait for the flag and convert the frame into the current-image plane
//   (rmdk::MatcherArgs);
//
//   in place: the ingest workgroups read the pinned ring themselves over the host link: no copy engine, no staging, no flag;
//
//   "+ ahead" (single sequences): frame n + 1 is converted during the SEARCH kernel of frame n whenever it has arrived (staged) or been
//   handed over (in place) by the time setup n runs (rmdk::MatcherArgs::ahead); setup n + 1 then finds nothing left to do.
//
// Defaults (measured): a single sequence uses STAGED on copy engines addressed directly (rmd_engines.hpp; profiles/r06_ab_copy_engines.txt)
// and staged + ahead where its frames travel on the copy stream (RMD_HIP_TUNE_COPY_ENGINES = 0; profiles/r03_h2d.txt) -- the copy engines
// do not touch the CUs, whereas link reads issued by a CU delay the loads of the workgroups it shares its memory pipeline with (in place +
// ahead: search +4.5 us per update).  A batch (rmd_batch.hip decides per step) is staged on one engine, eight staging buffers deep, while a
// step is at most 5 MB, and read in place beyond that or without the engines -- its setup kernels are long enough to hide most of the link
// time.  Frames that go through the lens-undistortion maps are always staged, without ahead (the remap gathers single bytes). A/B:
// RMD_HIP_TUNE_HOST_FRAMES.
enum { HOST_FRAMES_DEFAULT = -1, HOST_FRAMES_STAGED = 0, HOST_FRAMES_STAGED_AHEAD = 1, HOST_FRAMES_INPLACE = 2,
    HOST_FRAMES_INPLACE_AHEAD = 3 };
constexpr int AHEAD_WGS = 128;  // workgroups of the search kernel that bring the next frame in (MatcherArgs::ahead)

// Process-wide settings of the host side (rmd_hip_set_tunable; include/rmd_hip.h lists them).  None changes results.  THE ONE PLACE where
// the library reads its environment: rmdh::tunables() (rmd_capi.hip) fills the defaults from RMD_HIP_<NAME> at its first call; handles
// pick the values up when they are created.
namespace rmdh {
struct Tunables {
  int v[RMD_HIP_NUM_TUNABLES];
};
Tunables& tunables();
}  // namespace rmdh

inline int host_frames_mode(bool batch) {
  const int forced = rmdh::tunables().v[RMD_HIP_TUNE_HOST_FRAMES];
  if (forced != HOST_FRAMES_DEFAULT) return forced;
  // (a single sequence: with the frames on copy engines addressed directly they arrive early enough for the frame's OWN setup kernel, whose
  // ingest workgroups convert beside the tiles' latency chains for nothing; one step ahead costs the search kernel its bringers)
  return batch ? HOST_FRAMES_INPLACE : rmdh::tunables().v[RMD_HIP_TUNE_COPY_ENGINES] != 0 ? HOST_FRAMES_STAGED : HOST_FRAMES_STAGED_AHEAD;
}
inline bool frame_in_place(bool batch, bool remap) {
  const int m = host_frames_mode(batch);
  return !remap && (m == HOST_FRAMES_INPLACE || m == HOST_FRAMES_INPLACE_AHEAD);
}
inline bool frame_ahead(bool remap) {
  const int m = host_frames_mode(false);
  return !remap && (m == HOST_FRAMES_STAGED_AHEAD || m == HOST_FRAMES_INPLACE_AHEAD);
}
constexpr size_t FLAG_WORDS = 16384;
inline void fill_flag_block(unsigned int* block, unsigned int n, size_t words) {
  for (size_t i = 0; i < words; ++i) block[i] = n;
}
// how many words of the flag block to send behind frame `n`: 1 when the device has already started the previous frame (it is waiting for
// this one, or about to), the whole block while the host runs ahead of it
inline size_t flag_words(const unsigned int* h_progress, unsigned int n) {
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
inline hipError_t create_stream(hipStream_t* out, int level) {
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest)
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
  const int prio = level == 0 ? (least + greatest) / 2 : level == 1 ? greatest : least;  // numerically smaller = higher priority
  return hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio);
}

// diagnostics (RMD_HIP_INGEST_PROFILE): waits for a staging slot that ended because the stream was idle before the word was reached
extern unsigned long g_progress_timeouts;
extern double g_progress_max_wait_us;   // ... and the longest such wait

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
  // (unit target 2: a single sequence's search is a latency chain with a tail; units of half the size shorten the tail now that a unit's
  // staging is cheap: 45.4 -> 44.3 us per update, profiles/r04_unit_target.txt; a batch keeps 1, its tails are filled by the other stream
  // groups)
  int opt_matcher = 3, opt_timing = 0, opt_stats = 0, opt_unit_target = 2;
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
  // A second copy stream (RMD_HIP_TUNE_COPY_STREAMS = 2; OFF by default): consecutive host frames of the fused path alternate between the
  // two. The runtime serves both with ONE copy engine, but one copy's submission overlaps the other's transfer: 640x480 39.1 -> 38.5 us per
  // update (an engine needs ~6 us per command beyond the transfer and every frame is two commands: 22 us per VGA frame, as long as a light
  // update). Off because with it one update() call in ~5 000 -- update 4..7 of a pass -- takes 9-10 ms: 8 of 17 runs of 80 passes against 2
  // of 38 with one stream (profiles/r06_stall_hunt.txt; where the phase timers caught such a call it sat in the memcpy into the pinned
  // slot). A 1.5 % gain is not worth one 17-ms pass in forty.
  hipStream_t copy_stream2 = nullptr;
  // Staged frames on copy engines addressed directly (rmd_engines.hpp; RMD_HIP_TUNE_COPY_ENGINES: 0 = the copy stream above, 1..4 = that
  // many engines in rotation): one completion signal per ring slot for the frame, one for its flag.  engines == nullptr: they cannot be
  // used in this process (or a submission was refused once): the copy stream carries the frames.
  CopyEngines* engines = nullptr;
  int engine_route = 0;
  uint64_t sig_frame[8] = {}, sig_flag[8] = {};  // [RING_MAX]
  unsigned long staged_by_engines = 0, staged_by_stream = 0;  // diagnostics (RMD_HIP_INGEST_PROFILE)
  // frames the caller keeps in pinned memory (rmd_hip_seeds_update_u8_pinned): the next frame is one (set by the entry point, taken by
  // ingest_current_fused), tickets issued, and per ring slot the ticket of the caller's frame an engine may still be reading (0: none)
  bool next_frame_pinned = false;
  unsigned long long pinned_issued = 0, slot_ticket[8] = {};
  CopyEngines* engines_used = nullptr;  // = engines, kept when those are given up: the signals are still waited for
  void* cur_planes[SLOTS] = {};             // current-image planes (pitch of planes[CURR_IMG]), used in rotation; [0] is the handle's own
  int u8_pitch = 0, ingest_slot = 0;
  // fused ingest (tile pipeline): the caller's frame is copied into one of SLOTS pinned buffers, a copy engine moves it to a staging buffer
  // in HBM and then writes the frame's number next to it (copy stream); the setup kernel waits for that number itself, converts the frame
  // into the current-image plane and tells the host through `h_progress` which frames it has consumed. No events, no cross-stream waits:
  // neither queue ever holds a barrier packet for the other. RING slots: the caller may be RING - 1 frames ahead of the setup kernel that
  // has started last.  With three, frame n was handed over when setup n - 2 started and reached HBM 55-60 us later (host copy, submission,
  // 35-45 us of copy engine) -- after setup n - 1 had looked for it, so it was rarely converted one step ahead (rmdk::MatcherArgs::ahead);
  // with four it always is.  The depth is a run-time value (RMD_HIP_TUNE_RING_DEPTH, 3..8; default 4): deeper rings bought nothing at any
  // size (what a late frame waits for is the copy engine, not the caller).
  static constexpr int RING_MAX = 8;
  int ring = 4;
  unsigned char* h_zc_u8[RING_MAX] = {};
  float* h_zc_f32[RING_MAX] = {};
  unsigned char* d_zc_u8[RING_MAX] = {};
  float* d_zc_f32[RING_MAX] = {};
  unsigned int* h_seq = nullptr;            // pinned, one block per slot: the frame number the copy stream writes into d_zc_flag
  // device, one block per ring slot: number of the last frame whose copy into that slot's staging buffer has completed
  unsigned int* d_zc_flag = nullptr;
  // pinned, [kind * RING + slot]: number of the newest 8-bit (kind 0) / float (kind 1) frame that is complete in that ring slot (frames
  // read in place, one step ahead)
  unsigned int* h_submitted = nullptr;
  unsigned int* d_ahead = nullptr;          // device: the words of rmdk::MatcherArgs::ahead
  // float frames that are not 8-bit levels: the next pack_backoff_len frames are not examined (pack_float_rows_u8)
  int pack_backoff = 0, pack_backoff_len = 15;
  unsigned int* h_progress = nullptr;       // pinned: [0] number of the ingested frame whose setup kernel has started, [1] error bits
  unsigned long long zc_number = 0;         // ingested frames so far (the device sees the low 32 bits and compares modulo 2^32)
  int opt_fused_ingest = 1;                 // RMD_HIP_TUNE_FUSED_INGEST = 0 switches back to the copy-stream pipeline (A/B)
  bool ingest_ready = false;                // ingest_init has run
  // test hook (RMD_HIP_OPT_INJECT_FAULT): the arrival flag of the next staged host frame is not sent
  bool inject_withhold_flag = false;
  double ingest_us[4] = {0, 0, 0, 0};  // diagnostics (RMD_HIP_INGEST_PROFILE): host time waiting for a slot, copying, submitting; frames
  // ... and how many frames the caller was ahead of the newest setup kernel that had started when it handed a frame over (<= 0, 1, 2, 3, >=
  // 4)
  // ... the longest single ring wait / copy into the slot / queueing of the copy-engine commands / pair of launches, and the frame it
  // happened at
  double ingest_max_us[4] = {0, 0, 0, 0};
  unsigned long long ingest_max_at[4] = {0, 0, 0, 0};
  unsigned long ingest_lead[5] = {0, 0, 0, 0, 0};
  bool ingest_profile = false;
  rmdh::StageTimer timers[RMD_HIP_NUM_SEED_STAGES];
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
  float4* h_pc_points = nullptr;        // pinned + mapped, W x H points: the point-cloud kernel writes here over the host link
  unsigned int* h_pc_total = nullptr;   // pinned + mapped: their number
  // Publication off the update stream (rmd_hip_seeds_publish_async, rmd_publish.hip), everything allocated at the first request.  One slot
  // per publication in flight: the snapshot of the state it publishes and its products in pinned host memory; the TV-L1 workspace and the
  // point cloud's counters are shared -- the publications run one after the other on pub_stream.
  struct Publication {
    rmd_hip_image mu, sigma_sq, a, b, conv, ref;  // snapshot (written by the update stream, read by pub_stream)
    rmdk::Pose T_world_ref;
    float* h_depth = nullptr;            // pinned, W x H
    float4* h_points = nullptr;          // pinned + mapped, W x H: the point-cloud kernel writes here over the host link
    unsigned int* h_total = nullptr;     // pinned + mapped: number of points
    unsigned char* h_bgr = nullptr;      // pinned, W x H x 3
    int* h_conv = nullptr;               // pinned, W x H
    hipEvent_t snapped = nullptr, done = nullptr;
    unsigned int what = 0;
    int ticket = 0;
    bool pending = false;
    // the TV-L1 launch sequence of this slot (tv_prepare + the blocked iterations on this slot's snapshot) as an instantiated graph: a
    // publication is 50 launches that never change between two requests with the same (depth range, lambda, iterations) -- one
    // hipGraphLaunch instead of 0.15 ms of launch calls on the thread that also feeds the update stream
    hipGraphExec_t tv_exec = nullptr;
    float tv_range = 0.0f, tv_lambda = 0.0f;
    int tv_iterations = -1, tv_result = 0;
  };
  bool pub_graphs = true;  // false after a failed capture: direct launches from then on
  Publication* pub[RMD_HIP_PUBLISH_SLOTS] = {};
  hipStream_t pub_stream = nullptr;
  rmd_hip_image pub_u[2], pub_u_head[2], pub_p[2], pub_g;  // TV-L1 workspace of the publications
  unsigned int* pub_pc_counts = nullptr;
  unsigned char* pub_d_bgr = nullptr;
  int pub_oldest = 0, pub_pending = 0, pub_tickets = 0;
  rmdk::MatcherWorkspace matcher_ws;
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
    unsigned long long slot_step[8] = {};   // host frames: the step of this group's last launch that read staging slot k (0: none)
    unsigned long long last_step = 0;       // ... and of its last launch altogether
    hipEvent_t ev = nullptr;                // fork / join of the region timer
  };
  static constexpr int MAX_GROUPS = 4;
  int n_groups = 0;
  Group groups[MAX_GROUPS];
  hipStream_t copy_stream = nullptr;
  // frames handed over in host memory: the frames of one step, back to back, through SLOTS pinned buffers and SLOTS staging buffers
  // (see ingest_current_fused: the same protocol, one sequence number per step)
  static constexpr int SLOTS_MAX = 8;
  // steps in flight between the caller and the setup kernels (RMD_HIP_TUNE_RING_DEPTH; more than three bought nothing)
  int slots = 3;
  unsigned char* h_stage[SLOTS_MAX] = {};
  unsigned char* d_stage[SLOTS_MAX] = {};
  size_t stage_bytes = 0;                   // capacity of each of the buffers above
  unsigned int* h_seq = nullptr;
  unsigned int* d_flag = nullptr;
  // staged steps on ONE copy engine addressed directly (rmd_engines.hpp; the batch has one flag word and the steps' numbers must reach it
  // in order): the pinned blocks [0, slots) are free when the engine has read them, the staging buffers in HBM are SLOTS_MAX deep
  CopyEngines* engines = nullptr;       // to submit to: given up (nullptr) after a refusal
  CopyEngines* engines_used = nullptr;  // ... whose signals are still waited for
  uint64_t sig_frame[SLOTS_MAX] = {}, sig_flag[SLOTS_MAX] = {};
  unsigned long long pinned_in_place[SLOTS_MAX] = {};  // step whose kernels read pinned block [q] in place, 0: its last step was staged
  unsigned long long step_number = 0;
  int opt_timing = 0, opt_unit_target = 1;
  int pack_backoff = 0;
  // diagnostics (RMD_HIP_INGEST_PROFILE), as for a single SeedMatrix: host time per step waiting for a slot, copying, submitting; steps
  bool ingest_profile = false;
  double ingest_us[4] = {0, 0, 0, 0};
  // steps handed over <= 0, 1, 2, 3, >= 4 ahead of the newest setup kernel of the first group that had started
  unsigned long ingest_lead[5] = {0, 0, 0, 0, 0};
  hipEvent_t region_start = nullptr, region_stop = nullptr;
  long region_updates = 0;
  // TV-L1 for all members in one launch sequence (rmd_hip_batch_denoise), allocated at the first request: the denoiser's planes hold the
  // members back to back (one tall image each: member z = rows [z * height, (z + 1) * height)), the members' input planes come from `table`
  struct Denoise {
    rmd_hip_image u[2], u_head[2], p[2], g;
    unsigned long long* d_table = nullptr;   // rmdk::TV_MEMBER_WORDS words per member (device)
    float* h_staging = nullptr;              // pinned, n x W x H
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr; // around the iteration launches of a run
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
  rmdh::StageTimer timer;
  // around the iteration loop when opt_timing is set; created at first use, destroyed with the handle
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace rmdh {

// per-handle part of a host frame that the setup kernel's ingest workgroups bring in (see ingest_current_fused / batch_update_host)
struct PendingIngest {
  rmdk::IngestArgs common;
  const unsigned int* u8 = nullptr;
  const float* f32 = nullptr;
  const void* next_src = nullptr;  // one step ahead: the next frame's place in the ring and its plane
  bool no_remap = false;           // a float frame that travels as bytes: never through the undistortion maps
  float* next_dst = nullptr;
};

inline bool side_supported(int s) { return s == 3 || s == 5 || s == 7 || s == 9; }
inline double host_now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
inline void cpu_relax() {  // a polite spin, whatever the host architecture
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#else
  std::this_thread::yield();
#endif
}

// rmd_capi.hip
int image_settle(const rmd_hip_image* img);  // wait until the owner of an image (if any) has settled it
int image_alloc(rmd_hip_image* img, int kind, int width, int height);
// every observer of the seed state goes through here: settle deferred work, then wait for the stream
int seeds_sync(const rmd_hip_seeds* s);
int seeds_bind_device(const rmd_hip_seeds* s);
int ingest_error_check(unsigned int* h_progress);
int seeds_after_reference(rmd_hip_seeds* s, const float* T_curr_world, float min_depth, float max_depth);
void seeds_frame_pose(rmd_hip_seeds* s, const float* T_curr_world);
int seeds_after_frame(rmd_hip_seeds* s, const float* T_curr_world, const PendingIngest* ingest = nullptr);
int seeds_create_impl(int width, int height, float fx, float fy, float cx, float cy, int patch_side, int max_extent, rmd_hip_batch* batch,
    int seq, rmd_hip_seeds** out);
int seeds_destroy_impl(rmd_hip_seeds* s);
// rmd_update.hip
int seeds_flush(rmd_hip_seeds* s);           // the deferred finalisation of this handle's last update, as a kernel of its own
int seeds_launch_init(rmd_hip_seeds* s);
int seeds_launch_update(rmd_hip_seeds* s, const PendingIngest* ingest = nullptr);
int batch_launch(rmd_hip_batch* b, unsigned int active, const rmdk::IngestArgs* ingest, const unsigned char* d_stage, size_t frame_bytes);
// rmd_ingest.hip
int ingest_init(rmd_hip_seeds* s);
int ingest_current(rmd_hip_seeds* s, const unsigned char* host_gray, const float* host_f32, const float* T_curr_world);
int ingest_reference(rmd_hip_seeds* s, const unsigned char* host_gray, const float* host_f32, const float* T_curr_world, float min_depth,
    float max_depth);
int wait_for_progress(volatile unsigned int* progress, unsigned int need, hipStream_t stream);
void ingest_release_engines(rmd_hip_seeds* s);  // waits for what the copy engines still hold of this handle; destroys its signals
// rmd_batch.hip
int batch_bind_device(const rmd_hip_batch* b);
// rmd_publish.hip
// waits for the publications in flight and releases everything rmd_hip_seeds_publish_async allocated
void publish_release(rmd_hip_seeds* s);
// rmd_reduce.hip
void launch_count_eq(const int* img, int w, int h, int stride, int value, unsigned long long* out_dev, hipStream_t stream);
// rmd_denoise.hip
int tv_run(const rmdk::TvParams& P, float* const u[2], float* const uh[2], float2* const p[2], int n_z, int iterations,
    int opt_iters_per_launch, int opt_geometry,
           hipStream_t stream, hipEvent_t ev0, int* result_index, long* launches);

}  // namespace rmdh

#endif  // RMD_HOST_HPP
