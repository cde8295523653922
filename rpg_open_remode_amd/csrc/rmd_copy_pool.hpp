// Host threads that copy (and examine) callers' frames into pinned staging buffers; shared by the frame sources of a SeedMatrix
// (rmd_ingest.hip), of a batch (rmd_batch.hip) and the self test (rmd_reduce.hip).  One pool per process (the instance lives in an inline
// function).
#ifndef RMD_COPY_POOL_HPP
#define RMD_COPY_POOL_HPP

#include "rmd_host.hpp"

// Copy of a caller's frame into a pinned staging buffer.  A float VGA frame is 1.2 MB: one core moves it in ~50 us, which made the host,
// not the GPU, the bound of update(float*) -- the reference's own signature (seed_matrix.cu:120-128).  Frames of 256 KB and more are
// split across a few persistent helper threads (created at the first such copy, parked on a condition variable in between).
namespace rmdh {
// Float frames whose every pixel is an 8-bit level -- what the reference's own host path produces: Depthmap::inputImage converts the 8-bit
// camera image with convertTo(CV_32F, 1.0f / 255.0f) and hands the floats to SeedMatrix::update (depthmap.cpp:105, 75-77) -- travel to the
// device as bytes: a quarter of the copy-engine time, which is what bounds float frames (1.2 MB at the engine's 23 GB/s take longer than
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

// A frame's way into a pinned buffer that a copy engine reads next: streaming stores.  memcpy's ordinary stores first READ every line of
// the destination into the cache (write-allocate) and leave the frame there for the engine to snoop out again; 32-byte non-temporal stores
// write it to memory once (glibc switches to them only above ~3/4 of the shared cache, tens of megabytes).  For the large copies only (1 MB
// and more per frame or batch step: 1920x1080 27 400-28 400 -> 28 800-32 900 Mpix/s, a batch of 16 16 250-16 330 -> 16 450-16 510): a
// 307-KB frame that is read by the engine microseconds later is better left in the cache (640x480: 9.5-10.4 us per copy against 11.5
// streamed). `dst` 32-byte aligned (pinned buffers are page-aligned, the pool's chunks multiples of 4 KB), else memcpy.
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
typedef long long rmd_v4di __attribute__((vector_size(32), aligned(32)));
typedef long long rmd_v4di_u __attribute__((vector_size(32), aligned(1)));
__attribute__((target("avx2"))) static void frame_copy_avx2(char* dst, const char* src, size_t n) {
  size_t i = 0;
  for (; i + 128 <= n; i += 128) {
    const rmd_v4di a = *reinterpret_cast<const rmd_v4di_u*>(src + i), b = *reinterpret_cast<const rmd_v4di_u*>(src + i + 32);
    const rmd_v4di c = *reinterpret_cast<const rmd_v4di_u*>(src + i + 64), d = *reinterpret_cast<const rmd_v4di_u*>(src + i + 96);
    __builtin_nontemporal_store(a, reinterpret_cast<rmd_v4di*>(dst + i));
    __builtin_nontemporal_store(b, reinterpret_cast<rmd_v4di*>(dst + i + 32));
    __builtin_nontemporal_store(c, reinterpret_cast<rmd_v4di*>(dst + i + 64));
    __builtin_nontemporal_store(d, reinterpret_cast<rmd_v4di*>(dst + i + 96));
  }
  for (; i + 32 <= n; i += 32) {
    const rmd_v4di a = *reinterpret_cast<const rmd_v4di_u*>(src + i);
    __builtin_nontemporal_store(a, reinterpret_cast<rmd_v4di*>(dst + i));
  }
  if (i < n) memcpy(dst + i, src + i, n - i);
  __atomic_thread_fence(__ATOMIC_SEQ_CST);  // (the streamed lines are globally visible before the engine is told to read them)
}
inline void frame_copy(void* dst, const void* src, size_t n, bool stream) {
  static const bool avx2 = __builtin_cpu_supports("avx2") != 0;
  if (stream && avx2 && (reinterpret_cast<uintptr_t>(dst) & 31u) == 0)
    frame_copy_avx2(static_cast<char*>(dst), static_cast<const char*>(src), n);
  else
    memcpy(dst, src, n);
}
#else
inline void frame_copy(void* dst, const void* src, size_t n, bool) { memcpy(dst, src, n); }
#endif

class CopyPool {
 public:
  // the pool (RMD_HIP_TUNE_COPY_THREADS participants): examination of float frames (pack) and the frames of a batch step (copy_many)
  static CopyPool& instance() {
    static CopyPool pool(tunables().v[RMD_HIP_TUNE_COPY_THREADS] - 1);
    return pool;
  }
  // ... and a pool of ONE helper for the plain copy of one frame (copy): that is bound by waking the helpers, not by bandwidth --
  // 1920x1080 (2 MB per 60 us) with two participants 31 900-33 300 Mpix/s at 1.1 host cores, with four 29 700-31 000 at 2.0, with eight
  // 28 900-31 100 at 3.1-3.7, and one alone (73 us per frame) cannot keep up -- whereas the examination is arithmetic and wants them all
  // (1280x960 float frames: 11 300-12 300 Mpix/s with four, 9 060 with two).
  static CopyPool& pair() {
    static CopyPool pool(tunables().v[RMD_HIP_TUNE_COPY_THREADS] > 1 ? 1 : 0);
    return pool;
  }
  struct Segment { void* dst; const void* src; };
  // several buffers of `bytes` each (the frames of one batch step): the participants take whole buffers in turn
  void copy_many(const Segment* segs, int n, size_t bytes) {
    if (n == 1) { pair().copy(segs[0].dst, segs[0].src, bytes); return; }
    if (n_workers_ == 0 || bytes * static_cast<size_t>(n) < kMinBytes) {
      for (int i = 0; i < n; ++i) frame_copy(segs[i].dst, segs[i].src, bytes, bytes * static_cast<size_t>(n) >= kMinBytes);
      return;
    }
    std::lock_guard<std::mutex> one_copy_at_a_time(call_mutex_);
    segs_ = segs; n_segs_ = n; bytes_ = bytes;
    post();
    for (int i = n_workers_; i < n; i += n_workers_ + 1) frame_copy(segs[i].dst, segs[i].src, bytes, true);  // the caller's share
    wait();
    segs_ = nullptr; n_segs_ = 0;
  }
  // pack_float_rows_u8 over the rows of one frame, split over the participants; true if every row was accepted
  bool pack(const float* src, unsigned char* dst, int w, int h, int pitch) {
    // an image of other floats is turned down before anybody is woken
    if (!pack_float_rows_u8(src, dst, w, pitch, h / 2, h / 2 + 1)) return false;
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
      frame_copy(dst, src, bytes, bytes >= kMinBytes);
      return;
    }
    std::lock_guard<std::mutex> one_copy_at_a_time(call_mutex_);
    const size_t chunk = ((bytes + parts - 1) / parts + 4095) & ~static_cast<size_t>(4095);
    dst_ = static_cast<char*>(dst); src_ = static_cast<const char*>(src); bytes_ = bytes; chunk_ = chunk;
    post();
    const size_t mine = static_cast<size_t>(n_workers_) * chunk;  // the caller takes the last part
    if (mine < bytes) frame_copy(dst_ + mine, src_ + mine, bytes - mine, true);
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
  explicit CopyPool(int n) {  // n helpers (the tunable is read when a pool is created: at the first large host frame)
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
          cv_.wait(lk, [&] { return __atomic_load_n(&stop_, __ATOMIC_ACQUIRE) || __atomic_load_n(&generation_,
              __ATOMIC_SEQ_CST) != seen; });
          __atomic_fetch_sub(&parked_, 1, __ATOMIC_SEQ_CST);
        }
      }
      if (__atomic_load_n(&stop_, __ATOMIC_ACQUIRE)) return;
      seen = __atomic_load_n(&generation_, __ATOMIC_ACQUIRE);
      if (pack_src_) {
        const int y0 = index * pack_rows_, y1 = y0 + pack_rows_ < pack_h_ ? y0 + pack_rows_ : pack_h_;
        if (y0 < y1 && !pack_float_rows_u8(pack_src_, pack_dst_, pack_w_, pack_pitch_, y0, y1)) __atomic_store_n(&pack_ok_, 0,
            __ATOMIC_RELAXED);
      } else if (segs_) {
        for (int i = index; i < n_segs_; i += n_workers_ + 1) frame_copy(segs_[i].dst, segs_[i].src, bytes_, true);
      } else {
        const size_t off = static_cast<size_t>(index) * chunk_;
        if (off < bytes_) frame_copy(dst_ + off, src_ + off, bytes_ - off < chunk_ ? bytes_ - off : chunk_, true);
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
// (A/B: RMD_HIP_TUNE_FLOAT_AS_BYTES = 0 sends every float frame as floats)
inline bool float_frames_as_bytes() { return tunables().v[RMD_HIP_TUNE_FLOAT_AS_BYTES] != 0; }
inline void host_copy(void* dst, const void* src, size_t bytes) { CopyPool::pair().copy(dst, src, bytes); }
}  // namespace rmdh

#endif  // RMD_COPY_POOL_HPP
