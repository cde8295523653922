// What the host link of THIS box delivers for frame-sized transfers, by every route a frame can take into HBM (round 6: is a 1920x1080
// sequence with host frames bound by the link, or by the one copy engine hipMemcpyAsync uses?).  Stand-alone: links the HIP and the HSA
// runtime, nothing of the library.
//   A  hipMemcpyAsync, pinned -> device, back to back on ONE stream (the library's staged path), with and without the 64-KB arrival flag
//   B  the same, frames alternating between TWO streams
//   C  a kernel of the CUs reading the pinned block with 16-byte loads (the library's in-place path), N workgroups
//   D  hsa_amd_memory_async_copy_on_engine: every engine hsa_amd_memory_copy_engine_status reports for host -> device alone, then frames
//      alternating between two of them
//   E  A and C at once (each half of the frames)
//   F  the frame on a data engine, its 4-byte arrival flag a DEPENDENT copy (the engine waits for the frame's completion signal in hardware) on
//      another engine / the other data engine / the same engine; G: frames alternating between two engines, each flag on its frame's engine
// build: hipcc -O2 --offload-arch=gfx950 tools/link_probe.cpp -o build_ab/link_probe -lhsa-runtime64
// usage: link_probe [reps=200]
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define HSA(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = nullptr; hsa_status_string(s_, &m_); fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, m_ ? m_ : "?"); return 1; } } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void read_host_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * blockDim.x)
    dst[i] = __builtin_nontemporal_load(src + i);
}

struct Agents { hsa_agent_t gpu{}, cpu{}; bool have_gpu = false, have_cpu = false; };
static hsa_status_t agent_cb(hsa_agent_t a, void* data) {
  Agents* A = static_cast<Agents*>(data);
  hsa_device_type_t t;
  if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
  if (t == HSA_DEVICE_TYPE_GPU && !A->have_gpu) { A->gpu = a; A->have_gpu = true; }
  if (t == HSA_DEVICE_TYPE_CPU && !A->have_cpu) { A->cpu = a; A->have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  CHECK(hipSetDevice(0));
  const size_t sizes[] = {640 * 480, 1280 * 960, 1920 * 1080, size_t(64) << 20};
  const char* names[] = {"640x480 u8", "1280x960 u8", "1920x1080 u8", "64 MB"};
  const size_t max_bytes = size_t(64) << 20, flag_bytes = 65536;
  const int SLOTS = 4;
  unsigned char* h = nullptr; unsigned char* d = nullptr; unsigned char* hflag = nullptr; unsigned char* dflag = nullptr;
  CHECK(hipHostMalloc(&h, max_bytes * 2, hipHostMallocDefault));
  CHECK(hipMalloc(&d, max_bytes * 2));
  CHECK(hipHostMalloc(&hflag, flag_bytes, hipHostMallocDefault));
  CHECK(hipMalloc(&dflag, flag_bytes));
  memset(h, 7, max_bytes * 2); memset(hflag, 1, flag_bytes);
  hipStream_t s0, s1, sk;
  CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
  CHECK(hipMemcpyAsync(d, h, max_bytes, hipMemcpyHostToDevice, s0));  // warm the route
  CHECK(hipStreamSynchronize(s0));

  Agents A;
  bool hsa_ok = hsa_init() == HSA_STATUS_SUCCESS && hsa_iterate_agents(agent_cb, &A) == HSA_STATUS_SUCCESS && A.have_gpu && A.have_cpu;
  uint32_t engine_mask = 0;
  if (hsa_ok) {
    hsa_status_t st = hsa_amd_memory_copy_engine_status(A.gpu, A.cpu, &engine_mask);
    printf("hsa_amd_memory_copy_engine_status(dst gpu, src cpu): status %d, engines free 0x%x\n", static_cast<int>(st), engine_mask);
    uint32_t pref = 0;
    st = hsa_amd_memory_get_preferred_copy_engine(A.gpu, A.cpu, &pref);
    printf("hsa_amd_memory_get_preferred_copy_engine: status %d, mask 0x%x\n", static_cast<int>(st), pref);
  } else {
    printf("HSA agents not found: routes D skipped\n");
  }

  if (hsa_ok) {  // the runtime creates an engine's queue at its first use: not inside a timed loop
    hsa_signal_t w; HSA(hsa_signal_create(1, 0, nullptr, &w));
    for (uint32_t b = 1; b <= 0x40u; b <<= 1) {
      hsa_signal_store_relaxed(w, 1);
      if (hsa_amd_memory_async_copy_on_engine(d, A.gpu, h, A.cpu, 1 << 20, 0, nullptr, w, static_cast<hsa_amd_sdma_engine_id_t>(b), false) == HSA_STATUS_SUCCESS)
        while (hsa_signal_wait_scacquire(w, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
    }
    hsa_signal_destroy(w);
  }
  for (int rep = 0; rep < 64; ++rep) CHECK(hipMemcpyAsync(dflag, hflag, flag_bytes, hipMemcpyHostToDevice, s0));
  CHECK(hipStreamSynchronize(s0));
  for (int si = 0; si < 4; ++si) {
    const size_t bytes = sizes[si];
    const int n = si == 3 ? 8 : reps;
    auto slot_h = [&](int i) { return h + (si == 3 ? size_t(i & 1) * max_bytes : size_t(i % SLOTS) * ((bytes + 4095) & ~size_t(4095))); };
    auto slot_d = [&](int i) { return d + (si == 3 ? size_t(i & 1) * max_bytes : size_t(i % SLOTS) * ((bytes + 4095) & ~size_t(4095))); };
    auto report = [&](const char* what, double us_total, int frames) {
      printf("  %-14s %-58s %8.1f us per frame  %6.1f GB/s\n", names[si], what, us_total / frames, bytes * double(frames) / us_total / 1e3);
    };
    // A
    for (int with_flag = 0; with_flag < 2; ++with_flag) {
      CHECK(hipDeviceSynchronize());
      const double t0 = now_us();
      for (int i = 0; i < n; ++i) {
        CHECK(hipMemcpyAsync(slot_d(i), slot_h(i), bytes, hipMemcpyHostToDevice, s0));
        if (with_flag) CHECK(hipMemcpyAsync(dflag, hflag, flag_bytes, hipMemcpyHostToDevice, s0));
      }
      CHECK(hipStreamSynchronize(s0));
      report(with_flag ? "A  hipMemcpyAsync + 64-KB flag, one stream" : "A  hipMemcpyAsync, one stream", now_us() - t0, n);
    }
    // B
    {
      CHECK(hipDeviceSynchronize());
      const double t0 = now_us();
      for (int i = 0; i < n; ++i) {
        hipStream_t s = (i & 1) ? s1 : s0;
        CHECK(hipMemcpyAsync(slot_d(i), slot_h(i), bytes, hipMemcpyHostToDevice, s));
        CHECK(hipMemcpyAsync(dflag + (i & 1) * 4096, hflag, 4096, hipMemcpyHostToDevice, s));
      }
      CHECK(hipStreamSynchronize(s0)); CHECK(hipStreamSynchronize(s1));
      report("B  hipMemcpyAsync + 4-KB flag, two streams alternating", now_us() - t0, n);
    }
    // C
    for (int wgs : {16, 64, 256, 1024}) {
      if (si < 3 && size_t(wgs) * 256 * 16 > bytes * 2) continue;
      CHECK(hipDeviceSynchronize());
      const double t0 = now_us();
      for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL(read_host_kernel, dim3(wgs), dim3(256), 0, sk, reinterpret_cast<const u32x4*>(slot_h(i)), reinterpret_cast<u32x4*>(slot_d(i)), bytes / 16);
      CHECK(hipStreamSynchronize(sk));
      char what[96]; snprintf(what, sizeof what, "C  kernel reading the pinned block, %d workgroups", wgs);
      report(what, now_us() - t0, n);
    }
    // E
    {
      CHECK(hipDeviceSynchronize());
      const double t0 = now_us();
      for (int i = 0; i < n; ++i) {
        if (i & 1) CHECK(hipMemcpyAsync(slot_d(i), slot_h(i), bytes, hipMemcpyHostToDevice, s0));
        else hipLaunchKernelGGL(read_host_kernel, dim3(256), dim3(256), 0, sk, reinterpret_cast<const u32x4*>(slot_h(i)), reinterpret_cast<u32x4*>(slot_d(i)), bytes / 16);
      }
      CHECK(hipStreamSynchronize(s0)); CHECK(hipStreamSynchronize(sk));
      report("E  copy engine and kernel (256 wgs), frames alternating", now_us() - t0, n);
    }
    // D
    if (hsa_ok && engine_mask) {
      std::vector<uint32_t> engines;
      for (uint32_t b = 1; b <= 0x8000u; b <<= 1) if (engine_mask & b) engines.push_back(b);
      std::vector<hsa_signal_t> sig(n);
      for (auto& s : sig) HSA(hsa_signal_create(1, 0, nullptr, &s));
      auto run = [&](uint32_t e0, uint32_t e1, const char* what) -> int {
        for (auto& s : sig) hsa_signal_store_relaxed(s, 1);
        CHECK(hipDeviceSynchronize());
        const double t0 = now_us();
        for (int i = 0; i < n; ++i) {
          // (copies on one engine execute in submission order; a frame's completion signal is its own)
          hsa_status_t st = hsa_amd_memory_async_copy_on_engine(slot_d(i), A.gpu, slot_h(i), A.cpu, bytes, 0, nullptr, sig[i],
                                                                static_cast<hsa_amd_sdma_engine_id_t>((i & 1) ? e1 : e0), false);
          if (st != HSA_STATUS_SUCCESS) { const char* m = nullptr; hsa_status_string(st, &m); printf("  %-14s %s: refused (%s)\n", names[si], what, m ? m : "?"); return 0; }
        }
        for (int i = 0; i < n; ++i)
          while (hsa_signal_wait_scacquire(sig[i], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
        report(what, now_us() - t0, n);
        return 0;
      };
      char what[96];
      for (size_t e = 0; e < engines.size() && e < 4; ++e) {
        snprintf(what, sizeof what, "D  hsa copy on engine 0x%x alone", engines[e]);
        run(engines[e], engines[e], what);
      }
      if (engines.size() >= 2) {
        snprintf(what, sizeof what, "D  hsa copies alternating between engines 0x%x and 0x%x", engines[0], engines[1]);
        run(engines[0], engines[1], what);
      }
      if (engines.size() >= 3) {
        snprintf(what, sizeof what, "D  hsa copies alternating between engines 0x%x and 0x%x", engines[1], engines[2]);
        run(engines[1], engines[2], what);
        snprintf(what, sizeof what, "D  hsa copies alternating between engines 0x%x and 0x%x", engines[0], engines[2]);
        run(engines[0], engines[2], what);
      }
      if (engines.size() >= 5) {
        snprintf(what, sizeof what, "D  hsa copies alternating between engines 0x%x and 0x%x", engines[2], engines[3]);
        run(engines[2], engines[3], what);
        snprintf(what, sizeof what, "D  hsa copies alternating between engines 0x%x and 0x%x", engines[0], engines[4]);
        run(engines[0], engines[4], what);
      }
      // F: what the library would do -- the frame on a data engine, its 4-byte arrival flag on ANOTHER engine that waits for the frame's
      // completion signal (the data engine goes on with the next frame); done = every flag has landed
      {
        std::vector<hsa_signal_t> fsig(n);
        for (auto& s : fsig) HSA(hsa_signal_create(1, 0, nullptr, &s));
        auto run_f = [&](uint32_t e0, uint32_t e1, uint32_t ef, const char* what) -> int {
          for (auto& s : sig) hsa_signal_store_relaxed(s, 1);
          for (auto& s : fsig) hsa_signal_store_relaxed(s, 1);
          CHECK(hipDeviceSynchronize());
          const double t0 = now_us();
          for (int i = 0; i < n; ++i) {
            hsa_status_t st = hsa_amd_memory_async_copy_on_engine(slot_d(i), A.gpu, slot_h(i), A.cpu, bytes, 0, nullptr, sig[i],
                                                                  static_cast<hsa_amd_sdma_engine_id_t>((i & 1) ? e1 : e0), false);
            if (st == HSA_STATUS_SUCCESS)
              st = hsa_amd_memory_async_copy_on_engine(dflag + (i % SLOTS) * 64, A.gpu, hflag + (i % SLOTS) * 64, A.cpu, 4, 1, &sig[i], fsig[i],
                                                       static_cast<hsa_amd_sdma_engine_id_t>(ef == 0xffffu ? ((i & 1) ? e1 : e0) : ef ? ef : ((i & 1) ? e0 : e1)), false);
            if (st != HSA_STATUS_SUCCESS) { const char* m = nullptr; hsa_status_string(st, &m); printf("  %-14s %s: refused (%s)\n", names[si], what, m ? m : "?"); return 0; }
          }
          for (int i = 0; i < n; ++i)
            while (hsa_signal_wait_scacquire(fsig[i], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
          report(what, now_us() - t0, n);
          return 0;
        };
        run_f(1, 1, 4, "F  frames on engine 0x1, dependent 4-byte flags on 0x4");
        run_f(1, 2, 4, "F  frames alternating 0x1 / 0x2, dependent flags on 0x4");
        run_f(1, 2, 0, "F  frames alternating 0x1 / 0x2, flag on the other one");
        run_f(1, 1, 1, "F  frames and dependent flags all on engine 0x1");
        run_f(1, 2, 0xffffu, "G  frames alternating 0x1 / 0x2, a frame's flag on its own");
        run_f(1, 4, 0xffffu, "G  frames alternating 0x1 / 0x4, a frame's flag on its own");
        run_f(2, 4, 0xffffu, "G  frames alternating 0x2 / 0x4, a frame's flag on its own");
        run_f(4, 8, 0xffffu, "G  frames alternating 0x4 / 0x8, a frame's flag on its own");
        run_f(1, 0x10, 0xffffu, "G  frames alternating 0x1 / 0x10, a frame's flag on its own");
        // G with any number of engines: frame i and its flag on engine list[i % n]
        auto run_g = [&](std::vector<uint32_t> list, const char* what) -> int {
          for (auto& s : sig) hsa_signal_store_relaxed(s, 1);
          for (auto& s : fsig) hsa_signal_store_relaxed(s, 1);
          CHECK(hipDeviceSynchronize());
          const double t0 = now_us();
          for (int i = 0; i < n; ++i) {
            const hsa_amd_sdma_engine_id_t e = static_cast<hsa_amd_sdma_engine_id_t>(list[i % list.size()]);
            hsa_status_t st = hsa_amd_memory_async_copy_on_engine(slot_d(i), A.gpu, slot_h(i), A.cpu, bytes, 0, nullptr, sig[i], e, false);
            if (st == HSA_STATUS_SUCCESS)
              st = hsa_amd_memory_async_copy_on_engine(dflag + (i % SLOTS) * 64, A.gpu, hflag + (i % SLOTS) * 64, A.cpu, 4, 1, &sig[i], fsig[i], e, false);
            if (st != HSA_STATUS_SUCCESS) { const char* m = nullptr; hsa_status_string(st, &m); printf("  %-14s %s: refused (%s)\n", names[si], what, m ? m : "?"); return 0; }
          }
          for (int i = 0; i < n; ++i)
            while (hsa_signal_wait_scacquire(fsig[i], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
          report(what, now_us() - t0, n);
          return 0;
        };
        run_g({1, 4, 0x10}, "G  frames over 0x1 / 0x4 / 0x10, a frame's flag on its own");
        run_g({1, 4, 2}, "G  frames over 0x1 / 0x4 / 0x2, a frame's flag on its own");
        run_g({1, 4, 2, 8}, "G  frames over 0x1 / 0x4 / 0x2 / 0x8, a frame's flag on its own");
        run_g({1, 4, 0x10, 0x40}, "G  frames over 0x1 / 0x4 / 0x10 / 0x40, a frame's flag on its own");
        // latency of one frame + flag on an idle link (submit -> flag landed)
        {
          double worst = 0, sum = 0;
          for (int i = 0; i < 20; ++i) {
            hsa_signal_store_relaxed(sig[0], 1); hsa_signal_store_relaxed(fsig[0], 1);
            const double t0 = now_us();
            hsa_amd_memory_async_copy_on_engine(slot_d(0), A.gpu, slot_h(0), A.cpu, bytes, 0, nullptr, sig[0], HSA_AMD_SDMA_ENGINE_0, false);
            hsa_amd_memory_async_copy_on_engine(dflag, A.gpu, hflag, A.cpu, 4, 1, &sig[0], fsig[0], HSA_AMD_SDMA_ENGINE_2, false);
            while (hsa_signal_wait_scacquire(fsig[0], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
            const double dt = now_us() - t0; sum += dt; if (dt > worst) worst = dt;
          }
          printf("  %-14s F  one frame + dependent flag on an idle link: %.1f us avg, %.1f max (submit -> flag landed)\n", names[si], sum / 20, worst);
          sum = 0; worst = 0;
          for (int i = 0; i < 20; ++i) {
            const double t0 = now_us();
            CHECK(hipMemcpyAsync(slot_d(0), slot_h(0), bytes, hipMemcpyHostToDevice, s0));
            CHECK(hipMemcpyAsync(dflag, hflag, flag_bytes, hipMemcpyHostToDevice, s0));
            CHECK(hipStreamSynchronize(s0));
            const double dt = now_us() - t0; sum += dt; if (dt > worst) worst = dt;
          }
          printf("  %-14s A  one hipMemcpyAsync + 64-KB flag on an idle link:  %.1f us avg, %.1f max (submit -> stream idle)\n", names[si], sum / 20, worst);
        }
        for (auto& s : fsig) hsa_signal_destroy(s);
      }
      for (auto& s : sig) hsa_signal_destroy(s);
    }
  }
  return 0;
}
