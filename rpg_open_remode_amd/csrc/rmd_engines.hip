// librmd_hip.so -- staged host frames on copy engines addressed directly (rmd_engines.hpp).  Host code only; the one unit of the
// library that talks to the HSA runtime (libhsa-runtime64, which libamdhip64 itself sits on: it is in every process that runs HIP).
#include "rmd_engines.hpp"
#include "rmd_host.hpp"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <cstdio>
#include <cstring>
#include <mutex>

namespace rmdh {

namespace {

struct AgentSearch {
  uint32_t want_domain = 0, want_bdf = 0;  // of the HIP device
  hsa_agent_t gpu{}, cpu{};
  bool have_gpu = false, have_cpu = false;
  int n_gpus = 0;
  hsa_agent_t only_gpu{};
};

hsa_status_t visit_agent(hsa_agent_t a, void* data) {
  AgentSearch* S = static_cast<AgentSearch*>(data);
  hsa_device_type_t type;
  if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
  if (type == HSA_DEVICE_TYPE_CPU) {
    if (!S->have_cpu) { S->cpu = a; S->have_cpu = true; }  // (which CPU agent is named as the source only tells the runtime the direction)
  } else if (type == HSA_DEVICE_TYPE_GPU) {
    ++S->n_gpus;
    S->only_gpu = a;
    uint32_t bdf = 0, domain = 0;
    if (hsa_agent_get_info(a, static_cast<hsa_agent_info_t>(HSA_AMD_AGENT_INFO_BDFID),
        &bdf) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    (void)hsa_agent_get_info(a, static_cast<hsa_agent_info_t>(HSA_AMD_AGENT_INFO_DOMAIN), &domain);
    // BDFID: bus << 8 | device << 3 | function
    if (!S->have_gpu && (bdf >> 3) == (S->want_bdf >> 3) && domain == S->want_domain) { S->gpu = a; S->have_gpu = true; }
  }
  return HSA_STATUS_SUCCESS;
}

const char* status_text(hsa_status_t st) {
  const char* m = nullptr;
  return hsa_status_string(st, &m) == HSA_STATUS_SUCCESS && m ? m : "unknown HSA status";
}

hsa_signal_t as_signal(uint64_t h) { hsa_signal_t s; s.handle = h; return s; }
hsa_agent_t as_agent(uint64_t h) { hsa_agent_t a; a.handle = h; return a; }

// handles of several threads submit to one CopyEngines object: a refusal is reported to the thread that met it
thread_local char tl_submit_error[160] = {0};

std::mutex g_engines_mutex;
constexpr int MAX_DEVICES = 32;
CopyEngines* g_engines[MAX_DEVICES] = {nullptr};
bool g_engines_tried[MAX_DEVICES] = {false};

}  // namespace

CopyEngines* CopyEngines::for_device(int hip_device, int n_engines) {
  if (hip_device < 0 || hip_device >= MAX_DEVICES) return nullptr;
  std::lock_guard<std::mutex> lock(g_engines_mutex);
  if (!g_engines_tried[hip_device]) {
    g_engines_tried[hip_device] = true;
    CopyEngines* e = new CopyEngines();
    if (e->find_agents(hip_device)) g_engines[hip_device] = e;
    else {
      if (tunables().v[RMD_HIP_TUNE_INGEST_PROFILE]) fprintf(stderr, "[rmd_hip ingest] copy engines not used: %s\n", e->err_);
      delete e;
    }
  }
  CopyEngines* e = g_engines[hip_device];
  if (e && n_engines > e->n_ready_ && !e->first_copies(n_engines)) {
    if (tunables().v[RMD_HIP_TUNE_INGEST_PROFILE]) fprintf(stderr, "[rmd_hip ingest] copy engines not used: %s\n", e->err_);
    return nullptr;
  }
  return e;
}

bool CopyEngines::find_agents(int hip_device) {
  hsa_status_t st = hsa_init();  // (reference-counted: HIP holds the runtime open already)
  if (st != HSA_STATUS_SUCCESS) { snprintf(err_, sizeof err_, "hsa_init: %s", status_text(st)); return false; }
  int bus = 0, dev = 0, domain = 0;
  if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, hip_device) != hipSuccess ||
      hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, hip_device) != hipSuccess ||
      hipDeviceGetAttribute(&domain, hipDeviceAttributePciDomainID, hip_device) != hipSuccess) {
    snprintf(err_, sizeof err_, "the PCI address of HIP device %d is not known", hip_device);
    return false;
  }
  AgentSearch S;
  S.want_domain = static_cast<uint32_t>(domain);
  S.want_bdf = (static_cast<uint32_t>(bus) << 8) | (static_cast<uint32_t>(dev) << 3);
  st = hsa_iterate_agents(visit_agent, &S);
  if (st != HSA_STATUS_SUCCESS) { snprintf(err_, sizeof err_, "hsa_iterate_agents: %s", status_text(st)); return false; }
  const bool by_address = S.have_gpu;
  if (!S.have_gpu && S.n_gpus == 1) { S.gpu = S.only_gpu; S.have_gpu = true; }  // (one GPU in the process: no address needed to tell it)
  if (!S.have_gpu || !S.have_cpu) {
    snprintf(err_, sizeof err_, "no HSA agent at PCI %04x:%02x:%02x (HIP device %d)", domain, bus, dev, hip_device);
    return false;
  }
  gpu_ = S.gpu.handle; cpu_ = S.cpu.handle;
  // (one process per GPU on a node of eight: an engine of the WRONG agent would write this device's memory from another one)
  if (tunables().v[RMD_HIP_TUNE_INGEST_PROFILE])
    fprintf(stderr,
        "[rmd_hip ingest] copy engines of HIP device %d: the HSA agent at PCI %04x:%02x:%02x, %s (%d GPU agent%s in the process)\n",
            hip_device, domain, bus, dev, by_address ? "matched by address" : "the only GPU agent", S.n_gpus, S.n_gpus == 1 ? "" : "s");
  // The order frames rotate in: 0x1, 0x4, 0x2, 0x8 -- not 0x1, 0x2: two frames in flight with their flags behind them take 41 us per
  // 1920x1080 frame on 0x1 / 0x4 and 48 us on 0x1 / 0x2 (as on 0x4 / 0x8: neighbours share something; tools/link_probe.cpp, route G).
  engine_[0] = HSA_AMD_SDMA_ENGINE_0; engine_[1] = HSA_AMD_SDMA_ENGINE_2; engine_[2] = HSA_AMD_SDMA_ENGINE_1; engine_[3] =
      HSA_AMD_SDMA_ENGINE_3;
  return true;
}

// Engines [n_ready_, n) are tried once with a word of their own -- which also makes the runtime create each engine's queue here and not
// inside the first update() that uses it (2-4 ms).  Under g_engines_mutex.
bool CopyEngines::first_copies(int n) {
  if (n > 4) n = 4;
  const hsa_agent_t gpu = as_agent(gpu_), cpu = as_agent(cpu_);
  unsigned int* h_word = nullptr; unsigned int* d_word = nullptr;
  if (hipHostMalloc(reinterpret_cast<void**>(&h_word), 64, hipHostMallocDefault) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&d_word), 64) != hipSuccess) {
    (void)hipGetLastError();
    if (h_word) (void)hipHostFree(h_word);
    snprintf(err_, sizeof err_, "no memory for the engines' first copy");
    return false;
  }
  h_word[0] = 1u;
  const uint64_t sig = create_signal();
  bool ok = sig != 0;
  if (!ok) snprintf(err_, sizeof err_, "hsa_signal_create failed");
  for (int q = n_ready_; ok && q < n; ++q) {
    hsa_signal_store_relaxed(as_signal(sig), 1);
    const hsa_status_t st = hsa_amd_memory_async_copy_on_engine(d_word, gpu, h_word, cpu, 4, 0, nullptr, as_signal(sig),
                                                                static_cast<hsa_amd_sdma_engine_id_t>(engine_[q]), false);
    if (st != HSA_STATUS_SUCCESS) {
      hsa_signal_store_relaxed(as_signal(sig), 0);
      snprintf(err_, sizeof err_, "hsa_amd_memory_async_copy_on_engine (first copy, engine 0x%x): %s", engine_[q], status_text(st));
      ok = false;
    } else if (!wait_idle(sig, 2e6)) {
      snprintf(err_, sizeof err_, "engine 0x%x did not complete its first copy within 2 s", engine_[q]);
      return false;  // (signal and words stay allocated: the engine may still write them)
    } else {
      n_ready_ = q + 1;
    }
  }
  if (sig) destroy_signal(sig);
  (void)hipFree(d_word);
  (void)hipHostFree(h_word);
  return ok;
}

uint64_t CopyEngines::create_signal() {
  hsa_signal_t s;
  // Consumed by the device's engines only (a dependent copy waits for it) and READ by the host (idle()): no interrupt and no event of the
  // kernel driver behind it (an interrupt signal changed neither rate nor host time: gpurun r06_y).
  if (hsa_amd_signal_create(0, 0, nullptr, HSA_AMD_SIGNAL_AMD_GPU_ONLY, &s) != HSA_STATUS_SUCCESS) return 0;
  return s.handle;
}

void CopyEngines::destroy_signal(uint64_t sig) {
  if (sig) (void)hsa_signal_destroy(as_signal(sig));
}

bool CopyEngines::idle(uint64_t sig) const { return hsa_signal_load_scacquire(as_signal(sig)) <= 0; }

bool CopyEngines::wait_idle(uint64_t sig, double timeout_us) const {
  if (idle(sig)) return true;
  const double t0 = host_now_us();
  while (!idle(sig)) {
    if (host_now_us() - t0 > timeout_us) return false;
    cpu_relax();
  }
  return true;
}

const char* CopyEngines::last_error() const { return tl_submit_error; }

bool CopyEngines::submit(unsigned data_engine, unsigned flag_engine, void* dst, const void* src, size_t bytes, uint64_t frame_sig,
                         void* flag_dst, const void* flag_src, size_t flag_bytes, uint64_t flag_sig) {
  const hsa_agent_t gpu = as_agent(gpu_), cpu = as_agent(cpu_);
  const unsigned n_ready = static_cast<unsigned>(n_ready_ > 0 ? n_ready_ : 1);
  const unsigned de = engine_[data_engine % n_ready], fe = engine_[flag_engine % n_ready];
  const hsa_signal_t fs = as_signal(frame_sig);
  hsa_signal_store_relaxed(fs, 1);
  hsa_status_t st =
      hsa_amd_memory_async_copy_on_engine(dst, gpu, src, cpu, bytes, 0, nullptr, fs, static_cast<hsa_amd_sdma_engine_id_t>(de), false);
  if (st != HSA_STATUS_SUCCESS) {
    hsa_signal_store_relaxed(fs, 0);
    snprintf(tl_submit_error, sizeof tl_submit_error, "hsa_amd_memory_async_copy_on_engine (frame, engine 0x%x): %s", de, status_text(st));
    return false;
  }
  if (!flag_dst) return true;
  const hsa_signal_t gs = as_signal(flag_sig);
  hsa_signal_store_relaxed(gs, 1);
  // (the dependency is named on the frame's own engine as well: its queue is served in order, the engine finds the signal at zero)
  st = hsa_amd_memory_async_copy_on_engine(flag_dst, gpu, flag_src, cpu, flag_bytes, 1, &fs, gs,
                                           static_cast<hsa_amd_sdma_engine_id_t>(fe), false);
  if (st != HSA_STATUS_SUCCESS) {
    hsa_signal_store_relaxed(gs, 0);
    snprintf(tl_submit_error, sizeof tl_submit_error, "hsa_amd_memory_async_copy_on_engine (flag, engine 0x%x): %s", fe, status_text(st));
    // (the frame is in flight: the caller sends it again its other way -- the same bytes to the same place -- and waits for frame_sig
    // before the slot's next frame)
    return false;
  }
  return true;
}

}  // namespace rmdh
