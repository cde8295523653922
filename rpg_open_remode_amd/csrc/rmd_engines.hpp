// Copy engines addressed directly (the HSA runtime underneath HIP) for staged host frames -- rmd_ingest.hip, DESIGN.md 4.6.
//
// hipMemcpyAsync puts every host -> device copy of a process on ONE engine, and each command costs that engine ~10 us beyond the
// transfer (fetch, first read over the link, fence, completion).  A staged frame is two commands -- the frame, then its arrival flag --
// so a 640x480 frame occupied the engine for 25 us (a light update: 20 us), a 1920x1080 one for 57-93 us (update: 60 us):
// tools/link_probe.cpp, profiles/r06_link_probe.txt.  Here a frame goes to a DATA engine (consecutive frames to alternating ones: one
// engine's fixed cost overlaps the other's transfer) and its flag to a FLAG engine that waits for the frame's completion signal in
// hardware (a dependent copy: the engine polls the signal, the host is not involved); the data engine goes on with the next frame.
// What the kernels see is unchanged: the frame's number appears in the slot's flag word in HBM after the frame has landed.
//
// Nothing here is ordered against any HIP stream: the kernels wait for the flag words, the host learns from the progress word which
// slots are free (rmdh::wait_for_progress), and a handle waits for its signals before it frees what the engines write.
#ifndef RMD_ENGINES_HPP
#define RMD_ENGINES_HPP

#include <cstddef>
#include <cstdint>

namespace rmdh {

class CopyEngines {
 public:
  // The engines of a HIP device with (at least) the first n_engines of them (1..4) ready, or nullptr when they cannot be used (agents not
  // found, an engine refused its first copy): the caller stays on its copy stream.  The object lives as long as the process.
  static CopyEngines* for_device(int hip_device, int n_engines);

  // a completion signal (0 = idle); 0 on failure
  uint64_t create_signal();
  void destroy_signal(uint64_t sig);
  bool idle(uint64_t sig) const;
  // until the engine has completed what the signal stands for; false after `timeout_us`
  bool wait_idle(uint64_t sig, double timeout_us) const;

  // dst <- src (bytes) on engine `data_engine` (0..3, modulo the engines that are ready); then, once that has landed, flag_dst <- flag_src
  // (flag_bytes) on engine `flag_engine` -- the same engine: the next command in its queue; another one: that engine waits for the frame's
  // completion signal in hardware, i.e. it is busy polling until then (fastest for one process, -26 % when eight share a device).  flag_dst
  // == nullptr: no flag (fault injection: the frame arrives, its flag never does). Both signals must be idle.  false: the runtime refused
  // (last_error() says why) -- the frame copy may be in flight all the same: frame_sig tells.
  bool submit(unsigned data_engine, unsigned flag_engine, void* dst, const void* src, size_t bytes, uint64_t frame_sig, void* flag_dst,
              const void* flag_src, size_t flag_bytes, uint64_t flag_sig);
  int engines_ready() const { return n_ready_; }

  const char* last_error() const;  // of the calling thread's last refused submit()

 private:
  CopyEngines() = default;
  bool find_agents(int hip_device);
  bool first_copies(int n);
  uint64_t gpu_ = 0, cpu_ = 0;          // hsa_agent_t handles
  unsigned engine_[4] = {0, 0, 0, 0};  // hsa_amd_sdma_engine_id_t bits
  int n_ready_ = 0;                    // engines [0, n_ready_) have carried a first copy
  char err_[160] = {0};  // why the engines could not be had (written under the mutex of for_device)
};

}  // namespace rmdh

#endif  // RMD_ENGINES_HPP
