// TEST INFRASTRUCTURE ONLY -- kernel "launches" for the CPU emulation shim.
//
// The build recipe (oracle/Makefile) rewrites the reference's
//     kernel<<<grid, block[, smem]>>>(args...)
// token sequence to
//     RMD_SHIM_LAUNCH(kernel, grid, block[, smem])(args...)
// on the fly (sed through a pipe; nothing is written to disk) because the
// triple-chevron is not C++.  This header gives that macro a meaning.
//
// Two execution modes:
//   * plain  : kernels without barriers.  Blocks are spread over host threads
//              (OpenMP), the threads of a block run one after the other.
//   * fibres : kernels that call __syncthreads().  Blocks run serially in
//              row-major order (a legal CUDA schedule); the threads of a block
//              are ucontext fibres that are resumed round-robin, each running
//              until its next barrier, exactly like a hardware barrier: a
//              barrier releases when every thread of the block that has not
//              yet exited is waiting at *a* barrier.
#ifndef RMD_ORACLE_CUDA_SHIM_LAUNCH_H
#define RMD_ORACLE_CUDA_SHIM_LAUNCH_H

#include <cuda_runtime.h>
#include <stdio.h>
#include <string.h>
#include <ucontext.h>
#include <functional>
#include <vector>

thread_local uint3 threadIdx;
thread_local uint3 blockIdx;
thread_local dim3 blockDim;
thread_local dim3 gridDim;

namespace rmd_shim {

struct FibreBlock {
  std::vector<ucontext_t> ctx;
  std::vector<char> stacks;
  std::vector<char> done;
  ucontext_t scheduler;
  int current;
  std::function<void()> body;
  static const size_t kStack = 64 * 1024;
};
static FibreBlock* g_fibres = nullptr;  // non-null while a block runs in fibre mode

static void fibre_entry() {
  FibreBlock* fb = g_fibres;
  const int me = fb->current;
  fb->body();
  fb->done[me] = 1;
  swapcontext(&fb->ctx[me], &fb->scheduler);
}

inline void run_block_fibres(const dim3& block, const std::function<void()>& body) {
  FibreBlock fb;
  const int n = static_cast<int>(block.x * block.y * block.z);
  fb.ctx.resize(n);
  fb.done.assign(n, 0);
  fb.stacks.resize(static_cast<size_t>(n) * FibreBlock::kStack);
  fb.body = body;
  g_fibres = &fb;
  for (int t = 0; t < n; ++t) {
    getcontext(&fb.ctx[t]);
    fb.ctx[t].uc_stack.ss_sp = &fb.stacks[static_cast<size_t>(t) * FibreBlock::kStack];
    fb.ctx[t].uc_stack.ss_size = FibreBlock::kStack;
    fb.ctx[t].uc_link = &fb.scheduler;
    makecontext(&fb.ctx[t], fibre_entry, 0);
  }
  int live = n;
  while (live > 0) {  // one pass == "run every live thread up to its next barrier"
    for (int t = 0; t < n; ++t) {
      if (fb.done[t]) continue;
      fb.current = t;
      threadIdx.x = t % block.x;
      threadIdx.y = (t / block.x) % block.y;
      threadIdx.z = t / (block.x * block.y);
      swapcontext(&fb.scheduler, &fb.ctx[t]);
      if (fb.done[t]) --live;
    }
  }
  g_fibres = nullptr;
}

inline bool kernel_has_barriers(const char* name) {
  return strstr(name, "reduction") != nullptr || strstr(name, "TVL1") != nullptr;
}

template <typename Kernel>
struct Launcher {
  Kernel kernel;
  const char* name;
  dim3 grid, block;
  Launcher(Kernel k, const char* n, dim3 g, dim3 b, size_t /*smem*/ = 0) : kernel(k), name(n), grid(g), block(b) {}

  template <typename... Args>
  void operator()(Args... args) const {
    if (kernel_has_barriers(name)) {
      for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) {
          blockIdx.x = bx; blockIdx.y = by; blockIdx.z = 0;
          blockDim = block; gridDim = grid;
          Kernel k = kernel;
          run_block_fibres(block, [=]() { k(args...); });
        }
      return;
    }
    const long nblocks = static_cast<long>(grid.x) * grid.y;
#pragma omp parallel for schedule(dynamic, 4)
    for (long b = 0; b < nblocks; ++b) {
      blockIdx.x = static_cast<unsigned>(b % grid.x);
      blockIdx.y = static_cast<unsigned>(b / grid.x);
      blockIdx.z = 0;
      blockDim = block; gridDim = grid;
      for (unsigned ty = 0; ty < block.y; ++ty)
        for (unsigned tx = 0; tx < block.x; ++tx) {
          threadIdx.x = tx; threadIdx.y = ty; threadIdx.z = 0;
          kernel(args...);
        }
    }
  }
};

template <typename Kernel>
inline Launcher<Kernel> make_launcher(Kernel k, const char* name, dim3 g, dim3 b, size_t smem = 0) {
  return Launcher<Kernel>(k, name, g, b, smem);
}

}  // namespace rmd_shim

void __syncthreads() {
  rmd_shim::FibreBlock* fb = rmd_shim::g_fibres;
  if (!fb) {
    fprintf(stderr, "cuda_shim: __syncthreads() reached in a kernel launched in plain mode\n");
    abort();
  }
  const int me = fb->current;
  swapcontext(&fb->ctx[me], &fb->scheduler);
  // resumed: restore this fibre's identity (the scheduler set it before switching)
}

#define RMD_SHIM_LAUNCH(kernel, ...) rmd_shim::make_launcher(&kernel, #kernel, __VA_ARGS__)

#endif  // RMD_ORACLE_CUDA_SHIM_LAUNCH_H
