// Sustained rate of bringing one frame (default 640x480 floats = 1.2 MB) from pinned host memory into HBM, per frame, for the ways the ingest
// path could do it: (a) one hipMemcpyAsync on one stream (one SDMA engine), (b) the frame split into P parts on P streams, (c) a kernel that
// reads the pinned buffer directly over PCIe (128 workgroups, dwordx4 loads).  Wall time over N frames, device idle otherwise.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void pull_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * blockDim.x) dst[i] = src[i];
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 10) : 640ull * 480 * 4;
  const int n = 400, slots = 3;
  std::vector<hipStream_t> st(4);
  for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned char *h[slots], *d[slots];
  for (int k = 0; k < slots; ++k) { hipHostMalloc((void**)&h[k], bytes); hipMalloc((void**)&d[k], bytes); }
  for (int parts : {1, 2, 4}) {
    for (int rep = 0; rep < 2; ++rep) {
      const double t0 = now_us();
      for (int i = 0; i < n; ++i) {
        const int k = i % slots;
        const size_t part = (bytes / parts + 255) & ~size_t(255);
        for (int p = 0; p < parts; ++p) {
          const size_t off = p * part, len = off + part <= bytes ? part : bytes - off;
          hipMemcpyAsync(d[k] + off, h[k] + off, len, hipMemcpyHostToDevice, st[p]);
        }
      }
      for (auto& s : st) hipStreamSynchronize(s);
      const double us = (now_us() - t0) / n;
      if (rep) printf("%zu bytes, %d part(s) on %d stream(s): %.1f us per frame, %.1f GB/s\n", bytes, parts, parts, us, bytes / us * 1e-3);
    }
  }
  for (int wgs : {32, 128, 512}) {
    for (int rep = 0; rep < 2; ++rep) {
      const double t0 = now_us();
      for (int i = 0; i < n; ++i) {
        const int k = i % slots;
        hipLaunchKernelGGL(pull_kernel, dim3(wgs), dim3(256), 0, st[0], (const uint4*)h[k], (uint4*)d[k], bytes / 16);
      }
      hipStreamSynchronize(st[0]);
      const double us = (now_us() - t0) / n;
      if (rep) printf("%zu bytes, kernel reading pinned memory, %d workgroups: %.1f us per frame, %.1f GB/s\n", bytes, wgs, us, bytes / us * 1e-3);
    }
  }
  return 0;
}
