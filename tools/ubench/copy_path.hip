// Which engine carries a pinned-host -> device hipMemcpyAsync of a given size: run under `rocprofv3 --kernel-trace --memory-copy-trace`
// and count __amd_rocclr_copyBuffer dispatches (shader blit) against MEMORY_COPY_HOST_TO_DEVICE records (SDMA) per size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4;
  const int n = 200;
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned char *h, *d;
  hipHostMalloc((void**)&h, bytes < 64 ? 64 : bytes); hipMalloc((void**)&d, bytes < 64 ? 64 : bytes);
  for (int i = 0; i < n; ++i) { hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); if ((i & 15) == 15) hipStreamSynchronize(s); }
  hipStreamSynchronize(s);
  printf("%zu bytes x %d copies done\n", bytes, n);
  return 0;
}
