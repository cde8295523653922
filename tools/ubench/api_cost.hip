// Host-side cost of the HIP runtime calls a frame upload is made of (microseconds per call, submission only).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t a, b;
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  hipEvent_t e[2]; hipEventCreateWithFlags(&e[0], hipEventDisableTiming); hipEventCreateWithFlags(&e[1], hipEventDisableTiming);
  const size_t n = 640 * 480;
  unsigned char *h, *d, *pageable = (unsigned char*)malloc(n);
  hipHostMalloc((void**)&h, n); hipMalloc((void**)&d, n); memset(pageable, 1, n);
  const int N = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now(); for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1200), dim3(256), 0, a, nullptr); double t1 = now(); hipStreamSynchronize(a);
    if (rep) printf("kernel launch            %6.2f us/call\n", (t1 - t0) / N);
    t0 = now(); for (int i = 0; i < N; ++i) hipEventRecord(e[0], a); t1 = now(); hipStreamSynchronize(a);
    if (rep) printf("hipEventRecord           %6.2f us/call\n", (t1 - t0) / N);
    t0 = now(); for (int i = 0; i < N; ++i) hipStreamWaitEvent(b, e[0], 0); t1 = now(); hipStreamSynchronize(b);
    if (rep) printf("hipStreamWaitEvent       %6.2f us/call\n", (t1 - t0) / N);
    t0 = now(); for (int i = 0; i < N; ++i) hipEventSynchronize(e[0]); t1 = now();
    if (rep) printf("hipEventSynchronize(done)%6.2f us/call\n", (t1 - t0) / N);
    t0 = now(); for (int i = 0; i < N; ++i) hipEventQuery(e[0]); t1 = now();
    if (rep) printf("hipEventQuery(done)      %6.2f us/call\n", (t1 - t0) / N);
    t0 = now(); for (int i = 0; i < N; ++i) hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, b); t1 = now(); hipStreamSynchronize(b); double t2 = now();
    if (rep) printf("hipMemcpyAsync 300 KB pinned %6.2f us/call submit, %6.2f us/call incl. completion\n", (t1 - t0) / N, (t2 - t0) / N);
    t0 = now(); for (int i = 0; i < N; ++i) memcpy(h, pageable, n); t1 = now();
    if (rep) printf("memcpy 300 KB pageable -> pinned %6.2f us/call\n", (t1 - t0) / N);
    // the frame pattern: wait, copy, kernel, record | wait, 3 kernels, record
    t0 = now();
    for (int i = 0; i < N; ++i) {
      hipStreamWaitEvent(b, e[1], 0); hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, b); hipLaunchKernelGGL(empty_kernel, dim3(300), dim3(256), 0, b, nullptr); hipEventRecord(e[0], b);
      hipStreamWaitEvent(a, e[0], 0); for (int q = 0; q < 3; ++q) hipLaunchKernelGGL(empty_kernel, dim3(1200), dim3(256), 0, a, nullptr); hipEventRecord(e[1], a);
    }
    t1 = now(); hipStreamSynchronize(a); hipStreamSynchronize(b); t2 = now();
    if (rep) printf("two-stream frame pattern %6.2f us/frame submit, %6.2f us/frame incl. completion\n", (t1 - t0) / N, (t2 - t0) / N);
    t0 = now();
    for (int i = 0; i < N; ++i) {
      hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, a); for (int q = 0; q < 4; ++q) hipLaunchKernelGGL(empty_kernel, dim3(1200), dim3(256), 0, a, nullptr);
    }
    t1 = now(); hipStreamSynchronize(a); t2 = now();
    if (rep) printf("one-stream frame pattern %6.2f us/frame submit, %6.2f us/frame incl. completion\n", (t1 - t0) / N, (t2 - t0) / N);
  }
  return 0;
}
