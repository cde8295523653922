// rmd::checkCudaDevice (reference: include/rmd/check_cuda_device.cuh:24, src/check_cuda_device.cu:23-117):
// list the devices, honour --device=N, make it current.  Name kept for source compatibility.
#ifndef RMD_CHECK_CUDA_DEVICE_CUH
#define RMD_CHECK_CUDA_DEVICE_CUH

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rmd_hip.h>

namespace rmd {

inline bool checkCudaDevice(int argc, char** argv) {
  printf("Running executable: %s\nChecking available HIP devices...\n", argc > 0 ? argv[0] : "?");
  int n = 0;
  if (rmd_hip_device_count(&n) != RMD_HIP_OK) {
    printf("ERROR: %s\n", rmd_hip_last_error());
    return false;
  }
  printf("%d GPU detected:\n", n);
  char name[256];
  for (int d = 0; d < n; ++d)
    if (rmd_hip_device_name(d, name, sizeof(name)) == RMD_HIP_OK) printf("Device %d - %s\n", d, name);
  int dev = 0;
  const char* key = "--device=";
  for (int i = 1; i < argc; ++i)
    if (strncmp(argv[i], key, strlen(key)) == 0) {
      dev = atoi(argv[i] + strlen(key));
      printf("User-specified device: %d\n", dev);
      break;
    }
  if (rmd_hip_set_device(dev) != RMD_HIP_OK) {
    printf("ERROR: %s\n", rmd_hip_last_error());
    return false;
  }
  return true;
}

}  // namespace rmd

#endif  // RMD_CHECK_CUDA_DEVICE_CUH
