// test stub (NOT part of the library): test/main_test.cpp includes <cuda_runtime_api.h> for one call, cudaDeviceReset() after
// RUN_ALL_TESTS (main_test.cpp:31).  Every handle of the HIP library has been destroyed by then; nothing to do.
#ifndef RMD_TEST_STUB_CUDA_RUNTIME_API
#define RMD_TEST_STUB_CUDA_RUNTIME_API
inline int cudaDeviceReset() { return 0; }
#endif
