// librmd_hip.so -- rmd::ImageReducer<T> (reduction.cu), the self tests of the wave primitives / the float-frame examination / the device
// side of the arithmetic contract.
#include "rmd_host.hpp"
#include "rmd_copy_pool.hpp"

using namespace rmdh;

namespace rmdk {  // (kernels of this unit only: each kernel of the library has ONE home translation unit)

// Reductions (reduction_kernels.cu:57-159): wave-level shuffles, one atomic / one partial per block.
static __global__ __launch_bounds__(256) void count_eq_kernel(const int* __restrict__ img, int w, int h, int stride, int value,
                                                       unsigned long long* __restrict__ out) {
  __shared__ unsigned long long wave_part[4];
  unsigned long long c = 0;
  for (int y = blockIdx.y; y < h; y += gridDim.y) {
    const int* row = img + static_cast<size_t>(y) * stride;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x) c += (row[x] == value);
  }
  c = wave_sum_u64(c);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_part[wave] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = wave_part[0] + wave_part[1] + wave_part[2] + wave_part[3];
    if (t) atomicAdd(out, t);
  }
}

// pass 1: one fp64 partial per block, in a fixed order; pass 2: one block folds the partials.
static __global__ __launch_bounds__(256) void sum_partial_kernel(const float* __restrict__ img, int w, int h, int stride,
                                                          double* __restrict__ partials) {
  __shared__ double wave_part[4];
  double acc = 0.0;
  for (int y = blockIdx.y; y < h; y += gridDim.y) {
    const float* row = img + static_cast<size_t>(y) * stride;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x) acc += static_cast<double>(row[x]);
  }
  acc = wave_sum_f64(acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    partials[blockIdx.y * gridDim.x + blockIdx.x] = ((wave_part[0] + wave_part[1]) + wave_part[2]) + wave_part[3];
}
// integer image sum (ImageReducer<int>::sum, reduction.cu:186): exact in 64 bits, the caller truncates to int like the
// reference's int accumulation wraps
static __global__ __launch_bounds__(256) void sum_i32_kernel(const int* __restrict__ img, int w, int h, int stride,
    unsigned long long* __restrict__ out) {
  __shared__ unsigned long long wave_part[4];
  unsigned long long acc = 0;
  for (int y = blockIdx.y; y < h; y += gridDim.y) {
    const int* row = img + static_cast<size_t>(y) * stride;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x) acc += static_cast<unsigned long
        long>(static_cast<long long>(row[x]));
  }
  acc = wave_sum_u64(acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, wave_part[0] + wave_part[1] + wave_part[2] + wave_part[3]);
}

static __global__ __launch_bounds__(64) void sum_final_kernel(const double* __restrict__ partials, int n, float* __restrict__ out) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) acc += partials[i];
  acc = wave_sum_f64(acc);
  if (threadIdx.x == 0) *out = static_cast<float>(acc);
}

// ------------------------------------------------------------------------------------------
// device side of the arithmetic contract, for the self test
// self test of the DPP wave primitives of rmd_device.hpp against the shuffle forms: mismatching lanes -> *bad
static __global__ __launch_bounds__(64) void wave_primitives_selftest_kernel(unsigned int seed, unsigned int* bad) {
  const int lane = threadIdx.x;
  unsigned int h = seed * 2654435761u + static_cast<unsigned int>(lane) * 40503u + blockIdx.x * 97u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  const int v = static_cast<int>(h % 2001u) - 1000;
  int add = v, mn = v, mx = v;
  for (int off = 32; off > 0; off >>= 1) {
    add += __shfl_xor(add, off, 64);
    mn = min(mn, __shfl_xor(mn, off, 64));
    mx = max(mx, __shfl_xor(mx, off, 64));
  }
  int incl = v;
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  const bool ok = wave_reduce_i32<WaveAdd>(v) == add && wave_reduce_i32<WaveMin>(v) == mn && wave_reduce_i32<WaveMax>(v) == mx &&
                  wave_scan_i32<WaveAdd>(v) == incl;
  if (!ok) atomicAdd(bad, 1u);
}

static __global__ void math_eval_kernel(int op, const float* x, const float* y, const float* z, float* out, size_t n) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r;
  switch (op) {
    case 0: r = rmd_expf(x[i]); break;
    case 1: r = rmd_sinf(x[i]); break;
    case 2: r = rmd_acosf(x[i]); break;
    case 3: r = rmd_rsqrtf(x[i]); break;
    case 4: r = sqrtf(x[i]); break;
    case 5: r = x[i] / y[i]; break;
    default: r = rmd_lerp(x[i], y[i], z[i]); break;
  }
  out[i] = r;
}

}  // namespace rmdk

namespace rmdh {
// countEqual on `stream` into *out_dev (zeroed by the caller); for rmd_hip_seeds_converged_count (rmd_capi.hip)
void launch_count_eq(const int* img, int w, int h, int stride, int value, unsigned long long* out_dev, hipStream_t stream) {
  const dim3 block(256), grid((w + 255) / 256, h < 64 ? h : 64);
  hipLaunchKernelGGL(rmdk::count_eq_kernel, grid, block, 0, stream, img, w, h, stride, value, out_dev);
}
}  // namespace rmdh

// ---- ImageReducer ---------------------------------------------------------------------------
namespace {

// scratch of the reducers: allocated once per device (the reference keeps dev_partial_ / dev_final_ in the object,
// reduction.cu:29-73), used on the null stream under a lock
struct ReduceScratch {
  double* parts = nullptr;            // 8 x 64 fp64 partials
  float* out_f32 = nullptr;
  unsigned long long* out_u64 = nullptr;
};
constexpr int MAX_DEVICES = 64;
ReduceScratch g_reduce_scratch[MAX_DEVICES];
std::mutex g_reduce_mutex;

int reduce_scratch(ReduceScratch** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= MAX_DEVICES) return fail(RMD_HIP_ERR_RUNTIME, "reduce: device index %d", dev);
  ReduceScratch& r = g_reduce_scratch[dev];
  if (!r.parts) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&r.parts), 8 * 64 * sizeof(double)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&r.out_f32), sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&r.out_u64), sizeof(unsigned long long)));
  }
  *out = &r;
  return RMD_HIP_OK;
}

int reduce_sum_f32_dev(const float* data, size_t stride, size_t width, size_t height, float* sum) {
  if (width == 0 || height == 0 || width > 0x7fffffff || height > 0x7fffffff || stride > 0x7fffffff) return fail(RMD_HIP_ERR_INVALID_ARG,
      "reduce_sum: bad shape");
  std::lock_guard<std::mutex> lock(g_reduce_mutex);
  ReduceScratch* r = nullptr;
  TRY(reduce_scratch(&r));
  const int w = static_cast<int>(width), h = static_cast<int>(height);
  const dim3 block(256), grid((w + 255) / 256 < 8 ? (w + 255) / 256 : 8, h < 64 ? h : 64);
  hipLaunchKernelGGL(rmdk::sum_partial_kernel, grid, block, 0, nullptr, data, w, h, static_cast<int>(stride), r->parts);
  hipLaunchKernelGGL(rmdk::sum_final_kernel, dim3(1), dim3(64), 0, nullptr, r->parts, static_cast<int>(grid.x * grid.y), r->out_f32);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(sum, r->out_f32, sizeof(float), hipMemcpyDeviceToHost));
  return RMD_HIP_OK;
}

int reduce_u64_dev(bool count_eq, const int* data, size_t stride, size_t width, size_t height, int value, unsigned long long* result) {
  if (width == 0 || height == 0 || width > 0x7fffffff || height > 0x7fffffff || stride > 0x7fffffff) return fail(RMD_HIP_ERR_INVALID_ARG,
      "reduce: bad shape");
  std::lock_guard<std::mutex> lock(g_reduce_mutex);
  ReduceScratch* r = nullptr;
  TRY(reduce_scratch(&r));
  const int w = static_cast<int>(width), h = static_cast<int>(height);
  HIP_TRY(hipMemsetAsync(r->out_u64, 0, sizeof(unsigned long long), nullptr));
  const dim3 block(256), grid((w + 255) / 256, h < 64 ? h : 64);
  if (count_eq) hipLaunchKernelGGL(rmdk::count_eq_kernel, grid, block, 0, nullptr, data, w, h, static_cast<int>(stride), value, r->out_u64);
  else hipLaunchKernelGGL(rmdk::sum_i32_kernel, grid, block, 0, nullptr, data, w, h, static_cast<int>(stride), r->out_u64);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(result, r->out_u64, sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return RMD_HIP_OK;
}

}  // namespace

extern "C" {

int rmd_hip_reduce_sum_f32(const rmd_hip_image_t* img, float* sum) {
  if (!img || !sum) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_sum: null argument");
  if (img->kind != RMD_HIP_KIND_F32) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_sum: image is not f32");
  ScopedDevice dev(img->device);
  TRY(image_settle(img));
  return reduce_sum_f32_dev(static_cast<const float*>(img->data), img->stride, img->width, img->height, sum);
}

int rmd_hip_reduce_sum_i32(const rmd_hip_image_t* img, int* sum) {
  if (!img || !sum) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_sum_i32: null argument");
  if (img->kind != RMD_HIP_KIND_I32) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_sum_i32: image is not i32");
  ScopedDevice dev(img->device);
  TRY(image_settle(img));
  unsigned long long r = 0;
  TRY(reduce_u64_dev(false, static_cast<const int*>(img->data), img->stride, img->width, img->height, 0, &r));
  *sum = static_cast<int>(static_cast<unsigned int>(r));
  return RMD_HIP_OK;
}

int rmd_hip_reduce_count_eq_i32(const rmd_hip_image_t* img, int value, size_t* count) {
  if (!img || !count) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_count_eq: null argument");
  if (img->kind != RMD_HIP_KIND_I32) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_count_eq: image is not i32");
  ScopedDevice dev(img->device);
  TRY(image_settle(img));
  unsigned long long r = 0;
  TRY(reduce_u64_dev(true, static_cast<const int*>(img->data), img->stride, img->width, img->height, value, &r));
  *count = static_cast<size_t>(r);
  return RMD_HIP_OK;
}

// the raw-pointer forms of the reference (reduction.cuh:33-47): device pointers on the current device; the caller is
// responsible for the data being complete (as with the reference, which launches on the default stream)
int rmd_hip_reduce_sum_f32_raw(const float* dev_data, size_t stride_elems, size_t width, size_t height, float* sum) {
  if (!dev_data || !sum) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_sum_raw: null argument");
  HIP_TRY(hipDeviceSynchronize());
  return reduce_sum_f32_dev(dev_data, stride_elems, width, height, sum);
}
int rmd_hip_reduce_sum_i32_raw(const int* dev_data, size_t stride_elems, size_t width, size_t height, int* sum) {
  if (!dev_data || !sum) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_sum_i32_raw: null argument");
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long r = 0;
  TRY(reduce_u64_dev(false, dev_data, stride_elems, width, height, 0, &r));
  *sum = static_cast<int>(static_cast<unsigned int>(r));
  return RMD_HIP_OK;
}
int rmd_hip_reduce_count_eq_i32_raw(const int* dev_data, size_t stride_elems, size_t width, size_t height, int value, size_t* count) {
  if (!dev_data || !count) return fail(RMD_HIP_ERR_INVALID_ARG, "reduce_count_eq_raw: null argument");
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long r = 0;
  TRY(reduce_u64_dev(true, dev_data, stride_elems, width, height, value, &r));
  *count = static_cast<size_t>(r);
  return RMD_HIP_OK;
}

// ---- self test of the float-frame examination (host code, the same entry the update path uses) ----
int rmd_hip_selftest_pack_float_frame(const float* host_img, int width, int height, int pitch, unsigned char* bytes, int* all_levels) {
  if (!host_img || !bytes || !all_levels || width <= 0 || height <= 0 || pitch < width)
    return fail(RMD_HIP_ERR_INVALID_ARG, "selftest_pack_float_frame: bad argument");
  *all_levels = CopyPool::instance().pack(host_img, bytes, width, height, pitch) ? 1 : 0;
  return RMD_HIP_OK;
}

// ---- self test of the wave primitives the kernels rely on -----------------------------------
int rmd_hip_selftest_wave_primitives(int* mismatching_lanes) {
  if (!mismatching_lanes) return fail(RMD_HIP_ERR_INVALID_ARG, "selftest: null output");
  unsigned int* d = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), sizeof(unsigned int)));
  hipError_t e = hipMemset(d, 0, sizeof(unsigned int));
  if (e == hipSuccess) {
    for (unsigned int seed = 1; seed <= 8; ++seed) hipLaunchKernelGGL(rmdk::wave_primitives_selftest_kernel, dim3(64), dim3(64), 0, nullptr,
        seed, d);
    e = hipGetLastError();
  }
  unsigned int bad = 0;
  if (e == hipSuccess) e = hipMemcpy(&bad, d, sizeof(bad), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(RMD_HIP_ERR_RUNTIME, "selftest: %s", hipGetErrorString(e));
  *mismatching_lanes = static_cast<int>(bad);
  return RMD_HIP_OK;
}

// ---- arithmetic-contract self test ----------------------------------------------------------
int rmd_hip_math_eval(int op, const float* x, const float* y, const float* z, float* out, size_t n) {
  if (!x || !out || n == 0 || op < 0 || op > 6) return fail(RMD_HIP_ERR_INVALID_ARG, "math_eval: bad argument");
  if ((op == 5 && !y) || (op == 6 && (!y || !z))) return fail(RMD_HIP_ERR_INVALID_ARG, "math_eval: missing operand");
  float *dx = nullptr, *dy = nullptr, *dz = nullptr, *dout = nullptr;
  const size_t bytes = n * sizeof(float);
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dx), bytes));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dy), bytes));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dz), bytes));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dout), bytes));
  const hipError_t e_in = [&] {
    hipError_t r = hipMemcpy(dx, x, bytes, hipMemcpyHostToDevice);
    if (r == hipSuccess) r = hipMemcpy(dy, y ? y : x, bytes, hipMemcpyHostToDevice);
    if (r == hipSuccess) r = hipMemcpy(dz, z ? z : x, bytes, hipMemcpyHostToDevice);
    return r;
  }();
  if (e_in != hipSuccess) {
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dz); (void)hipFree(dout);
    return fail(RMD_HIP_ERR_RUNTIME, "math_eval: upload failed: %s", hipGetErrorString(e_in));
  }
  hipLaunchKernelGGL(rmdk::math_eval_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, op, dx, dy,
                     dz, dout, n);
  const hipError_t e = hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
  (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dz); (void)hipFree(dout);
  if (e != hipSuccess) return fail(RMD_HIP_ERR_RUNTIME, "math_eval: %s", hipGetErrorString(e));
  return RMD_HIP_OK;
}

}  // extern "C"

