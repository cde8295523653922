// VALU issue-rate micro-benchmark for gfx950: cycles per wave64 instruction of the fp32 operations the NCC search is made of,
// as a function of waves per SIMD.  Settles the peak the VALU roofline of bench.py is priced against.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  float b = 1.0001f, c = 0.5f;
  typedef float float2_ __attribute__((ext_vector_type(2)));
  float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
  float2_ pb = {b, b}, pc = {c, c};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (OP == 1) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
    if (OP == 2) { REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 3) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));) }
    if (OP == 4) { REP16(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));) }
    if (OP == 5) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));) }
    if (OP == 6) { REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 7) {  // one dependent chain
      REP16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(c));) }
    if (OP == 8) {  // fma dependent chain
      REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));) }
    if (OP == 9) {  // three interleaved dependent chains (the NCC accumulation pattern)
      REP16(asm volatile("v_add_f32 %0, %0, %3\n v_add_f32 %1, %1, %3\n v_add_f32 %2, %2, %3\n v_add_f32 %0, %0, %3\n v_add_f32 %1, %1, %3\n v_add_f32 %2, %2, %3\n v_add_f32 %0, %0, %3\n v_add_f32 %1, %1, %3" : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(c));) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, float* d_out, unsigned long long* d_cyc, int n_cu) {
  const int iters = 2000;
  for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {  // 256-thread WGs: 1 wave per SIMD each
    const int grid = n_cu * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d_out, 10, d_cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d_out, iters, d_cyc);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_wave = (double)iters * 16 * 8;
    // per SIMD: wg_per_cu waves, each insts_per_wave instructions
    const double ns_per_inst_per_simd = ms * 1e6 / (insts_per_wave * wg_per_cu);
    unsigned long long c0 = 0;
    hipMemcpy(&c0, d_cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s waves/SIMD %d: %.3f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz), chip %.1f G wave-inst/s; wave0 clock ticks/inst %.2f\n",
           name, wg_per_cu, ns_per_inst_per_simd, ns_per_inst_per_simd * 2.4, n_cu * 4 / ns_per_inst_per_simd, (double)c0 / insts_per_wave);
  }
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  printf("%s, %d CUs, clock %d kHz\n", prop.gcnArchName, n_cu, prop.clockRate);
  float* d_out; unsigned long long* d_cyc;
  hipMalloc(&d_out, sizeof(float) * 256 * n_cu * 8);
  hipMalloc(&d_cyc, 8 * n_cu * 8);
  run<0>("v_fma_f32 x8 independent", d_out, d_cyc, n_cu);
  run<1>("v_add_f32 x8 independent", d_out, d_cyc, n_cu);
  run<2>("v_mul_f32 x8 independent", d_out, d_cyc, n_cu);
  run<3>("v_pk_fma_f32 x8 independent", d_out, d_cyc, n_cu);
  run<4>("v_pk_add_f32 x8 independent", d_out, d_cyc, n_cu);
  run<5>("v_pk_mul_f32 x8 independent", d_out, d_cyc, n_cu);
  run<6>("v_mov_b32 ring", d_out, d_cyc, n_cu);
  run<7>("v_add_f32 dependent chain", d_out, d_cyc, n_cu);
  run<8>("v_fma_f32 dependent chain", d_out, d_cyc, n_cu);
  run<9>("v_add_f32 3 chains", d_out, d_cyc, n_cu);
  return 0;
}
