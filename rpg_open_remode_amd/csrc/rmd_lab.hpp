// Measurement hooks of the seed kernels.  EMPTY in the product build: LAB_PROF(...) expands to nothing, so rmd_frame.hpp reads as the
// kernel it is.  A lab build (tools/ab_make.sh prof "-DRMD_LAB_PROFILE_ROUNDS", selected on the GPU box with RMD_HIP_LIB) keeps per-phase
// ticks of every search workgroup -- window policy, staging, rounds with / without an evaluation served from L2 -- in eight LDS words and
// writes them into the timeline slot of the workgroup (tools/profile_rounds.py reads them).  Experiments that were measured and dropped
// (retired matchers, one launch per update, FMA contraction, other tile geometries, ...) are NOT kept behind switches any more: LAB.md
// lists them with the commit that last carried their code.
#ifndef RMD_LAB_HPP
#define RMD_LAB_HPP

// A/B of the round-5 sheared window (tools/ab_make.sh noshear "-DRMD_LAB_NO_SHEAR"): no tile ever asks for a band, i.e. all windows are
// boxes
#ifdef RMD_LAB_NO_SHEAR
#define LAB_WANT_BAND(flag) false
#else
#define LAB_WANT_BAND(flag) (flag)
#endif

// A/B of the LDS window's row stride (tools/ab_make.sh stride3 "-DRMD_LAB_STRIDE=3": ww | 3; stride5 "-DRMD_LAB_STRIDE=5": (ww + 2) | 1)
#if defined(RMD_LAB_STRIDE) && RMD_LAB_STRIDE == 3
#define LAB_WINDOW_STRIDE(ww) ((ww) | 3)
#elif defined(RMD_LAB_STRIDE) && RMD_LAB_STRIDE == 5
#define LAB_WINDOW_STRIDE(ww) (((ww) + 2) | 1)
#else
#define LAB_WINDOW_STRIDE(ww) ((ww) | 1)
#endif

#ifdef RMD_LAB_PROFILE_ROUNDS
#define LAB_PROF(...) __VA_ARGS__
namespace rmdk {
__device__ __forceinline__ unsigned long long prof_clock() {  // the 100 MHz wall clock, pinned in program order
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return t;
}
}  // namespace rmdk
#else
#define LAB_PROF(...)
#endif

#endif  // RMD_LAB_HPP
