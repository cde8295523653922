// Tile-cooperative fused seed update (check + epipolar match + Bayesian fusion).
// PLACEHOLDER: forwards to the per-pixel kernel until the tile kernel lands.
#ifndef RMD_MATCHER_HPP
#define RMD_MATCHER_HPP

#include "rmd_kernels.hpp"

namespace rmdk {

struct MatcherWorkspace {
  int allocate(int, int) { return 0; }
  void release() {}
};

template <int SIDE>
inline void launch_seed_update_tile(const SeedParams& P, MatcherWorkspace&, hipStream_t stream) {
  const dim3 block(64, 4), grid((P.w + 63) / 64, (P.h + 3) / 4);
  hipLaunchKernelGGL((seed_update_pixel_kernel<SIDE>), grid, block, 0, stream, P);
}

}  // namespace rmdk

#endif  // RMD_MATCHER_HPP
