// Test infrastructure: rmd_expf / rmd_sinf / rmd_acosf (csrc/rmd_math.h, the restated glibc 2.35 routines the kernels compile) against
// the expf / sinf / acosf of the libm this program is linked with, for ALL 2^32 float arguments or a strided sample of them.
//   usage: libm_exhaustive [stride=1] [threads=all]      (stride 1 = every argument: ~1 minute on 8 cores)
// Prints, per function, the number of arguments whose results differ (NaN == NaN whatever the payload) and the first few of them.
// Exit code 0 iff there is none.  Built by oracle/Makefile (g++ -O2 -ffp-contract=off -fopenmp, linked with -lm).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rmd_math.h"

static inline bool same(float a, float b) {
  if (a != a && b != b) return true;
  uint32_t x, y;
  memcpy(&x, &a, 4); memcpy(&y, &b, 4);
  return x == y;
}

int main(int argc, char** argv) {
  const unsigned long stride = argc > 1 ? strtoul(argv[1], nullptr, 10) : 1;
  const char* names[3] = {"expf", "sinf", "acosf"};
  unsigned long long total_bad = 0;
  for (int fn = 0; fn < 3; ++fn) {
    unsigned long long bad = 0, n = 0;
    uint32_t first[8];
    int n_first = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad, n)
    for (long long blk = 0; blk < 4096; ++blk) {
      for (unsigned long long k = static_cast<unsigned long long>(blk) << 20; k < (static_cast<unsigned long long>(blk + 1) << 20); k += stride) {
        const uint32_t bits = static_cast<uint32_t>(k);
        float x;
        memcpy(&x, &bits, 4);
        float a, b;
        if (fn == 0) { a = rmd_expf(x); b = expf(x); }
        else if (fn == 1) { a = rmd_sinf(x); b = sinf(x); }
        else { a = rmd_acosf(x); b = acosf(x); }
        ++n;
        if (!same(a, b)) {
          ++bad;
#pragma omp critical
          if (n_first < 8) first[n_first++] = bits;
        }
      }
    }
    printf("%-6s %llu arguments, %llu differ from the host libm", names[fn], n, bad);
    for (int i = 0; i < n_first; ++i) {
      float x; memcpy(&x, &first[i], 4);
      const float a = fn == 0 ? rmd_expf(x) : fn == 1 ? rmd_sinf(x) : rmd_acosf(x), b = fn == 0 ? expf(x) : fn == 1 ? sinf(x) : acosf(x);
      uint32_t ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
      printf("%s x=0x%08x (%a): rmd 0x%08x libm 0x%08x", i ? ";" : "  e.g.", first[i], x, ua, ub);
    }
    printf("\n");
    total_bad += bad;
  }
  return total_bad ? 1 : 0;
}
