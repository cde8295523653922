// Test-owned driver of the reference's rmd::Depthmap class.  The class itself (include/rmd/depthmap.h, src/depthmap.cpp) is
// compiled UNMODIFIED from /root/reference against this repository's include/rmd/ headers and the test-only third-party
// stubs (tests/cpp/stubs); this file only feeds it frames and writes what its getters return.
//   depthmap_check in.bin out.bin
// in.bin : int32 w, h, n, distorted, iterations; float32 K[4] (fx, fy, cx, cy), D[4], range[2], lambda;
//          then n x { u8 image w*h, float32 T_curr_world[12] (row-major 3x4) }
// out.bin: float32 depth w*h, int32 convergence w*h, float32 denoised w*h, u8 reference image w*h,
//          uint64 converged count, float32 converged percentage, float32 dist from ref, uint8 scaled+coloured depth w*h*3
#include <cstdio>
#include <cstring>
#include <iostream>
#include <vector>

#include <rmd/depthmap.h>

static bool read_all(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* in = fopen(argv[1], "rb");
  if (!in) return 3;
  int hdr[5];
  float K[4], D[4], range[2], lambda;
  if (!read_all(in, hdr, sizeof(hdr)) || !read_all(in, K, sizeof(K)) || !read_all(in, D, sizeof(D)) || !read_all(in, range, sizeof(range)) ||
      !read_all(in, &lambda, sizeof(lambda)))
    return 4;
  const int w = hdr[0], h = hdr[1], n = hdr[2];
  const size_t px = static_cast<size_t>(w) * h;
  rmd::Depthmap depthmap(w, h, K[0], K[2], K[1], K[3]);
  if (hdr[3]) depthmap.initUndistortionMap(D[0], D[1], D[2], D[3]);
  for (int k = 0; k < n; ++k) {
    cv::Mat img(h, w, CV_8UC1);
    float pose[12];
    if (!read_all(in, img.data, px) || !read_all(in, pose, sizeof(pose))) return 5;
    float r[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]}, t[3] = {pose[3], pose[7], pose[11]};
    const rmd::SE3<float> T_curr_world(r, t);
    if (k == 0) {
      if (!depthmap.setReferenceImage(img, T_curr_world, range[0], range[1])) return 6;
    } else {
      depthmap.update(img, T_curr_world);
    }
  }
  fclose(in);
  FILE* out = fopen(argv[2], "wb");
  if (!out) return 7;
  depthmap.downloadDepthmap();
  const cv::Mat raw = depthmap.getDepthmap().clone();
  fwrite(raw.data, 4, px, out);
  depthmap.downloadConvergenceMap();
  fwrite(depthmap.getConvergenceMap().data, 4, px, out);
  depthmap.downloadDenoisedDepthmap(lambda, hdr[4]);
  fwrite(depthmap.getDepthmap().data, 4, px, out);
  fwrite(depthmap.getReferenceImage().data, 1, px, out);
  const unsigned long long count = depthmap.getConvergedCount();
  const float pct = depthmap.getConvergedPercentage(), dist = depthmap.getDistFromRef();
  fwrite(&count, 8, 1, out);
  fwrite(&pct, 4, 1, out);
  fwrite(&dist, 4, 1, out);
  const cv::Mat coloured = rmd::Depthmap::scaleMat(raw);
  fwrite(coloured.data, 1, px * 3, out);
  fclose(out);
  std::cout << "depthmap_check: " << n << " frames, " << count << " converged (" << pct << " %)" << std::endl;
  return 0;
}
