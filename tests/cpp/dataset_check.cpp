// Test-owned driver of the reference's rmd::test::Dataset (test/dataset.cpp, compiled UNMODIFIED against include/rmd/ and the
// test-only third-party stand-ins): reads a dataset directory and prints, per entry, what the reader returned -- file names, the pose
// as the bits of SE3<float>::data, image size and byte sum, and for entries that have a depth file its element count, double sum and
// the bits of the first and last value.  tests/test_reference_host_sources.py compares this with rpg_open_remode_amd/dataset.py.
//   dataset_check <dataset dir> <sequence file> <width> <height>
#include <cstdio>
#include <cstring>
#include <iostream>

#include "dataset.h"

static unsigned bits(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  return u;
}

int main(int argc, char** argv) {
  if (argc != 5) return 2;
  rmd::test::Dataset dataset(argv[1], argv[2]);
  if (!dataset.readDataSequence()) return 3;
  const size_t width = static_cast<size_t>(atoi(argv[3])), height = static_cast<size_t>(atoi(argv[4]));
  for (const auto& entry : dataset) {
    rmd::SE3<float> T_world_curr;
    dataset.readCameraPose(T_world_curr, entry);
    printf("%s %s", entry.getImageFileName().c_str(), entry.getDepthmapFileName().c_str());
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) printf(" %08x", bits(T_world_curr.data(r, c)));
    cv::Mat img;
    if (dataset.readImage(img, entry)) {
      unsigned long long sum = 0;
      for (size_t i = 0; i < static_cast<size_t>(img.rows) * img.cols; ++i) sum += img.data[i];
      printf(" img %d %d %llu", img.cols, img.rows, sum);
    } else {
      printf(" img - - -");
    }
    cv::Mat depth;
    if (dataset.readDepthmap(depth, entry, width, height)) {
      double sum = 0.0;
      const float* d = reinterpret_cast<const float*>(depth.data);
      for (size_t i = 0; i < width * height; ++i) sum += d[i];
      printf(" depth %zu %.17g %08x %08x", width * height, sum, bits(d[0]), bits(d[width * height - 1]));
    } else {
      printf(" depth -");
    }
    printf("\n");
  }
  return 0;
}
