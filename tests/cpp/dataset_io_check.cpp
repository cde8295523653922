// Reads a dataset directory through apps/dataset.h and prints what it found, for comparison with the Python reader.
#include <cstdio>
#include <iostream>

#include "../../apps/dataset.h"

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  rmd::test::Dataset ds(argv[1], argv[2]);
  const size_t w = static_cast<size_t>(atol(argv[3])), h = static_cast<size_t>(atol(argv[4]));
  if (!ds.readDataSequence(argc > 5 ? atol(argv[5]) : 0, argc > 6 ? atol(argv[6]) : 0)) return 3;
  printf("entries %zu\n", ds.size());
  for (std::vector<rmd::test::DatasetEntry>::const_iterator it = ds.begin(); it != ds.end(); ++it) {
    rmd::test::GrayImage img;
    const bool ok = ds.readImage(img, *it);
    unsigned long long sum = 0, wsum = 0;
    for (size_t i = 0; i < img.data.size(); ++i) { sum += img.data[i]; wsum += static_cast<unsigned long long>(img.data[i]) * (i % 251 + 1); }
    rmd::SE3<float> T;
    ds.readCameraPose(T, *it);
    std::vector<float> depth;
    const bool dok = ds.readDepthmap(depth, *it, w, h);
    double dsum = 0.0;
    for (size_t i = 0; i < depth.size(); ++i) dsum += depth[i];
    printf("%s %s img %d %dx%d %llu %llu depth %d %.9g pose", it->image_file_name.c_str(), it->depthmap_file_name.c_str(), ok ? 1 : 0, img.width,
           img.height, sum, wsum, dok ? 1 : 0, dsum);
    for (int k = 0; k < 12; ++k) printf(" %.9g", T.data[k]);
    printf("\n");
  }
  return 0;
}
