// Exercises the C++ facade headers (include/rmd/*.cuh) the way the reference's own host code does
// (test/seed_matrix_test.cpp:70-110, src/depthmap.cpp:63-123, test/reduction_test.cpp:48-60):
// reads frames + poses from a file, runs SeedMatrix / DepthmapDenoiser / ImageReducer, writes the results.
//   facade_check --probe            -> only touches the device-independent part (used on CPU-only machines)
//   facade_check <in.bin> <out.bin>
#include <cstdio>
#include <cstring>
#include <vector>

#include <rmd/check_cuda_device.cuh>
#include <rmd/depthmap_denoiser.cuh>
#include <rmd/reduction.cuh>
#include <rmd/seed_matrix.cuh>

static bool read_all(FILE* f, void* dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes; }

int main(int argc, char** argv) {
  if (argc >= 2 && strcmp(argv[1], "--probe") == 0) {
    rmd::SE3<float> T(1.0f, 0.0f, 0.0f, 0.0f, 0.1f, 0.2f, 0.3f);
    const rmd::SE3<float> I = T * T.inv();
    rmd::PinholeCamera cam(481.2f, -480.0f, 319.5f, 239.5f);
    printf("version %d patch_side %d I(0,3)=%g one_pix=%g\n", rmd_hip_version(), RMD_CORR_PATCH_SIDE, I(0, 3), cam.getOnePixAngle());
    try {
      rmd::DeviceImage<float> img(0, 0);  // must throw, not crash
      return 2;
    } catch (const rmd::CudaException& e) {
      printf("caught: %s", e.what());
    }
    return 0;
  }
  if (argc < 3) return 64;
  if (!rmd::checkCudaDevice(argc, argv)) return 3;
  FILE* in = fopen(argv[1], "rb");
  if (!in) return 4;
  int hdr[3];
  float K[4], range[2];
  if (!read_all(in, hdr, sizeof(hdr)) || !read_all(in, K, sizeof(K)) || !read_all(in, range, sizeof(range))) return 5;
  const int w = hdr[0], h = hdr[1], n = hdr[2];
  const size_t px = static_cast<size_t>(w) * h;
  std::vector<float> img(px);
  float pose[12];
  rmd::SeedMatrix seeds(w, h, rmd::PinholeCamera(K[0], K[1], K[2], K[3]));
  rmd::DepthmapDenoiser denoiser(w, h);
  denoiser.setLargeSigmaSq(range[1] - range[0]);
  for (int k = 0; k < n; ++k) {
    if (!read_all(in, img.data(), px * 4) || !read_all(in, pose, sizeof(pose))) return 6;
    float r[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]}, t[3] = {pose[3], pose[7], pose[11]};
    const rmd::SE3<float> T_curr_world(r, t);
    if (k == 0) seeds.setReferenceImage(img.data(), T_curr_world, range[0], range[1]);
    else seeds.update(img.data(), T_curr_world);
  }
  fclose(in);
  std::vector<float> mu(px), sig(px), a(px), b(px), den(px);
  std::vector<int> conv(px);
  seeds.downloadDepthmap(mu.data());
  seeds.downloadSigmaSq(sig.data());
  seeds.downloadA(a.data());
  seeds.downloadB(b.data());
  seeds.downloadConvergence(conv.data());
  denoiser.denoise(seeds.getMu(), seeds.getSigmaSq(), seeds.getA(), seeds.getB(), den.data(), 0.5f, 30);
  rmd::ImageReducer<int> counter(dim3(16, 16), dim3(4, 4));
  const unsigned long long n_conv = counter.countEqual(seeds.getConvergence(), rmd::ConvergenceStates::CONVERGED);
  const unsigned long long n_conv2 = seeds.getConvergedCount();
  rmd::ImageReducer<float> summer(dim3(16, 16), dim3(4, 4));
  const float mu_sum = summer.sum(seeds.getMu());
  // the raw-pointer overloads and the int sum of the reference's header (reduction.cuh:33-47)
  const rmd::DeviceImage<int>& cv = seeds.getConvergence();
  const rmd::DeviceImage<float>& mu_img = seeds.getMu();
  if (counter.countEqual(cv.data, cv.stride, cv.width, cv.height, rmd::ConvergenceStates::CONVERGED) != n_conv) return 8;
  if (summer.sum(mu_img.data, mu_img.stride, mu_img.width, mu_img.height) != mu_sum) return 9;
  if (counter.sum(cv) != counter.sum(cv.data, cv.stride, cv.width, cv.height)) return 10;
  const float dist = seeds.getDistFromRef();
  FILE* out = fopen(argv[2], "wb");
  if (!out) return 7;
  fwrite(mu.data(), 4, px, out); fwrite(sig.data(), 4, px, out); fwrite(a.data(), 4, px, out); fwrite(b.data(), 4, px, out);
  fwrite(conv.data(), 4, px, out); fwrite(den.data(), 4, px, out);
  fwrite(&n_conv, 8, 1, out); fwrite(&n_conv2, 8, 1, out); fwrite(&mu_sum, 4, 1, out); fwrite(&dist, 4, 1, out);
  // extension: the publisher's point cloud, from the denoised depth still resident on the device
  std::vector<float> cloud(static_cast<size_t>(px) * 4);
  const unsigned long long n_points = seeds.downloadPointCloud(denoiser.resultHandle(), cloud.data(), px);
  fwrite(&n_points, 8, 1, out);
  fwrite(cloud.data(), 16, n_points, out);
  fclose(out);
  // extension: the same products off the update stream (publishAsync / collectPublication), requested while the handle moves on -- they must be
  // the synchronous ones above, bit for bit
  {
    std::vector<float> depth2(px), cloud2(static_cast<size_t>(px) * 4);
    std::vector<unsigned char> bgr(static_cast<size_t>(px) * 3), bgr2(static_cast<size_t>(px) * 3);
    std::vector<int> conv2(px);
    seeds.downloadConvergenceBGR8(bgr.data());
    const int ticket = seeds.publishAsync(RMD_HIP_PUBLISH_DEPTH | RMD_HIP_PUBLISH_CLOUD | RMD_HIP_PUBLISH_CONVERGENCE_BGR | RMD_HIP_PUBLISH_CONVERGENCE,
                                          range[1] - range[0], 0.5f, 30);
    unsigned int what = 0;
    int got_ticket = 0;
    size_t n2 = 0;
    while (!seeds.collectPublication(false, &what, &got_ticket, depth2.data(), cloud2.data(), px, &n2, bgr2.data(), conv2.data())) {}  // polling: false = still in flight
    if (got_ticket != ticket || n2 != n_points) return 11;
    if (memcmp(depth2.data(), den.data(), sizeof(float) * px) != 0) return 12;
    if (memcmp(cloud2.data(), cloud.data(), 16 * n_points) != 0) return 13;
    if (memcmp(bgr2.data(), bgr.data(), bgr.size()) != 0) return 14;
    if (memcmp(conv2.data(), conv.data(), sizeof(int) * px) != 0) return 15;
    bool threw = false;
    try { seeds.collectPublication(true, &what, &got_ticket, NULL, NULL, 0, &n2, NULL, NULL); } catch (const rmd::CudaException&) { threw = true; }  // nothing left
    if (!threw) return 16;
  }
  printf("facade_check OK: %dx%d, %d frames, converged %llu\n", w, h, n, n_conv);
  return 0;
}
