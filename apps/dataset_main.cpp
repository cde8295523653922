// The reference's stand-alone experiment, test/dataset_main.cpp:32-138, written against the drop-in headers of this
// repository (include/rmd/*.cuh over librmd_hip.so) and apps/dataset.h in place of OpenCV / boost / Eigen.
//
//   RMD_TEST_DATA_PATH=/data/remode_test_data ./dataset_main [--device=N] [--end=200] [--out=prefix]
//
// Frame 0 is the reference view (scene range = min / max of its ground-truth depth map), every further frame an update with
// T_world_curr.inv(); per-update wall time, then mean / variance / standard deviation as the reference prints them; the depth
// map and the TV-L1 denoised depth map (0.5, 200) are written as raw float32 to <prefix>depth.f32 / <prefix>denoised.f32
// instead of cv::imshow.  Unlike the reference, the ground-truth depth map is only required for the reference frame.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <iostream>
#include <numeric>
#include <string>
#include <vector>

#include <rmd/check_cuda_device.cuh>
#include <rmd/depthmap_denoiser.cuh>
#include <rmd/seed_matrix.cuh>

#include "dataset.h"

namespace {

// rmd::Depthmap (src/depthmap.cpp) reduced to what the experiment uses: 8-bit input scaled by 1/255 (:105), the seed
// matrix, the denoiser.  Argument order of the reference: (width, height, fx, cx, fy, cy).
class Depthmap {
 public:
  Depthmap(size_t width, size_t height, float fx, float cx, float fy, float cy)
      : width_(width), height_(height), seeds_(width, height, rmd::PinholeCamera(fx, fy, cx, cy)), denoiser_(width, height),
        img_32fc1_(width * height), output_depth_32fc1_(width * height) {}
  bool setReferenceImage(const rmd::test::GrayImage& img, const rmd::SE3<float>& T_curr_world, float min_depth, float max_depth) {
    denoiser_.setLargeSigmaSq(max_depth - min_depth);
    inputImage(img);
    return seeds_.setReferenceImage(img_32fc1_.data(), T_curr_world, min_depth, max_depth);
  }
  void update(const rmd::test::GrayImage& img, const rmd::SE3<float>& T_curr_world) {
    inputImage(img);
    seeds_.update(img_32fc1_.data(), T_curr_world);
  }
  void downloadDepthmap() { seeds_.downloadDepthmap(output_depth_32fc1_.data()); }
  void downloadDenoisedDepthmap(float lambda, int iterations) {
    denoiser_.denoise(seeds_.getMu(), seeds_.getSigmaSq(), seeds_.getA(), seeds_.getB(), output_depth_32fc1_.data(), lambda, iterations);
  }
  const std::vector<float>& getDepthmap() const { return output_depth_32fc1_; }
  float getConvergedPercentage() const {
    return static_cast<float>(seeds_.getConvergedCount()) / static_cast<float>(width_ * height_) * 100.0f;
  }

 private:
  void inputImage(const rmd::test::GrayImage& img) {
    for (size_t i = 0; i < img_32fc1_.size(); ++i) img_32fc1_[i] = static_cast<float>(img.data[i]) * (1.0f / 255.0f);
  }
  size_t width_, height_;
  rmd::SeedMatrix seeds_;
  rmd::DepthmapDenoiser denoiser_;
  std::vector<float> img_32fc1_, output_depth_32fc1_;
};

bool write_f32(const std::string& path, const std::vector<float>& v) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  const bool ok = fwrite(v.data(), sizeof(float), v.size(), f) == v.size();
  fclose(f);
  return ok;
}

}  // namespace

int main(int argc, char** argv) {
  size_t end = 200;
  std::string out_prefix;
  bool quiet = false;
  for (int i = 1; i < argc; ++i) {
    const std::string a(argv[i]);
    if (a.compare(0, 6, "--end=") == 0) end = static_cast<size_t>(atol(a.c_str() + 6));
    else if (a.compare(0, 6, "--out=") == 0) out_prefix = a.substr(6);
    else if (a == "--quiet") quiet = true;
  }
  if (!rmd::checkCudaDevice(argc, argv)) return EXIT_FAILURE;

  rmd::PinholeCamera cam(481.2f, -480.0f, 319.5f, 239.5f);  // dataset_main.cpp:37
  rmd::test::Dataset dataset("first_200_frames_traj_over_table_input_sequence.txt");
  if (!dataset.loadPathFromEnv())
    std::cerr << "ERROR: could not retrieve dataset path from the environment variable '" << rmd::test::Dataset::getDataPathEnvVar() << "'"
              << std::endl;
  if (!dataset.readDataSequence(0, end)) {
    std::cerr << "ERROR: could not read dataset" << std::endl;
    return EXIT_FAILURE;
  }

  const size_t width = 640, height = 480;
  bool first_img = true;
  Depthmap depthmap(width, height, cam.fx, cam.cx, cam.fy, cam.cy);
  std::vector<double> update_time;

  for (std::vector<rmd::test::DatasetEntry>::const_iterator it = dataset.begin(); it != dataset.end(); ++it) {
    const rmd::test::DatasetEntry& data = *it;
    rmd::test::GrayImage img;
    if (!dataset.readImage(img, data) || img.width != static_cast<int>(width) || img.height != static_cast<int>(height)) {
      std::cerr << "ERROR: could not read image " << data.getImageFileName() << std::endl;
      continue;
    }
    rmd::SE3<float> T_world_curr;
    dataset.readCameraPose(T_world_curr, data);
    if (!quiet) {
      std::cout << "RUN EXPERIMENT: inputting image " << data.getImageFileName() << std::endl;
      std::cout << "T_world_curr:" << std::endl;
      std::cout << T_world_curr << std::endl;
    }
    if (first_img) {
      std::vector<float> depth_32fc1;
      if (!dataset.readDepthmap(depth_32fc1, data, width, height)) {
        std::cerr << "ERROR: could not read depthmap " << data.getDepthmapFileName() << std::endl;
        continue;
      }
      float min_depth = depth_32fc1[0], max_depth = depth_32fc1[0];
      for (size_t i = 1; i < depth_32fc1.size(); ++i) {
        min_depth = depth_32fc1[i] < min_depth ? depth_32fc1[i] : min_depth;
        max_depth = depth_32fc1[i] > max_depth ? depth_32fc1[i] : max_depth;
      }
      if (depthmap.setReferenceImage(img, T_world_curr.inv(), min_depth, max_depth)) {
        first_img = false;
      } else {
        std::cerr << "ERROR: could not set reference image" << std::endl;
        return EXIT_FAILURE;
      }
    } else {
      const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
      depthmap.update(img, T_world_curr.inv());
      const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (!quiet) printf("\nUPDATE execution time: %f seconds.\n", t);
      update_time.push_back(t);
    }
  }
  if (first_img) {
    std::cerr << "ERROR: no reference frame could be set" << std::endl;
    return EXIT_FAILURE;
  }

  depthmap.downloadDepthmap();
  if (!out_prefix.empty() && !write_f32(out_prefix + "depth.f32", depthmap.getDepthmap())) return EXIT_FAILURE;
  const float converged = depthmap.getConvergedPercentage();
  depthmap.downloadDenoisedDepthmap(0.5f, 200);
  if (!out_prefix.empty() && !write_f32(out_prefix + "denoised.f32", depthmap.getDepthmap())) return EXIT_FAILURE;

  const double n = update_time.empty() ? 1.0 : static_cast<double>(update_time.size());
  const double time_mean = std::accumulate(update_time.begin(), update_time.end(), 0.0) / n;
  double time_var = 0.0;
  for (size_t i = 0; i < update_time.size(); ++i) time_var += (update_time[i] - time_mean) * (update_time[i] - time_mean);
  time_var /= n;
  std::cout << "\n\n";
  std::cout << "MEAN update time: " << time_mean << std::endl;
  std::cout << "VAR  update time: " << time_var << std::endl << "(STDDEV: " << std::sqrt(time_var) << ")" << std::endl;
  std::cout << "updates: " << update_time.size() << "   converged: " << converged << " %" << std::endl;
  return EXIT_SUCCESS;
}
