// Timed driver of the depth-filter path in C++ over the drop-in headers (include/rmd/) -- the reference's own measurement program is C++
// (test/dataset_main.cpp:101-105 times `depthmap.update(img, T_curr_world)`); bench.py measures the same thing through ctypes and pays ~30 us of
// interpreter per update() call, against a device update of ~39 us.  Same workload, bracketing and arithmetic as bench.py's timed region:
//
//   pass  = setReferenceImage(frame 0) + update(frame 1 .. F-1) of the synthetic over-table sequence (scene `--scene`, 640x480 x 200, patch side 9)
//   timed = W warm-up passes, synchronise, then K passes between two wall-clock reads with a synchronisation before the second one;
//           device time = ONE HIP event pair on the handle's stream around the region (RMD_HIP_OPT_TIMING = 2)
//
// in three frame modes: "u8" (the headline: every frame an 8-bit image in pageable host memory, rmd_hip_seeds_update_u8), "resident" (frames in
// HBM, read in place, rmd_hip_seeds_update_device), "float" (rmd::SeedMatrix::update(float*), the reference's own signature) and "pinned" (8-bit
// frames the caller keeps in pinned host memory, rmd_hip_seeds_update_u8_pinned: no copy into the library's ring).  Prints one
// JSON line per mode: Mpix/s, us per update (wall and device), host CPU seconds of the process over the region (getrusage: all threads) and
// the wall time per update() until the call returned.  --ranks-probe N forks N processes that run the u8 mode concurrently (scenes 0..N-1, device
// rank % device_count) and reports each one's rate and CPU time: what N ranks cost the host (profiles/r05_nranks/).
//
// Build: rpg_open_remode_amd/build.py::build_apps (g++, links librmd_hip.so and librmd_synth.so).  Nothing here touches oracle/.
#include <rmd/seed_matrix.cuh>

#include <sys/resource.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
void rmd_synth_pose(int k, unsigned seed, double* T_world_cam);
int rmd_synth_render(int w, int h, double fx, double fy, double cx, double cy, const double* T_world_cam, unsigned seed, unsigned char* gray, float* range);
void rmd_synth_set_threads(int n);
}

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
double cpu_s() {
  rusage u;
  getrusage(RUSAGE_SELF, &u);
  return u.ru_utime.tv_sec + u.ru_stime.tv_sec + 1e-6 * (u.ru_utime.tv_usec + u.ru_stime.tv_usec);
}
void check(int rc, const char* what) {
  if (rc != RMD_HIP_OK) {
    fprintf(stderr, "%s: %s\n", what, rmd_hip_last_error());
    exit(2);
  }
}
// T_curr_world (row-major 3x4, float) = inverse of T_world_cam (double)
void invert_pose(const double* T, float* out) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) out[4 * i + j] = static_cast<float>(T[4 * j + i]);
    out[4 * i + 3] = static_cast<float>(-(T[i] * T[3] + T[4 + i] * T[7] + T[8 + i] * T[11]));
  }
}

struct Sequence {
  int w, h, n;
  double K[4];
  std::vector<std::vector<unsigned char> > gray;
  std::vector<std::vector<float> > image;  // gray * (1.0f / 255.0f), the conversion of depthmap.cpp:105
  std::vector<std::vector<float> > pose;   // T_curr_world, 12 floats
  float min_depth, max_depth;
};

Sequence render(int w, int h, int n, unsigned scene, bool want_float) {
  Sequence s;
  s.w = w; s.h = h; s.n = n;
  s.K[0] = 481.2 * w / 640.0; s.K[1] = -480.0 * h / 480.0; s.K[2] = (w - 1) / 2.0; s.K[3] = (h - 1) / 2.0;  // test/dataset_main.cpp:37 scaled (synth.intrinsics)
  std::vector<float> range(static_cast<size_t>(w) * h);
  for (int k = 0; k < n; ++k) {
    double T[12];
    rmd_synth_pose(k, scene, T);
    s.gray.emplace_back(static_cast<size_t>(w) * h);
    if (rmd_synth_render(w, h, s.K[0], s.K[1], s.K[2], s.K[3], T, scene, s.gray.back().data(), k == 0 ? range.data() : nullptr) != 0) exit(3);
    s.pose.emplace_back(12);
    invert_pose(T, s.pose.back().data());
    if (want_float) {
      s.image.emplace_back(static_cast<size_t>(w) * h);
      for (size_t i = 0; i < s.image.back().size(); ++i) s.image.back()[i] = static_cast<float>(s.gray.back()[i]) * (1.0f / 255.0f);
    }
  }
  s.min_depth = s.max_depth = range[0];
  for (float v : range) { s.min_depth = std::min(s.min_depth, v); s.max_depth = std::max(s.max_depth, v); }
  return s;
}

struct Result { double wall_s, device_ms, cpu_s, submit_s; long updates; size_t converged; };
int g_unit_target = 0;  // --unit-target: RMD_HIP_OPT_UNIT_TARGET (0: the library's default)

Result run_mode(const Sequence& q, const std::string& mode, int steps, int warmup) {
  rmd::PinholeCamera cam(static_cast<float>(q.K[0]), static_cast<float>(q.K[1]), static_cast<float>(q.K[2]), static_cast<float>(q.K[3]));
  rmd::SeedMatrix seeds(q.w, q.h, cam);
  rmd_hip_seeds_t* h = seeds.handle();
  if (g_unit_target > 0) check(rmd_hip_seeds_set_option(h, RMD_HIP_OPT_UNIT_TARGET, g_unit_target), "set_option");
  std::vector<rmd_hip_image_t*> dev;
  std::vector<const float*> dev_ptr;
  size_t dev_stride = 0;
  if (mode == "resident") {
    for (int k = 0; k < q.n; ++k) {
      rmd_hip_image_t* im = nullptr;
      check(rmd_hip_image_create(RMD_HIP_KIND_F32, q.w, q.h, &im), "image_create");
      check(rmd_hip_image_upload(im, q.image[k].data()), "image_upload");
      void* p = nullptr;
      check(rmd_hip_image_info(im, nullptr, nullptr, nullptr, nullptr, &dev_stride, &p), "image_info");
      dev.push_back(im); dev_ptr.push_back(static_cast<const float*>(p));
    }
  }
  unsigned char* pinned = nullptr;  // mode "pinned": the whole sequence in ONE block of pinned host memory, as a producer would leave it there
  const size_t frame_bytes = static_cast<size_t>(q.w) * q.h;
  if (mode == "pinned") {
    check(rmd_hip_host_alloc(reinterpret_cast<void**>(&pinned), frame_bytes * q.n), "host_alloc");
    for (int k = 0; k < q.n; ++k) memcpy(pinned + frame_bytes * k, q.gray[k].data(), frame_bytes);
  }
  auto one_pass = [&]() {
    if (mode == "pinned") {
      check(rmd_hip_seeds_set_reference_u8(h, q.gray[0].data(), q.pose[0].data(), q.min_depth, q.max_depth), "set_reference_u8");
      for (int k = 1; k < q.n; ++k) check(rmd_hip_seeds_update_u8_pinned(h, pinned + frame_bytes * k, q.pose[k].data(), nullptr), "update_u8_pinned");
    } else if (mode == "u8") {
      check(rmd_hip_seeds_set_reference_u8(h, q.gray[0].data(), q.pose[0].data(), q.min_depth, q.max_depth), "set_reference_u8");
      for (int k = 1; k < q.n; ++k) check(rmd_hip_seeds_update_u8(h, q.gray[k].data(), q.pose[k].data()), "update_u8");
    } else if (mode == "resident") {
      check(rmd_hip_seeds_set_reference_device(h, dev_ptr[0], dev_stride, q.pose[0].data(), q.min_depth, q.max_depth), "set_reference_device");
      for (int k = 1; k < q.n; ++k) check(rmd_hip_seeds_update_device(h, dev_ptr[k], dev_stride, q.pose[k].data()), "update_device");
    } else {  // the reference's own calls: rmd::SeedMatrix::setReferenceImage / update with float frames in host memory
      rmd::SE3<float> T0;
      memcpy(T0.data.data, q.pose[0].data(), 12 * sizeof(float));
      seeds.setReferenceImage(const_cast<float*>(q.image[0].data()), T0, q.min_depth, q.max_depth);
      for (int k = 1; k < q.n; ++k) {
        rmd::SE3<float> T;
        memcpy(T.data.data, q.pose[k].data(), 12 * sizeof(float));
        seeds.update(const_cast<float*>(q.image[k].data()), T);
      }
    }
  };
  for (int i = 0; i < warmup; ++i) one_pass();
  check(rmd_hip_seeds_sync(h), "sync");
  check(rmd_hip_seeds_set_option(h, RMD_HIP_OPT_TIMING, 2), "set_option");
  check(rmd_hip_seeds_timing_reset(h), "timing_reset");
  const double c0 = cpu_s(), t0 = now_s();
  for (int i = 0; i < steps; ++i) one_pass();
  const double t_sub = now_s();
  check(rmd_hip_seeds_sync(h), "sync");
  Result r;
  r.wall_s = now_s() - t0;
  r.cpu_s = cpu_s() - c0;
  r.submit_s = t_sub - t0;
  check(rmd_hip_seeds_timing(h, RMD_HIP_STAGE_UPDATE, &r.device_ms, &r.updates), "timing");
  check(rmd_hip_seeds_set_option(h, RMD_HIP_OPT_TIMING, 0), "set_option");
  r.converged = seeds.getConvergedCount();
  for (rmd_hip_image_t* im : dev) rmd_hip_image_destroy(im);
  if (pinned) check(rmd_hip_host_free(pinned), "host_free");
  return r;
}

void print_line(const Sequence& q, const std::string& mode, int steps, int warmup, const Result& r, int rank) {
  const double upd = static_cast<double>(steps) * (q.n - 1);
  printf("{\"driver\": \"apps/bench_main.cpp\", \"mode\": \"%s\", \"rank\": %d, \"metric\": \"Mpix/s depth-filter updates\", \"value\": %.2f, \"unit\": \"Mpix/s\", "
         "\"workload\": \"%dx%d, %d frames, patch side %d\", \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.4f, \"us_per_update_wall\": %.3f, "
         "\"us_per_update_device\": %.3f, \"host_cpu_s\": %.4f, \"host_cores_busy\": %.3f, \"host_submit_us_per_update\": %.3f, \"converged\": %zu}\n",
         mode.c_str(), rank, q.w * static_cast<double>(q.h) * upd / r.wall_s / 1e6, q.w, q.h, q.n, RMD_CORR_PATCH_SIDE, steps, warmup, r.wall_s / steps * 1e3,
         r.wall_s / upd * 1e6, r.device_ms * 1e3 / std::max(1L, r.updates), r.cpu_s, r.cpu_s / r.wall_s, r.submit_s / upd * 1e6, r.converged);
  fflush(stdout);
}

}  // namespace

int main(int argc, char** argv) {
  int w = 640, h = 480, frames = 200, steps = 5, warmup = 1, scene = 0, ranks = 0, threads = 4;
  g_unit_target = 0;
  std::string modes = "u8,resident";
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() { return i + 1 < argc ? argv[++i] : (fprintf(stderr, "missing value of %s\n", a.c_str()), exit(1), ""); };
    if (a == "--size") sscanf(next(), "%dx%d", &w, &h);
    else if (a == "--frames") frames = atoi(next());
    else if (a == "--steps") steps = atoi(next());
    else if (a == "--warmup") warmup = atoi(next());
    else if (a == "--scene") scene = atoi(next());
    else if (a == "--modes") modes = next();
    else if (a == "--ranks-probe") ranks = atoi(next());
    else if (a == "--render-threads") threads = atoi(next());
    else if (a == "--unit-target") g_unit_target = atoi(next());
    else { fprintf(stderr, "usage: bench_main [--size WxH] [--frames F] [--steps K] [--warmup W] [--scene S] [--modes u8,resident,float,pinned] [--ranks-probe N]\n"); return 1; }
  }
  rmd_synth_set_threads(threads);
  int rank = 0;
  if (ranks > 1) {  // N concurrent processes on the visible device(s): rank r uses device r % device_count and scene r; the parent is rank 0
    for (int r = 1; r < ranks; ++r)
      if (fork() == 0) { rank = r; break; }
    modes = "u8";
    scene = rank;
  }
  int n_dev = 0;
  check(rmd_hip_device_count(&n_dev), "device_count");
  check(rmd_hip_set_device(rank % n_dev), "set_device");
  const Sequence q = render(w, h, frames, static_cast<unsigned>(scene), modes.find("resident") != std::string::npos || modes.find("float") != std::string::npos);
  size_t pos = 0;
  while (pos < modes.size()) {
    const size_t e = modes.find(',', pos);
    const std::string mode = modes.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
    pos = e == std::string::npos ? modes.size() : e + 1;
    if (mode != "u8" && mode != "resident" && mode != "float" && mode != "pinned") { fprintf(stderr, "unknown mode %s\n", mode.c_str()); return 1; }
    print_line(q, mode, steps, warmup, run_mode(q, mode, steps, warmup), rank);
  }
  if (ranks > 1 && rank == 0)
    while (wait(nullptr) > 0) {}
  return 0;
}
