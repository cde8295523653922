// Synthetic "over-table" sequence generator (host only, deterministic).
//
// The reference's test dataset (test/dataset_main.cpp:37-52: 200 frames, 640x480,
// camera fx 481.2 / fy -480 / cx 319.5 / cy 239.5, per-frame ground-truth depth) is not
// redistributable and not present offline, so every test and benchmark in this
// repository runs on frames rendered here: an analytically ray-cast scene (ground
// plane z=0 plus three axis-aligned boxes) under a camera ~1.5 m above it looking
// straight down, with a procedural multi-octave texture.  Outputs have the same
// shapes and conventions as the reference's inputs: 8-bit gray image, pose
// T_world_cam as 3x4 row-major [R|t] (se3.cuh:72-77), depth = range along the
// pixel's ray in metres (seed_update.cu:81 uses norm(P), dataset.cpp:178 stores cm).
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <string.h>

namespace {

inline uint32_t hash2(int32_t x, int32_t y, uint32_t seed) {
  uint32_t h = static_cast<uint32_t>(x) * 374761393u + static_cast<uint32_t>(y) * 668265263u + seed * 2246822519u + 3266489917u;
  h = (h ^ (h >> 15)) * 2246822519u;
  h = (h ^ (h >> 13)) * 3266489917u;
  return h ^ (h >> 16);
}
inline double lattice(int32_t x, int32_t y, uint32_t seed) { return hash2(x, y, seed) * (1.0 / 4294967296.0); }

// smooth value noise in [0,1], lattice pitch `cell` metres
double vnoise(double u, double v, double cell, uint32_t seed) {
  const double gu = u / cell, gv = v / cell;
  const double fu = floor(gu), fv = floor(gv);
  const int32_t iu = static_cast<int32_t>(fu), iv = static_cast<int32_t>(fv);
  double a = gu - fu, b = gv - fv;
  a = a * a * (3.0 - 2.0 * a);
  b = b * b * (3.0 - 2.0 * b);
  const double n00 = lattice(iu, iv, seed), n10 = lattice(iu + 1, iv, seed);
  const double n01 = lattice(iu, iv + 1, seed), n11 = lattice(iu + 1, iv + 1, seed);
  return (n00 * (1 - a) + n10 * a) * (1 - b) + (n01 * (1 - a) + n11 * a) * b;
}

double texture_at(double u, double v, uint32_t seed) {
  double t = 0.0;
  t += 0.34 * (vnoise(u, v, 0.007, seed + 11u) - 0.5);
  t += 0.30 * (vnoise(u, v, 0.019, seed + 23u) - 0.5);
  t += 0.22 * (vnoise(u, v, 0.055, seed + 37u) - 0.5);
  t += 0.16 * (vnoise(u, v, 0.170, seed + 41u) - 0.5);
  t += 0.05 * sin(37.0 * u + 0.3) * sin(29.0 * v + 1.1);
  t += 0.04 * sin(11.3 * u - 7.7 * v);
  double val = 0.5 + 1.35 * t;
  if (val < 0.0) val = 0.0;
  if (val > 1.0) val = 1.0;
  return val;
}

struct Box { double lo[3], hi[3]; };

void scene_boxes(uint32_t seed, Box* b) {
  // three boxes on the plane, heights 0.10 .. 0.40 m; positions jitter with the seed
  const double jx = 0.10 * (lattice(1, 2, seed) - 0.5), jy = 0.10 * (lattice(3, 4, seed) - 0.5);
  const Box base[3] = {
      {{-0.55 + jx, -0.35 + jy, 0.0}, {-0.15 + jx, 0.05 + jy, 0.25}},
      {{0.10 - jx, 0.10 + jy, 0.0}, {0.60 - jx, 0.45 + jy, 0.40}},
      {{0.05 + jy, -0.50 - jx, 0.0}, {0.45 + jy, -0.20 - jx, 0.10}},
  };
  memcpy(b, base, sizeof(base));
}

}  // namespace

extern "C" {

// Camera pose of frame `k` of a smooth looping sweep, T_world_cam, 3x4 row-major.
// Frame 0 is the reference view.  Mean step ~1.8 cm, like the paper's sequence
// (4.576 m over 200 frames -> 2.3 cm, ICRA14_Pizzoli.pdf Table I).
void rmd_synth_pose(int k, unsigned seed, double* T_world_cam) {
  const double two_pi = 6.283185307179586;
  const double s0 = lattice(7, 9, seed), s1 = lattice(5, 1, seed);
  const double period = 120.0 + 20.0 * s0;
  const double th = two_pi * k / period;
  const double R = 0.33 + 0.06 * s1;
  const double px = R * sin(th), py = 0.6 * R * (1.0 - cos(th)), pz = 1.50 + 0.04 * sin(2.0 * th);
  const double yaw = 0.030 * sin(0.7 * th), pitch = 0.020 * sin(1.3 * th), roll = 0.020 * (cos(0.9 * th) - 1.0);
  // camera looking down: x_cam = +X, y_cam = -Y, z_cam = -Z, then small rotations about the camera axes
  const double cz = cos(yaw), sz = sin(yaw), cx = cos(pitch), sx = sin(pitch), cy = cos(roll), sy = sin(roll);
  const double Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
  const double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
  const double Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
  const double D[9] = {1, 0, 0, 0, -1, 0, 0, 0, -1};
  double A[9], B[9], C[9];
  auto mul = [](const double* a, const double* b, double* c) {
    for (int r = 0; r < 3; ++r)
      for (int col = 0; col < 3; ++col) c[3 * r + col] = a[3 * r] * b[col] + a[3 * r + 1] * b[3 + col] + a[3 * r + 2] * b[6 + col];
  };
  mul(D, Rz, A);
  mul(A, Rx, B);
  mul(B, Ry, C);
  for (int r = 0; r < 3; ++r) {
    for (int col = 0; col < 3; ++col) T_world_cam[4 * r + col] = C[3 * r + col];
  }
  T_world_cam[3] = px; T_world_cam[7] = py; T_world_cam[11] = pz;
}

// threads of the row loop below (0: the OpenMP default).  A container with a CPU quota far below its visible core count (16 of 256 on
// the measurement box) is throttled for most of every scheduling period when all visible cores spin up.
static int g_render_threads = 0;
void rmd_synth_set_threads(int n) { g_render_threads = n > 0 ? n : 0; }

// Render one frame.  gray: w*h bytes; range: w*h floats (may be NULL).
int rmd_synth_render(int w, int h, double fx, double fy, double cx, double cy, const double* T_world_cam, unsigned seed,
                     uint8_t* gray, float* range) {
  if (w <= 0 || h <= 0 || !T_world_cam || !gray) return -1;
  Box boxes[3];
  scene_boxes(seed, boxes);
  const double* T = T_world_cam;
  const double o[3] = {T[3], T[7], T[11]};
  const int n_threads = g_render_threads > 0 ? g_render_threads : omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(n_threads)
  for (int v = 0; v < h; ++v) {
    for (int u = 0; u < w; ++u) {
      const double dc[3] = {(u - cx) / fx, (v - cy) / fy, 1.0};
      const double d[3] = {T[0] * dc[0] + T[1] * dc[1] + T[2] * dc[2], T[4] * dc[0] + T[5] * dc[1] + T[6] * dc[2],
                           T[8] * dc[0] + T[9] * dc[1] + T[10] * dc[2]};
      double best_t = 1e30;
      int surf = -1, axis = 2;
      if (d[2] < 0.0) {
        const double t = -o[2] / d[2];
        if (t > 0.0) { best_t = t; surf = 0; axis = 2; }
      }
      for (int b = 0; b < 3; ++b) {
        double tn = -1e30, tf = 1e30;
        int an = 0;
        bool miss = false;
        for (int a = 0; a < 3; ++a) {
          if (fabs(d[a]) < 1e-12) {
            if (o[a] < boxes[b].lo[a] || o[a] > boxes[b].hi[a]) { miss = true; break; }
            continue;
          }
          double t0 = (boxes[b].lo[a] - o[a]) / d[a], t1 = (boxes[b].hi[a] - o[a]) / d[a];
          if (t0 > t1) { const double tmp = t0; t0 = t1; t1 = tmp; }
          if (t0 > tn) { tn = t0; an = a; }
          if (t1 < tf) tf = t1;
        }
        if (miss || tn > tf || tn <= 0.0) continue;
        if (tn < best_t) { best_t = tn; surf = 1 + b; axis = an; }
      }
      double val = 0.5, rng = 0.0;
      if (surf >= 0) {
        const double P[3] = {o[0] + best_t * d[0], o[1] + best_t * d[1], o[2] + best_t * d[2]};
        double tu, tv;
        if (axis == 2) { tu = P[0]; tv = P[1]; }
        else if (axis == 0) { tu = P[1] + 3.1; tv = P[2] + 5.3; }
        else { tu = P[0] + 7.7; tv = P[2] + 2.9; }
        val = texture_at(tu + 1.37 * surf, tv - 0.73 * surf, seed * 8u + static_cast<uint32_t>(surf));
        rng = best_t * sqrt(dc[0] * dc[0] + dc[1] * dc[1] + 1.0);
      }
      gray[static_cast<size_t>(v) * w + u] = static_cast<uint8_t>(floor(255.0 * val + 0.5));
      if (range) range[static_cast<size_t>(v) * w + u] = static_cast<float>(rng);
    }
  }
  return 0;
}

}  // extern "C"
