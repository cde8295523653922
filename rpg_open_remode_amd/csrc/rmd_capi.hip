// librmd_hip.so: C ABI (include/rmd_hip.h) over the HIP kernels -- library, device selection, rmd::DeviceImage, rmd::SeedMatrix (everything
// but the sources of its frames, rmd_ingest.hip, and its kernel launches, rmd_update.hip); the other units are listed in rmd_host.hpp.
// Host orchestration of rmd::SeedMatrix (seed_matrix.cu:28-230), rmd::DepthmapDenoiser (depthmap_denoiser.cu:124-229), rmd::ImageReducer
// (reduction.cu) and rmd::DeviceImage (device_image.cuh), redesigned around per-handle HIP streams, kernarg parameter blocks and pinned
// staging instead of the reference's default stream, device-resident descriptor structs and global texture references.
#include "rmd_host.hpp"

using namespace rmdh;

namespace {
thread_local char g_last_error[512] = "";
}

namespace rmdh {

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  return code;
}
const char* last_error() { return g_last_error; }

// accepted values per tunable (rmd_hip_set_tunable AND the environment presets)
static const int tunable_lo[RMD_HIP_NUM_TUNABLES] = {-1, 0, 1, 0, 0, 1, 0, 0, 0, 0, 1, -1}, tunable_hi[RMD_HIP_NUM_TUNABLES] = {3, 4, 1024,
    1 << 20, 1, 16, 1, 1, 1, 8, 2, 4};

// The library's process-wide settings and THE ONE PLACE where it reads its environment (include/rmd_hip.h: RMD_HIP_TUNE_*): the
// defaults come from RMD_HIP_<NAME>, once, at the first call; rmd_hip_set_tunable overrides them for handles created afterwards.
Tunables& tunables() {
  static Tunables T = [] {
    Tunables t;
    t.v[RMD_HIP_TUNE_HOST_FRAMES] = HOST_FRAMES_DEFAULT;
    t.v[RMD_HIP_TUNE_BATCH_GROUPS] = 0;
    t.v[RMD_HIP_TUNE_AHEAD_WGS] = AHEAD_WGS;
    t.v[RMD_HIP_TUNE_PACK_BACKOFF] = 15;
    t.v[RMD_HIP_TUNE_FLOAT_AS_BYTES] = 1;
    t.v[RMD_HIP_TUNE_COPY_THREADS] = 4;
    t.v[RMD_HIP_TUNE_FUSED_INGEST] = 1;
    t.v[RMD_HIP_TUNE_INGEST_PROFILE] = 0;
    t.v[RMD_HIP_TUNE_HOST_WAIT] = 1;
    t.v[RMD_HIP_TUNE_RING_DEPTH] = 0;
    t.v[RMD_HIP_TUNE_COPY_STREAMS] = 1;
    t.v[RMD_HIP_TUNE_COPY_ENGINES] = -1;
    static const char* const names[RMD_HIP_NUM_TUNABLES] = {"RMD_HIP_HOST_FRAMES", "RMD_HIP_BATCH_GROUPS", "RMD_HIP_AHEAD_WGS",
        "RMD_HIP_PACK_BACKOFF",
                                                            "RMD_HIP_FLOAT_AS_BYTES", "RMD_HIP_COPY_THREADS", "RMD_HIP_FUSED_INGEST",
                                                                "RMD_HIP_INGEST_PROFILE", "RMD_HIP_HOST_WAIT",
                                                            "RMD_HIP_RING_DEPTH", "RMD_HIP_COPY_STREAMS", "RMD_HIP_COPY_ENGINES"};
    // A preset from the environment passes the same range check as rmd_hip_set_tunable; one that fails it -- or does not parse -- is
    // IGNORED with a line on stderr (a negative RMD_HIP_AHEAD_WGS used to go straight into the search kernel's grid arithmetic).
    for (int k = 0; k < RMD_HIP_NUM_TUNABLES; ++k) {
      const char* e = getenv(names[k]);
      if (!e || !e[0]) continue;
      long v = 0;
      bool parsed = false;
      if (k == RMD_HIP_TUNE_HOST_FRAMES) {  // by name, or by number
        static const char* const modes[4] = {"staged", "staged_ahead", "inplace", "inplace_ahead"};
        for (int m = 0; m < 4; ++m)
          if (!strcmp(e, modes[m])) { v = m; parsed = true; }
      }
      if (!parsed) {
        char* end = nullptr;
        v = strtol(e, &end, 10);
        parsed = end != e && *end == '\0';
      }
      if (!parsed || v < tunable_lo[k] || v > tunable_hi[k]) {
        fprintf(stderr, "[rmd_hip] %s=%s ignored: %s [%d, %d]\n", names[k], e, parsed ? "outside" : "not a number in", tunable_lo[k],
            tunable_hi[k]);
        continue;
      }
      t.v[k] = static_cast<int>(v);
    }
    return t;
  }();
  return T;
}

// wait until the owner of an image (if any) has settled it
int image_settle(const rmd_hip_image* img) {
  ScopedDevice dev(img->device);  // the owner's stream belongs to the image's device, whatever the caller's current one is
  if (img->owner_seeds) return seeds_sync(img->owner_seeds);
  if (img->owner_stream) HIP_TRY(hipStreamSynchronize(img->owner_stream));
  return RMD_HIP_OK;
}

int image_alloc(rmd_hip_image* img, int kind, int width, int height) {
  if (width <= 0 || height <= 0 || kind < 0 || kind > RMD_HIP_KIND_F32X2)
    return fail(RMD_HIP_ERR_INVALID_ARG, "image: bad kind/size (%d, %dx%d)", kind, width, height);
  const size_t es = kind_size(kind);
  // rows padded to 256 B so that every row starts on a full HBM burst / 64-lane dword access
  const size_t pitch = (static_cast<size_t>(width) * es + 255) / 256 * 256;
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, pitch * height));
  // hipMemset runs on the null stream; the handles' kernels run on non-blocking streams that do NOT order against it,
  // so the fill must have completed before the buffer is handed out
  HIP_TRY(hipMemset(p, 0, pitch * height));
  HIP_TRY(hipStreamSynchronize(nullptr));
  img->kind = kind; img->width = width; img->height = height;
  img->pitch = pitch; img->stride = pitch / es; img->data = p; img->owns = true;
  (void)hipGetDevice(&img->device);
  return RMD_HIP_OK;
}

// the error word the setup kernel's ingest workgroups raise when a staging copy never arrived: reported ONCE (the frames of that update are
// invalid), then cleared, so that the handle is usable again from the next reference frame on
int ingest_error_check(unsigned int* h_progress) {
  if (h_progress && h_progress[1] != 0u) {
    h_progress[1] = 0u;
    return fail(RMD_HIP_ERR_RUNTIME,
        "seed update: the staging copy of a host frame did not complete within the kernel's bounded wait (about "
                                     "0.1 s); the seed state is invalid until the next setReferenceImage");
  }
  return RMD_HIP_OK;
}

// every observer of the seed state goes through here: settle deferred work, then wait for the stream
int seeds_sync(const rmd_hip_seeds* s) {
  rmd_hip_seeds* m = const_cast<rmd_hip_seeds*>(s);
  TRY(seeds_flush(m));
  HIP_TRY(hipStreamSynchronize(s->stream));
  TRY(ingest_error_check(m->batch ? m->batch->group_of(m->batch_index).h_progress : m->h_progress));
  for (auto& t : m->timers) t.drain();
  if (m->stats_pending) {
    for (int k = 0; k < 16; ++k) m->last_stats[k] = static_cast<long long>(m->h_scalars[1 + k]);
    m->stats_pending = false;
  }
  return RMD_HIP_OK;
}

int seeds_bind_device(const rmd_hip_seeds* s) {
  int cur = -1;
  HIP_TRY(hipGetDevice(&cur));
  if (cur != s->device) HIP_TRY(hipSetDevice(s->device));
  return RMD_HIP_OK;
}

// common tail of setReferenceImage (seed_matrix.cu:95-113) once the frame is in planes[REF_IMG]
int seeds_after_reference(rmd_hip_seeds* s, const float* T_curr_world, float min_depth, float max_depth) {
  TRY(seeds_flush(s));  // a deferred finalisation of the old reference must not run after the re-initialisation
  rmdk::SeedParams& P = s->P;
  P.avg_depth = (min_depth + max_depth) / 2.0f;
  P.depth_range = max_depth - min_depth;
  P.sigma_sq_max = P.depth_range * P.depth_range / 36.0f;
  P.eta_inlier = 0.7f;
  P.eta_outlier = 0.05f;
  P.epsilon = P.depth_range / 1000.0f;
  rmdk::Pose T;
  memcpy(T.d, T_curr_world, sizeof(T.d));
  s->T_world_ref = pose_inverse(T);
  TRY(seeds_launch_init(s));
  s->async_count_valid = false;
  // the pipeline's load statistics (unit size from the previous frame's work) belong to the old sequence; in a batch the other
  // members' next frame simply starts from the largest unit size again
  s->mws->frame = 0;
  HIP_TRY(hipMemsetAsync(s->mws->d_shards, 0, 3 * rmdk::UNIT_SHARDS * sizeof(unsigned long long), s->stream));
  s->has_reference = true;
  // the reference synchronises here (seed_matrix.cu:113); so do we: the host image is borrowed
  return seeds_sync(s);
}

// common tail of update (seed_matrix.cu:124-157) once the frame is in planes[CURR_IMG]
void seeds_frame_pose(rmd_hip_seeds* s, const float* T_curr_world) {
  rmdk::Pose T;
  memcpy(T.d, T_curr_world, sizeof(T.d));
  const rmdk::Pose T_curr_ref = pose_compose(T, s->T_world_ref);
  const float tx = T_curr_ref.d[3], ty = T_curr_ref.d[7], tz = T_curr_ref.d[11];
  s->dist_from_ref = sqrtf(tx * tx + ty * ty + tz * tz);
  s->P.T_curr_ref = T_curr_ref;
  s->P.T_ref_curr = pose_inverse(T_curr_ref);
}
int seeds_after_frame(rmd_hip_seeds* s, const float* T_curr_world, const PendingIngest* ingest) {
  seeds_frame_pose(s, T_curr_world);
  return seeds_launch_update(s, ingest);
}

}  // namespace rmdh

extern "C" {

const char* rmd_hip_last_error(void) { return g_last_error; }
int rmd_hip_version(void) { return RMD_HIP_VERSION_NUMBER; }

int rmd_hip_set_tunable(int tunable, int value) {
  if (tunable < 0 || tunable >= RMD_HIP_NUM_TUNABLES) return fail(RMD_HIP_ERR_INVALID_ARG, "set_tunable: unknown tunable %d", tunable);
  if (value < tunable_lo[tunable] || value > tunable_hi[tunable])
    return fail(RMD_HIP_ERR_INVALID_ARG, "set_tunable: value %d of tunable %d outside [%d, %d]", value, tunable, tunable_lo[tunable],
        tunable_hi[tunable]);
  tunables().v[tunable] = value;
  return RMD_HIP_OK;
}
int rmd_hip_get_tunable(int tunable, int* value) {
  if (tunable < 0 || tunable >= RMD_HIP_NUM_TUNABLES || !value) return fail(RMD_HIP_ERR_INVALID_ARG, "get_tunable: bad argument");
  *value = tunables().v[tunable];
  return RMD_HIP_OK;
}

// ---- device selection (check_cuda_device.cu:23-117) -------------------------------------------
int rmd_hip_device_count(int* count) {
  if (!count) return fail(RMD_HIP_ERR_INVALID_ARG, "device_count: null output");
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    *count = 0;
    return fail(RMD_HIP_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(e));
  }
  *count = n;
  return RMD_HIP_OK;
}
int rmd_hip_set_device(int device_id) {
  int n = 0;
  TRY(rmd_hip_device_count(&n));
  if (device_id < 0 || device_id >= n)
    return fail(RMD_HIP_ERR_INVALID_ARG, "invalid device id %d: specify a value in [0, %d]", device_id, n - 1);
  HIP_TRY(hipSetDevice(device_id));
  return RMD_HIP_OK;
}
int rmd_hip_device_name(int device_id, char* buf, size_t buf_len) {
  if (!buf || buf_len == 0) return fail(RMD_HIP_ERR_INVALID_ARG, "device_name: null buffer");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  snprintf(buf, buf_len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return RMD_HIP_OK;
}

// ---- DeviceImage ----------------------------------------------------------------------------
int rmd_hip_image_create(int kind, int width, int height, rmd_hip_image_t** out) {
  if (!out) return fail(RMD_HIP_ERR_INVALID_ARG, "image_create: null output");
  rmd_hip_image* img = new (std::nothrow) rmd_hip_image();
  if (!img) return fail(RMD_HIP_ERR_RUNTIME, "image_create: out of host memory");
  const int rc = image_alloc(img, kind, width, height);
  if (rc != RMD_HIP_OK) {
    delete img;
    return rc;
  }
  *out = img;
  return RMD_HIP_OK;
}
int rmd_hip_image_destroy(rmd_hip_image_t* img) {
  if (!img) return RMD_HIP_OK;
  ScopedDevice dev(img->device);
  if (img->owns && img->data) (void)hipFree(img->data);  // destructors must not throw (device_image.cuh:124-132 does)
  delete img;
  return RMD_HIP_OK;
}
int rmd_hip_image_upload(rmd_hip_image_t* img, const void* host) {
  if (!img || !host) return fail(RMD_HIP_ERR_INVALID_ARG, "image_upload: null argument");
  TRY(image_settle(img));
  ScopedDevice dev(img->device);
  const size_t row = static_cast<size_t>(img->width) * kind_size(img->kind);
  HIP_TRY(hipMemcpy2D(img->data, img->pitch, host, row, row, img->height, hipMemcpyHostToDevice));
  return RMD_HIP_OK;
}
int rmd_hip_image_download(const rmd_hip_image_t* img, void* host) {
  if (!img || !host) return fail(RMD_HIP_ERR_INVALID_ARG, "image_download: null argument");
  TRY(image_settle(img));
  ScopedDevice dev(img->device);
  const size_t row = static_cast<size_t>(img->width) * kind_size(img->kind);
  HIP_TRY(hipMemcpy2D(host, row, img->data, img->pitch, row, img->height, hipMemcpyDeviceToHost));
  return RMD_HIP_OK;
}
int rmd_hip_image_zero(rmd_hip_image_t* img) {
  if (!img) return fail(RMD_HIP_ERR_INVALID_ARG, "image_zero: null image");
  TRY(image_settle(img));
  ScopedDevice dev(img->device);  // the null stream that is waited on below is the image's device's
  HIP_TRY(hipMemset(img->data, 0, img->pitch * img->height));
  HIP_TRY(hipStreamSynchronize(nullptr));
  return RMD_HIP_OK;
}
int rmd_hip_image_copy(rmd_hip_image_t* dst, const rmd_hip_image_t* src) {
  if (!dst || !src) return fail(RMD_HIP_ERR_INVALID_ARG, "image_copy: null argument");
  if (dst == src) return RMD_HIP_OK;
  if (dst->kind != src->kind || dst->width != src->width || dst->height != src->height)
    return fail(RMD_HIP_ERR_INVALID_ARG, "image_copy: shape mismatch");
  TRY(image_settle(src));
  TRY(image_settle(dst));
  ScopedDevice dev(dst->device);
  const size_t row = static_cast<size_t>(src->width) * kind_size(src->kind);
  HIP_TRY(hipMemcpy2D(dst->data, dst->pitch, src->data, src->pitch, row, src->height, hipMemcpyDeviceToDevice));
  return RMD_HIP_OK;
}
int rmd_hip_image_info(const rmd_hip_image_t* img, int* kind, int* width, int* height, size_t* pitch_bytes,
                       size_t* stride_elems, void** device_data) {
  if (!img) return fail(RMD_HIP_ERR_INVALID_ARG, "image_info: null image");
  if (kind) *kind = img->kind;
  if (width) *width = img->width;
  if (height) *height = img->height;
  if (pitch_bytes) *pitch_bytes = img->pitch;
  if (stride_elems) *stride_elems = img->stride;
  if (device_data) *device_data = img->data;
  return RMD_HIP_OK;
}

// ---- SeedMatrix -----------------------------------------------------------------------------
int rmd_hip_seeds_destroy(rmd_hip_seeds_t* s) {
  if (s && s->batch) return fail(RMD_HIP_ERR_INVALID_ARG,
      "seeds_destroy: this SeedMatrix belongs to a batch (rmd_hip_batch_destroy releases it)");
  return seeds_destroy_impl(s);
}
}  // extern "C"
int rmdh::seeds_destroy_impl(rmd_hip_seeds* s) {
  if (!s) return RMD_HIP_OK;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (auto& t : s->timers) t.destroy();
  if (s->copy_stream) (void)hipStreamSynchronize(s->copy_stream);
  if (s->copy_stream2) (void)hipStreamSynchronize(s->copy_stream2);
  ingest_release_engines(s);
  publish_release(s);
  if (s->ingest_profile && s->ingest_us[3] > 0) {
    fprintf(stderr, "[rmd_hip ingest] %.0f frames: wait for slot %.2f us, host copy %.2f us, submit %.2f us per frame; "
                    "longest wait %.0f us, %lu waits gave up (stream idle, word not reached)\n",
            s->ingest_us[3], s->ingest_us[0] / s->ingest_us[3], s->ingest_us[1] / s->ingest_us[3], s->ingest_us[2] / s->ingest_us[3],
                g_progress_max_wait_us,
            g_progress_timeouts);
    fprintf(stderr,
        "[rmd_hip ingest] longest single phase of an update() call: ring wait %.0f us (frame %llu), copy into the slot %.0f us (%llu), "
                    "copy-engine commands %.0f us (%llu), launches %.0f us (%llu)\n",
            s->ingest_max_us[0], s->ingest_max_at[0], s->ingest_max_us[1], s->ingest_max_at[1], s->ingest_max_us[2], s->ingest_max_at[2],
                s->ingest_max_us[3],
            s->ingest_max_at[3]);
    if (s->h_progress) {
      fprintf(stderr, "[rmd_hip ingest] staged frames: %lu on copy engines addressed directly (route %d), %lu on the copy stream\n",
              s->staged_by_engines, s->engine_route, s->staged_by_stream);
      fprintf(stderr, "[rmd_hip ingest] frames handed over <=0 / 1 / 2 / 3 / >=4 ahead of the newest setup kernel that had started: "
                      "%lu / %lu / %lu / %lu / %lu; "
                      "converted by their own setup kernel (not one step ahead) %u, of which the kernel waited for %u (%u polls)\n",
              s->ingest_lead[0], s->ingest_lead[1], s->ingest_lead[2], s->ingest_lead[3], s->ingest_lead[4], s->h_progress[2],
                  s->h_progress[3], s->h_progress[4]);
    }
  }
  for (int k = 0; k < rmd_hip_seeds::RING_MAX; ++k) {
    if (s->h_zc_u8[k]) (void)hipHostFree(s->h_zc_u8[k]);
    if (s->h_zc_f32[k]) (void)hipHostFree(s->h_zc_f32[k]);
    if (s->d_zc_u8[k]) (void)hipFree(s->d_zc_u8[k]);
    if (s->d_zc_f32[k]) (void)hipFree(s->d_zc_f32[k]);
  }
  for (int k = 0; k < rmd_hip_seeds::SLOTS; ++k) {
    if (s->h_u8[k]) (void)hipHostFree(s->h_u8[k]);
    if (s->h_f32[k]) (void)hipHostFree(s->h_f32[k]);
    if (s->d_u8[k]) (void)hipFree(s->d_u8[k]);
    if (s->staged[k]) (void)hipEventDestroy(s->staged[k]);
    if (s->frame_done[k]) (void)hipEventDestroy(s->frame_done[k]);
    if (k > 0 && s->cur_planes[k]) (void)hipFree(s->cur_planes[k]);  // [0] belongs to planes[]
  }
  if (s->cur_planes[0]) s->planes[RMD_HIP_PLANE_CURR_IMG].data = s->cur_planes[0];
  if (s->copy_stream && !s->batch) (void)hipStreamDestroy(s->copy_stream);
  if (s->copy_stream2) (void)hipStreamDestroy(s->copy_stream2);
  if (s->region_start) (void)hipEventDestroy(s->region_start);
  if (s->region_stop) (void)hipEventDestroy(s->region_stop);
  for (auto& pl : s->planes)
    if (pl.owns && pl.data) (void)hipFree(pl.data);
  s->matcher_ws.release();
  if (s->d_undist_map1) (void)hipFree(s->d_undist_map1);
  if (s->d_undist_map2) (void)hipFree(s->d_undist_map2);
  if (s->d_bgr) (void)hipFree(s->d_bgr);
  if (s->h_bgr) (void)hipHostFree(s->h_bgr);
  if (s->d_pc_counts) (void)hipFree(s->d_pc_counts);
  if (s->h_pc_points) (void)hipHostFree(s->h_pc_points);
  if (s->h_pc_total) (void)hipHostFree(s->h_pc_total);
  if (s->d_scalars) (void)hipFree(s->d_scalars);
  if (s->h_scalars) (void)hipHostFree(s->h_scalars);
  if (s->h_progress) (void)hipHostFree(s->h_progress);
  if (s->h_seq) (void)hipHostFree(s->h_seq);
  if (s->d_zc_flag) (void)hipFree(s->d_zc_flag);
  if (s->h_submitted) (void)hipHostFree(s->h_submitted);
  if (s->d_ahead) (void)hipFree(s->d_ahead);
  if (s->stream && !s->batch) (void)hipStreamDestroy(s->stream);
  delete s;
  return RMD_HIP_OK;
}

extern "C" {

int rmd_hip_seeds_create(int width, int height, float fx, float fy, float cx, float cy, int patch_side, int max_extent,
                         rmd_hip_seeds_t** out) {
  return seeds_create_impl(width, height, fx, fy, cx, cy, patch_side, max_extent, nullptr, 0, out);
}

}  // extern "C"
// batch != null: member `seq` of that batch -- the batch's streams and update workspace instead of its own
int rmdh::seeds_create_impl(int width, int height, float fx, float fy, float cx, float cy, int patch_side, int max_extent,
    rmd_hip_batch* batch, int seq,
                            rmd_hip_seeds** out) {
  if (!out) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_create: null output");
  *out = nullptr;
  if (width <= 0 || height <= 0) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_create: bad size %dx%d", width, height);
  if (!side_supported(patch_side))
    return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_create: unsupported patch side %d (3, 5, 7, 9)", patch_side);
  if (max_extent <= 0 || max_extent > rmdk::MAX_EXTENT_LIMIT)
    return fail(RMD_HIP_ERR_INVALID_ARG,
        "seeds_create: max_extent %d outside (0, %d] (a seed's search steps are numbered in 8 bits: 255 steps of 0.7 pixels)", max_extent,
                rmdk::MAX_EXTENT_LIMIT);
  int ndev = 0;
  TRY(rmd_hip_device_count(&ndev));
  rmd_hip_seeds* s = new (std::nothrow) rmd_hip_seeds();
  if (!s) return fail(RMD_HIP_ERR_RUNTIME, "seeds_create: out of host memory");
  s->width = width; s->height = height; s->patch_side = patch_side;
  (void)hipGetDevice(&s->device);
  s->batch = batch; s->batch_index = seq;
  rmd_hip_batch::Group* grp = batch ? &batch->group_of(seq) : nullptr;
  s->seq = batch ? seq - grp->first : 0;
  s->mws = batch ? &grp->ws : &s->matcher_ws;
  auto bail = [&](int rc) { seeds_destroy_impl(s); return rc; };
  if (batch) { s->stream = grp->stream; s->copy_stream = batch->copy_stream; }
  else if (create_stream(&s->stream, 0) != hipSuccess)
    return bail(fail(RMD_HIP_ERR_RUNTIME, "seeds_create: hipStreamCreate failed"));
  for (int p = 0; p < RMD_HIP_NUM_PLANES; ++p) {
    const int kind = p == RMD_HIP_PLANE_CONVERGENCE ? RMD_HIP_KIND_I32
                     : p == RMD_HIP_PLANE_EPIPOLAR_MATCHES ? RMD_HIP_KIND_F32X2 : RMD_HIP_KIND_F32;
    const int rc = image_alloc(&s->planes[p], kind, width, height);
    if (rc != RMD_HIP_OK) return bail(rc);
    s->planes[p].owner_stream = s->stream;
    s->planes[p].owner_seeds = s;
  }
  if (hipMalloc(reinterpret_cast<void**>(&s->d_scalars), 17 * sizeof(unsigned long long)) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&s->h_scalars), 17 * sizeof(unsigned long long)) != hipSuccess)
    return bail(fail(RMD_HIP_ERR_RUNTIME, "seeds_create: scalar buffers"));
  if (hipMemset(s->d_scalars, 0, 17 * sizeof(unsigned long long)) != hipSuccess) return bail(fail(RMD_HIP_ERR_RUNTIME,
      "seeds_create: scalar buffers"));
  memset(s->h_scalars, 0, 17 * sizeof(unsigned long long));
  rmdk::SeedParams& P = s->P;
  memset(&P, 0, sizeof(P));
  P.w = width; P.h = height;
  P.stride = static_cast<int>(s->planes[RMD_HIP_PLANE_MU].stride);
  P.stride2 = static_cast<int>(s->planes[RMD_HIP_PLANE_EPIPOLAR_MATCHES].stride);
  P.ref = static_cast<const float*>(s->planes[RMD_HIP_PLANE_REF_IMG].data);
  P.cur = static_cast<const float*>(s->planes[RMD_HIP_PLANE_CURR_IMG].data);
  P.cur_stride = P.stride;
  P.sum_templ = static_cast<float*>(s->planes[RMD_HIP_PLANE_SUM_TEMPL].data);
  P.denom = static_cast<float*>(s->planes[RMD_HIP_PLANE_CONST_TEMPL_DENOM].data);
  P.mu = static_cast<float*>(s->planes[RMD_HIP_PLANE_MU].data);
  P.sigma_sq = static_cast<float*>(s->planes[RMD_HIP_PLANE_SIGMA_SQ].data);
  P.a = static_cast<float*>(s->planes[RMD_HIP_PLANE_A].data);
  P.b = static_cast<float*>(s->planes[RMD_HIP_PLANE_B].data);
  P.conv = static_cast<int*>(s->planes[RMD_HIP_PLANE_CONVERGENCE].data);
  P.match = static_cast<float2*>(s->planes[RMD_HIP_PLANE_EPIPOLAR_MATCHES].data);
  P.cam = rmdk::Cam{fx, fy, cx, cy};
  P.one_pix_angle = atan2f(1.0f, 2.0f * fx) * 2.0f;  // pinhole_camera.cuh:56-59
  P.max_extent = static_cast<float>(max_extent);
  if (!batch && s->matcher_ws.allocate(width, height, P.stride, 1, max_extent) != 0) return bail(fail(RMD_HIP_ERR_RUNTIME,
      "seeds_create: update workspace"));
  if (batch && (grp->ws.stride != P.stride || grp->ws.tiles_x != (width + rmdk::TILE_W - 1) / rmdk::TILE_W
      || grp->ws.tiles_y != (height + rmdk::TILE_H - 1) / rmdk::TILE_H))
    return bail(fail(RMD_HIP_ERR_RUNTIME, "seeds_create: the batch's workspace has another geometry"));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, s->device) == hipSuccess && prop.multiProcessorCount > 0) s->num_cus = prop.multiProcessorCount;
  int lds = 0;
  if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor,
      s->device) == hipSuccess && lds > 0) s->mws->lds_bytes = lds;
  // all fills done
  if (hipDeviceSynchronize() != hipSuccess) return bail(fail(RMD_HIP_ERR_RUNTIME, "seeds_create: device synchronisation failed"));
  // the copy stream is created right next to the compute stream.  A batch member's updates go through the BATCH's staging buffers and
  // progress words: it gets its own (the staging slots of set_reference*) lazily, at its first host reference frame (ingest_reference)
  if (!batch && ingest_init(s) != RMD_HIP_OK) return bail(RMD_HIP_ERR_RUNTIME);
  *out = s;
  return RMD_HIP_OK;
}

extern "C" {

int rmd_hip_seeds_set_reference_device(rmd_hip_seeds_t* s, const float* dev_img, size_t stride_elems,
                                       const float* T_curr_world, float min_depth, float max_depth) {
  if (!s || !dev_img || !T_curr_world) return fail(RMD_HIP_ERR_INVALID_ARG, "set_reference_device: null argument");
  if (stride_elems < static_cast<size_t>(s->width)) return fail(RMD_HIP_ERR_INVALID_ARG, "set_reference_device: stride < width");
  TRY(seeds_bind_device(s));
  const rmd_hip_image& im = s->planes[RMD_HIP_PLANE_REF_IMG];
  const size_t row = static_cast<size_t>(s->width) * 4;
  HIP_TRY(hipMemcpy2DAsync(im.data, im.pitch, dev_img, stride_elems * 4, row, s->height, hipMemcpyDeviceToDevice, s->stream));
  return seeds_after_reference(s, T_curr_world, min_depth, max_depth);
}

int rmd_hip_seeds_update_device(rmd_hip_seeds_t* s, const float* dev_img, size_t stride_elems, const float* T_curr_world) {
  if (!s || !dev_img || !T_curr_world) return fail(RMD_HIP_ERR_INVALID_ARG, "update_device: null argument");
  if (s->batch) return fail(RMD_HIP_ERR_INVALID_ARG,
      "update_device: this SeedMatrix is a member of a batch; its updates are issued with rmd_hip_batch_update*");
  if (!s->has_reference) return fail(RMD_HIP_ERR_NOT_READY, "update_device: setReferenceImage has not been called");
  if (stride_elems < static_cast<size_t>(s->width)) return fail(RMD_HIP_ERR_INVALID_ARG, "update_device: stride < width");
  TRY(seeds_bind_device(s));
  // zero copy: the kernels read the caller's buffer in place (see the header for the lifetime rule)
  s->P.cur = dev_img;
  s->P.cur_stride = static_cast<int>(stride_elems);
  return seeds_after_frame(s, T_curr_world);
}

int rmd_hip_seeds_download(const rmd_hip_seeds_t* s, int plane, void* host_dst) {
  if (!s || !host_dst) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_download: null argument");
  if (plane < 0 || plane >= RMD_HIP_NUM_PLANES) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_download: bad plane %d", plane);
  TRY(seeds_bind_device(s));
  TRY(seeds_sync(s));
  const rmd_hip_image& im = s->planes[plane];
  const size_t row = static_cast<size_t>(im.width) * kind_size(im.kind);
  HIP_TRY(hipMemcpy2D(host_dst, row, im.data, im.pitch, row, im.height, hipMemcpyDeviceToHost));
  return RMD_HIP_OK;
}

int rmd_hip_seeds_upload(rmd_hip_seeds_t* s, int plane, const float* host_src) {
  if (!s || !host_src) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_upload: null argument");
  if (plane < RMD_HIP_PLANE_MU || plane > RMD_HIP_PLANE_B) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_upload: plane %d", plane);
  TRY(seeds_bind_device(s));
  TRY(seeds_sync(s));
  s->async_count_valid = false;
  const rmd_hip_image& im = s->planes[plane];
  const size_t row = static_cast<size_t>(im.width) * 4;
  HIP_TRY(hipMemcpy2D(im.data, im.pitch, host_src, row, row, im.height, hipMemcpyHostToDevice));
  return RMD_HIP_OK;
}

int rmd_hip_seeds_plane(const rmd_hip_seeds_t* s, int plane, const rmd_hip_image_t** view) {
  if (!s || !view) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_plane: null argument");
  if (plane < 0 || plane >= RMD_HIP_NUM_PLANES) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_plane: bad plane %d", plane);
  *view = &s->planes[plane];
  return RMD_HIP_OK;
}

int rmd_hip_seeds_converged_count(const rmd_hip_seeds_t* s, size_t* count) {
  if (!s || !count) return fail(RMD_HIP_ERR_INVALID_ARG, "converged_count: null argument");
  TRY(seeds_bind_device(s));
  rmd_hip_seeds* m = const_cast<rmd_hip_seeds*>(s);
  if (m->async_count_valid) {
    // The latest thing that happened to the seed state is an update of the tile pipeline: its setup kernel -- which IS seed_check
    // (seed_matrix.cu:139-142) -- counted the seeds it found CONVERGED and the search kernel's last workgroup mirrored the sum,
    // stamped with the update's number, to pinned memory.  Nothing later in the update changes that count (the matcher only turns
    // UPDATE into NO_MATCH), so it is what countEqual(convergence, CONVERGED) returns after the whole update -- without waiting
    // for the search, without a kernel and without flushing the deferred finalisation.
    volatile unsigned long long* word = m->mws->h_conv + m->seq;
    const unsigned int want = m->async_number;
    const double t0 = host_now_us();
    bool synced = false;
    for (;;) {
      const unsigned long long v = *word;
      if (static_cast<unsigned int>(v >> 32) == want) {
        *count = static_cast<size_t>(v & 0xffffffffull);
        return ingest_error_check(m->batch ? m->batch->group_of(m->batch_index).h_progress : m->h_progress);
      }
      if (synced) break;  // cannot happen; fall through to the counting kernel
      if (host_now_us() - t0 > 2000.0) {
        HIP_TRY(hipStreamSynchronize(m->stream));
        synced = true;
        continue;
      }
      cpu_relax();
    }
  }
  TRY(seeds_flush(m));
  HIP_TRY(hipMemsetAsync(m->d_scalars, 0, sizeof(unsigned long long), m->stream));
  {
    ScopedStage st(m->opt_timing ? &m->timers[RMD_HIP_STAGE_COUNT] : nullptr, m->stream);
    launch_count_eq(static_cast<const int*>(s->planes[RMD_HIP_PLANE_CONVERGENCE].data), s->width, s->height, s->P.stride,
                    static_cast<int>(RMD_HIP_STATE_CONVERGED), m->d_scalars, m->stream);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipMemcpyAsync(m->h_scalars, m->d_scalars, sizeof(unsigned long long), hipMemcpyDeviceToHost, m->stream));
  TRY(seeds_sync(s));
  *count = static_cast<size_t>(m->h_scalars[0]);
  return RMD_HIP_OK;
}

int rmd_hip_seeds_dist_from_ref(const rmd_hip_seeds_t* s, float* dist) {
  if (!s || !dist) return fail(RMD_HIP_ERR_INVALID_ARG, "dist_from_ref: null argument");
  *dist = s->dist_from_ref;
  return RMD_HIP_OK;
}

int rmd_hip_seeds_staged_frames(const rmd_hip_seeds_t* s, unsigned long long counts[2]) {
  if (!s || !counts) return fail(RMD_HIP_ERR_INVALID_ARG, "staged_frames: null argument");
  counts[0] = s->staged_by_engines;
  counts[1] = s->staged_by_stream;
  return RMD_HIP_OK;
}

int rmd_hip_seeds_sync(const rmd_hip_seeds_t* s) {
  if (!s) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_sync: null handle");
  TRY(seeds_bind_device(s));
  return seeds_sync(s);
}

int rmd_hip_seeds_set_option(rmd_hip_seeds_t* s, int option, int value) {
  if (!s) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: null handle");
  // a batch member's updates are launched by the batch (batch_launch), which knows nothing of per-member statistics, per-update event
  // pairs, eager finalisation or unit targets: accepting such a setting and then ignoring it would leave last_stats / timing stale
  if (s->batch && ((option == RMD_HIP_OPT_COLLECT_STATS && value != 0) || (option == RMD_HIP_OPT_TIMING && value != 0) ||
                   (option == RMD_HIP_OPT_LAZY_FINALIZE && value == 0) || option == RMD_HIP_OPT_UNIT_TARGET))
    return fail(RMD_HIP_ERR_INVALID_ARG,
        "set_option: option %d has no effect on a member of a batch (rmd_hip_batch_set_option sets the batch's timing and unit "
                                         "target)", option);
  switch (option) {
    case RMD_HIP_OPT_MATCHER:
      if (value != 0 && value != 3)
        return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: matcher %d (0 = per-pixel kernel, 3 = tile pipeline)", value);
      if (s->batch && value != 3) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: members of a batch use the tile pipeline");
      if (value != s->opt_matcher) {  // each matcher keeps its own per-frame state: settle the old one first
        TRY(seeds_bind_device(s));
        TRY(seeds_sync(s));
        s->mws->frame = 0;
        HIP_TRY(hipMemsetAsync(s->mws->d_shards, 0, 3 * rmdk::UNIT_SHARDS * sizeof(unsigned long long), s->stream));
        s->async_count_valid = false;
      }
      s->opt_matcher = value;
      return RMD_HIP_OK;
    case RMD_HIP_OPT_UNIT_TARGET:
      if (value < 1 || value > 4) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: unit target %d outside 1..4", value);
      s->opt_unit_target = value;
      return RMD_HIP_OK;
    case RMD_HIP_OPT_TIMING:
      if (value < 0 || value > 2) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: timing mode %d", value);
      s->opt_timing = value;
      return RMD_HIP_OK;
    case RMD_HIP_OPT_LAZY_FINALIZE:
      s->opt_lazy = value != 0;
      return RMD_HIP_OK;
    case RMD_HIP_OPT_INJECT_FAULT:
      if (value != 0 && value != 1) return fail(RMD_HIP_ERR_INVALID_ARG,
          "set_option: fault %d (1 = withhold the arrival flag of the next staged host frame)", value);
      if (s->batch) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: fault injection is for plain SeedMatrix handles");
      s->inject_withhold_flag = value == 1;
      return RMD_HIP_OK;
    case RMD_HIP_OPT_COLLECT_STATS:
      s->opt_stats = value == 2 ? 2 : (value != 0);
      if (s->opt_stats == 2) {  // (re)start a timeline: zeroed buffer, frame counter 0
        if (s->batch) return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: timeline probes are for plain SeedMatrix handles");
        TRY(seeds_bind_device(s));
        rmdk::MatcherWorkspace& ws = s->matcher_ws;
        const size_t wbytes = ws.wg_trace_slice_u64() * rmdk::FR_TRACE_FRAMES * sizeof(unsigned long long);
        if (!ws.d_wg_trace) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ws.d_wg_trace), wbytes));
        HIP_TRY(hipStreamSynchronize(s->stream));
        HIP_TRY(hipMemset(ws.d_wg_trace, 0, wbytes));
        HIP_TRY(hipDeviceSynchronize());
        s->trace_frame = 0;
      }
      return RMD_HIP_OK;
    default: return fail(RMD_HIP_ERR_INVALID_ARG, "set_option: unknown option %d", option);
  }
}

int rmd_hip_seeds_timing(const rmd_hip_seeds_t* s, int stage, double* total_ms, long* launches) {
  if (!s || stage < 0 || stage >= RMD_HIP_NUM_SEED_STAGES) return fail(RMD_HIP_ERR_INVALID_ARG, "seeds_timing: bad argument");
  TRY(seeds_bind_device(s));
  if (s->opt_timing == 2 && stage == RMD_HIP_STAGE_UPDATE) {
    // region mode: ONE event pair around everything queued since timing_reset (no per-launch markers in the stream)
    rmd_hip_seeds* m = const_cast<rmd_hip_seeds*>(s);
    if (!m->region_start) return fail(RMD_HIP_ERR_NOT_READY, "seeds_timing: call timing_reset first");
    if (!m->region_stop) HIP_TRY(hipEventCreate(&m->region_stop));
    TRY(seeds_flush(m));  // the deferred finalisation of the last update belongs to the region
    HIP_TRY(hipEventRecord(m->region_stop, m->stream));
    HIP_TRY(hipEventSynchronize(m->region_stop));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, m->region_start, m->region_stop));
    if (total_ms) *total_ms = ms;
    if (launches) *launches = m->region_updates;
    return RMD_HIP_OK;
  }
  TRY(seeds_sync(s));
  if (total_ms) *total_ms = s->timers[stage].total_ms;
  if (launches) *launches = s->timers[stage].launches;
  return RMD_HIP_OK;
}
int rmd_hip_seeds_timing_reset(rmd_hip_seeds_t* s) {
  if (!s) return fail(RMD_HIP_ERR_INVALID_ARG, "timing_reset: null handle");
  TRY(seeds_bind_device(s));
  TRY(seeds_sync(s));
  for (auto& t : s->timers) t.reset();
  if (s->opt_timing == 2) {
    if (!s->region_start) HIP_TRY(hipEventCreate(&s->region_start));
    HIP_TRY(hipEventRecord(s->region_start, s->stream));
    s->region_updates = 0;
  }
  return RMD_HIP_OK;
}
int rmd_hip_seeds_last_stats(const rmd_hip_seeds_t* s, long long* out3) {
  if (!s || !out3) return fail(RMD_HIP_ERR_INVALID_ARG, "last_stats: null argument");
  TRY(seeds_bind_device(s));
  TRY(seeds_sync(s));
  for (int k = 0; k < 3; ++k) out3[k] = s->last_stats[k];
  return RMD_HIP_OK;
}
int rmd_hip_seeds_last_diagnostics(const rmd_hip_seeds_t* s, long long* out16) {
  if (!s || !out16) return fail(RMD_HIP_ERR_INVALID_ARG, "last_diagnostics: null argument");
  TRY(seeds_bind_device(s));
  TRY(seeds_sync(s));
  for (int k = 0; k < 16; ++k) out16[k] = s->last_stats[k];
  return RMD_HIP_OK;
}

int rmd_hip_seeds_trace_download(rmd_hip_seeds_t* s, int frame, unsigned long long* out, size_t capacity, size_t* written) {
  if (!s || !out || !written) return fail(RMD_HIP_ERR_INVALID_ARG, "trace_download: null argument");
  const rmdk::MatcherWorkspace& ws = s->matcher_ws;
  if (!ws.d_wg_trace) return fail(RMD_HIP_ERR_NOT_READY, "trace_download: set RMD_HIP_OPT_COLLECT_STATS to 2 first");
  if (frame < 0 || frame >= s->trace_frame || frame < s->trace_frame - rmdk::FR_TRACE_FRAMES)
    return fail(RMD_HIP_ERR_INVALID_ARG, "trace_download: frame not in the buffer");
  if (s->opt_matcher == 3) {  // tile pipeline: FR_TRACE_WORDS words per tile / search workgroup
    const size_t fn = ws.wg_trace_slice_u64();
    if (capacity < fn) return fail(RMD_HIP_ERR_INVALID_ARG, "trace_download: buffer too small (%zu words needed)", fn);
    TRY(seeds_bind_device(s));
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(out, ws.d_wg_trace + static_cast<size_t>(frame % rmdk::FR_TRACE_FRAMES) * fn, fn * sizeof(unsigned long long),
        hipMemcpyDeviceToHost));
    *written = fn;
    return RMD_HIP_OK;
  }
  return fail(RMD_HIP_ERR_INVALID_ARG, "trace_download: timeline probes exist for the tile pipeline (matcher 3) only in this build");
}

}  // extern "C"

