/*
 * librmd_hip.so -- C ABI of the MI355X-native REMODE depth-filter path.
 *
 * This is the drop-in boundary.  The reference exposes the path as C++ classes compiled
 * by nvcc into the static library `rpg_open_remode_cuda` (CMakeLists.txt:92-107):
 *     rmd::SeedMatrix        include/rmd/seed_matrix.cuh:45-109,  src/seed_matrix.cu
 *     rmd::DepthmapDenoiser  include/rmd/depthmap_denoiser.cuh:27-54, src/depthmap_denoiser.cu
 *     rmd::ImageReducer<T>   include/rmd/reduction.cuh:26-62,    src/reduction.cu
 *     rmd::DeviceImage<T>    include/rmd/device_image.cuh:34-180
 *     rmd::checkCudaDevice   include/rmd/check_cuda_device.cuh:24, src/check_cuda_device.cu
 * Each entry point below replaces one method of those classes (cited per function).  The
 * C++ headers under include/rmd/ in THIS repository re-create the classes, name for name,
 * as thin inline wrappers over this ABI, so the reference's host code (src/depthmap.cpp,
 * src/depthmap_node.cpp, the gtest sources) compiles against them unchanged -- see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns RMD_HIP_OK (0) or a negative RMD_HIP_ERR_*; a description of the
 *     last failure on the calling thread is available from rmd_hip_last_error();
 *   - handles are opaque, own their device memory and a HIP stream, are bound to the device
 *     that was current when they were created, and may be used from one thread at a time;
 *     distinct handles are fully independent (no process-global state, unlike the reference's
 *     texture references / __constant__ symbols, texture_memory.cuh:27-42);
 *   - host images are contiguous row-major W x H, borrowed for the duration of the call;
 *   - poses cross as 12 floats, the row-major 3x4 [R|t] of rmd::SE3<float>::data
 *     (se3.cuh:141, matrix.cuh:34);
 *   - `update` may leave device work in flight on return exactly like the reference
 *     (seed_matrix.cu:155-157); every download / count / denoise synchronises first.
 */
#ifndef RMD_HIP_H
#define RMD_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RMD_HIP_OK 0
#define RMD_HIP_ERR_INVALID_ARG (-1) /* null handle, bad size, unsupported patch side, ... */
#define RMD_HIP_ERR_RUNTIME (-2)     /* a HIP runtime call failed (rmd::CudaException in the reference) */
#define RMD_HIP_ERR_NOT_READY (-3)   /* e.g. denoise() before set_large_sigma_sq(), update() before set_reference() */
#define RMD_HIP_ERR_NO_DEVICE (-4)   /* no usable gfx950 device */

/* rmd::ConvergenceStates (seed_matrix.cuh:33-41) */
#define RMD_HIP_STATE_UPDATE 0
#define RMD_HIP_STATE_CONVERGED 1
#define RMD_HIP_STATE_BORDER 2
#define RMD_HIP_STATE_DIVERGED 3
#define RMD_HIP_STATE_NO_MATCH 4
#define RMD_HIP_STATE_NOT_VISIBLE 5

/* planes of a SeedMatrix (seed_matrix.cuh:89-97) */
#define RMD_HIP_PLANE_MU 0               /* f32  depth estimate                         */
#define RMD_HIP_PLANE_SIGMA_SQ 1         /* f32  variance of the depth estimate         */
#define RMD_HIP_PLANE_A 2                /* f32  Beta a                                 */
#define RMD_HIP_PLANE_B 3                /* f32  Beta b                                 */
#define RMD_HIP_PLANE_CONVERGENCE 4      /* i32  RMD_HIP_STATE_*                        */
#define RMD_HIP_PLANE_SUM_TEMPL 5        /* f32  NCC template sum                       */
#define RMD_HIP_PLANE_CONST_TEMPL_DENOM 6 /* f32 NCC template denominator              */
#define RMD_HIP_PLANE_EPIPOLAR_MATCHES 7 /* f32x2 matched pixel in the current frame   */
#define RMD_HIP_PLANE_REF_IMG 8          /* f32  reference image                        */
#define RMD_HIP_PLANE_CURR_IMG 9         /* f32  current image                          */
#define RMD_HIP_NUM_PLANES 10

/* element kinds of rmd::DeviceImage<T> */
#define RMD_HIP_KIND_F32 0
#define RMD_HIP_KIND_I32 1
#define RMD_HIP_KIND_F32X2 2

typedef struct rmd_hip_image rmd_hip_image_t;
typedef struct rmd_hip_seeds rmd_hip_seeds_t;
typedef struct rmd_hip_denoiser rmd_hip_denoiser_t;
typedef struct rmd_hip_batch rmd_hip_batch_t;

/* ---- process-wide settings of the host side ------------------------------------------------
 * A/B and diagnostics switches; none changes results (tests/test_host_frame_modes.py, tests/test_full_speed.py run the frame-path modes
 * against the resident path bit for bit).  Defaults are what was measured to be fastest.  A handle picks the values
 * up when it is created.  The environment is read in ONE place (rmdh::tunables(), csrc/rmd_capi.hip), once, at the library's first
 * use: RMD_HIP_<NAME> = value presets tunable <NAME>.  (The Python loader additionally honours RMD_HIP_LIB = path of another build of
 * this library, tools/ab_make.sh.) */
#define RMD_HIP_TUNE_HOST_FRAMES 0    /* how a frame handed over in host memory reaches the current-image plane: 0 staged (a copy engine fills a staging
                                         buffer in HBM), 1 staged_ahead (... and the frame is converted during the previous update's search kernel), 2 inplace
                                         (the kernels read the pinned ring themselves), 3 inplace_ahead; -1 (default) = staged for a SeedMatrix (staged_ahead
                                         when RMD_HIP_TUNE_COPY_ENGINES is 0); a batch: staged (on one copy engine, eight staging buffers deep) while a step
                                         is at most 5 MB and the engines can be addressed, else inplace; frames with lens undistortion are always staged.
                                         Environment: the names or the numbers */
#define RMD_HIP_TUNE_BATCH_GROUPS 1   /* stream groups of a batch, 1..4; 0 (default) = min(n, 3) */
#define RMD_HIP_TUNE_AHEAD_WGS 2      /* workgroups that convert a host frame one step ahead (128) */
#define RMD_HIP_TUNE_PACK_BACKOFF 3   /* after a float frame that is not made of 8-bit levels the next n frames are not examined (15; tests use 0) */
#define RMD_HIP_TUNE_FLOAT_AS_BYTES 4 /* 1 (default): float frames of 8-bit levels travel as bytes; 0: always as floats */
#define RMD_HIP_TUNE_COPY_THREADS 5   /* host threads that examine a large float frame / copy the frames of a batch step into the pinned ring, 1..16 (4); one
                                         large frame is copied by two threads whatever this says (by one if it is 1); read when the first large frame arrives */
#define RMD_HIP_TUNE_FUSED_INGEST 6   /* 1 (default): host frames are converted by the update's own kernels; 0: upload + conversion kernel on the copy stream */
#define RMD_HIP_TUNE_INGEST_PROFILE 7 /* 1: a handle prints the host time per frame it spent waiting / copying / submitting when it is destroyed */
#define RMD_HIP_TUNE_HOST_WAIT 8      /* how update() waits for a free slot of its pinned frame ring (the device is up to three frames behind the caller): 1
                                         (default) = spin for 2 us, then sleep in steps of ~15 us (the calling thread's timer slack is lowered to 2 us for the duration of
                                         the wait and put back before update() returns; if it cannot be changed the steps are coarser);
                                         0 = spin only (one core per handle that is fed host frames at full speed) */
#define RMD_HIP_TUNE_RING_DEPTH 9     /* frames (a batch: steps) that may be in flight between update() and the setup kernel that consumes them = slots of the pinned
                                         frame ring, 3..8; 0 (default) = the library's choice: 4 for a SeedMatrix, 3 for a batch */
#define RMD_HIP_TUNE_COPY_STREAMS 10   /* copy streams a SeedMatrix spreads its staged host frames over, 1..2 (1; 2 is 1.5 % faster at 640x480 and stalls one update() in ~5 000 for 9 ms) */
#define RMD_HIP_TUNE_COPY_ENGINES 11  /* how a SeedMatrix's staged host frames travel: 0 = hipMemcpyAsync on the handle's copy stream (one engine for every
                                         host-to-device copy of the process; the arrival flag a second, 64-KB command behind the frame); 1..4 = on that many
                                         copy engines addressed directly (csrc/rmd_engines.hip), consecutive frames on consecutive engines, each frame's
                                         flag a 4-byte dependent copy behind it on its own engine; -1 (default) = two engines, three for frames of 1.5
                                         Mpixel or more.  Falls back to 0 by itself where the engines cannot be addressed (rmd_hip_seeds_staged_frames
                                         tells) */
#define RMD_HIP_NUM_TUNABLES 12
int rmd_hip_set_tunable(int tunable, int value);
int rmd_hip_get_tunable(int tunable, int* value);

/* ---- library ------------------------------------------------------------------------ */
const char* rmd_hip_last_error(void);
int rmd_hip_version(void);

/* ---- device selection: rmd::checkCudaDevice (check_cuda_device.cu:23-117) ------------- */
int rmd_hip_device_count(int* count);
int rmd_hip_set_device(int device_id);
int rmd_hip_device_name(int device_id, char* buf, size_t buf_len);

/* ---- rmd::DeviceImage<T> (device_image.cuh) ------------------------------------------- */
/* ctor :38-65 (pitched allocation) / dtor :124-132 */
int rmd_hip_image_create(int kind, int width, int height, rmd_hip_image_t** out);
int rmd_hip_image_destroy(rmd_hip_image_t* img);
/* setDevData :93-105 / getDevData :109-121 / zero :141-151 / operator= :154-171 */
int rmd_hip_image_upload(rmd_hip_image_t* img, const void* host_row_major);
int rmd_hip_image_download(const rmd_hip_image_t* img, void* host_row_major);
int rmd_hip_image_zero(rmd_hip_image_t* img);
int rmd_hip_image_copy(rmd_hip_image_t* dst, const rmd_hip_image_t* src);
/* public fields width/height/pitch/stride/data :174-179 (data is a device pointer) */
int rmd_hip_image_info(const rmd_hip_image_t* img, int* kind, int* width, int* height, size_t* pitch_bytes,
                       size_t* stride_elems, void** device_data);

/* ---- rmd::SeedMatrix (seed_matrix.cu) -------------------------------------------------- */
/* ctor :28-80.  patch_side = RMD_CORR_PATCH_SIDE (3, 5, 7 or 9; CMakeLists.txt:50-51: "must be odd", default 5),
 * max_extent = RMD_MAX_EXTENT_EPIPOLAR_SEARCH in pixels (CMakeLists.txt:52-53, default 100), 1..178: both are compile-time constants of the
 * reference and run-time arguments here.  Bounds: the NCC block is instantiated for the four sides above; a seed's search steps are numbered in
 * 8 bits (255 steps of 0.7 pixels = 178 pixels).  Anything else is RMD_HIP_ERR_INVALID_ARG at creation. */
int rmd_hip_seeds_create(int width, int height, float fx, float fy, float cx, float cy, int patch_side, int max_extent,
                         rmd_hip_seeds_t** out);
int rmd_hip_seeds_destroy(rmd_hip_seeds_t* s);
/* setReferenceImage :87-118 */
int rmd_hip_seeds_set_reference(rmd_hip_seeds_t* s, const float* host_img, const float* T_curr_world, float min_depth,
                                float max_depth);
/* update :120-158 (check -> epipolar match -> triangulate + fuse).  host_img (contiguous W x H floats, pageable memory is fine) has
 * been copied when the call returns, like the reference's synchronous cudaMemcpy2D (:128); the device work is in flight (up to four
 * frames inside the library), the next synchronising call (download, converged count, denoise, sync) waits for it.  A frame whose every
 * pixel has the bit pattern of (float)k * (1.0f / 255.0f), k = 0..255 -- what Depthmap::inputImage hands over (depthmap.cpp:105) --
 * is sent to the device as bytes and multiplied there by the same constant: the current image is the caller's image bit for bit either
 * way, the byte form just costs a quarter of the transfer. */
int rmd_hip_seeds_update(rmd_hip_seeds_t* s, const float* host_img, const float* T_curr_world);
/* same two calls for a frame that is already resident in device memory (row stride in elements).
 * set_reference_device copies the frame into the handle's own plane.  update_device reads the caller's buffer IN
 * PLACE (zero copy): it must stay valid and unmodified until the next call on this handle that synchronises
 * (download / converged_count / sync / denoise) or until the next update has been issued and synchronised. */
int rmd_hip_seeds_set_reference_device(rmd_hip_seeds_t* s, const float* dev_img, size_t stride_elems,
                                       const float* T_curr_world, float min_depth, float max_depth);
int rmd_hip_seeds_update_device(rmd_hip_seeds_t* s, const float* dev_img, size_t stride_elems,
                                const float* T_curr_world);
/* same two calls for an 8-bit gray frame (contiguous W x H bytes): what rmd::Depthmap::inputImage does on the host
 * (src/depthmap.cpp:95-106: cv::remap through the undistortion maps if rmd_hip_seeds_init_undistortion_map was called, then
 * cv::Mat::convertTo(CV_32F, 1.0f/255.0f)) happens on the device, bit for bit.  A quarter of the bytes of a float frame cross the bus;
 * update_u8 returns as soon as the frame has been copied into one of four pinned buffers; copy engines bring it into HBM and the update's own
 * first kernel converts it (no extra launch, no synchronisation between the upload and the compute queue): 97-99 % of the rate of frames that
 * are already resident. */
int rmd_hip_seeds_set_reference_u8(rmd_hip_seeds_t* s, const unsigned char* host_gray, const float* T_curr_world,
                                   float min_depth, float max_depth);
int rmd_hip_seeds_update_u8(rmd_hip_seeds_t* s, const unsigned char* host_gray, const float* T_curr_world);
/* Not in the reference: 8-bit frames the caller KEEPS in pinned host memory (rmd_hip_host_alloc, hipHostMalloc) -- the producer writes them there,
 * e.g. a capture driver or a decoder -- are read by the copy engine where they lie: update() without the copy into the library's ring (at
 * 1920x1080 the 2 MB per frame that keep one to two host cores busy).  The frame must stay unchanged until it has been read:
 * rmd_hip_seeds_pinned_frames_done returns the newest ticket up to which every frame handed over this way has been read (tickets count these
 * calls from 1; after rmd_hip_seeds_sync all are done).  Rows dense (width bytes apart).  Where the engine cannot take the frame from there (copy
 * engines not addressable, RMD_HIP_TUNE_COPY_ENGINES = 0, another RMD_HIP_TUNE_HOST_FRAMES mode, lens undistortion, a batch member's reference)
 * the call copies like rmd_hip_seeds_update_u8 and the ticket is done when it returns.  Results are those of rmd_hip_seeds_update_u8, bit for bit. */
int rmd_hip_host_alloc(void** ptr, size_t bytes);
int rmd_hip_host_free(void* ptr);
int rmd_hip_seeds_update_u8_pinned(rmd_hip_seeds_t* s, const unsigned char* pinned_gray, const float* T_curr_world, unsigned long long* ticket);
int rmd_hip_seeds_pinned_frames_done(rmd_hip_seeds_t* s, unsigned long long* ticket_done);
/* downloadDepthmap/downloadConvergence :160-168 and the RMD_BUILD_TESTS downloads :205-230 */
int rmd_hip_seeds_download(const rmd_hip_seeds_t* s, int plane, void* host_dst);
/* test hook: overwrite mu / sigma_sq / a / b (planes 0..3) */
int rmd_hip_seeds_upload(rmd_hip_seeds_t* s, int plane, const float* host_src);
/* getMu/getSigmaSq/getA/getB/getConvergence :170-193: a borrowed view, valid while `s` lives */
int rmd_hip_seeds_plane(const rmd_hip_seeds_t* s, int plane, const rmd_hip_image_t** view);
/* getConvergedCount :195-198 (count of CONVERGED in the convergence plane).  Right after an update of the tile pipeline this needs
 * neither a kernel nor a device synchronisation: seed_check -- the update's first kernel -- decides the count, the update mirrors it to
 * pinned memory, and the call returns as soon as that has happened (the NCC search of the same update may still be running). */
int rmd_hip_seeds_converged_count(const rmd_hip_seeds_t* s, size_t* count);
/* getDistFromRef :200-203 */
/* Lens undistortion of the 8-bit frames handed to set_reference_u8 / update_u8, in front of the x(1/255) conversion:
 * Depthmap::initUndistortionMap + inputImage (depthmap.cpp:45-61,95-106), i.e. cv::initUndistortRectifyMap(K, (k1, k2, r1, r2),
 * I, K, size, CV_16SC2) once (host, double precision) and cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) per frame (device, integer
 * arithmetic).  OpenCV itself is not linked; its published algorithm is restated (parity with a particular OpenCV build is not
 * pinned). */
int rmd_hip_seeds_init_undistortion_map(rmd_hip_seeds_t* s, float k1, float k2, float r1, float r2);
/* the map computation alone (host only, no device needed): what cv::initUndistortRectifyMap(..., CV_16SC2, map1, map2) returns */
int rmd_hip_compute_undistortion_map(int width, int height, float fx, float fy, float cx, float cy, float k1, float k2, float r1, float r2,
                                     short* map1_xy, unsigned short* map2);
/* the maps as computed: map1_xy = W*H (x, y) int16 pairs, map2 = W*H uint16 (fy * 32 + fx) */
int rmd_hip_seeds_undistortion_map(const rmd_hip_seeds_t* s, short* map1_xy, unsigned short* map2);
/* CONVERGED-masked back-projection to world-frame XYZI points, the loop of Publisher::publishPointCloud (publisher.cpp:54-104)
 * on the device: for every pixel (x, y) in row-major order whose state is CONVERGED,
 *   f = normalize(((x - cx) / fx, (y - cy) / fy, 1));  (X, Y, Z) = T_world_ref * (f * depth(x, y));  I = 8-bit reference image.
 * depth: an f32 W x H device image (e.g. rmd_hip_denoiser_result) or NULL for the seeds' own mu.  out_xyzi: `capacity` points
 * of 4 floats.  *n_points = number of converged seeds; if it exceeds `capacity` only the first `capacity` points are written. */
int rmd_hip_seeds_point_cloud(rmd_hip_seeds_t* s, const rmd_hip_image_t* depth, float* out_xyzi, size_t capacity, size_t* n_points);
/* The coloured convergence map of Publisher::publishConvergenceMap (publisher.cpp:112-147) on the device: the 8-bit reference image as
 * B = G = R (cv::cvtColor GRAY2BGR), blue = 255 where the seed is CONVERGED, red = 255 where it is DIVERGED.  host_bgr: W x H x 3 bytes,
 * packed (what cv::Mat CV_8UC3 / sensor_msgs BGR8 hold).  Replaces downloadConvergence (4 B/pixel) + a host loop by a 3 B/pixel download. */
int rmd_hip_seeds_convergence_bgr8(rmd_hip_seeds_t* s, unsigned char* host_bgr);
/* Publication OFF the update stream (live use: DepthmapNode::denoiseAndPublishResults / publishConvergenceMap, depthmap_node.cpp:165-182, whose
 * std::async only blocks by accident of a discarded future).  rmd_hip_seeds_publish_async snapshots what the requested products need -- mu,
 * sigma_sq, a, b, convergence, reference image, T_world_ref: one device-to-device kernel on the handle's stream -- and queues, on a second stream
 * of the handle, the work the reference does between two messages: DepthmapDenoiser::denoise(lambda, iterations) of the snapshot
 * (depthmap_denoiser.cu:179-224, setLargeSigmaSq(depth_range)), the CONVERGED-masked point cloud of the denoised map (publisher.cpp:54-104), the
 * coloured convergence map (:112-147), the int32 convergence plane, and their transfers into pinned host memory.  It returns at once: the caller
 * goes on with setReferenceImage / update, whose kernels run beside the publication's.  rmd_hip_seeds_publish_collect hands over the OLDEST
 * publication that has not been collected: RMD_HIP_OK and the products in the caller's buffers (any of them may be NULL), RMD_HIP_BUSY when it is
 * still in flight and `wait` is 0, RMD_HIP_ERR_NOT_READY when there is none.  Up to RMD_HIP_PUBLISH_SLOTS publications may be uncollected at a
 * time; results are those of the synchronous calls on the state at the time of the request, bit for bit (tests/test_publish_async.py). */
#define RMD_HIP_BUSY 1
#define RMD_HIP_PUBLISH_DEPTH 1u            /* TV-L1 denoised depth map, W x H floats */
#define RMD_HIP_PUBLISH_CLOUD 2u            /* XYZI points of the converged seeds from that map (implies DEPTH) */
#define RMD_HIP_PUBLISH_CONVERGENCE_BGR 4u  /* coloured convergence map, W x H x 3 bytes */
#define RMD_HIP_PUBLISH_CONVERGENCE 8u      /* convergence states, W x H int32 */
#define RMD_HIP_PUBLISH_SLOTS 3
int rmd_hip_seeds_publish_async(rmd_hip_seeds_t* s, unsigned int what, float depth_range, float lambda, int iterations, int* ticket);
int rmd_hip_seeds_publish_collect(rmd_hip_seeds_t* s, int wait, unsigned int* what, int* ticket, float* host_depth, float* host_xyzi, size_t capacity,
                                  size_t* n_points, unsigned char* host_bgr, int* host_convergence);
/* The same without the copies: _peek hands out POINTERS into the pinned host buffers of the oldest publication (NULL for a product that was
 * not requested; the points: *n_points x 4 floats) and leaves it in its slot; they stay valid until rmd_hip_seeds_publish_release gives the slot
 * back.  collect = peek + memcpy + release. */
int rmd_hip_seeds_publish_peek(rmd_hip_seeds_t* s, int wait, unsigned int* what, int* ticket, const float** depth, const float** xyzi, size_t* n_points,
                               const unsigned char** bgr, const int** convergence);
int rmd_hip_seeds_publish_release(rmd_hip_seeds_t* s);
int rmd_hip_seeds_dist_from_ref(const rmd_hip_seeds_t* s, float* dist);
/* diagnostics (not in the reference): how this handle's staged host frames have travelled so far -- counts[0] on copy engines addressed directly
 * (RMD_HIP_TUNE_COPY_ENGINES 1..3), counts[1] on the handle's copy stream (route 0, and the fallback where the engines cannot be addressed) */
int rmd_hip_seeds_staged_frames(const rmd_hip_seeds_t* s, unsigned long long counts[2]);
/* blocks until all work queued by this handle has finished */
int rmd_hip_seeds_sync(const rmd_hip_seeds_t* s);

/* knobs (not in the reference) */
#define RMD_HIP_OPT_MATCHER 0      /* 0 = per-pixel kernel (the reference's shape: one lane per seed, every lane gathers its own texels; the A/B baseline the
                                      full-size tests compare the tile pipeline with), 3 = two-launch tile pipeline (default) */
#define RMD_HIP_OPT_TIMING 1       /* 1 = bracket every update with HIP events on the handle's stream; 2 = one event pair
                                      around everything between timing_reset and the timing query (no markers in between) */
#define RMD_HIP_OPT_COLLECT_STATS 2 /* 1 = count live seeds / search steps / NCC evaluations per update;
                                      2 = in-kernel timeline probes instead (see rmd_hip_seeds_trace_download) */
#define RMD_HIP_OPT_LAZY_FINALIZE 4 /* 1 (default) = defer an update's last kernel and fuse it into the next update */
#define RMD_HIP_OPT_UNIT_TARGET 7   /* tile pipeline: work units aimed at per frame, in multiples (1..4) of the resident search workgroups (default 2 for a
                                      SeedMatrix, 1 for a batch); the unit size (1..4 rounds of 256 NCC evaluations) follows from the previous frame's work */
#define RMD_HIP_OPT_INJECT_FAULT 9  /* test hook: 1 = the arrival flag of the NEXT host frame that travels through a staging buffer is withheld once; that
                                      update's bounded in-kernel wait (about 0.1 s) runs out, the next synchronising call reports RMD_HIP_ERR_RUNTIME
                                      once, and the handle is usable again from the next setReferenceImage on (tests/test_full_speed.py) */
/* (option numbers 3, 5, 6, 8, 10 belonged to experiments that are no longer part of the library -- LAB.md -- and are rejected) */
int rmd_hip_seeds_set_option(rmd_hip_seeds_t* s, int option, int value);
/* kernels of the seed path, for rmd_hip_seeds_timing */
#define RMD_HIP_STAGE_SEED_INIT 0
#define RMD_HIP_STAGE_UPDATE 1  /* fused seed_check + epipolar_match + seed_update */
#define RMD_HIP_STAGE_COUNT 2
#define RMD_HIP_NUM_SEED_STAGES 3
/* accumulated device time and launch count of one stage since the last reset (needs RMD_HIP_OPT_TIMING) */
int rmd_hip_seeds_timing(const rmd_hip_seeds_t* s, int stage, double* total_ms, long* launches);
int rmd_hip_seeds_timing_reset(rmd_hip_seeds_t* s);
/* out[0..2] = live seeds, epipolar steps visited, NCC evaluations of the last update (needs COLLECT_STATS) */
int rmd_hip_seeds_last_stats(const rmd_hip_seeds_t* s, long long* out3);
/* diagnostics of the last update (needs COLLECT_STATS = 1): [0..2] as last_stats, [3] NCC evaluations that read their texels from L2 instead
 * of the LDS window, [4] work units searched, [5] windows staged inside the search kernel; the rest is zero (per-workgroup probes are the
 * timeline, COLLECT_STATS = 2) */
int rmd_hip_seeds_last_diagnostics(const rmd_hip_seeds_t* s, long long* out16);
/* timeline of update number `frame` (0 = first update after COLLECT_STATS was set to 2; the last 256 are kept), tile pipeline: 8 words
 * per 16x16 tile / search workgroup (row-major tiles; workgroup w of the search shares slot w): [0] search workgroup start, [1] its first unit
 * staged, [2] setup tile: start | state ready << 32 | end << 48 (relative), [3] search workgroup end, [4] work items, [5] units,
 * [6] last tile, [7] fallbacks | windows << 32; 10 ns ticks of the device wall clock.  Needs 8 * tiles words. */
int rmd_hip_seeds_trace_download(rmd_hip_seeds_t* s, int frame, unsigned long long* out, size_t capacity, size_t* written);

/* ---- batched mode: several independent SeedMatrix objects stepped by ONE launch pair ------------------------------------------
 * BASELINE configs[3] / north_star "a batched mode shards independent image sequences": sequences do not interact (seed_matrix.cu
 * keeps no state outside the object once the global texture references are gone), and a single 640x480 frame cannot occupy 256 CUs.
 * A batch owns `n` (1..24: up to three stream groups of up to eight sequences, one launch pair per group and step) SeedMatrix objects of one size on the current device; each is a full rmd_hip_seeds_t -- set_reference*,
 * download, plane views, converged_count, point_cloud, denoise all work per member, exactly as for a stand-alone object -- except
 * that update* is issued for all members at once with the calls below (seed_matrix.cu:120-158 per member: same arithmetic, same
 * results bit for bit as that member stepped alone) and that rmd_hip_batch_destroy releases the members. */
int rmd_hip_batch_create(int n, int width, int height, float fx, float fy, float cx, float cy, int patch_side, int max_extent,
                         rmd_hip_batch_t** out);
int rmd_hip_batch_destroy(rmd_hip_batch_t* b);
int rmd_hip_batch_size(const rmd_hip_batch_t* b, int* n);
/* member `index` (borrowed: valid while the batch lives, not to be destroyed) */
int rmd_hip_batch_member(rmd_hip_batch_t* b, int index, rmd_hip_seeds_t** member);
/* SeedMatrix::update for every member i whose frame pointer is not NULL (a member without a frame in this step is left alone).
 * T_curr_world: n x 12 floats.  Frames: device-resident (read in place until the next synchronising call, like
 * rmd_hip_seeds_update_device), 8-bit gray host frames (x(1/255) and the member's lens undistortion on the device, like
 * rmd_hip_seeds_update_u8) or float host frames (like rmd_hip_seeds_update); host frames have been copied when the call returns. */
int rmd_hip_batch_update_device(rmd_hip_batch_t* b, const float* const* dev_imgs, const size_t* stride_elems, const float* T_curr_world);
int rmd_hip_batch_update_u8(rmd_hip_batch_t* b, const unsigned char* const* host_gray, const float* T_curr_world);
int rmd_hip_batch_update(rmd_hip_batch_t* b, const float* const* host_imgs, const float* T_curr_world);
/* blocks until all work queued by the batch and its members has finished */
int rmd_hip_batch_sync(rmd_hip_batch_t* b);
/* RMD_HIP_OPT_TIMING (0 / 2: one HIP event pair on the batch's stream between timing_reset and timing) and RMD_HIP_OPT_UNIT_TARGET */
int rmd_hip_batch_set_option(rmd_hip_batch_t* b, int option, int value);
int rmd_hip_batch_timing_reset(rmd_hip_batch_t* b);
int rmd_hip_batch_timing(rmd_hip_batch_t* b, double* total_ms, long* steps);

/* DepthmapDenoiser::denoise (depthmap_denoiser.cu:179-224) for EVERY member of the batch in one launch sequence (grid z = member: the B depth
 * maps of a step share each launch instead of queueing B x 50 latency-bound launches one behind the other); per member the arithmetic and the
 * result are those of a rmd_hip_denoiser_t on that member's planes, bit for bit.  depth_range: n floats (setLargeSigmaSq per member, :226-229);
 * host_denoised: NULL, or n pointers to W x H floats (NULL entries: that member's map stays on the device).  Synchronises like denoise(). */
int rmd_hip_batch_denoise(rmd_hip_batch_t* b, const float* depth_range, float lambda, int iterations, float* const* host_denoised);
/* member `index`'s result of the last rmd_hip_batch_denoise, in device memory (a view, valid until the next one; e.g. for rmd_hip_seeds_point_cloud) */
int rmd_hip_batch_denoise_result(const rmd_hip_batch_t* b, int index, const rmd_hip_image_t** view);
/* device time of the iteration launches of the last rmd_hip_batch_denoise (one HIP event pair on its stream) and their number */
int rmd_hip_batch_denoise_timing(const rmd_hip_batch_t* b, double* total_ms, long* launches);

/* ---- rmd::DepthmapDenoiser (depthmap_denoiser.cu) --------------------------------------- */
int rmd_hip_denoiser_create(int width, int height, rmd_hip_denoiser_t** out); /* ctor :143-169 */
int rmd_hip_denoiser_destroy(rmd_hip_denoiser_t* d);
int rmd_hip_denoiser_set_large_sigma_sq(rmd_hip_denoiser_t* d, float depth_range); /* :226-229 */
/* denoise :179-224.  Returns RMD_HIP_ERR_NOT_READY if set_large_sigma_sq was never called
 * (the reference prints to cerr and returns, :189-193). */
int rmd_hip_denoiser_denoise(rmd_hip_denoiser_t* d, const rmd_hip_image_t* mu, const rmd_hip_image_t* sigma_sq,
                             const rmd_hip_image_t* a, const rmd_hip_image_t* b, float* host_denoised, float lambda,
                             int iterations);
/* as above but leaves the result in device memory (view valid until the next denoise); host_denoised may be NULL */
int rmd_hip_denoiser_result(const rmd_hip_denoiser_t* d, const rmd_hip_image_t** view);
/* L, tau, sigma, theta of denoise::DeviceData (depthmap_denoiser.cu:124-141) */
int rmd_hip_denoiser_constants(const rmd_hip_denoiser_t* d, float* out4);
#define RMD_HIP_DENOISE_OPT_TIMING 1
#define RMD_HIP_DENOISE_OPT_ITERS_PER_LAUNCH 2 /* TV iterations per launch: 0 = chosen from the image size (default), 1 = one launch per iteration, 2..8 = temporal blocking depth (capped by the geometry's) */
#define RMD_HIP_DENOISE_OPT_GEOMETRY 3 /* tile geometry of the blocked kernel: 0 = default (16x16 tiles, 4 iterations per launch), 1 = 32x8 K2, 2 = 64x16 K4,
                                          3 = 32x16 K4, 4 = 16x16 K4, 5 = 16x16 K8 (experiments, tools/denoise_sweep.py) */
int rmd_hip_denoiser_set_option(rmd_hip_denoiser_t* d, int option, int value);
/* accumulated device time / launches of the TV iteration kernel since the last denoise() started */
int rmd_hip_denoiser_timing(const rmd_hip_denoiser_t* d, double* total_ms, long* launches);

/* ---- rmd::ImageReducer<T> (reduction.cu) ------------------------------------------------ */
int rmd_hip_reduce_sum_f32(const rmd_hip_image_t* img, float* sum);                    /* ImageReducer<float>::sum(const DeviceImage&) :132-139 */
int rmd_hip_reduce_sum_i32(const rmd_hip_image_t* img, int* sum);                      /* ImageReducer<int>::sum (explicit instantiation :186) */
int rmd_hip_reduce_count_eq_i32(const rmd_hip_image_t* img, int value, size_t* count); /* countEqual(const DeviceImage<int>&, int) :175-183 */
/* the raw-pointer overloads (reduction.cuh:33-36, 41-45; reduction.cu:81-130, 145-173): DEVICE pointers on the current device,
 * row stride in elements */
int rmd_hip_reduce_sum_f32_raw(const float* dev_data, size_t stride_elems, size_t width, size_t height, float* sum);
int rmd_hip_reduce_sum_i32_raw(const int* dev_data, size_t stride_elems, size_t width, size_t height, int* sum);
int rmd_hip_reduce_count_eq_i32_raw(const int* dev_data, size_t stride_elems, size_t width, size_t height, int value, size_t* count);

/* ---- self test: the VALU (DPP) wave reductions / scans of the kernels against their shuffle forms; *mismatching_lanes must be 0 */
int rmd_hip_selftest_wave_primitives(int* mismatching_lanes);
/* ---- self test (host only, no device needed): the examination that lets a float frame of 8-bit levels travel as bytes (see
 * rmd_hip_seeds_update).  *all_levels = 1 and `bytes` (height rows of `pitch` bytes) filled if every pixel of the W x H float image has the
 * bit pattern of (float)k * (1.0f / 255.0f), else 0 (bytes then unspecified). */
int rmd_hip_selftest_pack_float_frame(const float* host_img, int width, int height, int pitch, unsigned char* bytes, int* all_levels);

/* ---- arithmetic-contract self test (device side of csrc/rmd_math.h) ---------------------- */
/* op: 0 expf, 1 sinf, 2 acosf, 3 rsqrtf, 4 sqrtf, 5 x/y, 6 lerp(t=x, a=y, b=z); n host floats in, n out */
int rmd_hip_math_eval(int op, const float* x, const float* y, const float* z, float* out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* RMD_HIP_H */
