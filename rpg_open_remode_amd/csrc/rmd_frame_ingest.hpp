// Frames handed over in host memory, device side: the conversions that bring a frame from the pinned ring (in place, over the host link) or
// from its staging buffer in HBM into a current-image plane.  Run by the ingest workgroups of the setup kernel (rmd_frame_setup.hpp) and by
// the bringers of the search kernel one step ahead (rmd_frame_search.hpp); the host side is rmd_ingest.hip / rmd_batch.hip.
#ifndef RMD_FRAME_INGEST_HPP
#define RMD_FRAME_INGEST_HPP

#include "rmd_matcher.hpp"

namespace rmdk {

// A frame in pinned HOST memory converted into a current-image plane by workgroup `part` of `parts`: plain 16-byte loads over the host
// link (pinned host memory is mapped uncached: every load fetches; 8-byte system-scope atomic loads reached 10 GB/s, these reach the
// link's 50).  kind 1: 8-bit rows of `pitch` bytes, x (1/255) like Depthmap::inputImage's convertTo(CV_32F, 1.0f / 255.0f)
// (depthmap.cpp:105); kind 2: w x h floats, unpadded.  The buffer is padded to a multiple of 16 bytes.
RMDK_D void ingest_in_place(int kind, int pitch, const void* src_v, float* __restrict__ dst, int w, int h, int stride, int part, int parts,
    int tid) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4* src = static_cast<const u32x4*>(src_v);
  if (kind == 1) {
    const int total_bytes = pitch * h, requests = (total_bytes + 15) >> 4;
    for (int d = part * TILE_PIX + tid; d < requests; d += parts * TILE_PIX) {
      const u32x4 v = __builtin_nontemporal_load(src + d);
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // four pixels per dword; rows are a multiple of 4 bytes, so a dword never straddles two rows
        const int byte = d * 16 + q * 4;
        if (byte >= total_bytes) break;
        const int row = byte / pitch, x4 = byte - row * pitch;
        const unsigned int u = v[q];
        float* out = dst + static_cast<size_t>(row) * stride + x4;
        const float f0 = static_cast<float>(u & 0xffu) * (1.0f / 255.0f), f1 = static_cast<float>((u >> 8) & 0xffu) * (1.0f / 255.0f);
        const float f2 = static_cast<float>((u >> 16) & 0xffu) * (1.0f / 255.0f), f3 = static_cast<float>(u >> 24) * (1.0f / 255.0f);
        if (x4 + 3 < w) *reinterpret_cast<float4*>(out) = make_float4(f0, f1, f2, f3);  // plane rows and x4 are multiples of 16 bytes
        else {
          if (x4 < w) out[0] = f0;
          if (x4 + 1 < w) out[1] = f1;
          if (x4 + 2 < w) out[2] = f2;
        }
      }
    }
  } else {
    const int total = w * h, requests = (total + 3) >> 2;
    for (int d = part * TILE_PIX + tid; d < requests; d += parts * TILE_PIX) {
      const u32x4 v = __builtin_nontemporal_load(src + d);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = d * 4 + q;
        if (e >= total) break;
        const int row = e / w;
        dst[static_cast<size_t>(row) * stride + (e - row * w)] = __uint_as_float(v[q]);
      }
    }
  }
}

// Dword `d` of an 8-bit frame (rows of `per_row` dwords, four pixels each) -> its four floats in the current-image plane: x (1/255) like
// Depthmap::inputImage's convertTo(CV_32F, 1.0f / 255.0f) (depthmap.cpp:105), one fp32 multiply per pixel.
RMDK_D void store_u8x4(float* __restrict__ dst, int stride, int w, int per_row, int d, unsigned int u) {
  const int row = d / per_row, x4 = (d - row * per_row) * 4;
  float* out = dst + static_cast<size_t>(row) * stride + x4;
  const float f0 = static_cast<float>(u & 0xffu) * (1.0f / 255.0f), f1 = static_cast<float>((u >> 8) & 0xffu) * (1.0f / 255.0f);
  const float f2 = static_cast<float>((u >> 16) & 0xffu) * (1.0f / 255.0f), f3 = static_cast<float>(u >> 24) * (1.0f / 255.0f);
  if (x4 + 3 < w) *reinterpret_cast<float4*>(out) = make_float4(f0, f1, f2, f3);  // plane rows and x4 are multiples of 16 bytes
  else {
    if (x4 < w) out[0] = f0;
    if (x4 + 1 < w) out[1] = f1;
    if (x4 + 2 < w) out[2] = f2;
  }
}

// A frame in a staging buffer in HBM (MatcherArgs::ingest_kind: 1 = 8-bit rows of `pitch` bytes, 2 = w x h floats) converted into a
// current-image plane by workgroup `part` of `parts`.  INGEST_BATCH dwords per lane are requested before the first is used: a lane's
// requests are `parts` x 256 dwords apart (every instruction fully coalesced), and the whole frame is a round trip or two instead of one
// per dword -- at 1920x1080 a lane of 128 workgroups has sixteen dwords to fetch, and fetched one by one (the loads could not be moved
// across the stores of the iteration before) the conversion took longer than the setup kernel it rides in.  LOAD: plain loads for the
// search kernel's bringers (the copy had completed before an EARLIER kernel saw its flag), agent-scope loads for a frame's own setup kernel
// (the copy may complete while the kernel runs).
constexpr int INGEST_BATCH = 4;
template <typename LOAD>
RMDK_D void ingest_staged(int kind, int pitch, const void* src_v, float* __restrict__ dst, int w, int h, int stride, int part, int parts,
    int tid, LOAD load) {
  const unsigned int* src = static_cast<const unsigned int*>(src_v);
  const int per_row = kind == 1 ? pitch >> 2 : w, total = per_row * h, step = parts * TILE_PIX;
  auto put = [&](int d, unsigned int u) {
    if (d >= total) return;
    if (kind == 1) store_u8x4(dst, stride, w, per_row, d, u);
    else {
      const int row = d / w;
      dst[static_cast<size_t>(row) * stride + (d - row * w)] = __uint_as_float(u);
    }
  };
  static_assert(INGEST_BATCH == 4, "the four requests below");
  for (int d0 = part * TILE_PIX + tid; d0 < total; d0 += INGEST_BATCH * step) {
    const int d1 = d0 + step, d2 = d1 + step, d3 = d2 + step;
    const unsigned int v0 = load(src + d0), v1 = load(src + min(d1, total - 1)), v2 = load(src + min(d2, total - 1)),
        v3 = load(src + min(d3, total - 1));
    put(d0, v0); put(d1, v1); put(d2, v2); put(d3, v3);
  }
}
// workgroups that convert a staged frame of `dwords` dwords: INGEST_BATCH requests per lane, at most `cap`
inline int ingest_workgroups(long long dwords, int cap) {
  const long long want = (dwords + static_cast<long long>(TILE_PIX) * INGEST_BATCH - 1) / (static_cast<long long>(TILE_PIX) * INGEST_BATCH);
  return static_cast<int>(want < cap ? (want < 1 ? 1 : want) : cap);
}

}  // namespace rmdk

#endif  // RMD_FRAME_INGEST_HPP
