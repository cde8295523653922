"""Oracle for the host-side steps either side of the depth-filter path: numpy restatements of
  * Publisher::publishPointCloud           src/publisher.cpp:54-104      -> point_cloud()
  * Depthmap::initUndistortionMap          src/depthmap.cpp:45-61        -> undistort_maps()   (cv::initUndistortRectifyMap)
  * Depthmap::inputImage, distorted case   src/depthmap.cpp:95-106       -> remap_u8()         (cv::remap, INTER_LINEAR)

TEST INFRASTRUCTURE ONLY: imported by tests/ (through tests/oracles.py), never by the product package.

Pinning.  point_cloud() restates reference source that is present in /root/reference but cannot be compiled here (ROS, PCL,
OpenCV); it is float32 arithmetic, one rounding per operation in the reference's order.  undistort_maps() / remap_u8() restate
OpenCV, a third-party dependency that is absent from /root/reference and from this image and whose version the reference does
not fix (find_package(OpenCV REQUIRED), CMakeLists.txt:55): they follow OpenCV's published scalar algorithm (imgproc/undistort.cpp,
imgproc/imgwarp.cpp, cv::invert's closed form for 3x3) and are NOT pinned against an OpenCV build -- PARITY UNPINNED for the
undistortion step.  The reference's own tests hold no vectors for any of the three.
"""
import numpy as np


def point_cloud(depth, convergence, ref_img_u8, K, T_world_ref):
    """CPU restatement of Publisher::publishPointCloud (src/publisher.cpp:54-104) in numpy float32, one rounding per
    operation in the reference's order: f = normalize(((x-cx)/fx, (y-cy)/fy, 1)) with normalize = v * (1/sqrt(v.v))
    (helper_math.h:1248-1251,1309-1313, rsqrtf as 1/sqrtf off-device), xyz = T_world_ref.translate(rotate(f * depth))
    (se3.cuh:111-124,165-168); CONVERGED pixels only, row-major; intensity = the 8-bit reference image.
    Test infrastructure only."""
    f32 = np.float32
    fx, fy, cx, cy = (f32(v) for v in K)
    depth = np.asarray(depth, f32)
    h, w = depth.shape
    xs = np.broadcast_to(np.arange(w, dtype=f32)[None, :], (h, w))
    ys = np.broadcast_to(np.arange(h, dtype=f32)[:, None], (h, w))
    vx, vy, vz = (xs - cx) / fx, (ys - cy) / fy, np.ones((h, w), f32)
    dot = vx * vx + vy * vy + vz * vz
    inv = f32(1.0) / np.sqrt(dot)
    px, py, pz = (vx * inv) * depth, (vy * inv) * depth, (vz * inv) * depth
    T = np.asarray(T_world_ref, f32).reshape(12)
    X = (T[0] * px + T[1] * py + T[2] * pz) + T[3]
    Y = (T[4] * px + T[5] * py + T[6] * pz) + T[7]
    Z = (T[8] * px + T[9] * py + T[10] * pz) + T[11]
    keep = np.asarray(convergence) == 1
    return np.stack([X[keep], Y[keep], Z[keep], np.asarray(ref_img_u8)[keep].astype(f32)], axis=1).astype(f32)


def undistort_maps(width, height, K, D):
    """CPU restatement (numpy float64) of cv::initUndistortRectifyMap(K, D, I, K, (width, height), CV_16SC2) as called by
    Depthmap::initUndistortionMap (src/depthmap.cpp:45-61).  OpenCV is a third-party dependency absent from /root/reference and
    from this image, and the reference does not pin its version (CMakeLists.txt:55), so this follows OpenCV's published scalar
    algorithm (imgproc/undistort.cpp; cv::invert's closed form for 3x3) and is NOT pinned against an OpenCV build:
    PARITY UNPINNED for the undistortion step.  Test infrastructure only.
    K = (fx, fy, cx, cy), D = (k1, k2, r1, r2), all taken as float32 values widened to double (cv::Mat_<float> inputs)."""
    f64 = np.float64
    fx, fy, u0, v0 = (f64(np.float32(v)) for v in K)
    k1, k2, p1, p2 = (f64(np.float32(v)) for v in D)
    k3 = k4 = k5 = k6 = f64(0.0)
    S = [[fx, f64(0), u0], [f64(0), fy, v0], [f64(0), f64(0), f64(1)]]
    d = S[0][0] * (S[1][1] * S[2][2] - S[1][2] * S[2][1]) - S[0][1] * (S[1][0] * S[2][2] - S[1][2] * S[2][0]) + \
        S[0][2] * (S[1][0] * S[2][1] - S[1][1] * S[2][0])
    d = f64(1.0) / d
    ir = [(S[1][1] * S[2][2] - S[1][2] * S[2][1]) * d, (S[0][2] * S[2][1] - S[0][1] * S[2][2]) * d, (S[0][1] * S[1][2] - S[0][2] * S[1][1]) * d,
          (S[1][2] * S[2][0] - S[1][0] * S[2][2]) * d, (S[0][0] * S[2][2] - S[0][2] * S[2][0]) * d, (S[0][2] * S[1][0] - S[0][0] * S[1][2]) * d,
          (S[1][0] * S[2][1] - S[1][1] * S[2][0]) * d, (S[0][1] * S[2][0] - S[0][0] * S[2][1]) * d, (S[0][0] * S[1][1] - S[0][1] * S[1][0]) * d]
    rows = np.arange(height, dtype=f64)
    _x, _y, _w = rows * ir[1] + ir[2], rows * ir[4] + ir[5], rows * ir[7] + ir[8]
    map1 = np.zeros((height, width, 2), np.int16)
    map2 = np.zeros((height, width), np.uint16)
    for j in range(width):  # the column loop accumulates (_x += ir[0], ...), one rounding per column, as OpenCV's does
        w = f64(1.0) / _w
        x, y = _x * w, _y * w
        x2, y2 = x * x, y * y
        r2, _2xy = x2 + y2, 2 * x * y
        kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
        u = fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + u0
        v = fy * (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + v0
        iu, iv = np.rint(u * 32).astype(np.int64), np.rint(v * 32).astype(np.int64)  # cvRound: ties to even
        map1[:, j, 0] = (iu >> 5).astype(np.int16)
        map1[:, j, 1] = (iv >> 5).astype(np.int16)
        map2[:, j] = ((iv & 31) * 32 + (iu & 31)).astype(np.uint16)
        _x, _y, _w = _x + ir[0], _y + ir[3], _w + ir[6]
    return map1, map2


def remap_u8(img, map1, map2):
    """cv::remap(img 8UC1, map1 CV_16SC2, map2 CV_16UC1, INTER_LINEAR, BORDER_CONSTANT 0) (depthmap.cpp:99): OpenCV's fixed-point
    bilinear path -- 15-bit weights (32-fy)(32-fx)*32 ... from its interpolation table, (sum + 2^14) >> 15, taps outside the
    image read 0.  Integer arithmetic; same caveat as undistort_maps.  Test infrastructure only."""
    img = np.asarray(img, np.uint8)
    h, w = img.shape
    sx, sy = map1[..., 0].astype(np.int64), map1[..., 1].astype(np.int64)
    f = map2.astype(np.int64) & 1023
    fx, fy = f & 31, f >> 5

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        return np.where(ok, img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64), 0)
    s = tap(sy, sx) * ((32 - fy) * (32 - fx) * 32) + tap(sy, sx + 1) * ((32 - fy) * fx * 32) + \
        tap(sy + 1, sx) * (fy * (32 - fx) * 32) + tap(sy + 1, sx + 1) * (fy * fx * 32)
    return ((s + (1 << 14)) >> 15).astype(np.uint8)
