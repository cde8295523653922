"""ctypes bindings for the two CPU oracles (test infrastructure only).

  OracleLib("ref", side)       -> oracle/_ref/libremode_ref_s<side>.so     Oracle A: the reference's own sources on CPU (glibc libm)
  OracleLib("ref_rmd", side)   -> oracle/_ref/libremode_ref_rmd_s<side>.so Oracle A with expf/sinf/acosf from csrc/rmd_math.h
  OracleLib("port", side)      -> oracle/libremode_oracle_s<side>.so       Oracle B: our restatement, rmd_math.h transcendentals
  OracleLib("port_libm", side) -> oracle/libremode_oracle_libm_s<side>.so  Oracle B with glibc transcendentals (== A bit for bit)

All three expose the same Seeds / Denoiser objects so tests can swap them freely.
"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

PLANE_MU, PLANE_SIGMA_SQ, PLANE_A, PLANE_B, PLANE_CONV, PLANE_SUM_TEMPL, PLANE_DENOM, PLANE_MATCH = range(8)
UPDATE, CONVERGED, BORDER, DIVERGED, NO_MATCH, NOT_VISIBLE = range(6)

_c_f, _c_i, _c_p = ctypes.c_float, ctypes.c_int, ctypes.c_void_p


def lib_path(kind, side):
    if kind == "ref":
        return os.path.join(ORACLE_DIR, "_ref", f"libremode_ref_s{side}.so")
    if kind == "ref_rmd":
        return os.path.join(ORACLE_DIR, "_ref", f"libremode_ref_rmd_s{side}.so")
    if kind == "port":
        return os.path.join(ORACLE_DIR, f"libremode_oracle_s{side}.so")
    if kind == "port_libm":
        return os.path.join(ORACLE_DIR, f"libremode_oracle_libm_s{side}.so")
    raise ValueError(kind)


def available(kind, side):
    return os.path.exists(lib_path(kind, side))


class OracleLib:
    _cache = {}

    def __new__(cls, kind, side):
        key = (kind, side)
        if key not in cls._cache:
            obj = super().__new__(cls)
            obj._init(kind, side)
            cls._cache[key] = obj
        return cls._cache[key]

    def _init(self, kind, side):
        self.kind, self.side = kind, side
        self.prefix = "ref_" if kind.startswith("ref") else "orc_"
        path = lib_path(kind, side)
        if not os.path.exists(path):
            raise RuntimeError(f"oracle library {path} missing (run `make -C oracle`)")
        self.lib = L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        p = self.prefix
        f = lambda name: getattr(L, p + name)
        f("seeds_create").restype = _c_p
        f("seeds_create").argtypes = [_c_i, _c_i, _c_f, _c_f, _c_f, _c_f]
        f("seeds_destroy").argtypes = [_c_p]
        f("seeds_set_reference").argtypes = [_c_p, _c_p, _c_p, _c_f, _c_f]
        f("seeds_update").argtypes = [_c_p, _c_p, _c_p]
        f("seeds_download").argtypes = [_c_p, _c_i, _c_p]
        f("seeds_upload").argtypes = [_c_p, _c_i, _c_p]
        f("seeds_converged_count").restype = ctypes.c_ulong
        f("seeds_converged_count").argtypes = [_c_p]
        f("seeds_dist_from_ref").restype = _c_f
        f("seeds_dist_from_ref").argtypes = [_c_p]
        f("denoiser_create").restype = _c_p
        f("denoiser_create").argtypes = [_c_i, _c_i]
        f("denoiser_destroy").argtypes = [_c_p]
        f("denoiser_set_large_sigma_sq").argtypes = [_c_p, _c_f]
        f("denoiser_denoise").argtypes = [_c_p, _c_p, _c_p, _c_f, _c_i]
        f("denoiser_constants").argtypes = [_c_p, _c_p]
        f("reduce_sum_f32").restype = _c_f
        f("reduce_sum_f32").argtypes = [_c_p, _c_i, _c_i]
        f("reduce_count_eq_i32").restype = ctypes.c_ulong
        f("reduce_count_eq_i32").argtypes = [_c_p, _c_i, _c_i, _c_i]
        f("patch_side").restype = _c_i
        assert f("patch_side")() == side
        if not kind.startswith("ref"):
            L.orc_seeds_last_stats.argtypes = [_c_p, _c_p]
            L.orc_set_num_threads.argtypes = [_c_i]
            L.orc_max_threads.restype = _c_i
            for name in ("expf", "sinf", "acosf", "rsqrtf"):
                fn = getattr(L, "orc_math_" + name)
                fn.restype, fn.argtypes = _c_f, [_c_f]
            L.orc_math_lerp.restype, L.orc_math_lerp.argtypes = _c_f, [_c_f, _c_f, _c_f]
            L.orc_tex_linear.restype, L.orc_tex_linear.argtypes = _c_f, [_c_p, _c_i, _c_i, _c_f, _c_f]
            L.orc_denoiser_denoise_planes.argtypes = [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_f, _c_i]
            L.orc_math_eval_array.argtypes = [_c_i, _c_p, _c_p, _c_p, _c_p, ctypes.c_long]
            L.orc_math_eval_array.restype = None

    def fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def reduce_sum(self, img):
        img = np.ascontiguousarray(img, np.float32)
        return float(self.fn("reduce_sum_f32")(img.ctypes.data, img.shape[1], img.shape[0]))

    def reduce_count_eq(self, img, value):
        img = np.ascontiguousarray(img, np.int32)
        return int(self.fn("reduce_count_eq_i32")(img.ctypes.data, img.shape[1], img.shape[0], int(value)))


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


class Seeds:
    """rmd::SeedMatrix as run by one of the oracles."""

    def __init__(self, olib, width, height, K):
        self.o, self.w, self.h = olib, width, height
        self.ptr = olib.fn("seeds_create")(width, height, *[float(k) for k in K])
        if not self.ptr:
            raise RuntimeError("oracle seeds_create failed")

    def close(self):
        if self.ptr:
            self.o.fn("seeds_destroy")(self.ptr)
            self.ptr = None

    def __del__(self):
        self.close()

    def set_reference(self, img, T_curr_world, min_depth, max_depth):
        img, T = _f32(img), _f32(T_curr_world).reshape(12)
        return self.o.fn("seeds_set_reference")(self.ptr, img.ctypes.data, T.ctypes.data, float(min_depth), float(max_depth))

    def update(self, img, T_curr_world):
        img, T = _f32(img), _f32(T_curr_world).reshape(12)
        return self.o.fn("seeds_update")(self.ptr, img.ctypes.data, T.ctypes.data)

    def download(self, plane):
        if plane == PLANE_CONV:
            out = np.empty((self.h, self.w), np.int32)
        elif plane == PLANE_MATCH:
            out = np.empty((self.h, self.w, 2), np.float32)
        else:
            out = np.empty((self.h, self.w), np.float32)
        rc = self.o.fn("seeds_download")(self.ptr, plane, out.ctypes.data)
        assert rc == 0
        return out

    def upload(self, plane, arr):
        arr = _f32(arr)
        assert arr.shape == (self.h, self.w)
        rc = self.o.fn("seeds_upload")(self.ptr, plane, arr.ctypes.data)
        assert rc == 0

    def state(self):
        return {p: self.download(p) for p in range(8)}

    def converged_count(self):
        return int(self.o.fn("seeds_converged_count")(self.ptr))

    def dist_from_ref(self):
        return float(self.o.fn("seeds_dist_from_ref")(self.ptr))

    def last_stats(self):
        out = np.zeros(3, np.int64)
        self.o.lib.orc_seeds_last_stats(self.ptr, out.ctypes.data)
        return {"live_seeds": int(out[0]), "ncc_evals": int(out[1]), "steps": int(out[2])}


class Denoiser:
    """rmd::DepthmapDenoiser as run by one of the oracles."""

    def __init__(self, olib, width, height):
        self.o, self.w, self.h = olib, width, height
        self.ptr = olib.fn("denoiser_create")(width, height)

    def close(self):
        if self.ptr:
            self.o.fn("denoiser_destroy")(self.ptr)
            self.ptr = None

    def __del__(self):
        self.close()

    def set_large_sigma_sq(self, depth_range):
        self.o.fn("denoiser_set_large_sigma_sq")(self.ptr, float(depth_range))

    def constants(self):
        out = np.zeros(4, np.float32)
        self.o.fn("denoiser_constants")(self.ptr, out.ctypes.data)
        return out

    def denoise(self, seeds, lam, iterations):
        out = np.empty((self.h, self.w), np.float32)
        self.o.fn("denoiser_denoise")(self.ptr, seeds.ptr, out.ctypes.data, float(lam), int(iterations))
        return out


def planes_equal(a, b):
    """bit-level equality that treats any NaN as equal to any NaN"""
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        return bool(np.array_equal(a, b, equal_nan=True))
    return bool(np.array_equal(a, b))


def count_mismatch(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        return int(np.count_nonzero(~((a == b) | (np.isnan(a) & np.isnan(b)))))
    return int(np.count_nonzero(a != b))


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
