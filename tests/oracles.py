"""ctypes bindings for the two CPU oracles (test infrastructure only).

  OracleLib("ref", side)       -> oracle/_ref/libremode_ref_s<side>.so     Oracle A: the reference's own sources on CPU (glibc libm)
  OracleLib("ref_rmd", side)   -> oracle/_ref/libremode_ref_rmd_s<side>.so Oracle A with expf/sinf/acosf from csrc/rmd_math.h
  OracleLib("port", side)      -> oracle/libremode_oracle_s<side>.so       Oracle B: our restatement, rmd_math.h transcendentals
  OracleLib("port_libm", side) -> oracle/libremode_oracle_libm_s<side>.so  Oracle B with glibc transcendentals (== A bit for bit)
  OracleLib("cudalike", 9)     -> oracle/libremode_oracle_cudalike_s9.so   the libm build with run-time switches (set_cudalike) that model the
                                  reference's real CUDA build: 8-bit texture weights, -use_fast_math; "cudalike_fma": contracted as well

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
    if kind in ("port_e150", "port_e178"):  # Oracle B / Oracle A (shared math) built with RMD_MAX_EXTENT_EPIPOLAR_SEARCH = 150 / 178 (side 9 only)
        return os.path.join(ORACLE_DIR, f"libremode_oracle_{kind[5:]}_s{side}.so")
    if kind in ("ref_rmd_e150", "ref_rmd_e178"):
        return os.path.join(ORACLE_DIR, "_ref", f"libremode_ref_rmd_{kind[8:]}_s{side}.so")
    if kind in ("cudalike", "cudalike_fma"):  # Oracle B with run-time switches towards the reference's real CUDA build (side 9 only; remode_oracle.cpp)
        return os.path.join(ORACLE_DIR, f"libremode_oracle_{kind}_s{side}.so")
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
            L.orc_set_cudalike.argtypes = [_c_i]
            L.orc_set_cudalike.restype = _c_i
            L.orc_set_cudalike_amplitude.argtypes = [_c_i]
            L.orc_set_cudalike_amplitude.restype = _c_i
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

    # switches of the "cudalike" builds (remode_oracle.cpp); a no-op returning False on the other builds
    TEX8, TEX8_TRUNC, DIV, SQRT, EXP, SIN, SIN_ABS, ACOS, FTZ = 1, 2, 4, 8, 16, 32, 64, 128, 256

    def set_cudalike(self, flags, amplitude=0):
        """flags: which operations follow the model of the reference's CUDA build; amplitude: 0 the model's fixed forms of the guessed operations,
        n >= 1 the correctly rounded result moved by up to n ulp, -1 by each operation's documented maximum (oracle/remode_oracle.cpp)"""
        return bool(self.lib.orc_set_cudalike(int(flags))) and bool(self.lib.orc_set_cudalike_amplitude(int(amplitude)))

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


def _bits(a):
    """the bit patterns of a float array with every NaN mapped to one value: +0 and -0 stay different"""
    a = np.ascontiguousarray(a)
    u = a.view(np.uint32 if a.dtype.itemsize == 4 else np.uint64).copy()
    u[np.isnan(a)] = 0x7fc00000 if a.dtype.itemsize == 4 else 0x7ff8000000000000
    return u


def planes_equal(a, b):
    """bit-level equality (the sign of zero included) that treats any NaN as equal to any NaN"""
    return count_mismatch(a, b) == 0


def count_mismatch(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return max(a.size, b.size)
    if a.dtype.kind == "f":
        return int(np.count_nonzero(_bits(a) != _bits(b)))
    return int(np.count_nonzero(a != b))


# numpy restatements of the host-side steps around the path (point cloud, lens undistortion) live with the other oracles
import importlib.util as _ilu
_spec = _ilu.spec_from_file_location("remode_host_steps", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "host_steps.py"))
_host_steps = _ilu.module_from_spec(_spec)
_spec.loader.exec_module(_host_steps)
point_cloud, undistort_maps, remap_u8 = _host_steps.point_cloud, _host_steps.undistort_maps, _host_steps.remap_u8
