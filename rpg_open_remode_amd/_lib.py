"""ctypes loader for librmd_hip.so (the C ABI declared in include/rmd_hip.h).

There is no CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "librmd_hip.so")
HEADER_PATH = os.path.join(ROOT, "include", "rmd_hip.h")

OK, ERR_INVALID_ARG, ERR_RUNTIME, ERR_NOT_READY, ERR_NO_DEVICE = 0, -1, -2, -3, -4
BUSY = 1  # rmd_hip_seeds_publish_collect: the oldest publication is still in flight (not an error)


class RmdHipError(RuntimeError):
    """Raised where the reference throws rmd::CudaException (cuda_exception.cuh:27-45)."""

    def __init__(self, code, message):
        super().__init__(f"rmd_hip error {code}: {message}")
        self.code = code


_c = ctypes
_p, _i, _f, _sz = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t
_pp = _c.POINTER(_c.c_void_p)

# name -> (restype, argtypes); must list every function of include/rmd_hip.h (checked by tests/test_abi.py)
SIGNATURES = {
    "rmd_hip_last_error": (_c.c_char_p, []),
    "rmd_hip_version": (_i, []),
    "rmd_hip_set_tunable": (_i, [_i, _i]),
    "rmd_hip_get_tunable": (_i, [_i, _c.POINTER(_i)]),
    "rmd_hip_device_count": (_i, [_c.POINTER(_i)]),
    "rmd_hip_set_device": (_i, [_i]),
    "rmd_hip_device_name": (_i, [_i, _c.c_char_p, _sz]),
    "rmd_hip_image_create": (_i, [_i, _i, _i, _pp]),
    "rmd_hip_image_destroy": (_i, [_p]),
    "rmd_hip_image_upload": (_i, [_p, _p]),
    "rmd_hip_image_download": (_i, [_p, _p]),
    "rmd_hip_image_zero": (_i, [_p]),
    "rmd_hip_image_copy": (_i, [_p, _p]),
    "rmd_hip_image_info": (_i, [_p, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_sz), _c.POINTER(_sz), _pp]),
    "rmd_hip_seeds_create": (_i, [_i, _i, _f, _f, _f, _f, _i, _i, _pp]),
    "rmd_hip_seeds_destroy": (_i, [_p]),
    "rmd_hip_seeds_set_reference": (_i, [_p, _p, _p, _f, _f]),
    "rmd_hip_seeds_update": (_i, [_p, _p, _p]),
    "rmd_hip_seeds_set_reference_device": (_i, [_p, _p, _sz, _p, _f, _f]),
    "rmd_hip_seeds_update_device": (_i, [_p, _p, _sz, _p]),
    "rmd_hip_seeds_set_reference_u8": (_i, [_p, _p, _p, _f, _f]),
    "rmd_hip_seeds_update_u8": (_i, [_p, _p, _p]),
    "rmd_hip_host_alloc": (_i, [_c.POINTER(_p), _sz]),
    "rmd_hip_host_free": (_i, [_p]),
    "rmd_hip_seeds_update_u8_pinned": (_i, [_p, _p, _p, _c.POINTER(_c.c_ulonglong)]),
    "rmd_hip_seeds_pinned_frames_done": (_i, [_p, _c.POINTER(_c.c_ulonglong)]),
    "rmd_hip_seeds_download": (_i, [_p, _i, _p]),
    "rmd_hip_seeds_upload": (_i, [_p, _i, _p]),
    "rmd_hip_seeds_plane": (_i, [_p, _i, _pp]),
    "rmd_hip_seeds_converged_count": (_i, [_p, _c.POINTER(_sz)]),
    "rmd_hip_seeds_convergence_bgr8": (_i, [_p, _p]),
    "rmd_hip_seeds_dist_from_ref": (_i, [_p, _c.POINTER(_f)]),
    "rmd_hip_seeds_staged_frames": (_i, [_p, _c.POINTER(_c.c_ulonglong)]),
    "rmd_hip_seeds_sync": (_i, [_p]),
    "rmd_hip_seeds_set_option": (_i, [_p, _i, _i]),
    "rmd_hip_seeds_timing": (_i, [_p, _i, _c.POINTER(_c.c_double), _c.POINTER(_c.c_long)]),
    "rmd_hip_seeds_timing_reset": (_i, [_p]),
    "rmd_hip_seeds_last_stats": (_i, [_p, _p]),
    "rmd_hip_seeds_last_diagnostics": (_i, [_p, _p]),
    "rmd_hip_seeds_trace_download": (_i, [_p, _i, _p, _sz, _c.POINTER(_sz)]),
    "rmd_hip_seeds_point_cloud": (_i, [_p, _p, _p, _sz, _c.POINTER(_sz)]),
    "rmd_hip_seeds_publish_async": (_i, [_p, _c.c_uint, _f, _f, _i, _c.POINTER(_i)]),
    "rmd_hip_seeds_publish_collect": (_i, [_p, _i, _c.POINTER(_c.c_uint), _c.POINTER(_i), _p, _p, _sz, _c.POINTER(_sz), _p, _p]),
    "rmd_hip_seeds_publish_peek": (_i, [_p, _i, _c.POINTER(_c.c_uint), _c.POINTER(_i), _pp, _pp, _c.POINTER(_sz), _pp, _pp]),
    "rmd_hip_seeds_publish_release": (_i, [_p]),
    "rmd_hip_seeds_init_undistortion_map": (_i, [_p, _f, _f, _f, _f]),
    "rmd_hip_seeds_undistortion_map": (_i, [_p, _p, _p]),
    "rmd_hip_compute_undistortion_map": (_i, [_i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _p, _p]),
    "rmd_hip_batch_create": (_i, [_i, _i, _i, _f, _f, _f, _f, _i, _i, _pp]),
    "rmd_hip_batch_destroy": (_i, [_p]),
    "rmd_hip_batch_size": (_i, [_p, _c.POINTER(_i)]),
    "rmd_hip_batch_member": (_i, [_p, _i, _pp]),
    "rmd_hip_batch_update_device": (_i, [_p, _p, _p, _p]),
    "rmd_hip_batch_update_u8": (_i, [_p, _p, _p]),
    "rmd_hip_batch_update": (_i, [_p, _p, _p]),
    "rmd_hip_batch_sync": (_i, [_p]),
    "rmd_hip_batch_set_option": (_i, [_p, _i, _i]),
    "rmd_hip_batch_timing_reset": (_i, [_p]),
    "rmd_hip_batch_timing": (_i, [_p, _c.POINTER(_c.c_double), _c.POINTER(_c.c_long)]),
    "rmd_hip_batch_denoise": (_i, [_p, _p, _f, _i, _p]),
    "rmd_hip_batch_denoise_result": (_i, [_p, _i, _pp]),
    "rmd_hip_batch_denoise_timing": (_i, [_p, _c.POINTER(_c.c_double), _c.POINTER(_c.c_long)]),
    "rmd_hip_denoiser_create": (_i, [_i, _i, _pp]),
    "rmd_hip_denoiser_destroy": (_i, [_p]),
    "rmd_hip_denoiser_set_large_sigma_sq": (_i, [_p, _f]),
    "rmd_hip_denoiser_denoise": (_i, [_p, _p, _p, _p, _p, _p, _f, _i]),
    "rmd_hip_denoiser_result": (_i, [_p, _pp]),
    "rmd_hip_denoiser_constants": (_i, [_p, _p]),
    "rmd_hip_denoiser_set_option": (_i, [_p, _i, _i]),
    "rmd_hip_denoiser_timing": (_i, [_p, _c.POINTER(_c.c_double), _c.POINTER(_c.c_long)]),
    "rmd_hip_reduce_sum_f32": (_i, [_p, _c.POINTER(_f)]),
    "rmd_hip_reduce_count_eq_i32": (_i, [_p, _i, _c.POINTER(_sz)]),
    "rmd_hip_reduce_sum_i32": (_i, [_p, _c.POINTER(_i)]),
    "rmd_hip_reduce_sum_f32_raw": (_i, [_p, _sz, _sz, _sz, _c.POINTER(_f)]),
    "rmd_hip_reduce_sum_i32_raw": (_i, [_p, _sz, _sz, _sz, _c.POINTER(_i)]),
    "rmd_hip_reduce_count_eq_i32_raw": (_i, [_p, _sz, _sz, _sz, _i, _c.POINTER(_sz)]),
    "rmd_hip_selftest_wave_primitives": (_i, [_c.POINTER(_i)]),
    "rmd_hip_selftest_pack_float_frame": (_i, [_c.c_void_p, _i, _i, _i, _c.c_void_p, _c.POINTER(_i)]),
    "rmd_hip_math_eval": (_i, [_i, _p, _p, _p, _p, _sz]),
}

_LIB = None


def header_functions():
    """Names of all functions declared in include/rmd_hip.h."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rmd_hip_[a-z0-9_]+)\s*\(", text)))


def lib():
    global _LIB
    if _LIB is None:
        # RMD_HIP_LIB: another build of the SAME library (A/B variants under build_ab/, tools/ab_make.sh) -- a measurement switch, not a
        # fallback: the file must exist and export every symbol of the header like the product build
        path = os.environ.get("RMD_HIP_LIB") or LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found: the HIP extension has not been built "
                "(run `python -m rpg_open_remode_amd.build`); there is no CPU fallback")
        L = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _LIB = L
    return _LIB


def check(rc):
    if rc != OK:
        msg = lib().rmd_hip_last_error()
        raise RmdHipError(rc, msg.decode() if msg else "")
    return rc
