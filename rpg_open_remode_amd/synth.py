"""Synthetic 'over-table' sequence generator (ctypes wrapper around csrc/synth.cpp).

Stands in for the reference's test dataset (test/dataset_main.cpp:37-52), which is not
available offline.  Conventions are the reference's: 8-bit gray frames, pose = T_world_cam
(3x4 row-major [R|t], se3.cuh:72-77), depth = range along the pixel ray in metres.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "librmd_synth.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python __graft_entry__.py` (build()) first")
        lib = ctypes.CDLL(path)
        lib.rmd_synth_pose.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]
        lib.rmd_synth_pose.restype = None
        lib.rmd_synth_render.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_double] * 4 + [
            ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
        lib.rmd_synth_render.restype = ctypes.c_int
        lib.rmd_synth_set_threads.argtypes = [ctypes.c_int]
        lib.rmd_synth_set_threads.restype = None
        lib.rmd_synth_set_threads(effective_cpus())
        _LIB = lib
    return _LIB


def effective_cpus():
    """CPUs this process may really use: the cgroup CPU quota if there is one (a container that sees 256 cores may be limited to
    16 CPUs' worth of time per period -- spinning up all visible cores gets it throttled), else the affinity mask / core count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except Exception:
            continue
    return max(1, n)


def intrinsics(width, height):
    """ICL-style intrinsics of test/dataset_main.cpp:37 scaled to the image size: (fx, fy, cx, cy)."""
    return (481.2 * width / 640.0, -480.0 * height / 480.0, (width - 1) / 2.0, (height - 1) / 2.0)


def pose(frame, seed=0):
    """T_world_cam of `frame`, float64 3x4."""
    T = np.zeros(12, np.float64)
    _lib().rmd_synth_pose(int(frame), int(seed), T.ctypes.data)
    return T.reshape(3, 4)


def invert_pose(T):
    """Inverse of a 3x4 [R|t] pose (float64)."""
    R, t = T[:, :3], T[:, 3]
    out = np.empty((3, 4), np.float64)
    out[:, :3] = R.T
    out[:, 3] = -R.T @ t
    return out


def render(width, height, T_world_cam, seed=0, want_range=True, K=None):
    """Returns (gray uint8 HxW, range float32 HxW or None)."""
    fx, fy, cx, cy = K if K is not None else intrinsics(width, height)
    T = np.ascontiguousarray(T_world_cam, np.float64).reshape(12)
    gray = np.empty((height, width), np.uint8)
    rng = np.empty((height, width), np.float32) if want_range else None
    rc = _lib().rmd_synth_render(width, height, fx, fy, cx, cy, T.ctypes.data, int(seed), gray.ctypes.data,
                                 rng.ctypes.data if want_range else None)
    if rc != 0:
        raise RuntimeError("rmd_synth_render failed")
    return gray, rng


def to_float_image(gray):
    """uint8 -> float32 exactly as depthmap.cpp:105 (cv::Mat::convertTo(CV_32F, 1.0f/255.0f))."""
    return (gray.astype(np.float32) * np.float32(1.0 / 255.0)).astype(np.float32)


class Sequence:
    """frames[0] is the reference view; every frame: (float image, T_curr_world float32[12], range or None)."""

    def __init__(self, width, height, n_frames, seed=0):
        self.width, self.height, self.n_frames, self.seed = width, height, n_frames, seed
        self.K = intrinsics(width, height)
        self.images = []
        self.gray = []
        self.T_curr_world = []
        self.T_world_cam = []
        self.range0 = None
        for k in range(n_frames):
            T = pose(k, seed)
            gray, rng = render(width, height, T, seed, want_range=(k == 0), K=self.K)
            if k == 0:
                self.range0 = rng
            self.gray.append(gray)
            self.images.append(to_float_image(gray))
            self.T_world_cam.append(T)
            self.T_curr_world.append(np.ascontiguousarray(invert_pose(T).astype(np.float32).reshape(12)))
        self.min_depth = float(self.range0.min())
        self.max_depth = float(self.range0.max())
