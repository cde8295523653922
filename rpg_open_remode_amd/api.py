"""Host-side mirror of the reference's C++ interface for the depth-filter path, over the C ABI.

Class and method names, argument order and error behaviour follow the reference so that the
parity tests read like the reference's own tests (test/seed_matrix_test.cpp, test/reduction_test.cpp,
test/dataset_main.cpp):

    rmd::PinholeCamera      include/rmd/pinhole_camera.cuh:27-63
    rmd::SE3<float>         include/rmd/se3.cuh:27-168
    rmd::DeviceImage<T>     include/rmd/device_image.cuh:34-180
    rmd::SeedMatrix         include/rmd/seed_matrix.cuh:45-109
    rmd::DepthmapDenoiser   include/rmd/depthmap_denoiser.cuh:27-54
    rmd::ImageReducer<T>    include/rmd/reduction.cuh:26-62
    rmd::Depthmap           include/rmd/depthmap.h:34-129   (numpy arrays instead of cv::Mat)
    rmd::checkCudaDevice    include/rmd/check_cuda_device.cuh:24
    SeedMatrixBatch         (not in the reference) several SeedMatrix objects whose update() calls are issued as one launch pair

Everything computes on the GPU through librmd_hip.so; nothing here falls back to the CPU.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import RmdHipError, check

__all__ = ["PinholeCamera", "SE3", "DeviceImage", "SeedMatrix", "SeedMatrixBatch", "DepthmapDenoiser", "ImageReducer", "Depthmap",
           "ConvergenceStates", "checkCudaDevice", "RmdHipError"]

PLANE_MU, PLANE_SIGMA_SQ, PLANE_A, PLANE_B, PLANE_CONVERGENCE = 0, 1, 2, 3, 4
PLANE_SUM_TEMPL, PLANE_CONST_TEMPL_DENOM, PLANE_EPIPOLAR_MATCHES, PLANE_REF_IMG, PLANE_CURR_IMG = 5, 6, 7, 8, 9
KIND_F32, KIND_I32, KIND_F32X2 = 0, 1, 2
OPT_MATCHER, OPT_TIMING, OPT_COLLECT_STATS, OPT_LAZY_FINALIZE, OPT_UNIT_TARGET = 0, 1, 2, 4, 7  # include/rmd_hip.h: RMD_HIP_OPT_*
OPT_INJECT_FAULT = 9  # test hook, see include/rmd_hip.h
PUBLISH_DEPTH, PUBLISH_CLOUD, PUBLISH_CONVERGENCE_BGR, PUBLISH_CONVERGENCE = 1, 2, 4, 8  # include/rmd_hip.h: RMD_HIP_PUBLISH_*

# process-wide settings of the host side (include/rmd_hip.h: RMD_HIP_TUNE_*; the environment presets them: RMD_HIP_<NAME>)
TUNE_HOST_FRAMES, TUNE_BATCH_GROUPS, TUNE_AHEAD_WGS, TUNE_PACK_BACKOFF, TUNE_FLOAT_AS_BYTES, TUNE_COPY_THREADS, TUNE_FUSED_INGEST, TUNE_INGEST_PROFILE, TUNE_HOST_WAIT, TUNE_RING_DEPTH, TUNE_COPY_STREAMS, TUNE_COPY_ENGINES = range(12)
HOST_FRAMES_DEFAULT, HOST_FRAMES_STAGED, HOST_FRAMES_STAGED_AHEAD, HOST_FRAMES_INPLACE, HOST_FRAMES_INPLACE_AHEAD = -1, 0, 1, 2, 3


def setTunable(tunable, value):
    """rmd_hip_set_tunable: a process-wide setting of the host side; handles created afterwards pick it up"""
    check(_lib.lib().rmd_hip_set_tunable(int(tunable), int(value)))


def getTunable(tunable):
    v = ctypes.c_int()
    check(_lib.lib().rmd_hip_get_tunable(int(tunable), ctypes.byref(v)))
    return v.value
MAX_BATCH = 24  # sequences one SeedMatrixBatch can hold (rmdk::MAX_BATCH: three stream groups of up to eight)
MATCHER_PIXEL, MATCHER_PIPELINE = 0, 3
STAGE_SEED_INIT, STAGE_UPDATE, STAGE_COUNT = 0, 1, 2
DENOISE_OPT_TIMING, DENOISE_OPT_ITERS_PER_LAUNCH, DENOISE_OPT_GEOMETRY = 1, 2, 3


class ConvergenceStates:  # seed_matrix.cuh:33-41
    UPDATE, CONVERGED, BORDER, DIVERGED, NO_MATCH, NOT_VISIBLE = range(6)


def checkCudaDevice(device=0, verbose=False):
    """check_cuda_device.cu:23-117: enumerate devices, select `device` (the reference's --device=N)."""
    L = _lib.lib()
    n = ctypes.c_int(0)
    if L.rmd_hip_device_count(ctypes.byref(n)) != _lib.OK:
        if verbose:
            print("ERROR: no HIP-capable device found.")
        return False
    if verbose:
        buf = ctypes.create_string_buffer(256)
        for d in range(n.value):
            L.rmd_hip_device_name(d, buf, 256)
            print(f"Device {d} - {buf.value.decode()}")
    if L.rmd_hip_set_device(int(device)) != _lib.OK:
        if verbose:
            print(f"ERROR: invalid device ID specified. Please specify a value in [0, {n.value - 1}].")
        return False
    return True


class PinholeCamera:
    def __init__(self, fx=0.0, fy=0.0, cx=0.0, cy=0.0):
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)


class SE3:
    """3x4 row-major [R|t], float32 (se3.cuh).  Host arithmetic in float32, in the reference's order."""

    def __init__(self, *args):
        self.data = np.zeros(12, np.float32)
        if len(args) == 7:  # (qw, qx, qy, qz, tx, ty, tz), se3.cuh:38-66
            f = np.float32
            qw, qx, qy, qz, tx, ty, tz = (f(a) for a in args)
            x, y, z = f(2) * qx, f(2) * qy, f(2) * qz
            wx, wy, wz = x * qw, y * qw, z * qw
            xx, xy, xz = x * qx, y * qx, z * qx
            yy, yz, zz = y * qy, z * qy, z * qz
            d = self.data
            d[0], d[1], d[2] = f(1) - (yy + zz), xy - wz, xz + wy
            d[4], d[5], d[6] = xy + wz, f(1) - (xx + zz), yz - wx
            d[8], d[9], d[10] = xz - wy, yz + wx, f(1) - (xx + yy)
            d[3], d[7], d[11] = tx, ty, tz
        elif len(args) == 2:  # (r[9] row-major, t[3]), se3.cuh:68-77
            r, t = np.asarray(args[0], np.float32).reshape(3, 3), np.asarray(args[1], np.float32).reshape(3)
            m = self.data.reshape(3, 4)
            m[:, :3], m[:, 3] = r, t
        elif len(args) == 1:
            self.data[:] = np.asarray(args[0], np.float32).reshape(12)
        elif len(args) != 0:
            raise TypeError("SE3(): 0, 1, 2 or 7 arguments")

    def inv(self):  # se3.cuh:78-95
        d, r = self.data, SE3()
        o = r.data
        for i in range(3):
            for j in range(3):
                o[4 * i + j] = d[4 * j + i]
            o[4 * i + 3] = -d[i] * d[3] - d[4 + i] * d[7] - d[8 + i] * d[11]
        return r

    def __mul__(self, rhs):  # se3.cuh:144-162
        l, r, out = self.data, rhs.data, SE3()
        o = out.data
        for row in range(3):
            l0, l1, l2, lt = l[4 * row], l[4 * row + 1], l[4 * row + 2], l[4 * row + 3]
            for col in range(3):
                o[4 * row + col] = l0 * r[col] + l1 * r[4 + col] + l2 * r[8 + col]
            o[4 * row + 3] = lt + l0 * r[3] + l1 * r[7] + l2 * r[11]
        return out

    def getTranslation(self):
        return self.data[[3, 7, 11]].copy()

    def __str__(self):  # matrix.cuh:56-67: setprecision(9), a space after every element, one line per row
        return "".join("".join(f"{float(v):.9g} " for v in self.data[4 * r:4 * r + 4]) + "\n" for r in range(3))


def _as_pose(T):
    if isinstance(T, SE3):
        T = T.data
    if type(T) is np.ndarray and T.dtype == np.float32 and T.size == 12 and T.flags.c_contiguous:
        return T.reshape(12)  # a view (the per-frame calls of a streaming host: no temporary)
    return np.ascontiguousarray(T, np.float32).reshape(12)


def _ptr(a):
    """address of an ndarray's data (ndarray.ctypes builds a helper object on every access: microseconds per frame)"""
    return a.__array_interface__["data"][0]


_KIND_OF = {np.dtype(np.float32): KIND_F32, np.dtype(np.int32): KIND_I32}


def computeUndistortionMap(width, height, fx, fy, cx, cy, k1, k2, r1, r2):
    """cv::initUndistortRectifyMap(K, (k1, k2, r1, r2), I, K, (width, height), CV_16SC2) as restated in the library (host code):
    (map1 int16 HxWx2 (x, y), map2 uint16 HxW)."""
    m1 = np.zeros((height, width, 2), np.int16)
    m2 = np.zeros((height, width), np.uint16)
    check(_lib.lib().rmd_hip_compute_undistortion_map(int(width), int(height), float(fx), float(fy), float(cx), float(cy), float(k1),
                                                      float(k2), float(r1), float(r2), m1.ctypes.data, m2.ctypes.data))
    return m1, m2


class DeviceImage:
    """rmd::DeviceImage<T>: a pitched 2-D device buffer.  dtype float32, int32, or 'float2'."""

    def __init__(self, width, height, dtype=np.float32, _view=None):
        self._owns = _view is None
        if _view is not None:
            self.ptr = _view
        else:
            kind = KIND_F32X2 if dtype == "float2" else _KIND_OF[np.dtype(dtype)]
            h = ctypes.c_void_p()
            check(_lib.lib().rmd_hip_image_create(kind, int(width), int(height), ctypes.byref(h)))
            self.ptr = h.value
        k, w, hh = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        pitch, stride, data = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_void_p()
        check(_lib.lib().rmd_hip_image_info(self.ptr, ctypes.byref(k), ctypes.byref(w), ctypes.byref(hh),
                                            ctypes.byref(pitch), ctypes.byref(stride), ctypes.byref(data)))
        self.kind, self.width, self.height = k.value, w.value, hh.value
        self.pitch, self.stride, self.data = pitch.value, stride.value, data.value

    def _host_shape(self):
        return (self.height, self.width, 2) if self.kind == KIND_F32X2 else (self.height, self.width)

    def _host_dtype(self):
        return np.int32 if self.kind == KIND_I32 else np.float32

    def setDevData(self, host):  # device_image.cuh:93-105
        arr = np.ascontiguousarray(host, self._host_dtype())
        assert arr.shape == self._host_shape(), (arr.shape, self._host_shape())
        check(_lib.lib().rmd_hip_image_upload(self.ptr, arr.ctypes.data))

    def getDevData(self):  # device_image.cuh:109-121
        out = np.empty(self._host_shape(), self._host_dtype())
        check(_lib.lib().rmd_hip_image_download(self.ptr, out.ctypes.data))
        return out

    def zero(self):  # device_image.cuh:141-151
        check(_lib.lib().rmd_hip_image_zero(self.ptr))

    def assign(self, other):  # operator=, device_image.cuh:154-171
        check(_lib.lib().rmd_hip_image_copy(self.ptr, other.ptr))
        return self

    def close(self):
        if self._owns and self.ptr:
            _lib.lib().rmd_hip_image_destroy(self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ImageReducer:
    """rmd::ImageReducer<T>.  The launch-shape arguments of the reference constructor are accepted and ignored."""

    def __init__(self, num_threads_per_block=None, num_blocks_per_grid=None):
        pass

    def sum(self, img, stride=None, width=None, height=None):
        """sum(const DeviceImage<T>&) (reduction.cu:132-139), or -- with stride / width / height -- the raw-pointer form
        sum(const T*, stride, width, height) (:81-130) on a device pointer (an int).  float or int images."""
        L = _lib.lib()
        if stride is not None:
            out = ctypes.c_float()
            check(L.rmd_hip_reduce_sum_f32_raw(int(img), int(stride), int(width), int(height), ctypes.byref(out)))
            return float(out.value)
        if img.kind == KIND_I32:
            out = ctypes.c_int()
            check(L.rmd_hip_reduce_sum_i32(img.ptr, ctypes.byref(out)))
            return int(out.value)
        out = ctypes.c_float()
        check(L.rmd_hip_reduce_sum_f32(img.ptr, ctypes.byref(out)))
        return float(out.value)

    def sumIntRaw(self, dev_ptr, stride, width, height):  # ImageReducer<int>::sum(const int*, ...)
        out = ctypes.c_int()
        check(_lib.lib().rmd_hip_reduce_sum_i32_raw(int(dev_ptr), int(stride), int(width), int(height), ctypes.byref(out)))
        return int(out.value)

    def countEqual(self, img, value, stride=None, width=None, height=None):
        """countEqual(const DeviceImage<int>&, value) (reduction.cu:175-183) or the raw-pointer form (:145-173)"""
        out = ctypes.c_size_t()
        if stride is not None:
            check(_lib.lib().rmd_hip_reduce_count_eq_i32_raw(int(img), int(stride), int(width), int(height), int(value), ctypes.byref(out)))
        else:
            check(_lib.lib().rmd_hip_reduce_count_eq_i32(img.ptr, int(value), ctypes.byref(out)))
        return int(out.value)


class PinnedFrames:
    """count frames of height x width bytes in ONE block of pinned host memory (rmd_hip_host_alloc): .frames[i] are numpy views a producer writes
    into and SeedMatrix.updateU8Pinned reads from without a copy"""

    def __init__(self, count, height, width):
        self.ptr = ctypes.c_void_p()
        self.nbytes = int(count) * int(height) * int(width)
        check(_lib.lib().rmd_hip_host_alloc(ctypes.byref(self.ptr), self.nbytes))
        buf = (ctypes.c_ubyte * self.nbytes).from_address(self.ptr.value)
        self.block = np.frombuffer(buf, np.uint8).reshape(count, height, width)
        self.frames = [self.block[i] for i in range(count)]

    def close(self):
        if self.ptr:
            self.frames, self.block = [], None
            _lib.lib().rmd_hip_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SeedMatrix:
    def __init__(self, width, height, cam, patch_side=5, max_extent=100, _member_of=None, _ptr=None):
        self.width, self.height, self.patch_side = int(width), int(height), int(patch_side)
        self._batch = _member_of  # a member of a SeedMatrixBatch is owned by the batch; this wrapper keeps the batch alive
        if _member_of is not None:
            self._handle = _ptr
            return
        h = ctypes.c_void_p()
        check(_lib.lib().rmd_hip_seeds_create(self.width, self.height, cam.fx, cam.fy, cam.cx, cam.cy, int(patch_side),
                                              int(max_extent), ctypes.byref(h)))
        self._handle = h.value

    @property
    def ptr(self):
        """the rmd_hip_seeds_t* of this object.  A batch member's handle dies with its batch (rmd_hip_batch_destroy releases the members): a
        wrapper that outlives an explicit SeedMatrixBatch.close() raises instead of handing a dangling pointer to the library."""
        if self._batch is not None and not self._batch.ptr:
            raise RmdHipError(_lib.ERR_INVALID_ARG, "this SeedMatrix was a member of a batch that has been closed")
        return self._handle

    def close(self):
        if getattr(self, "_handle", None) and self._batch is None:
            _lib.lib().rmd_hip_seeds_destroy(self._handle)
        self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- seed_matrix.cu:87-158 ---
    def setReferenceImage(self, host_ref_img, T_curr_world, min_depth, max_depth):
        img = np.ascontiguousarray(host_ref_img, np.float32)
        assert img.shape == (self.height, self.width)
        T = _as_pose(T_curr_world)
        check(_lib.lib().rmd_hip_seeds_set_reference(self.ptr, img.ctypes.data, T.ctypes.data, float(min_depth), float(max_depth)))
        return True

    def update(self, host_curr_img, T_curr_world):
        img = np.ascontiguousarray(host_curr_img, np.float32)
        assert img.shape == (self.height, self.width)
        T = _as_pose(T_curr_world)
        check(_lib.lib().rmd_hip_seeds_update(self.ptr, _ptr(img), _ptr(T)))
        return True

    def setReferenceImageU8(self, gray_u8, T_curr_world, min_depth, max_depth):
        img = np.ascontiguousarray(gray_u8, np.uint8)
        assert img.shape == (self.height, self.width)
        T = _as_pose(T_curr_world)
        check(_lib.lib().rmd_hip_seeds_set_reference_u8(self.ptr, img.ctypes.data, T.ctypes.data, float(min_depth), float(max_depth)))
        return True

    def updateU8(self, gray_u8, T_curr_world):
        img = np.ascontiguousarray(gray_u8, np.uint8)
        assert img.shape == (self.height, self.width)
        T = _as_pose(T_curr_world)
        check(_lib.lib().rmd_hip_seeds_update_u8(self.ptr, _ptr(img), _ptr(T)))
        return True

    def updateU8Pinned(self, pinned_gray_u8, T_curr_world):
        """rmd_hip_seeds_update_u8_pinned: an 8-bit frame the caller KEEPS in pinned host memory (a view of a PinnedFrames block) is read by the copy
        engine where it lies; it must stay unchanged until pinnedFramesDone() has reached the ticket this returns"""
        img = pinned_gray_u8
        assert img.dtype == np.uint8 and img.shape == (self.height, self.width) and img.flags["C_CONTIGUOUS"]
        T = _as_pose(T_curr_world)
        ticket = ctypes.c_ulonglong()
        check(_lib.lib().rmd_hip_seeds_update_u8_pinned(self.ptr, img.ctypes.data, _ptr(T), ctypes.byref(ticket)))
        return int(ticket.value)

    def pinnedFramesDone(self):
        out = ctypes.c_ulonglong()
        check(_lib.lib().rmd_hip_seeds_pinned_frames_done(self.ptr, ctypes.byref(out)))
        return int(out.value)

    def setReferenceImageDevice(self, dev_ptr, stride_elems, T_curr_world, min_depth, max_depth):
        T = _as_pose(T_curr_world)
        check(_lib.lib().rmd_hip_seeds_set_reference_device(self.ptr, dev_ptr, int(stride_elems), T.ctypes.data,
                                                            float(min_depth), float(max_depth)))
        return True

    def updateDevice(self, dev_ptr, stride_elems, T_curr_world):
        T = _as_pose(T_curr_world)
        check(_lib.lib().rmd_hip_seeds_update_device(self.ptr, dev_ptr, int(stride_elems), _ptr(T)))
        return True

    # --- downloads, seed_matrix.cu:160-168,205-230 ---
    def download(self, plane):
        if plane == PLANE_CONVERGENCE:
            out = np.empty((self.height, self.width), np.int32)
        elif plane == PLANE_EPIPOLAR_MATCHES:
            out = np.empty((self.height, self.width, 2), np.float32)
        else:
            out = np.empty((self.height, self.width), np.float32)
        check(_lib.lib().rmd_hip_seeds_download(self.ptr, int(plane), out.ctypes.data))
        return out

    def downloadDepthmap(self): return self.download(PLANE_MU)
    def downloadConvergence(self): return self.download(PLANE_CONVERGENCE)
    def downloadSigmaSq(self): return self.download(PLANE_SIGMA_SQ)
    def downloadA(self): return self.download(PLANE_A)
    def downloadB(self): return self.download(PLANE_B)
    def downloadSumTempl(self): return self.download(PLANE_SUM_TEMPL)
    def downloadConstTemplDenom(self): return self.download(PLANE_CONST_TEMPL_DENOM)
    def downloadEpipolarMatches(self): return self.download(PLANE_EPIPOLAR_MATCHES)

    def upload(self, plane, arr):
        arr = np.ascontiguousarray(arr, np.float32)
        assert arr.shape == (self.height, self.width)
        check(_lib.lib().rmd_hip_seeds_upload(self.ptr, int(plane), arr.ctypes.data))

    def state(self):
        return {p: self.download(p) for p in range(8)}

    # --- getters, seed_matrix.cu:170-203 ---
    def _plane(self, plane):
        v = ctypes.c_void_p()
        check(_lib.lib().rmd_hip_seeds_plane(self.ptr, int(plane), ctypes.byref(v)))
        img = DeviceImage(0, 0, _view=v.value)
        img._keepalive = self
        return img

    def getMu(self): return self._plane(PLANE_MU)
    def getSigmaSq(self): return self._plane(PLANE_SIGMA_SQ)
    def getA(self): return self._plane(PLANE_A)
    def getB(self): return self._plane(PLANE_B)
    def getConvergence(self): return self._plane(PLANE_CONVERGENCE)

    def getConvergedCount(self):
        out = ctypes.c_size_t()
        check(_lib.lib().rmd_hip_seeds_converged_count(self.ptr, ctypes.byref(out)))
        return int(out.value)

    def initUndistortionMap(self, k1, k2, r1, r2):
        """Lens undistortion of the 8-bit frames of setReferenceImageU8 / updateU8 (Depthmap::initUndistortionMap,
        depthmap.cpp:45-61: cv::initUndistortRectifyMap + cv::remap, restated)."""
        check(_lib.lib().rmd_hip_seeds_init_undistortion_map(self.ptr, float(k1), float(k2), float(r1), float(r2)))

    def undistortionMap(self):
        """(map1 int16 HxWx2 (x, y), map2 uint16 HxW) as computed by initUndistortionMap."""
        m1 = np.zeros((self.height, self.width, 2), np.int16)
        m2 = np.zeros((self.height, self.width), np.uint16)
        check(_lib.lib().rmd_hip_seeds_undistortion_map(self.ptr, m1.ctypes.data, m2.ctypes.data))
        return m1, m2

    def pointCloud(self, depth=None):
        """World-frame XYZI points of the CONVERGED seeds, row-major pixel order (Publisher::publishPointCloud,
        publisher.cpp:54-104, computed on the device).  depth: a DeviceImage (e.g. DepthmapDenoiser.result()) or None for mu.
        Returns an (N, 4) float32 array."""
        cap = self.width * self.height
        out = np.empty((cap, 4), np.float32)
        n = ctypes.c_size_t()
        check(_lib.lib().rmd_hip_seeds_point_cloud(self.ptr, depth.ptr if depth is not None else None, out.ctypes.data, cap, ctypes.byref(n)))
        return out[:int(n.value)].copy()

    def convergenceBGR8(self, out=None):
        """The coloured convergence map of Publisher::publishConvergenceMap (publisher.cpp:112-147), computed on the device: (H, W, 3)
        uint8, the reference image as gray BGR with blue = 255 where CONVERGED, red = 255 where DIVERGED."""
        if out is None:
            out = np.empty((self.height, self.width, 3), np.uint8)
        check(_lib.lib().rmd_hip_seeds_convergence_bgr8(self.ptr, _ptr(out)))
        return out

    def getDistFromRef(self):
        out = ctypes.c_float()
        check(_lib.lib().rmd_hip_seeds_dist_from_ref(self.ptr, ctypes.byref(out)))
        return float(out.value)

    def stagedFrames(self):
        """(frames staged on copy engines addressed directly, frames staged on the copy stream) so far: which way RMD_HIP_TUNE_COPY_ENGINES really went"""
        out = (ctypes.c_ulonglong * 2)()
        check(_lib.lib().rmd_hip_seeds_staged_frames(self.ptr, out))
        return int(out[0]), int(out[1])

    def publishAsync(self, what, depth_range=0.0, lam=0.5, iterations=200):
        """rmd_hip_seeds_publish_async: snapshot the state and queue its publication products (PUBLISH_* bits: TV-L1 denoised depth map, point
        cloud of the converged seeds, coloured convergence map, int32 convergence plane) on the handle's second stream; returns the ticket
        at once -- setReferenceImage / update may follow immediately (DepthmapNode::denoiseAndPublishResults, depthmap_node.cpp:165-182)."""
        t = ctypes.c_int()
        check(_lib.lib().rmd_hip_seeds_publish_async(self.ptr, int(what), float(depth_range), float(lam), int(iterations), ctypes.byref(t)))
        return int(t.value)

    def peekPublication(self, wait=True):
        """rmd_hip_seeds_publish_peek: like collectPublication, but the arrays are VIEWS of the library's pinned buffers -- no copy -- valid until
        releasePublication() (which must follow); None when the oldest publication is still in flight and wait is False."""
        what, ticket, n_pts = ctypes.c_uint(), ctypes.c_int(), ctypes.c_size_t()
        d, x, b, c = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        rc = _lib.lib().rmd_hip_seeds_publish_peek(self.ptr, 1 if wait else 0, ctypes.byref(what), ctypes.byref(ticket), ctypes.byref(d), ctypes.byref(x),
                                                   ctypes.byref(n_pts), ctypes.byref(b), ctypes.byref(c))
        if rc == _lib.BUSY:
            return None
        check(rc)
        h, w, n = self.height, self.width, int(n_pts.value)

        def view(ptr, ctype, count, dtype, shape):
            return np.frombuffer((ctype * count).from_address(ptr), dtype).reshape(shape) if ptr else None
        return {"ticket": int(ticket.value), "what": int(what.value),
                "depth": view(d.value, ctypes.c_float, h * w, np.float32, (h, w)),
                "points": (view(x.value, ctypes.c_float, max(n, 1) * 4, np.float32, (max(n, 1), 4))[:n] if x.value else None),
                "bgr": view(b.value, ctypes.c_ubyte, h * w * 3, np.uint8, (h, w, 3)),
                "convergence": view(c.value, ctypes.c_int, h * w, np.int32, (h, w))}

    def releasePublication(self):
        check(_lib.lib().rmd_hip_seeds_publish_release(self.ptr))

    def collectPublication(self, wait=True):
        """rmd_hip_seeds_publish_collect: the OLDEST uncollected publication as a dict {"ticket", "what", "depth", "points", "bgr", "convergence"}
        (the products that were requested), or None when it is still in flight and wait is False.  Raises when there is none."""
        if not hasattr(self, "_pub_bufs"):
            n = self.width * self.height
            self._pub_bufs = (np.empty((self.height, self.width), np.float32), np.empty((n, 4), np.float32),
                              np.empty((self.height, self.width, 3), np.uint8), np.empty((self.height, self.width), np.int32))
        depth, pts, bgr, conv = self._pub_bufs
        what, ticket, n_pts = ctypes.c_uint(), ctypes.c_int(), ctypes.c_size_t()
        rc = _lib.lib().rmd_hip_seeds_publish_collect(self.ptr, 1 if wait else 0, ctypes.byref(what), ctypes.byref(ticket), depth.ctypes.data, pts.ctypes.data,
                                                      pts.shape[0], ctypes.byref(n_pts), bgr.ctypes.data, conv.ctypes.data)
        if rc == _lib.BUSY:
            return None
        check(rc)
        w = int(what.value)
        return {"ticket": int(ticket.value), "what": w,
                "depth": depth.copy() if w & PUBLISH_DEPTH else None, "points": pts[:int(n_pts.value)].copy() if w & PUBLISH_CLOUD else None,
                "bgr": bgr.copy() if w & PUBLISH_CONVERGENCE_BGR else None, "convergence": conv.copy() if w & PUBLISH_CONVERGENCE else None}

    # --- extras ---
    def sync(self):
        check(_lib.lib().rmd_hip_seeds_sync(self.ptr))

    def setOption(self, option, value):
        check(_lib.lib().rmd_hip_seeds_set_option(self.ptr, int(option), int(value)))

    def timing(self, stage):
        ms, n = ctypes.c_double(), ctypes.c_long()
        check(_lib.lib().rmd_hip_seeds_timing(self.ptr, int(stage), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def timingReset(self):
        check(_lib.lib().rmd_hip_seeds_timing_reset(self.ptr))

    def lastStats(self):
        out = np.zeros(3, np.int64)
        check(_lib.lib().rmd_hip_seeds_last_stats(self.ptr, out.ctypes.data))
        return {"live_seeds": int(out[0]), "steps": int(out[1]), "ncc_evals": int(out[2])}


    def traceDownload(self, frame):
        """Timeline of update `frame` of the tile pipeline since setOption(OPT_COLLECT_STATS, 2): an (n_tiles, 8) uint64 array, see
        rmd_hip_seeds_trace_download in include/rmd_hip.h (row t: setup tile t in word 2, search workgroup t in the other words)."""
        tiles = ((self.width + 15) // 16) * ((self.height + 15) // 16)
        n = tiles * 8
        out = np.zeros(n, np.uint64)
        written = ctypes.c_size_t()
        check(_lib.lib().rmd_hip_seeds_trace_download(self.ptr, int(frame), out.ctypes.data, n, ctypes.byref(written)))
        return out.reshape(-1, 8)

    frameTraceDownload = traceDownload  # (name used by the timeline tools)

    def lastDiagnosticsRaw(self):
        out = np.zeros(16, np.int64)
        check(_lib.lib().rmd_hip_seeds_last_diagnostics(self.ptr, out.ctypes.data))
        return [int(v) for v in out]

    def lastDiagnostics(self):
        out = np.zeros(16, np.int64)
        check(_lib.lib().rmd_hip_seeds_last_diagnostics(self.ptr, out.ctypes.data))
        names = ["live_seeds", "steps", "ncc_evals", "evals_lds", "evals_global", "evals_irregular", "max_tile_items",
                 "cycles_setup", "cycles_stage", "cycles_search", "_10", "max_wg_cycles", "tiles_with_work", "rounds",
                 "max_setup_cycles", "max_search_cycles"]
        return {n: int(v) for n, v in zip(names, out) if not n.startswith("_")}


class SeedMatrixBatch:
    """Several independent SeedMatrix objects of one size on one GPU whose update() calls are issued TOGETHER, as one launch pair
    (rmd_hip_batch_*; BASELINE configs[3]: independent sequences).  `batch[i]` is a full SeedMatrix -- setReferenceImage*, downloads,
    getters, getConvergedCount, pointCloud, use with DepthmapDenoiser -- whose update* methods are replaced by the batch's: per member
    the arithmetic and the results are those of that member stepped alone (seed_matrix.cu:120-158)."""

    def __init__(self, n, width, height, cam, patch_side=5, max_extent=100):
        self.n, self.width, self.height, self.patch_side = int(n), int(width), int(height), int(patch_side)
        h = ctypes.c_void_p()
        check(_lib.lib().rmd_hip_batch_create(self.n, self.width, self.height, cam.fx, cam.fy, cam.cx, cam.cy, int(patch_side), int(max_extent),
                                              ctypes.byref(h)))
        self.ptr = h.value
        self._T = np.zeros((self.n, 12), np.float32)  # per-step argument blocks, reused
        self._ptrs = (ctypes.c_void_p * self.n)()
        self._strides = (ctypes.c_size_t * self.n)()
        self._cam, self._max_extent = cam, max_extent
        self._member_ptrs = []
        for i in range(self.n):
            m = ctypes.c_void_p()
            check(_lib.lib().rmd_hip_batch_member(self.ptr, i, ctypes.byref(m)))
            self._member_ptrs.append(m.value)

    def __len__(self): return self.n

    def __getitem__(self, i):
        """member i as a SeedMatrix.  A fresh wrapper every time: it keeps the batch alive, the batch does not keep it (a reference cycle
        would leave destroyed-looking batches -- and their streams -- around until the garbage collector runs)"""
        if not self.ptr:
            raise RmdHipError(_lib.ERR_INVALID_ARG, "the batch has been closed")
        return SeedMatrix(self.width, self.height, self._cam, self.patch_side, self._max_extent, _member_of=self, _ptr=self._member_ptrs[i])

    @property
    def members(self):
        return [self[i] for i in range(self.n)]

    def close(self):
        if getattr(self, "ptr", None):
            _lib.lib().rmd_hip_batch_destroy(self.ptr)
            self.ptr = None
            self._member_ptrs = [None] * self.n

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _poses(self, poses):
        T = self._T
        for i, p in enumerate(poses):
            if p is not None:
                T[i] = _as_pose(p)
        return T

    def updateDevice(self, dev_ptrs, strides, poses):
        """dev_ptrs[i]: device address of member i's frame (None / 0: no frame for that member in this step)"""
        ptrs, st = self._ptrs, self._strides
        for i in range(self.n):
            ptrs[i] = int(dev_ptrs[i]) if dev_ptrs[i] else None
            st[i] = int(strides[i])
        T = self._poses(poses)
        check(_lib.lib().rmd_hip_batch_update_device(self.ptr, ptrs, st, _ptr(T)))
        return True

    def _host(self, fn, imgs, poses, dtype):
        keep = [None if im is None else np.ascontiguousarray(im, dtype) for im in imgs]
        ptrs = self._ptrs
        for i, im in enumerate(keep):
            assert im is None or im.shape == (self.height, self.width)
            ptrs[i] = None if im is None else _ptr(im)
        T = self._poses(poses)
        check(fn(self.ptr, ptrs, _ptr(T)))
        return True

    def updateU8(self, gray_frames, poses):
        return self._host(_lib.lib().rmd_hip_batch_update_u8, gray_frames, poses, np.uint8)

    def update(self, float_frames, poses):
        return self._host(_lib.lib().rmd_hip_batch_update, float_frames, poses, np.float32)

    def denoise(self, depth_ranges, lam, iterations, download=True):
        """DepthmapDenoiser::denoise for every member in ONE launch sequence (rmd_hip_batch_denoise): depth_ranges[i] = max_depth - min_depth of
        member i (setLargeSigmaSq).  Returns the n denoised maps (download=True) or None; denoiseResult(i) is member i's map on the device."""
        dr = np.ascontiguousarray(depth_ranges, np.float32).reshape(self.n)
        outs = [np.empty((self.height, self.width), np.float32) for _ in range(self.n)] if download else None
        ptrs = (ctypes.c_void_p * self.n)(*[_ptr(o) for o in outs]) if download else None
        check(_lib.lib().rmd_hip_batch_denoise(self.ptr, _ptr(dr), float(lam), int(iterations), ptrs))
        return outs

    def denoiseResult(self, i):
        v = ctypes.c_void_p()
        check(_lib.lib().rmd_hip_batch_denoise_result(self.ptr, int(i), ctypes.byref(v)))
        img = DeviceImage(0, 0, _view=v.value)
        img._keepalive = self
        return img

    def denoiseTiming(self):
        ms, n = ctypes.c_double(), ctypes.c_long()
        check(_lib.lib().rmd_hip_batch_denoise_timing(self.ptr, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def sync(self):
        check(_lib.lib().rmd_hip_batch_sync(self.ptr))

    def setOption(self, option, value):
        check(_lib.lib().rmd_hip_batch_set_option(self.ptr, int(option), int(value)))

    def timingReset(self):
        check(_lib.lib().rmd_hip_batch_timing_reset(self.ptr))

    def timing(self):
        ms, n = ctypes.c_double(), ctypes.c_long()
        check(_lib.lib().rmd_hip_batch_timing(self.ptr, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value


class DepthmapDenoiser:
    def __init__(self, width, height):
        self.width, self.height = int(width), int(height)
        h = ctypes.c_void_p()
        check(_lib.lib().rmd_hip_denoiser_create(self.width, self.height, ctypes.byref(h)))
        self.ptr = h.value

    def close(self):
        if getattr(self, "ptr", None):
            _lib.lib().rmd_hip_denoiser_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setLargeSigmaSq(self, depth_range):  # depthmap_denoiser.cu:226-229
        check(_lib.lib().rmd_hip_denoiser_set_large_sigma_sq(self.ptr, float(depth_range)))

    def denoise(self, mu, sigma_sq, a, b, lam, iterations, download=True):  # depthmap_denoiser.cu:179-224
        out = np.empty((self.height, self.width), np.float32) if download else None
        check(_lib.lib().rmd_hip_denoiser_denoise(self.ptr, mu.ptr, sigma_sq.ptr, a.ptr, b.ptr,
                                                  out.ctypes.data if download else None, float(lam), int(iterations)))
        return out

    def result(self):
        """Device-resident output of the last denoise() (a view; valid until the next denoise() of this object)."""
        v = ctypes.c_void_p()
        check(_lib.lib().rmd_hip_denoiser_result(self.ptr, ctypes.byref(v)))
        img = DeviceImage(0, 0, _view=v.value)
        img._keepalive = self
        return img

    def constants(self):
        out = np.zeros(4, np.float32)
        check(_lib.lib().rmd_hip_denoiser_constants(self.ptr, out.ctypes.data))
        return out

    def setOption(self, option, value):
        check(_lib.lib().rmd_hip_denoiser_set_option(self.ptr, int(option), int(value)))

    def timing(self):
        ms, n = ctypes.c_double(), ctypes.c_long()
        check(_lib.lib().rmd_hip_denoiser_timing(self.ptr, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value


class Depthmap:
    """rmd::Depthmap (depthmap.h:34-129, depthmap.cpp) with numpy uint8 images in place of cv::Mat.
    Constructor argument order is the reference's: (width, height, fx, cx, fy, cy)."""

    def __init__(self, width, height, fx, cx, fy, cy, patch_side=5, max_extent=100):
        self.width_, self.height_ = int(width), int(height)
        self.fx_, self.fy_, self.cx_, self.cy_ = fx, fy, cx, cy
        self.seeds_ = SeedMatrix(width, height, PinholeCamera(fx, fy, cx, cy), patch_side, max_extent)
        self.denoiser_ = DepthmapDenoiser(width, height)
        self.output_depth_32fc1_ = np.zeros((self.height_, self.width_), np.float32)
        self.output_convergence_int_ = np.zeros((self.height_, self.width_), np.int32)
        self.ref_img_8uc1_ = None
        self.T_world_ref_ = SE3()

    @staticmethod
    def _check_u8(img_8uc1):  # depthmap.cpp:95-106: the x(1/255) conversion runs on the device (undistortion is outside this path)
        img = np.asarray(img_8uc1)
        if img.dtype != np.uint8:
            raise TypeError("Depthmap expects 8-bit gray images (CV_8UC1)")
        return img

    def initUndistortionMap(self, k1, k2, r1, r2):  # depthmap.cpp:45-61
        self.seeds_.initUndistortionMap(k1, k2, r1, r2)
        self.is_distorted_ = True

    def setReferenceImage(self, img_curr, T_curr_world, min_depth, max_depth):  # depthmap.cpp:63-83
        self.denoiser_.setLargeSigmaSq(max_depth - min_depth)
        self.depth_range_ = float(np.float32(max_depth) - np.float32(min_depth))  # what setLargeSigmaSq was given: the publications' TV-L1 uses it too
        ret = self.seeds_.setReferenceImageU8(self._check_u8(img_curr), T_curr_world, min_depth, max_depth)
        if getattr(self, "is_distorted_", False):  # ref_img_undistorted_8uc1_ (depthmap.cpp:74-79): the undistorted frame, back in 8 bits
            self.ref_img_8uc1_ = np.rint(self.seeds_.download(PLANE_REF_IMG) * np.float32(255.0)).astype(np.uint8)
        else:
            self.ref_img_8uc1_ = np.array(img_curr, copy=True)
        self.T_world_ref_ = (T_curr_world if isinstance(T_curr_world, SE3) else SE3(T_curr_world)).inv()
        return ret

    def update(self, img_curr, T_curr_world):  # depthmap.cpp:85-93
        self.seeds_.updateU8(self._check_u8(img_curr), T_curr_world)

    def downloadDepthmap(self):
        self.output_depth_32fc1_ = self.seeds_.downloadDepthmap()

    def downloadDenoisedDepthmap(self, lam, iterations):  # depthmap.cpp:113-123
        self.output_depth_32fc1_ = self.denoiser_.denoise(self.seeds_.getMu(), self.seeds_.getSigmaSq(), self.seeds_.getA(),
                                                          self.seeds_.getB(), lam, iterations)

    def getDepthmap(self): return self.output_depth_32fc1_

    def downloadConvergenceMap(self):
        self.output_convergence_int_ = self.seeds_.downloadConvergence()

    def getConvergenceMap(self): return self.output_convergence_int_

    def downloadPointCloud(self, denoised=True):
        """(N, 4) XYZI points of the converged seeds from the depth map of the last downloadDenoisedDepthmap() (or from
        the raw depth estimate): what rmd::Publisher builds on the host from two downloaded images."""
        return self.seeds_.pointCloud(self.denoiser_.result() if denoised else None)
    def getReferenceImage(self): return self.ref_img_8uc1_
    def getConvergedCount(self): return self.seeds_.getConvergedCount()

    def convergenceBGR8(self):
        """what Publisher::publishConvergenceMap builds from getConvergenceMap() and getReferenceImage() (publisher.cpp:112-147), on the
        device (no int32 download, no host loop)"""
        return self.seeds_.convergenceBGR8()

    def publishAsync(self, what, lam=0.5, iterations=200):
        """downloadDenoisedDepthmap / downloadConvergenceMap / the publisher's point cloud and coloured map as ONE request that does not block
        (SeedMatrix.publishAsync): the products belong to the state at this moment; the next setReferenceImage / update may follow at once."""
        return self.seeds_.publishAsync(what, getattr(self, "depth_range_", 0.0), lam, iterations)

    def collectPublication(self, wait=True, views=False):
        """the oldest publication requested with publishAsync (None: still in flight and wait is False); refreshes the host mirrors the
        reference's getters return (getDepthmap, getConvergenceMap: ONE copy each out of the library's pinned buffers, and the copy is what the
        dict holds).  views=True: the other products (points, coloured map) are views of those buffers, valid until releasePublication() --
        which the caller then owes; views=False: everything is copied and the slot released here."""
        pub = self.seeds_.peekPublication(wait)
        if pub is not None:
            if pub["depth"] is not None:
                self.output_depth_32fc1_ = pub["depth"] = pub["depth"].copy()
            if pub["convergence"] is not None:
                self.output_convergence_int_ = pub["convergence"] = pub["convergence"].copy()
            if not views:
                for key in ("points", "bgr"):
                    if pub[key] is not None:
                        pub[key] = pub[key].copy()
                self.seeds_.releasePublication()
        return pub

    def releasePublication(self):
        self.seeds_.releasePublication()

    def getConvergedPercentage(self):  # depthmap.cpp:152-156
        return float(np.float32(self.getConvergedCount()) / np.float32(self.width_ * self.height_) * np.float32(100.0))

    def getDistFromRef(self): return self.seeds_.getDistFromRef()
    def getT_world_ref(self): return self.T_world_ref_
    def getWidth(self): return self.width_
    def getHeight(self): return self.height_
    def getFx(self): return self.fx_
    def getFy(self): return self.fy_
    def getCx(self): return self.cx_
    def getCy(self): return self.cy_
