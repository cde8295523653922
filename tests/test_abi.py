"""CPU-only checks of the drop-in boundary: librmd_hip.so loads, exports every symbol of include/rmd_hip.h,
and reports errors the way the header says (no compute calls here)."""
import ctypes
import os

import pytest

from rpg_open_remode_amd import _lib


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    declared = _lib.header_functions()
    assert len(declared) >= 35
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, f"declared in rmd_hip.h but not exported: {missing}"


def test_python_binding_covers_the_header():
    assert sorted(_lib.SIGNATURES) == _lib.header_functions()


def test_version_and_error_string():
    L = _lib.lib()
    assert L.rmd_hip_version() >= 100
    assert L.rmd_hip_seeds_create(64, 48, 1.0, 1.0, 1.0, 1.0, 5, 100, None) == _lib.ERR_INVALID_ARG
    assert b"null output" in L.rmd_hip_last_error()


def test_argument_validation_precedes_device_use():
    L = _lib.lib()
    h = ctypes.c_void_p()
    # unsupported patch side, bad size, bad extent: rejected before any HIP call
    assert L.rmd_hip_seeds_create(64, 48, 1.0, 1.0, 1.0, 1.0, 4, 100, ctypes.byref(h)) == _lib.ERR_INVALID_ARG
    assert L.rmd_hip_seeds_create(0, 48, 1.0, 1.0, 1.0, 1.0, 5, 100, ctypes.byref(h)) == _lib.ERR_INVALID_ARG
    assert L.rmd_hip_seeds_create(64, 48, 1.0, 1.0, 1.0, 1.0, 5, 1000, ctypes.byref(h)) == _lib.ERR_INVALID_ARG
    assert L.rmd_hip_image_create(7, 8, 8, ctypes.byref(h)) == _lib.ERR_INVALID_ARG
    assert L.rmd_hip_seeds_update(None, None, None) == _lib.ERR_INVALID_ARG
    assert L.rmd_hip_denoiser_denoise(None, None, None, None, None, None, 0.5, 1) == _lib.ERR_INVALID_ARG
    # destroying null handles is a no-op, as deleting a null pointer is
    assert L.rmd_hip_seeds_destroy(None) == 0 and L.rmd_hip_denoiser_destroy(None) == 0 and L.rmd_hip_image_destroy(None) == 0


def test_device_count_reports_absence_cleanly():
    L = _lib.lib()
    n = ctypes.c_int(-1)
    rc = L.rmd_hip_device_count(ctypes.byref(n))
    assert rc in (_lib.OK, _lib.ERR_NO_DEVICE)
    if rc == _lib.ERR_NO_DEVICE:
        assert n.value == 0
        h = ctypes.c_void_p()
        # no silent CPU path: creating a handle without a GPU is an error
        assert L.rmd_hip_seeds_create(64, 48, 1.0, 1.0, 1.0, 1.0, 5, 100, ctypes.byref(h)) == _lib.ERR_NO_DEVICE
        assert L.rmd_hip_denoiser_create(64, 48, ctypes.byref(h)) == _lib.ERR_NO_DEVICE


def test_product_never_touches_the_oracle():
    """The product path must not import, link or execute anything under oracle/."""
    root = _lib.ROOT
    pkg = os.path.join(root, "rpg_open_remode_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("libremode_oracle", "libremode_ref", "import oracles", "oracle/_ref", "#include \"remode_oracle"):
                    if needle in text and f != "build.py":
                        offenders.append((f, needle))
    assert not offenders, offenders


@pytest.mark.gpu
def test_wave_primitives_selftest():
    """the DPP reductions / scans the kernels use equal their shuffle forms on the device"""
    import ctypes
    from rpg_open_remode_amd import _lib
    bad = ctypes.c_int(-1)
    _lib.check(_lib.lib().rmd_hip_selftest_wave_primitives(ctypes.byref(bad)))
    assert bad.value == 0


@pytest.mark.parametrize("wh", [(640, 480), (101, 67), (7, 3)])
def test_float_frames_of_8bit_levels_are_recognised_on_the_host(wh):
    """host logic of update(float*): a frame travels as bytes only if EVERY pixel has the bit pattern of (float)k * (1.0f / 255.0f)
    (depthmap.cpp:105); the bytes are the k.  No device involved."""
    import ctypes
    import numpy as np
    from rpg_open_remode_amd import _lib
    w, h = wh
    pitch = (w + 3) // 4 * 4
    rng = np.random.default_rng(w * 31 + h)
    gray = rng.integers(0, 256, (h, w), dtype=np.uint8)
    gray.flat[:256 if w * h >= 256 else w * h] = np.arange(256, dtype=np.uint8)[:min(256, w * h)]  # every level at least once
    levels = (gray.astype(np.float32) * np.float32(1.0 / 255.0)).astype(np.float32)

    def pack(img):
        img = np.ascontiguousarray(img, np.float32)
        out = np.zeros((h, pitch), np.uint8)
        ok = ctypes.c_int(-1)
        _lib.check(_lib.lib().rmd_hip_selftest_pack_float_frame(img.ctypes.data, w, h, pitch, out.ctypes.data, ctypes.byref(ok)))
        return ok.value, out[:, :w]

    ok, got = pack(levels)
    assert ok == 1 and np.array_equal(got, gray)
    y, x = h // 3, w // 2
    for bad in (np.float32(np.nan), np.float32(-0.0), np.float32(1.5), np.float32(2.0), np.float32(-0.25), np.float32(np.inf),
                np.nextafter(levels[y, x], np.float32(2.0)), np.nextafter(np.float32(0.5), np.float32(0.0)), np.float32(1e-40)):
        img = levels.copy()
        img[y, x] = bad
        assert pack(img)[0] == 0, f"a frame with the pixel {bad!r} was taken for 8-bit levels"
    for pos in ((0, 0), (h - 1, w - 1), (h // 2, 0)):  # the examination looks at the middle row first and splits the rest: every part must see its pixels
        img = levels.copy()
        img[pos] = np.float32(0.1234)
        assert pack(img)[0] == 0
    assert pack((levels * np.float32(0.999)).astype(np.float32))[0] == 0


def test_tunables_are_set_and_read_in_one_place():
    """process-wide settings: set / get round trip, range checks, and the environment preset (read once, by rmdh::tunables(), in a fresh process)"""
    import subprocess
    import sys
    from rpg_open_remode_amd import api
    old = api.getTunable(api.TUNE_PACK_BACKOFF)
    api.setTunable(api.TUNE_PACK_BACKOFF, 3)
    assert api.getTunable(api.TUNE_PACK_BACKOFF) == 3
    api.setTunable(api.TUNE_PACK_BACKOFF, old)
    for bad in ((api.TUNE_HOST_FRAMES, 4), (api.TUNE_BATCH_GROUPS, 5), (api.TUNE_COPY_THREADS, 0), (99, 0)):
        with pytest.raises(api.RmdHipError):
            api.setTunable(*bad)
    code = ("from rpg_open_remode_amd import api; "
            "print(api.getTunable(api.TUNE_HOST_FRAMES), api.getTunable(api.TUNE_BATCH_GROUPS), api.getTunable(api.TUNE_COPY_THREADS), api.getTunable(api.TUNE_FUSED_INGEST))")
    env = dict(os.environ, RMD_HIP_HOST_FRAMES="inplace_ahead", RMD_HIP_BATCH_GROUPS="2", RMD_HIP_FUSED_INGEST="0")
    env.pop("RMD_HIP_COPY_THREADS", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), check=True).stdout
    assert out.split() == ["3", "2", "4", "0"], out
    # a preset passes the same range check as rmd_hip_set_tunable: values outside it, and strings that are no number / no mode name, are ignored
    # with a line on stderr (a negative RMD_HIP_AHEAD_WGS once went straight into the search kernel's grid arithmetic)
    code = "from rpg_open_remode_amd import api; print(api.getTunable(api.TUNE_AHEAD_WGS), api.getTunable(api.TUNE_HOST_FRAMES), api.getTunable(api.TUNE_COPY_THREADS), api.getTunable(api.TUNE_PACK_BACKOFF))"
    env = dict(os.environ, RMD_HIP_AHEAD_WGS="-7", RMD_HIP_HOST_FRAMES="sideways", RMD_HIP_COPY_THREADS="0", RMD_HIP_PACK_BACKOFF="12x")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), check=True)
    assert res.stdout.split() == ["128", "-1", "4", "15"], res.stdout
    assert res.stderr.count("ignored") == 4, res.stderr
    env = dict(os.environ, RMD_HIP_AHEAD_WGS="0")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), check=True)
    assert res.stdout.split()[0] == "128" and "ignored" in res.stderr
    # the library reads its environment in ONE place
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rpg_open_remode_amd", "csrc")
    hits = [f for f in sorted(os.listdir(csrc)) if os.path.isfile(os.path.join(csrc, f)) and "getenv(" in open(os.path.join(csrc, f)).read()]
    assert hits == ["rmd_capi.hip"], hits
