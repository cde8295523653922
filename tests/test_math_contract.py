"""The shared arithmetic contract (csrc/rmd_math.h) on the host: expf / sinf / acosf restate glibc 2.35's routines and must equal
the host libm's results bit for bit (oracle/libm_exhaustive covers all 2^32 arguments; a strided sample of it runs here), and must
lie within 1 ulp of the mathematically exact values.  The device side of the same header is compared bit for bit in
tests/test_hip_parity.py."""
import os
import subprocess

import numpy as np

import oracles as O
from common import ulp_distance


def _eval(fn, xs):
    return np.array([fn(float(v)) for v in xs], np.float32)


def test_transcendentals_equal_the_host_libm_on_a_strided_sample_of_all_floats():
    """every 1021st float (4.2 million arguments per function, all exponents, both signs, NaNs and infinities included); the full
    run (`oracle/libm_exhaustive 1`, ~30 s on 8 cores) was 0 differences of 3 x 2^32 on glibc 2.35 / x86-64 with FMA"""
    import glibc_parity
    glibc_parity.require_pinned_glibc()
    exe = os.path.join(O.ORACLE_DIR, "libm_exhaustive")
    if not os.path.exists(exe):
        import pytest
        pytest.skip("oracle/libm_exhaustive not built")
    res = subprocess.run([exe, "1021"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout
    assert res.stdout.count(" 0 differ") == 3, res.stdout


def test_transcendentals_within_one_ulp_of_exact():
    lib = O.OracleLib("port", 5).lib
    rng = np.random.default_rng(3)
    cases = [
        (lib.orc_math_expf, np.exp, rng.uniform(-100, 88, 20000).astype(np.float32)),
        (lib.orc_math_sinf, np.sin, rng.uniform(-7, 7, 20000).astype(np.float32)),
        (lib.orc_math_acosf, np.arccos, rng.uniform(-1, 1, 20000).astype(np.float32)),
        (lib.orc_math_acosf, np.arccos, (1.0 - 10.0 ** rng.uniform(-7, -1, 4000)).astype(np.float32)),
    ]
    for fn, ref, xs in cases:
        got = _eval(fn, xs)
        exact = ref(xs.astype(np.float64)).astype(np.float32)  # numpy fp64 libm, rounded once
        d = ulp_distance(got, exact)
        assert d.max() <= 1, (fn, d.max())


def test_special_values():
    lib = O.OracleLib("port", 5).lib
    assert lib.orc_math_expf(0.0) == 1.0 and lib.orc_math_expf(-200.0) == 0.0 and lib.orc_math_expf(100.0) == np.inf
    assert np.isnan(lib.orc_math_expf(float("nan"))) and np.isnan(lib.orc_math_sinf(float("inf")))
    assert lib.orc_math_acosf(1.0) == 0.0 and lib.orc_math_acosf(-1.0) == np.float32(np.pi) and np.isnan(lib.orc_math_acosf(1.0000001))
    assert lib.orc_math_sinf(0.0) == 0.0
    assert lib.orc_math_rsqrtf(4.0) == 0.5 and lib.orc_math_rsqrtf(0.0) == np.inf
    assert lib.orc_math_lerp(0.0, 3.0, 7.0) == 3.0 and lib.orc_math_lerp(1.0, 3.0, 7.0) == 7.0 and lib.orc_math_lerp(0.5, 3.0, 7.0) == 5.0


def test_texture_filter_rule():
    """clamp addressing, texel centres at +0.5, separable fma lerp, exact texel at zero weight, NaN for non-finite coords"""
    lib = O.OracleLib("port", 5).lib
    img = np.arange(12, dtype=np.float32).reshape(3, 4) * np.float32(1.25)
    f = lambda x, y: lib.orc_tex_linear(img.ctypes.data, 4, 3, x, y)
    assert f(0.5, 0.5) == img[0, 0] and f(3.5, 2.5) == img[2, 3]
    assert f(-5.0, -5.0) == img[0, 0] and f(50.0, 50.0) == img[2, 3]  # clamp to edge
    assert f(1.0, 0.5) == np.float32(0.5) * (img[0, 0] + img[0, 1])
    a, b = np.float32(0.25), np.float32(0.75)
    h0 = np.float32(img[1, 1] + a * (img[1, 2] - img[1, 1]))
    h1 = np.float32(img[2, 1] + a * (img[2, 2] - img[2, 1]))
    assert f(1.75, 2.25) == np.float32(h0 + b * (h1 - h0))
    assert np.isnan(f(float("nan"), 1.0)) and np.isnan(f(1.0, float("inf")))


def test_contract_equals_this_process_libm():
    """the same comparison through Python's ctypes on the libm of this process (random arguments in the ranges the filter uses)"""
    import ctypes
    import ctypes.util
    libm = ctypes.CDLL(ctypes.util.find_library("m"))
    lib = O.OracleLib("port", 5).lib
    rng = np.random.default_rng(5)
    for name, mine, xs in (("expf", lib.orc_math_expf, rng.uniform(-110, 0, 20000)), ("sinf", lib.orc_math_sinf, rng.uniform(0, 3.2, 20000)),
                           ("acosf", lib.orc_math_acosf, rng.uniform(-1, 1, 20000))):
        f = getattr(libm, name)
        f.restype, f.argtypes = ctypes.c_float, [ctypes.c_float]
        xs = xs.astype(np.float32)
        a = _eval(mine, xs)
        b = np.array([f(float(v)) for v in xs], np.float32)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name
