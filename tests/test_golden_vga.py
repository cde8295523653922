"""BASELINE-size golden vectors from the reference itself (tests/golden/make_golden_vga.py: the benchmark sequence, 640x480,
patch side 9, the first 60 updates through the reference's own kernels compiled for the CPU).  Pins the CPU oracle and -- on
the GPU -- the device path DIRECTLY to the reference at the size at which the search's load balancing, per-unit windows and
fall-back paths are actually exercised: convergence masks after every update bit for bit, every f32 plane by SHA-256 after
every update (NaN == NaN), the TV-L1 output after the last one."""
import hashlib
import os

import numpy as np
import pytest

import oracles as O
from common import MATCHERS, apply_matcher
from rpg_open_remode_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
PLANES = (("mu", O.PLANE_MU), ("sigma_sq", O.PLANE_SIGMA_SQ), ("a", O.PLANE_A), ("b", O.PLANE_B), ("match", O.PLANE_MATCH))


def plane_digest(a):
    a = np.ascontiguousarray(a, np.float32).copy()
    bits = a.view(np.uint32)
    bits[np.isnan(a)] = 0x7FC00000
    return np.frombuffer(hashlib.sha256(bits.tobytes()).digest(), np.uint8)


@pytest.fixture(scope="module")
def golden():
    g = np.load(os.path.join(HERE, "golden", "remode_golden_vga_s9.npz"))
    seq = synth.Sequence(int(g["width"]), int(g["height"]), int(g["updates"]) + 1, seed=int(g["scene"]))
    # the generator is deterministic: same frames, poses and depth range as when the fixture was made
    for k, gray in enumerate(seq.gray):
        assert np.array_equal(np.frombuffer(hashlib.sha256(gray.tobytes()).digest(), np.uint8), g["frame_sha256"][k]), f"frame {k} changed"
    assert np.array_equal(np.stack(seq.T_curr_world), g["T_curr_world"])
    assert np.array_equal(np.asarray([seq.min_depth, seq.max_depth], np.float32), g["depth_range"])
    return g, seq


def _check_update(g, k, conv, planes, count, what):
    assert np.array_equal(conv.astype(np.int8), g["conv_per_update"][k - 1]), f"{what}: convergence mask after update {k}"
    for name, arr in planes.items():
        assert np.array_equal(plane_digest(arr), g[f"{name}_sha256_per_update"][k - 1]), f"{what}: plane {name} after update {k}"
    assert count == int(g["converged_count_per_update"][k - 1]), f"{what}: converged count after update {k}"


def test_oracle_reproduces_the_vga_golden(golden):
    g, seq = golden
    lib = O.OracleLib("port", int(g["side"]))
    before = lib.fn("max_threads")()
    lib.fn("set_num_threads")(max(1, min(48, os.cpu_count() or 1)))
    try:
        s = O.Seeds(lib, seq.width, seq.height, seq.K)
        s.set_reference(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
        assert np.array_equal(plane_digest(s.download(O.PLANE_SUM_TEMPL)), g["sum_templ_sha256"])
        assert np.array_equal(plane_digest(s.download(O.PLANE_DENOM)), g["denom_sha256"])
        n = int(g["updates"])
        for k in range(1, n + 1):
            s.update(seq.images[k], seq.T_curr_world[k])
            if k <= 6 or k % 9 == 0 or k == n:  # every plane at a spread of updates; the masks at every update
                _check_update(g, k, s.download(O.PLANE_CONV), {name: s.download(p) for name, p in PLANES}, s.converged_count(), "oracle B")
            else:
                assert np.array_equal(s.download(O.PLANE_CONV).astype(np.int8), g["conv_per_update"][k - 1]), f"oracle B: mask after update {k}"
        d = O.Denoiser(lib, seq.width, seq.height)
        d.set_large_sigma_sq(seq.max_depth - seq.min_depth)
        assert np.array_equal(plane_digest(d.denoise(s, float(g["tv_lambda"]), int(g["tv_iterations"]))), g["denoised_sha256"])
    finally:
        lib.fn("set_num_threads")(before)


@pytest.mark.gpu
@pytest.mark.parametrize("matcher", MATCHERS)
def test_hip_reproduces_the_vga_golden(golden, matcher):
    from rpg_open_remode_amd import api
    g, seq = golden
    s = api.SeedMatrix(seq.width, seq.height, api.PinholeCamera(*seq.K), patch_side=int(g["side"]))
    apply_matcher(s, matcher)
    s.setReferenceImage(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    assert np.array_equal(plane_digest(s.downloadSumTempl()), g["sum_templ_sha256"])
    assert np.array_equal(plane_digest(s.downloadConstTemplDenom()), g["denom_sha256"])
    n = int(g["updates"])
    for k in range(1, n + 1):
        s.update(seq.images[k], seq.T_curr_world[k])
        _check_update(g, k, s.downloadConvergence(),
                      {"mu": s.downloadDepthmap(), "sigma_sq": s.downloadSigmaSq(), "a": s.downloadA(), "b": s.downloadB(),
                       "match": s.downloadEpipolarMatches()}, s.getConvergedCount(), f"HIP matcher {matcher}")
    d = api.DepthmapDenoiser(seq.width, seq.height)
    d.setLargeSigmaSq(seq.max_depth - seq.min_depth)
    den = d.denoise(s.getMu(), s.getSigmaSq(), s.getA(), s.getB(), float(g["tv_lambda"]), int(g["tv_iterations"]))
    assert np.array_equal(plane_digest(den), g["denoised_sha256"])


@pytest.mark.gpu
def test_hip_streaming_reproduces_the_vga_golden(golden):
    """the same without an observer between the updates (deferred finalisation fused into the next frame's setup, device-resident
    8-bit ingest): only the last update is checked"""
    from rpg_open_remode_amd import api
    g, seq = golden
    s = api.SeedMatrix(seq.width, seq.height, api.PinholeCamera(*seq.K), patch_side=int(g["side"]))
    s.setReferenceImageU8(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    n = int(g["updates"])
    for k in range(1, n + 1):
        s.updateU8(seq.gray[k], seq.T_curr_world[k])
    _check_update(g, n, s.downloadConvergence(),
                  {"mu": s.downloadDepthmap(), "sigma_sq": s.downloadSigmaSq(), "a": s.downloadA(), "b": s.downloadB(),
                   "match": s.downloadEpipolarMatches()}, s.getConvergedCount(), "HIP streaming")
