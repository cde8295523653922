"""Golden vectors produced by the reference itself (tests/golden/make_golden.py, run where /root/reference exists).
They travel with the repository, so the oracle and the HIP path are pinned to the reference on any machine."""
import os

import numpy as np
import pytest

import oracles as O
from common import MATCHERS, apply_matcher
from rpg_open_remode_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(side):
    g = np.load(os.path.join(HERE, "golden", f"remode_golden_s{side}.npz"))
    images = [synth.to_float_image(f) for f in g["frames_u8"]]
    return g, images


def _check(g, tag, seeds_state_fn, conv_per_update, denoised, count, dist):
    for name, arr in seeds_state_fn.items():
        assert O.planes_equal(g[f"{tag}_{name}"], arr), f"{tag}_{name}: {O.count_mismatch(g[f'{tag}_{name}'], arr)} elements differ"
    assert np.array_equal(g[f"{tag}_conv_per_update"], conv_per_update)
    assert O.planes_equal(g[f"{tag}_denoised"], denoised)
    assert int(g[f"{tag}_converged_count"]) == count
    assert np.float32(g[f"{tag}_dist_from_ref"]) == np.float32(dist)


def _run_oracle(kind, side, g, images):
    w, h = int(g["width"]), int(g["height"])
    lib = O.OracleLib(kind, side)
    s = O.Seeds(lib, w, h, g["K"])
    dmin, dmax = g["depth_range"]
    s.set_reference(images[0], g["T_curr_world"][0], dmin, dmax)
    init = {"sum_templ": s.download(O.PLANE_SUM_TEMPL), "denom": s.download(O.PLANE_DENOM)}
    convs = []
    for k in range(1, len(images)):
        s.update(images[k], g["T_curr_world"][k])
        convs.append(s.download(O.PLANE_CONV).astype(np.int8))
    state = dict(init, mu=s.download(O.PLANE_MU), sigma_sq=s.download(O.PLANE_SIGMA_SQ), a=s.download(O.PLANE_A),
                 b=s.download(O.PLANE_B), match=s.download(O.PLANE_MATCH))
    d = O.Denoiser(lib, w, h)
    d.set_large_sigma_sq(dmax - dmin)
    den = d.denoise(s, float(g["tv_lambda"]), int(g["tv_iterations"]))
    return state, np.stack(convs), den, s.converged_count(), s.dist_from_ref()


@pytest.mark.parametrize("side", [5, 9])
def test_oracle_reproduces_the_reference_golden_vectors(side):
    g, images = _load(side)
    _check(g, "ref", *_run_oracle("port_libm", side, g, images))      # glibc libm on both sides
    _check(g, "refrmd", *_run_oracle("port", side, g, images))        # rmd_math.h on both sides


@pytest.mark.gpu
@pytest.mark.parametrize("matcher", MATCHERS)
@pytest.mark.parametrize("side", [5, 9])
def test_hip_reproduces_the_reference_golden_vectors(side, matcher):
    from rpg_open_remode_amd import api
    g, images = _load(side)
    w, h = int(g["width"]), int(g["height"])
    dmin, dmax = g["depth_range"]
    s = api.SeedMatrix(w, h, api.PinholeCamera(*g["K"]), patch_side=side)
    apply_matcher(s, matcher)
    s.setReferenceImage(images[0], g["T_curr_world"][0], dmin, dmax)
    init = {"sum_templ": s.downloadSumTempl(), "denom": s.downloadConstTemplDenom()}
    convs = []
    for k in range(1, len(images)):
        s.update(images[k], g["T_curr_world"][k])
        convs.append(s.downloadConvergence().astype(np.int8))
    state = dict(init, mu=s.downloadDepthmap(), sigma_sq=s.downloadSigmaSq(), a=s.downloadA(), b=s.downloadB(),
                 match=s.downloadEpipolarMatches())
    d = api.DepthmapDenoiser(w, h)
    d.setLargeSigmaSq(dmax - dmin)
    den = d.denoise(s.getMu(), s.getSigmaSq(), s.getA(), s.getB(), float(g["tv_lambda"]), int(g["tv_iterations"]))
    _check(g, "refrmd", state, np.stack(convs), den, s.getConvergedCount(), s.getDistFromRef())


# ---- host-side steps (point cloud, undistortion): vectors frozen from oracle/host_steps.py (make_golden_host_steps.py)
def _host_golden():
    return np.load(os.path.join(HERE, "golden", "host_steps_golden.npz"))


def test_host_steps_oracle_and_library_map_equal_golden():
    from rpg_open_remode_amd import api
    g = _host_golden()
    w, h = int(g["width"]), int(g["height"])
    m1, m2 = O.undistort_maps(w, h, g["K"], g["D"])
    assert np.array_equal(m1, g["map1"]) and np.array_equal(m2, g["map2"])
    l1, l2 = api.computeUndistortionMap(w, h, *g["K"], *g["D"])  # the library's host code
    assert np.array_equal(l1, g["map1"]) and np.array_equal(l2, g["map2"])
    assert np.array_equal(O.remap_u8(g["gray"], g["map1"], g["map2"]), g["remapped"])
    assert O.count_mismatch(O.point_cloud(g["depth"], g["convergence"], g["gray"], g["K"], g["T_world_ref"]), g["cloud"]) == 0


@pytest.mark.gpu
def test_hip_remap_equals_golden():
    from rpg_open_remode_amd import api
    g = _host_golden()
    w, h = int(g["width"]), int(g["height"])
    s = api.SeedMatrix(w, h, api.PinholeCamera(*g["K"]), patch_side=3)
    s.initUndistortionMap(*g["D"])
    s.setReferenceImageU8(g["gray"], np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), 1.0, 2.0)
    assert O.count_mismatch(s.download(api.PLANE_REF_IMG), g["remapped"].astype(np.float32) * np.float32(1.0 / 255.0)) == 0
