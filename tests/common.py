"""Shared helpers of the test-suite: cached synthetic sequences, random seed states, comparisons."""
import functools

import numpy as np

import oracles as O
from rpg_open_remode_amd import api, synth

PLANE_NAMES = ["mu", "sigma_sq", "a", "b", "convergence", "sum_templ", "const_templ_denom", "epipolar_matches"]


@functools.lru_cache(maxsize=16)
def sequence(width, height, n_frames, seed=0):
    return synth.Sequence(width, height, n_frames, seed)


def random_state(width, height, seq, rng, side):
    """Seed planes that exercise every branch of check/match/update: fresh, nearly converged, diverged,
    tiny and huge variances (search capped at max_extent), NaN/Inf/negative values, depths behind the camera."""
    n = (height, width)
    rngd = seq.max_depth - seq.min_depth
    mu = rng.uniform(seq.min_depth - 0.2 * rngd, seq.max_depth + 0.2 * rngd, n).astype(np.float32)
    sig = (10.0 ** rng.uniform(-7, 0.5, n)).astype(np.float32) * np.float32(rngd * rngd)
    a = rng.uniform(1.0, 40.0, n).astype(np.float32)
    b = rng.uniform(1.0, 40.0, n).astype(np.float32)
    # make a good share converge / diverge
    m = rng.random(n) < 0.15
    a[m], b[m], sig[m] = 30.0, 2.0, np.float32(rngd / 5000.0)
    m = rng.random(n) < 0.1
    a[m], b[m] = 1.2, 30.0
    # pathological values
    flat = rng.choice(width * height, size=max(8, width * height // 200), replace=False)
    specials = np.array([np.nan, np.inf, -np.inf, 0.0, -1.0, 1e-30, 1e30, -0.0], np.float32)
    for plane in (mu, sig, a, b):
        idx = rng.choice(flat, size=len(flat) // 4, replace=False)
        plane.reshape(-1)[idx] = rng.choice(specials, size=len(idx))
    return mu, sig, a, b


def assert_states_equal(ref_state, got_state, what="", planes=range(8)):
    bad = {}
    for p in planes:
        n = O.count_mismatch(ref_state[p], got_state[p])
        if n:
            bad[PLANE_NAMES[p]] = n
    assert not bad, f"{what}: mismatching elements per plane: {bad}"


def rmse(a, b, mask=None):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = a - b
    if mask is not None:
        d = d[mask]
    d = d[np.isfinite(d)]
    return float(np.sqrt(np.mean(d * d))) if d.size else 0.0


def ulp_distance(a, b):
    """distance in units in the last place between two float32 arrays (same-sign finite values)"""
    ia = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2**31) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2**31) - ib, ib)
    return np.abs(ia - ib)


# Matcher variants every parity test runs: 0 = per-pixel kernel (the reference's shape), 3 = the two-launch tile pipeline (the
# default).
MATCHERS = [0, 3]


def apply_matcher(seeds, matcher):
    from rpg_open_remode_amd import api
    seeds.setOption(api.OPT_MATCHER, matcher)
    return seeds


class OracleDepthmap:
    """rmd::Depthmap's interface (depthmap.h:37-102) on Oracle B.  Test infrastructure."""

    def __init__(self, seq, side):
        self.seq, self.olib = seq, O.OracleLib("port", side)
        self.seeds = O.Seeds(self.olib, seq.width, seq.height, seq.K)
        self.den = O.Denoiser(self.olib, seq.width, seq.height)
        self.depth, self.conv, self.ref, self.T_world_ref = None, None, None, api.SE3()

    def setReferenceImage(self, img, T_curr_world, min_depth, max_depth):
        self.den.set_large_sigma_sq(max_depth - min_depth)
        self.seeds.set_reference(img.astype(np.float32) * np.float32(1.0 / 255.0), T_curr_world.data, min_depth, max_depth)
        self.ref, self.T_world_ref = img.copy(), T_curr_world.inv()
        return True

    def update(self, img, T_curr_world):
        self.seeds.update(img.astype(np.float32) * np.float32(1.0 / 255.0), T_curr_world.data)

    def getConvergedPercentage(self):
        return float(np.float32(self.seeds.converged_count()) / np.float32(self.seq.width * self.seq.height) * np.float32(100.0))

    def getDistFromRef(self): return self.seeds.dist_from_ref()
    def downloadDenoisedDepthmap(self, lam, iters): self.depth = self.den.denoise(self.seeds, lam, iters)
    def getDepthmap(self): return self.depth
    def downloadConvergenceMap(self): self.conv = self.seeds.download(4)
    def getConvergenceMap(self): return self.conv
    def getReferenceImage(self): return self.ref
    def downloadPointCloud(self, denoised=True): return O.point_cloud(self.depth, self.conv, self.ref, self.seq.K, self.T_world_ref.data)
