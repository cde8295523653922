#!/usr/bin/env python3
"""TV-L1 kernel geometry sweep: device time per iteration (HIP events around the iteration loop) for every geometry of the
blocked kernel, at 640x480 (200 iterations) and 1920x1080 (500), checked against the one-iteration-per-launch kernel bit for bit.
usage: python tools/denoise_sweep.py"""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rpg_open_remode_amd import api

NAMES = {0: "default", 1: "32x8 K2", 2: "64x16 K4", 3: "32x16 K4", 4: "16x16 K4", 5: "16x16 K8"}
for (w, h, iters) in ((640, 480, 200), (1920, 1080, 500)):
    rng = np.random.default_rng(1)
    imgs = []
    for lo, hi in ((1, 2), (1e-6, 1e-2), (1, 30), (1, 30)):
        im = api.DeviceImage(w, h, np.float32); im.setDevData(rng.uniform(lo, hi, (h, w)).astype(np.float32)); imgs.append(im)
    d = api.DepthmapDenoiser(w, h); d.setLargeSigmaSq(1.0); d.setOption(api.DENOISE_OPT_TIMING, 1)
    d.setOption(api.DENOISE_OPT_ITERS_PER_LAUNCH, 1)
    ref = d.denoise(*imgs, 0.5, iters)
    ms1, n1 = d.timing()
    print(f"{w}x{h}, {iters} iterations: one iteration per launch {ms1 * 1e3 / iters:7.2f} us/iteration ({n1} launches)")
    d.setOption(api.DENOISE_OPT_ITERS_PER_LAUNCH, 0)
    for g in range(0, 6):
        d.setOption(api.DENOISE_OPT_GEOMETRY, g)
        out = d.denoise(*imgs, 0.5, iters)
        best = 1e9
        for _ in range(5):
            d.denoise(*imgs, 0.5, iters, download=False)
            ms, n = d.timing()
            best = min(best, ms)
        same = np.array_equal(out, ref, equal_nan=True)
        print(f"   geometry {g} ({NAMES[g]:9s}): {best * 1e3 / iters:7.2f} us/iteration, {n:4d} launches, {best:7.3f} ms total, "
              f"{40 * w * h * iters / best / 1e6:8.1f} GB/s algorithmic  {'bit-identical' if same else 'MISMATCH'}")
