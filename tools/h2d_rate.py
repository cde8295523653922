#!/usr/bin/env python3
"""H2D-inclusive update rate (frames start in pageable host memory) next to the resident rate, over complete passes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rpg_open_remode_amd import api, synth
W, H, F = 640, 480, 200
seq = synth.Sequence(W, H, F)
frames = []
for im in seq.images:
    d = api.DeviceImage(W, H, np.float32); d.setDevData(im); frames.append(d)
other = [(im * np.float32(0.999)).astype(np.float32) for im in seq.images]
def rate(kind):
    s = api.SeedMatrix(W, H, api.PinholeCamera(*seq.K), patch_side=9)
    if kind == "u8+undistort":  # 8-bit frames through the lens undistortion (Depthmap::initUndistortionMap)
        s.initUndistortionMap(-0.25, 0.08, 1e-3, -7e-4)
        kind = "u8"
    def one():
        if kind == "u8":
            s.setReferenceImageU8(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
            for k in range(1, F): s.updateU8(seq.gray[k], seq.T_curr_world[k])
        elif kind == "f32":  # the reference's host path: 8-bit levels as floats (convertTo(CV_32F, 1 / 255)); they travel as bytes
            s.setReferenceImage(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
            for k in range(1, F): s.update(seq.images[k], seq.T_curr_world[k])
        elif kind == "f32 any":  # floats that are NOT 8-bit levels: 1.2 MB per frame through the copy engine
            s.setReferenceImage(other[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
            for k in range(1, F): s.update(other[k], seq.T_curr_world[k])
        else:
            s.setReferenceImageDevice(frames[0].data, frames[0].stride, seq.T_curr_world[0], seq.min_depth, seq.max_depth)
            for k in range(1, F): s.updateDevice(frames[k].data, frames[k].stride, seq.T_curr_world[k])
    one(); s.sync()
    t0 = time.perf_counter()
    for _ in range(3): one()
    t_sub = time.perf_counter() - t0
    s.sync()
    t = time.perf_counter() - t0
    return t / 3 / (F - 1) * 1e6, t_sub / 3 / (F - 1) * 1e6
for kind in ("resident", "u8", "f32", "f32 any", "resident", "u8", "u8+undistort"):
    us, sub = rate(kind)
    print(f"{kind:12s}: {us:7.2f} us per update ({W * H / us:8.1f} Mpix/s); host submission {sub:6.2f} us per update")
