#!/usr/bin/env python3
"""Exercises the kernels either side of the path (u8 ingest with and without undistortion, point cloud) for a rocprofv3 run:
  rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python tools/aux_profile.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rpg_open_remode_amd import api, synth
W, H, N = 640, 480, 45
seq = synth.Sequence(W, H, N)
for distortion in (None, (-0.28, 0.07, 1e-3, -5e-4)):
    dm = api.Depthmap(W, H, seq.K[0], seq.K[2], seq.K[1], seq.K[3], patch_side=9)
    if distortion:
        dm.initUndistortionMap(*distortion)
    dm.setReferenceImage(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    for k in range(1, N):
        dm.update(seq.gray[k], seq.T_curr_world[k])
    dm.downloadDenoisedDepthmap(0.5, 200)
    for _ in range(10):
        pts = dm.downloadPointCloud()
    print("distortion" if distortion else "plain", len(pts), "points")
