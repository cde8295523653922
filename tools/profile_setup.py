#!/usr/bin/env python3
"""Phases of the setup kernel's chain in live tiles (needs librmd_hip.so built with -DRMD_PROFILE_ROUNDS): loads, triangulation,
uncertainty (acosf x2, sinf x2), normpdf (expf), rest of the Bayesian update, check + epipolar segment, valid run.  Tiles 1024.. only
(the slots of the lower tiles are overwritten by the search workgroups' own probes)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rpg_open_remode_amd import api, synth
W, H, F = 640, 480, 200
seq = synth.Sequence(W, H, F)
frames = []
for im in seq.images:
    d = api.DeviceImage(W, H, np.float32); d.setDevData(im); frames.append(d)
for rep in range(2):
    s = api.SeedMatrix(W, H, api.PinholeCamera(*seq.K), patch_side=9)
    s.setReferenceImageDevice(frames[0].data, frames[0].stride, seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    s.sync(); s.setOption(api.OPT_COLLECT_STATS, 2)
    for k in range(1, F):
        s.updateDevice(frames[k].data, frames[k].stride, seq.T_curr_world[k])
    s.sync()
names = ["loads+decode", "triangulate", "uncertainty", "normpdf", "rest of fuse", "check+segment", "valid run"]
print("setup kernel, live lanes of tiles 1024..1199, mean us per phase: " + ", ".join(names))
for k in (3, 14, 24, 59, 89, 124, 189):
    t = s.frameTraceDownload(k).astype(np.uint64)[1024:, 2]
    t = t[t != 0]
    if not len(t):
        continue
    f = [((t >> np.uint64(9 * i)) & np.uint64(511)).astype(np.float64) / 100 for i in range(7)]
    print(f"update {k + 1:3d} ({len(t):3d} tiles): " + "  ".join(f"{v.mean():5.2f}" for v in f) + f"   sum {sum(v.mean() for v in f):5.2f}")
