#!/usr/bin/env python3
"""Time of update 1 after setReferenceImage (every seed live, its whole depth range searched): the same work whatever the NCC values are, so
diagnostic builds that change the arithmetic (tools/ab_make.sh with -DRMD_EXPERIMENT_LDS=...) can be compared on it.  Stand-alone sequence
and batches; frames resident in HBM.  usage: python tools/first_update_bench.py [--b 1,8] [--size 640x480] [--reps 40]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rpg_open_remode_amd import api, synth

ap = argparse.ArgumentParser()
ap.add_argument("--b", default="1,8"); ap.add_argument("--size", default="640x480"); ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--side", type=int, default=9); ap.add_argument("--label", default=""); ap.add_argument("--unit-target", type=int, default=1)
a = ap.parse_args()
W, H = (int(v) for v in a.size.split("x"))
seq = synth.Sequence(W, H, 3, 0)
dev = []
for k in range(2):
    d = api.DeviceImage(W, H, np.float32)
    d.setDevData(seq.images[k])
    dev.append(d)
for n in (int(v) for v in a.b.split(",")):
    b = api.SeedMatrixBatch(n, W, H, api.PinholeCamera(*seq.K), patch_side=a.side)
    b.setOption(api.OPT_UNIT_TARGET, a.unit_target)
    ts = []
    for rep in range(a.reps + 3):
        for i in range(n):
            b[i].setReferenceImageDevice(dev[0].data, dev[0].stride, seq.T_curr_world[0], seq.min_depth, seq.max_depth)
        b.sync()
        t0 = time.perf_counter()
        b.updateDevice([dev[1].data] * n, [dev[1].stride] * n, [seq.T_curr_world[1]] * n)
        b.sync()
        if rep >= 3:
            ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    print(f"{a.label} {W}x{H} batch of {n}: update 1 takes {ts[len(ts) // 2]:.1f} us (median of {len(ts)}; min {ts[0]:.1f}) "
          f"= {ts[len(ts) // 2] / n:.1f} us per sequence", flush=True)
    del b
