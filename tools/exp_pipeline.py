#!/usr/bin/env python3
"""Experiment RMD_HIP_OPT_PIPELINE (csrc/rmd_pipelined.hpp: one launch per update, resident frames): bit-exact against the product pipeline
with observers in between (every look at the state forces the pending search) and at full speed, then the rate of both.
usage: python tools/exp_pipeline.py [--size 640x480] [--frames 200] [--passes 3]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rpg_open_remode_amd import api, synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="640x480"); ap.add_argument("--frames", type=int, default=200); ap.add_argument("--passes", type=int, default=3)
ap.add_argument("--side", type=int, default=9); ap.add_argument("--skip-parity", action="store_true")
a = ap.parse_args()
W, H = (int(v) for v in a.size.split("x")); F = a.frames
seq = synth.Sequence(W, H, F, 0)
cam = api.PinholeCamera(*seq.K)
dev = []
for k in range(F):
    d = api.DeviceImage(W, H, np.float32); d.setDevData(seq.images[k]); dev.append(d)


def mismatches(sa, sb):
    bad = 0
    for p in range(len(sa)):
        x, y = sa[p], sb[p]
        bad += int(np.count_nonzero(~((x == y) | (np.isnan(x) & np.isnan(y))))) if x.dtype.kind == "f" else int((x != y).sum())
    return bad


def make(pipeline):
    s = api.SeedMatrix(W, H, cam, patch_side=a.side)
    s.setOption(api.OPT_PIPELINE, pipeline)
    return s


def run(s, observe=()):
    s.setReferenceImageDevice(dev[0].data, dev[0].stride, seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    out = {}
    for k in range(1, F):
        s.updateDevice(dev[k].data, dev[k].stride, seq.T_curr_world[k])
        if k in observe:
            out[k] = s.state()
    return out


rc = 0
if not a.skip_parity:
    A, B = make(0), make(1)
    obs = (1, 2, 3, 7, 20, 21, 22, 60, 100, 101, 150, F - 1)
    sa, sb = run(A, obs), run(B, obs)
    for k in obs:
        m = mismatches(sa[k], sb[k])
        if m:
            rc = 1
        print(f"with observers: update {k}: {m} mismatches", flush=True)
    print("converged", A.getConvergedCount(), B.getConvergedCount())
    # full speed: nobody looks until the end; twice (the second pass restarts from a new reference with a search still pending)
    for rep in range(2):
        run(A); run(B)
        m = mismatches(A.state(), B.state())
        print(f"full speed, pass {rep + 1}: {m} mismatches after {F - 1} updates; converged {A.getConvergedCount()} / {B.getConvergedCount()}", flush=True)
        rc |= m != 0
    del A, B
for pipeline in (0, 1, 0, 1):
    s = make(pipeline)
    run(s); s.sync()
    t0 = time.perf_counter()
    for _ in range(a.passes):
        run(s)
    s.sync()
    dt = time.perf_counter() - t0
    print(f"pipeline {pipeline}: {dt / (a.passes * (F - 1)) * 1e6:.2f} us per update, {W * H * a.passes * (F - 1) / dt / 1e6:.0f} Mpix/s; converged {s.getConvergedCount()}", flush=True)
    del s
sys.exit(rc)
