#!/usr/bin/env python3
"""Rate of the batched mode (rmd_hip_batch_*): B independent 640x480 sequences (scene seeds 0..B-1) stepped together, one launch pair
per step, next to the same sequences stepped alone.  Frames resident in HBM, and 8-bit frames from host memory (--u8).
usage: python tools/batch_bench.py [--b 1,2,4,8] [--frames 200] [--passes 3] [--u8] [--unit-target 1]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rpg_open_remode_amd import api, synth

ap = argparse.ArgumentParser()
ap.add_argument("--b", default="1,2,4,8"); ap.add_argument("--size", default="640x480"); ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--passes", type=int, default=3); ap.add_argument("--side", type=int, default=9); ap.add_argument("--u8", action="store_true")
ap.add_argument("--unit-target", type=int, default=1); ap.add_argument("--same-scene", action="store_true")
ap.add_argument("--flags", default="6", help="(retired; kept so that old command lines still parse)")
ap.add_argument("--per-step", action="store_true", help="also: one more pass with a synchronisation after every step, wall time of selected steps")
a = ap.parse_args()
W, H = (int(v) for v in a.size.split("x"))
F = a.frames
sizes = [int(v) for v in a.b.split(",")]
n_max = max(sizes)
seqs = [synth.Sequence(W, H, F, s) for s in range(1 if a.same_scene else n_max)]
if a.same_scene:
    seqs = seqs * n_max  # one rendering, the same work for every member
dev = []
for seq in seqs[:1] if a.same_scene else seqs:
    fr = []
    for k in range(F):
        d = api.DeviceImage(W, H, np.float32)
        d.setDevData(seq.images[k])
        fr.append(d)
    dev.append(fr)
if a.same_scene:
    dev = dev * n_max


def run_pass(b, n):
    for i in range(n):
        seq = seqs[i]
        b[i].setReferenceImageDevice(dev[i][0].data, dev[i][0].stride, seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    for k in range(1, F):
        poses = [seqs[i].T_curr_world[k] for i in range(n)]
        if a.u8:
            b.updateU8([seqs[i].gray[k] for i in range(n)], poses)
        else:
            b.updateDevice([dev[i][k].data for i in range(n)], [dev[i][k].stride for i in range(n)], poses)


for n, flags in [(n, f) for f in (int(v) for v in a.flags.split(",")) for n in sizes]:
    b = api.SeedMatrixBatch(n, W, H, api.PinholeCamera(*seqs[0].K), patch_side=a.side)
    b.setOption(api.OPT_UNIT_TARGET, a.unit_target)
    run_pass(b, n)
    b.sync()
    b.setOption(api.OPT_TIMING, 2)
    b.timingReset()
    t0 = time.perf_counter()
    for _ in range(a.passes):
        run_pass(b, n)
    b.sync()
    dt = time.perf_counter() - t0
    ms, steps = b.timing()
    upd = n * a.passes * (F - 1)
    conv = [b[i].getConvergedCount() for i in range(n)]
    print(f"[flags {flags}] batch of {n} ({'8-bit host frames' if a.u8 else 'resident frames'}): {W * H * upd / dt / 1e6:.0f} Mpix/s, {dt / (a.passes * (F - 1)) * 1e6:.1f} us per step "
          f"({dt / upd * 1e6:.1f} us per sequence update; device {ms / max(steps, 1) * 1e3:.1f} us per step); converged {conv}", flush=True)
    if a.per_step:
        for i in range(n):
            b[i].setReferenceImageDevice(dev[i][0].data, dev[i][0].stride, seqs[i].T_curr_world[0], seqs[i].min_depth, seqs[i].max_depth)
        b.sync()
        ts = []
        for k in range(1, F):
            t1 = time.perf_counter()
            b.updateDevice([dev[i][k].data for i in range(n)], [dev[i][k].stride for i in range(n)], [seqs[i].T_curr_world[k] for i in range(n)])
            b.sync()  # (runs the stand-alone finalisation too: ~7 us per member on top of the pipelined step)
            ts.append((time.perf_counter() - t1) * 1e6)
        sel = [1, 2, 4, 8, 12, 16, 20, 25, 30, 35, 40, 50, 60, 70, 80, 90, 100, 120, 150, 180, 199]
        print("   step:us (synchronised after every step) " + " ".join(f"{k}:{ts[k - 1]:.0f}" for k in sel if k < F) + f"  | sum {sum(ts) / 1e3:.1f} ms", flush=True)
    del b
