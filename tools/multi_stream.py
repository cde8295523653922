#!/usr/bin/env python3
"""Zero-code experiment for the batched mode: N independent SeedMatrix handles (each has its own HIP stream) driven over the
same resident 640x480 sequence from N host threads of one process (ctypes releases the GIL inside the library calls).
Prints the aggregate rate per N.  usage: python tools/multi_stream.py [--n 1,2,4] [--frames 200] [--passes 3]"""
import argparse, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rpg_open_remode_amd import api, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", default="1,2,4"); ap.add_argument("--size", default="640x480"); ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--passes", type=int, default=3); ap.add_argument("--side", type=int, default=9)
a = ap.parse_args()
W, H = (int(v) for v in a.size.split("x"))
F = a.frames
seq = synth.Sequence(W, H, F)
frames = []
for k in range(F):
    d = api.DeviceImage(W, H, np.float32)
    d.setDevData(seq.images[k])
    frames.append(d)
poses = [np.ascontiguousarray(np.asarray(T, np.float32).reshape(12)) for T in seq.T_curr_world]


def run_pass(s):
    s.setReferenceImageDevice(frames[0].data, frames[0].stride, poses[0], seq.min_depth, seq.max_depth)
    for k in range(1, F):
        s.updateDevice(frames[k].data, frames[k].stride, poses[k])


for n in (int(v) for v in a.n.split(",")):
    handles = [api.SeedMatrix(W, H, api.PinholeCamera(*seq.K), patch_side=a.side) for _ in range(n)]
    for s in handles:
        run_pass(s)
        s.sync()
    start = threading.Barrier(n + 1)

    def worker(s):
        start.wait()
        for _ in range(a.passes):
            run_pass(s)
        s.sync()

    ts = [threading.Thread(target=worker, args=(s,)) for s in handles]
    for t in ts:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    upd = n * a.passes * (F - 1)
    print(f"{n} handles / streams / host threads: {W * H * upd / dt / 1e6:.0f} Mpix/s aggregate, {dt / (a.passes * (F - 1)) * 1e6:.1f} us per "
          f"update step of all {n} sequences ({dt / upd * 1e6:.1f} us per update)", flush=True)
    del handles
