#!/usr/bin/env python3
"""Hunting an intermittent multi-millisecond stall of the 8-bit-host-frame path: N passes of the headline workload one by one (wall time per
pass, the caller at most ring - 1 frames ahead of the device), the slowest passes, and -- with RMD_HIP_INGEST_PROFILE=1 -- the library's own
account (longest wait for a ring slot, waits that gave up, frames the setup kernel waited for).   usage: python tools/r06_stall.py [passes=60]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rpg_open_remode_amd import api, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
W, H, F = 640, 480, 200
seq = synth.Sequence(W, H, F, 0)
s = api.SeedMatrix(W, H, api.PinholeCamera(*seq.K), patch_side=9)


def one_pass():
    marks = []
    s.setReferenceImageU8(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    for k in range(1, F):
        t = time.perf_counter()
        s.updateU8(seq.gray[k], seq.T_curr_world[k])
        marks.append(time.perf_counter() - t)
    return marks


for _ in range(3):
    one_pass()
s.sync()
times, worst_calls = [], []
for p in range(N):
    t0 = time.perf_counter()
    m = one_pass()
    times.append((time.perf_counter() - t0) * 1e3)
    k = int(np.argmax(m))
    worst_calls.append((max(m) * 1e6, k + 1))
s.sync()
t = np.array(times)
print(f"{N} passes: median {np.median(t):.3f} ms, min {t.min():.3f}, max {t.max():.3f}; passes more than 1 ms over the median: "
      f"{[(i, round(float(v), 2), 'longest update() call %.0f us at update %d' % worst_calls[i]) for i, v in enumerate(t) if v > np.median(t) + 1.0]}")
del s
