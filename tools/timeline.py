#!/usr/bin/env python3
"""In-kernel timeline of the update pipeline on the benchmark sequence (RMD_HIP_OPT_COLLECT_STATS = 2): when do the
workgroups of seed_setup / seed_plan / seed_search start and end inside one update, in streaming use (no observers).
usage: python tools/timeline.py [--size WxH] [--frames N] [--window 0|1|2] [--side S]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rpg_open_remode_amd import api, synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="640x480"); ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--window", type=int, default=0); ap.add_argument("--side", type=int, default=9)
a = ap.parse_args()
W, H = (int(v) for v in a.size.split("x"))
seq = synth.Sequence(W, H, a.frames)
frames = []
for im in seq.images:
    d = api.DeviceImage(W, H, np.float32); d.setDevData(im); frames.append(d)
for rep in range(2):  # first pass warms up
    s = api.SeedMatrix(W, H, api.PinholeCamera(*seq.K), patch_side=a.side)
    s.setOption(api.OPT_WINDOW, a.window)
    s.setReferenceImageDevice(frames[0].data, frames[0].stride, seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    s.sync()
    s.setOption(api.OPT_COLLECT_STATS, 2)
    for k in range(1, a.frames):
        s.updateDevice(frames[k].data, frames[k].stride, seq.T_curr_world[k])
    s.sync()
us = lambda t: float(t) / 100.0
print("us relative to the first seed_setup workgroup start of the frame; L = workgroup lifetime; prev = end of the previous frame's search")
print(f"{'frame':>5} {'prev':>6} | setup: {'lastStart':>9} {'end':>6} {'avgL':>5} {'p99L':>5} {'maxL':>5} | plan: {'start':>6} {'end':>6} | "
      f"search: {'n':>4} {'start':>6} {'lastSt':>6} {'end':>6} {'avgL':>6} {'minEnd':>6}")
rows = []
prev_end = None
for k in range(a.frames - 1):
    t = s.traceDownload(k)
    S, P, Q = t["setup"].astype(np.int64), t["plan"].astype(np.int64), t["search"].astype(np.int64)
    unit_rounds = int(P[0, 1] >> 56); P[0, 1] &= (1 << 56) - 1
    t0 = S[:, 0].min()
    L = S[:, 1] - S[:, 0]
    row = [us(prev_end - t0) if prev_end is not None else 0.0, us(S[:, 0].max() - t0), us(S[:, 1].max() - t0), us(L.mean()), us(np.percentile(L, 99)),
           us(L.max()), us(P[0, 0] - t0), us(P[0, 1] - t0), len(Q), us(Q[:, 0].min() - t0), us(Q[:, 0].max() - t0), us(Q[:, 1].max() - t0),
           us((Q[:, 1] - Q[:, 0]).mean()), us(Q[:, 1].min() - t0)]
    prev_end = Q[:, 1].max()
    rows.append(row)
    if k + 1 in (100, 150):
        order = np.argsort(-L)[:14]
        tx = (W + 15) // 16
        print("      slowest setup tiles (tile_x, tile_y: start..end us):", ", ".join(f"({i % tx},{i // tx}: {us(S[i, 0] - t0):.1f}..{us(S[i, 1] - t0):.1f})" for i in order))
    QL = np.sort(Q[:, 1] - Q[:, 0])
    if k + 1 in (20, 40, 100, 150, a.frames - 1):
        busy = QL[QL > 50]  # > 0.5 us: had at least one unit
        print(f"      search lifetimes of the {len(busy)} busy workgroups: p10 {us(np.percentile(busy, 10)):.1f} p50 {us(np.percentile(busy, 50)):.1f} "
              f"p90 {us(np.percentile(busy, 90)):.1f} p99 {us(np.percentile(busy, 99)):.1f} max {us(busy.max()):.1f};  setup lifetimes: p50 {us(np.percentile(L, 50)):.1f} "
              f"p90 {us(np.percentile(L, 90)):.1f} p99 {us(np.percentile(L, 99)):.1f}; setup ends: p50 {us(np.percentile(S[:, 1] - t0, 50)):.1f} p90 {us(np.percentile(S[:, 1] - t0, 90)):.1f}")
    if k + 1 in (1, 2, 5, 10, 20, 40, 60, 100, 150, a.frames - 1):
        print(f"{k + 1:5d} {row[0]:6.1f} | {row[1]:16.1f} {row[2]:6.1f} {row[3]:5.1f} {row[4]:5.1f} {row[5]:5.1f} | {row[6]:12.1f} {row[7]:6.1f} | "
              f"r{unit_rounds} {row[8]:9d} {row[9]:6.1f} {row[10]:6.1f} {row[11]:6.1f} {row[12]:6.1f} {row[13]:6.1f}")
A = np.array(rows)[1:]
m = A.mean(0)
print(f" mean {m[0]:6.1f} | {m[1]:16.1f} {m[2]:6.1f} {m[3]:5.1f} {m[4]:5.1f} {m[5]:5.1f} | {m[6]:12.1f} {m[7]:6.1f} | {m[8]:12.0f} {m[9]:6.1f} {m[10]:6.1f} {m[11]:6.1f} {m[12]:6.1f} {m[13]:6.1f}")
