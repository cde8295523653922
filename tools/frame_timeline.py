#!/usr/bin/env python3
"""In-kernel timeline of the one-launch frame kernel on the benchmark sequence (RMD_HIP_OPT_COLLECT_STATS = 2).
usage: python tools/frame_timeline.py [--size WxH] [--frames N] [--side S] [--local-max M] [--show 1,2,5,...]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rpg_open_remode_amd import api, synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="640x480"); ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--side", type=int, default=9); ap.add_argument("--local-max", type=int, default=0)
ap.add_argument("--show", default="1,2,3,5,10,20,40,60,100,150,199")
a = ap.parse_args()
W, H = (int(v) for v in a.size.split("x"))
seq = synth.Sequence(W, H, a.frames)
frames = []
for im in seq.images:
    d = api.DeviceImage(W, H, np.float32); d.setDevData(im); frames.append(d)
for rep in range(2):  # first pass warms up
    s = api.SeedMatrix(W, H, api.PinholeCamera(*seq.K), patch_side=a.side)
    s.setOption(api.OPT_MATCHER, 2); s.setOption(api.OPT_LOCAL_MAX, a.local_max)
    s.setReferenceImageDevice(frames[0].data, frames[0].stride, seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    s.sync()
    s.setOption(api.OPT_COLLECT_STATS, 2)
    for k in range(1, a.frames):
        s.updateDevice(frames[k].data, frames[k].stride, seq.T_curr_world[k])
    s.sync()
us = lambda t: float(t) / 100.0
show = set(int(v) for v in a.show.split(","))
print("us relative to the first workgroup start of the frame; gap = first start - last exit of the previous frame")
print(f"{'frame':>5} {'gap':>5} | {'wgs':>5} {'lastSt':>6} | 1st tile set up p50 {'p99':>6} | tiles exhausted p50 {'max':>6} | exit p50 {'p90':>6} {'p99':>6} {'end':>7} | "
      f"{'items':>8} {'handed':>6} {'units':>5} {'tiles/wg max':>12}")
prev_end = None
ends = []
for k in range(a.frames - 1):
    t = s.frameTraceDownload(k).astype(np.int64)
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    sd, td, ex = t[:, 1] - t0, t[:, 2] - t0, t[:, 3] - t0
    sd = sd[t[:, 1] != 0]
    items = t[:, 4] & 0xffffffff
    pub = t[:, 4] >> 32
    gap = us(t0 - prev_end) if prev_end is not None else 0.0
    prev_end = t[:, 3].max()
    ends.append(us(ex.max()))
    if k + 1 in show:
        print(f"{k + 1:5d} {gap:5.1f} | {len(t):5d} {us((t[:, 0] - t0).max()):6.1f} | {us(np.percentile(sd, 50)):18.1f} {us(np.percentile(sd, 99)):6.1f} | "
              f"{us(np.percentile(td, 50)):19.1f} {us(td.max()):6.1f} | {us(np.percentile(ex, 50)):8.1f} {us(np.percentile(ex, 90)):6.1f} {us(np.percentile(ex, 99)):6.1f} "
              f"{us(ex.max()):7.1f} | {int(items.sum()):8d} {int(pub.sum()):6d} {int(t[:, 5].sum()):5d} {int(t[:, 6].max()):12d}")
    if k + 1 in show:
        order = np.argsort(-(t[:, 3] - t0))[:4]
        print("      slowest workgroups (start, 1st tile set up, tiles exhausted, exit; items, handed, units, tiles, fallback evals of wave 0, windows staged):")
        for i in order:
            print(f"        {us(t[i, 0] - t0):6.1f} {us(t[i, 1] - t0):6.1f} {us(t[i, 2] - t0):6.1f} {us(t[i, 3] - t0):6.1f}; {int(t[i, 4] & 0xffffffff):6d} {int(t[i, 4] >> 32):2d} {int(t[i, 5]):3d} "
                  f"{int(t[i, 6]):2d} {int(t[i, 7] & 0xffffffff):5d} {int(t[i, 7] >> 32):3d}")
print(f"mean frame span (first start -> last exit): {np.mean(ends):.1f} us; with the gaps {np.mean(ends) + 2.4:.1f} us")
