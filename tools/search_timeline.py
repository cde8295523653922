#!/usr/bin/env python3
"""In-kernel timeline of the compact search kernel (tile pipeline, OPT_MATCHER 3) on the benchmark sequence.
usage: python tools/search_timeline.py [--size WxH] [--frames N] [--side S] [--show 1,5,...]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rpg_open_remode_amd import api, synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="640x480"); ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--side", type=int, default=9)
ap.add_argument("--show", default="1,5,20,60,100,150,199")
ap.add_argument("--u8", action="store_true", help="frames handed over as 8-bit host images (ingest fused into the setup kernel)")
ap.add_argument("--brief", action="store_true", help="one line for EVERY update, nothing else")
a = ap.parse_args()
W, H = (int(v) for v in a.size.split("x"))
seq = synth.Sequence(W, H, a.frames)
frames = []
for im in seq.images:
    d = api.DeviceImage(W, H, np.float32); d.setDevData(im); frames.append(d)
for rep in range(2):  # first pass warms up
    s = api.SeedMatrix(W, H, api.PinholeCamera(*seq.K), patch_side=a.side)
    s.setOption(api.OPT_MATCHER, 3)
    s.setReferenceImageDevice(frames[0].data, frames[0].stride, seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    s.sync()
    s.setOption(api.OPT_COLLECT_STATS, 2)
    for k in range(1, a.frames):
        if a.u8:
            s.updateU8(seq.gray[k], seq.T_curr_world[k])
        else:
            s.updateDevice(frames[k].data, frames[k].stride, seq.T_curr_world[k])
    s.sync()
us = lambda t: float(t) / 100.0
show = set(int(v) for v in a.show.split(","))
print("search kernel only; us relative to its first workgroup start; busy = workgroups that searched at least one unit")
print(f"{'frame':>5} | {'busy':>5} | 1st unit ready p50 {'p99':>6} | end p50 {'p90':>6} {'p99':>6} {'max':>7} | {'items':>8} {'units':>6} {'units/wg max':>12} {'windows':>7}")
for k in range(a.frames - 1):
    t_all = s.frameTraceDownload(k).astype(np.int64)
    t = t_all[t_all[:, 0] != 0]
    if a.brief and len(t):
        b = t[t[:, 5] > 0]
        t0 = t[:, 0].min()
        ex = b[:, 3] - t0
        print(f"{k + 1:5d} | busy {len(b):5d} | end p50 {us(np.percentile(ex, 50)):6.1f} max {us(ex.max()):6.1f} | items {int(b[:, 4].sum()):8d} units {int(b[:, 5].sum()):6d} windows staged in the search {int((b[:, 7] >> 32).sum()):6d}")
        continue
    if k + 1 not in show or len(t) == 0:
        continue
    t0 = t[:, 0].min()
    b = t[t[:, 5] > 0]
    rd, ex = b[:, 1] - t0, b[:, 3] - t0
    print(f"{k + 1:5d} | {len(b):5d} | {us(np.percentile(rd, 50)):17.1f} {us(np.percentile(rd, 99)):6.1f} | {us(np.percentile(ex, 50)):7.1f} {us(np.percentile(ex, 90)):6.1f} "
          f"{us(np.percentile(ex, 99)):6.1f} {us(ex.max()):7.1f} | {int(b[:, 4].sum()):8d} {int(b[:, 5].sum()):6d} {int(b[:, 5].max()):12d} {int((b[:, 7] >> 32).sum()):7d}")
    w2 = t_all[:, 2]
    w2 = w2[w2 != 0]
    if len(w2):  # setup kernel probes: start (low 32 bits), state ready, end (deltas)
        st = w2 & 0xffffffff
        st = st - st.min()
        rd, en = (w2 >> 32) & 0xffff, (w2 >> 48) & 0xffff
        gap = (t0 & 0xffffffff) - ((w2 & 0xffffffff).min())
        print(f"      setup kernel: workgroup start p50 {us(np.percentile(st, 50)):.1f} p99 {us(np.percentile(st, 99)):.1f} max {us(st.max()):.1f}; state ready after "
              f"p50 {us(np.percentile(rd, 50)):.1f} p99 {us(np.percentile(rd, 99)):.1f}; workgroup lifetime p50 {us(np.percentile(en, 50)):.1f} p99 {us(np.percentile(en, 99)):.1f}; "
              f"last end {us((st + en).max()):.1f}; first search workgroup starts {us(gap):.1f} after the first setup workgroup")
    order = np.argsort(-(b[:, 3] - t0))[:5]
    print("      slowest workgroups (start, 1st unit ready, end; items, units, last tile (x, y), fallback evals of wave 0, windows staged):")
    tx = (W + 15) // 16
    for i in order:
        print(f"        {us(b[i, 0] - t0):6.1f} {us(b[i, 1] - t0):6.1f} {us(b[i, 3] - t0):6.1f}; {int(b[i, 4]):6d} {int(b[i, 5]):3d} ({int(b[i, 6]) % tx:2d},{int(b[i, 6]) // tx:2d}) "
              f"{int(b[i, 7] & 0xffffffff):5d} {int(b[i, 7] >> 32):3d}")
