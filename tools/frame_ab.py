#!/usr/bin/env python3
"""GPU-only regression + timing of the seed-update variants on a full-size sequence.

    frame_ab.py [--size WxH] [--frames N] [--side S] [--variants 0,1,2,21] [--lib path/to/librmd_hip.so] [--no-parity]

Parity: every variant against the first one listed (the per-pixel kernel 0 is itself pinned to the oracle by
tests/test_hip_parity.py), every plane bit for bit at a few frames.  Timing: wall time of complete passes over
device-resident frames (what bench.py times), plus the time of the first 20 updates (every seed live) on their own.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="640x480")
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--side", type=int, default=9)
    ap.add_argument("--variants", default="0,2,21,1")
    ap.add_argument("--lib", default="")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--passes", type=int, default=3)
    a = ap.parse_args()
    if a.lib:
        from rpg_open_remode_amd import _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
    from rpg_open_remode_amd import api, synth
    w, h = (int(v) for v in a.size.split("x"))
    n = a.frames
    variants = [int(v) for v in a.variants.split(",")]
    seq = synth.Sequence(w, h, n, seed=0)
    cam = api.PinholeCamera(*seq.K)
    frames = []
    for im in seq.images:
        d = api.DeviceImage(w, h, np.float32)
        d.setDevData(im)
        frames.append(d)

    def make(v):
        s = api.SeedMatrix(w, h, cam, patch_side=a.side)
        if v in (31, 32):  # tile pipeline with 2x / 3x as many (smaller) work units
            s.setOption(api.OPT_MATCHER, 3); s.setOption(api.OPT_UNIT_TARGET, v - 29)
        elif v >= 20:  # 21: everything beyond one round handed out in 1-round units; 2xy: local_max x * 256, y rounds per unit
            s.setOption(api.OPT_MATCHER, 2)
            if v == 21:
                s.setOption(api.OPT_LOCAL_MAX, 256); s.setOption(api.OPT_UNIT_ROUNDS, 1)
            else:
                s.setOption(api.OPT_LOCAL_MAX, ((v // 10) % 100) * 256); s.setOption(api.OPT_UNIT_ROUNDS, v % 10)
        else:
            s.setOption(api.OPT_MATCHER, v)
        return s

    def set_ref(s):
        s.setReferenceImageDevice(frames[0].data, frames[0].stride, seq.T_curr_world[0], seq.min_depth, seq.max_depth)

    bad = 0
    if not a.no_parity:
        check_at = sorted(set([1, 2, 3, 5, 8, 13, 20, 40, 80, 120, n - 1]) & set(range(1, n)))
        seeds = [make(v) for v in variants]
        for s in seeds:
            set_ref(s)
        for k in range(1, n):
            for s in seeds:
                s.updateDevice(frames[k].data, frames[k].stride, seq.T_curr_world[k])
            if k in check_at:
                ref = seeds[0].state()
                for v, s in zip(variants[1:], seeds[1:]):
                    st = s.state()
                    for p in range(8):
                        x, y = ref[p], st[p]
                        m = int(np.count_nonzero(~((x == y) | (np.isnan(x) & np.isnan(y))))) if x.dtype.kind == "f" else int((x != y).sum())
                        if m:
                            bad += m
                            print(f"frame {k} variant {v} plane {p}: {m} mismatches vs variant {variants[0]}")
        print(f"parity over {n - 1} updates at {w}x{h} side {a.side}: {'OK' if not bad else f'{bad} MISMATCHES'} (variants {variants})")
        del seeds

    for v in variants:
        s = make(v)
        set_ref(s)
        for k in range(1, n):
            s.updateDevice(frames[k].data, frames[k].stride, seq.T_curr_world[k])
        s.sync()
        t0 = time.perf_counter()
        for _ in range(a.passes):
            set_ref(s)
            for k in range(1, n):
                s.updateDevice(frames[k].data, frames[k].stride, seq.T_curr_world[k])
        s.sync()
        t_pass = (time.perf_counter() - t0) / a.passes
        n_head = min(20, n - 1)
        t0 = time.perf_counter()
        for _ in range(a.passes):
            set_ref(s)
            for k in range(1, n_head + 1):
                s.updateDevice(frames[k].data, frames[k].stride, seq.T_curr_world[k])
        s.sync()
        t_head = (time.perf_counter() - t0) / a.passes
        print(f"variant {v:3d}: {t_pass / (n - 1) * 1e6:8.2f} us/update over {n - 1} updates = {w * h * (n - 1) / t_pass / 1e6:9.1f} Mpix/s;"
              f"  first {n_head} updates {t_head / n_head * 1e6:8.2f} us/update", flush=True)
        del s
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
