#!/usr/bin/env python3
"""Experiment: S batches of B sequences each, every batch on its own stream pair and host thread (same scene for all members).
usage: python tools/multi_batch.py --configs 1x4,2x2,2x4,4x2,1x8 [--frames 200] [--passes 2]"""
import argparse, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rpg_open_remode_amd import api, synth

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="1x4,2x2,2x4,4x2,1x8"); ap.add_argument("--frames", type=int, default=200); ap.add_argument("--passes", type=int, default=2)
a = ap.parse_args()
W, H, F, SIDE = 640, 480, a.frames, 9
seq = synth.Sequence(W, H, F, 0)
dev = []
for k in range(F):
    d = api.DeviceImage(W, H, np.float32); d.setDevData(seq.images[k]); dev.append(d)


def run_pass(b, n):
    for i in range(n):
        b[i].setReferenceImageDevice(dev[0].data, dev[0].stride, seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    for k in range(1, F):
        b.updateDevice([dev[k].data] * n, [dev[k].stride] * n, [seq.T_curr_world[k]] * n)


for cfg in a.configs.split(","):
    S, B = (int(v) for v in cfg.split("x"))
    batches = [api.SeedMatrixBatch(B, W, H, api.PinholeCamera(*seq.K), patch_side=SIDE) for _ in range(S)]
    for b in batches:
        run_pass(b, B); b.sync()
    start = threading.Barrier(S + 1)

    def worker(b):
        start.wait()
        for _ in range(a.passes):
            run_pass(b, B)
        b.sync()
    ts = [threading.Thread(target=worker, args=(b,)) for b in batches]
    for t in ts: t.start()
    start.wait(); t0 = time.perf_counter()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    upd = S * B * a.passes * (F - 1)
    print(f"{S} stream(s) x batch of {B}: {W * H * upd / dt / 1e6:.0f} Mpix/s ({dt / upd * 1e6:.1f} us per sequence update)", flush=True)
    del batches
