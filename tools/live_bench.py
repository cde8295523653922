#!/usr/bin/env python3
"""Throughput in LIVE use: the node's state machine (rpg_open_remode_amd/depthmap_node.py) over the benchmark
sequence -- 8-bit frames from host memory, a converged-seed count after every update, and at every reference change a TV-L1
denoise (0.5, 200), the convergence map and the point cloud.  usage: python tools/live_bench.py [--size WxH] [--side S]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rpg_open_remode_amd import api, synth
from rpg_open_remode_amd.depthmap_node import DepthmapNode

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="640x480"); ap.add_argument("--side", type=int, default=9); ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--ref-compl-perc", type=float, default=10.0); ap.add_argument("--max-dist-from-ref", type=float, default=0.5)
ap.add_argument("--breakdown", action="store_true", help="host wall time per kind of call (timers around the Depthmap methods the node uses)")
a = ap.parse_args()
W, H = (int(v) for v in a.size.split("x"))
seq = synth.Sequence(W, H, a.frames)
poses = [api.SE3(T).inv() for T in seq.T_curr_world]
published = {"depth": 0, "points": 0, "conv": 0}
def on_depth(d): published["depth"] += 1
def on_pc(p): published["points"] = len(p)
def on_conv(c): published["conv"] += 1
node = DepthmapNode(W, H, *seq.K, ref_compl_perc=a.ref_compl_perc, max_dist_from_ref=a.max_dist_from_ref, patch_side=a.side,
                    on_depthmap=on_depth, on_pointcloud=on_pc, on_convergence=on_conv)
from rpg_open_remode_amd.depthmap_node import State
spent = {}
if a.breakdown:
    def timed(obj, name, key):
        f = getattr(obj, name)
        def g(*args, **kw):
            t = time.perf_counter()
            try:
                return f(*args, **kw)
            finally:
                spent[key] = spent.get(key, 0.0) + time.perf_counter() - t
        setattr(obj, name, g)
    dm = node.depthmap_
    timed(dm, "setReferenceImage", "setReferenceImage (upload + seed_init)")
    timed(dm, "update", "update (8-bit host frame)")
    timed(dm, "getConvergedPercentage", "getConvergedPercentage (waits for the update's setup kernel)")
    timed(dm, "downloadDenoisedDepthmap", "TV-L1 denoise + download")
    timed(dm, "downloadConvergenceMap", "convergence map download")
    timed(dm, "downloadPointCloud", "point cloud (device compaction + download)")
    timed(node.publisher_, "publishConvergenceMap", "coloured convergence map (device kernel + 3 B/pixel download)")
    timed(node.publisher_, "publishPointCloud", "point cloud publication incl. the growing host-side concatenation")
for rep in range(2):  # ONE node, two passes over the sequence: the first pays for every buffer the library allocates on first use
    node.state_ = State.TAKE_REFERENCE_FRAME
    node.num_msgs_ = 0
    node.references_taken = node.updates_done = 0
    node.publisher_.pc_ = np.zeros((0, 4), np.float32)
    for key in published: published[key] = 0
    spent.clear()
    t0 = time.perf_counter()
    for k in range(a.frames):
        node.denseInput(seq.gray[k], poses[k], seq.min_depth, seq.max_depth)
    node.depthmap_.seeds_.sync()
    dt = time.perf_counter() - t0
for key, v in sorted(spent.items(), key=lambda kv: -kv[1]):
    print(f"   {v * 1e3:7.2f} ms  {key}")
print(f"live mode {W}x{H} side {a.side}: {a.frames} messages in {dt * 1e3:.1f} ms = {W * H * a.frames / dt / 1e6:.0f} Mpix/s "
      f"({dt / a.frames * 1e6:.0f} us per message); {node.references_taken} reference frames, {node.updates_done} updates, "
      f"{published['depth']} depth maps / point clouds published ({published['points']} points accumulated), {published['conv']} convergence maps")
