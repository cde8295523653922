#!/usr/bin/env python3
"""Throughput in LIVE use: the node's state machine (rpg_open_remode_amd/depthmap_node.py) over the benchmark
sequence (rpg_open_remode_amd/live.py), with publication inside the callback like the reference and off the update stream.
usage: python tools/live_bench.py [--size WxH] [--side S] [--breakdown] [--modes sync,async]"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rpg_open_remode_amd import live, synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="640x480"); ap.add_argument("--side", type=int, default=9); ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--ref-compl-perc", type=float, default=10.0); ap.add_argument("--max-dist-from-ref", type=float, default=0.5)
ap.add_argument("--breakdown", action="store_true", help="host wall time per kind of call (timers around the Depthmap methods the node uses)")
ap.add_argument("--modes", default="sync,async"); ap.add_argument("--passes", type=int, default=3)
a = ap.parse_args()
W, H = (int(v) for v in a.size.split("x"))
seq = synth.Sequence(W, H, a.frames)
for mode in a.modes.split(","):
    r = live.run_live(W, H, a.frames, a.side, a.ref_compl_perc, a.max_dist_from_ref, async_publish=(mode == "async"), passes=a.passes, breakdown=a.breakdown, seq=seq)
    for key, v in (r["host_ms_by_call"] or {}).items():
        print(f"   {v:7.2f} ms  {key}")
    print(f"live mode {W}x{H} side {a.side}, publication {mode}: {r['messages']} messages in {r['ms']:.1f} ms = {r['value']:.0f} Mpix/s ({r['us_per_message']:.0f} us per message); "
          f"{r['reference_frames']} reference frames, {r['updates']} updates, {r['depth_maps_and_clouds']} depth maps / point clouds published "
          f"({r['points_accumulated']} points accumulated), {r['convergence_maps']} convergence maps")
