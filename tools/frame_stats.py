#!/usr/bin/env python3
"""Per-update work of the benchmark sequence (live seeds, search steps visited, NCC evaluations): one pass with the diagnostics counters on.
usage: python tools/frame_stats.py out.json [--size WxH] [--frames F]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rpg_open_remode_amd import api, synth

ap = argparse.ArgumentParser()
ap.add_argument("out"); ap.add_argument("--size", default="640x480"); ap.add_argument("--frames", type=int, default=200); ap.add_argument("--side", type=int, default=9)
a = ap.parse_args()
W, H = (int(v) for v in a.size.split("x"))
K = synth.intrinsics(W, H)
s = api.SeedMatrix(W, H, api.PinholeCamera(*K), patch_side=a.side)
s.setOption(api.OPT_COLLECT_STATS, 1)
out = {"live": [], "steps": [], "evals": [], "evals_from_l2": [], "units": [], "windows_in_search": []}
for k in range(a.frames):
    T = synth.pose(k, 0)
    g, rng = synth.render(W, H, T, 0, want_range=(k == 0), K=K)
    Tcw = np.ascontiguousarray(synth.invert_pose(T).astype(np.float32).reshape(12))
    if k == 0:
        s.setReferenceImageU8(g, Tcw, float(rng.min()), float(rng.max()))
        continue
    s.updateU8(g, Tcw)
    st = s.lastStats()
    out["live"].append(st["live_seeds"]); out["steps"].append(st["steps"]); out["evals"].append(st["ncc_evals"])
    d = s.lastDiagnosticsRaw()
    out["evals_from_l2"].append(d[3]); out["units"].append(d[4]); out["windows_in_search"].append(d[5])
json.dump(out, open(a.out, "w"))
print(f"{a.frames - 1} updates: mean live seeds {np.mean(out['live']):.0f}, mean NCC evaluations {np.mean(out['evals']):.0f}, of which from L2 {np.mean(out['evals_from_l2']):.0f}; "
      f"units {np.mean(out['units']):.0f}, windows staged in the search {np.mean(out['windows_in_search']):.0f}")
for k in (1, 4, 20, 24, 30, 40, 60, 100, 130, 160, 190):
    if k <= len(out["evals"]):
        print(f"update {k}: live {out['live'][k - 1]}, evaluations {out['evals'][k - 1]} (from L2 {out['evals_from_l2'][k - 1]}), units {out['units'][k - 1]}, windows staged in the search {out['windows_in_search'][k - 1]}")
