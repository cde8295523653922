#!/usr/bin/env python3
"""Per-frame tile-kernel diagnostics on the bench workload. usage: diag.py [frames]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpg_open_remode_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
w, h, side = 640, 480, 9
seq = synth.Sequence(w, h, n, seed=0)
s = api.SeedMatrix(w, h, api.PinholeCamera(*seq.K), patch_side=side)
s.setOption(api.OPT_COLLECT_STATS, 1); s.setOption(api.OPT_TIMING, 1)
s.setReferenceImage(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
print("frame us evals lds glob irr stageMcyc searchMcyc maxwgKcyc units")
for k in range(1, n):
    s.timingReset()
    s.update(seq.images[k], seq.T_curr_world[k])
    d = s.lastDiagnostics(); ms, _ = s.timing(api.STAGE_UPDATE)
    if k < 12 or k % 5 == 0:
        print(k, round(ms * 1e3, 1), d["ncc_evals"], d["evals_lds"], d["evals_global"], d["evals_irregular"],
              round(d["cycles_stage"] / 1e6, 2), round(d["cycles_search"] / 1e6, 2), round(d["max_wg_cycles"] / 1e3, 1), d["rounds"])
