#!/usr/bin/env python3
"""Per-update instruction counts of the two update kernels from a rocprofv3 --pmc run (per-dispatch counter_collection.csv), joined with the
per-update work (live seeds, NCC evaluations) of the same sequence: which part of the VALU instructions is the NCC block (851 per wave-
evaluation of 64 lanes at side 9) and which is everything around it.
usage: python tools/pmc_frames.py <dir with *counter_collection.csv> [stats.json]   (stats.json: written by tools/frame_stats.py)"""
import csv, glob, json, os, sys, collections

NCC_VALU_PER_WAVE_EVAL = 851  # side 9, DESIGN.md 4.1

d = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        kind = "setup" if "seed_setup_compact" in name else "search" if "seed_search_compact" in name else None
        if kind:
            d.setdefault((int(r["Dispatch_Id"]), kind), {})[r["Counter_Name"]] = float(r["Counter_Value"])
setup = [v for (i, k), v in sorted(d.items()) if k == "setup"]
search = [v for (i, k), v in sorted(d.items()) if k == "search"]
stats = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None
n = min(len(setup), len(search))
per_pass = len(stats["evals"]) if stats else n
print(f"# {n} update() launches; columns: update, setup VALU, search VALU, NCC evaluations, useful = evals / 64 * {NCC_VALU_PER_WAVE_EVAL}, useful / (setup + search)")
tot = [0.0, 0.0, 0.0]
for i in range(n):
    sv, qv = setup[i].get("SQ_INSTS_VALU", 0.0), search[i].get("SQ_INSTS_VALU", 0.0)
    ev = stats["evals"][i % per_pass] if stats else 0
    useful = ev / 64.0 * NCC_VALU_PER_WAVE_EVAL
    tot[0] += sv; tot[1] += qv; tot[2] += useful
    k = i % per_pass + 1
    if k <= 24 or k % 10 == 0:
        print(f"{k:4d} {sv:12.0f} {qv:12.0f} {ev:9d} {useful:12.0f} {useful / max(sv + qv, 1):6.3f}" + (f"  live {stats['live'][i % per_pass]}" if stats else ""))
print(f"mean per update: setup {tot[0] / n:.0f}, search {tot[1] / n:.0f}, total {(tot[0] + tot[1]) / n:.0f}, useful {tot[2] / n:.0f} = useful_valu_frac {tot[2] / max(tot[0] + tot[1], 1):.4f}")
