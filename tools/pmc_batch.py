#!/usr/bin/env python3
"""Counter totals of the update kernels of a BATCH run from a rocprofv3 --pmc run (per-dispatch counter_collection.csv): instructions per
step and per sequence-update -- measured on the batch itself, next to the single sequence's per-update counts that bench.py multiplies by B.
usage: python tools/pmc_batch.py <dir with *counter_collection.csv> <sequences> <updates per sequence in the profiled run>"""
import csv, glob, os, sys, collections

d = collections.defaultdict(lambda: collections.defaultdict(float))
n_disp = collections.Counter()
seen = set()
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        kind = "seed_setup" if "seed_setup_compact" in name else "seed_search" if "seed_search_compact" in name else "seed_finalize" if "seed_finalize" in name else None
        if not kind:
            continue
        d[kind][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], kind) not in seen:
            seen.add((r["Dispatch_Id"], kind)); n_disp[kind] += 1
B, U = int(sys.argv[2]), int(sys.argv[3])
print(f"# batch of {B}, {U} updates per sequence in the profiled run (warm-up pass + timed passes); launches: " + ", ".join(f"{k} {v}" for k, v in n_disp.items()))
tot = collections.defaultdict(float)
for kind, c in d.items():
    for k, v in c.items():
        tot[k] += v
    print(f"{kind:14s} " + "  ".join(f"{k} {v / (B * U) / 1e6:.3f} M per sequence-update" for k, v in sorted(c.items())))
print("all            " + "  ".join(f"{k} {v / (B * U) / 1e6:.3f} M per sequence-update ({v / U / 1e6:.2f} M per step of {B})" for k, v in sorted(tot.items())))
