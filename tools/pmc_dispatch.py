#!/usr/bin/env python3
"""Per-dispatch view of a rocprofv3 --pmc counter_collection.csv: duration + counters for kernels matching a substring.
usage: pmc_dispatch.py <dir> <kernel-substring> [max_rows]"""
import csv, glob, os, sys
from collections import OrderedDict, defaultdict
d, sub = sys.argv[1], sys.argv[2]
maxr = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
rows = OrderedDict()
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if sub not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        e = rows.setdefault(k, {"dur_us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "grid": int(r["Grid_Size"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = sorted({c for e in rows.values() for c in e if c not in ("dur_us", "grid")})
print("disp".rjust(6), "dur_us".rjust(8), "grid".rjust(7), " ".join(n.replace("SQ_", "")[:14].rjust(14) for n in names))
tot = defaultdict(float)
for i, (k, e) in enumerate(rows.items()):
    for n in ["dur_us"] + names:
        tot[n] += e.get(n, 0.0)
    if i < maxr:
        print(str(k).rjust(6), f"{e['dur_us']:8.1f}", str(e["grid"]).rjust(7), " ".join(f"{e.get(n, 0):14.0f}" for n in names))
print("TOTAL".rjust(6), f"{tot['dur_us']:8.1f}", " " * 7, " ".join(f"{tot[n]:14.0f}" for n in names), f" n={len(rows)}")
