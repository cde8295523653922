#!/usr/bin/env python3
"""Per-launch durations of the one-launch-per-update experiment from a rocprofv3 kernel trace. usage: pipe_trace.py <trace dir> <frames>"""
import csv, glob, os, sys
F = int(sys.argv[2])
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(("merged" if "pipe_search_setup" in r["Kernel_Name"] else "setup" if "pipe_setup" in r["Kernel_Name"] else "search" if "seed_search" in r["Kernel_Name"] else "fin" if "seed_finalize" in r["Kernel_Name"] else None),
       int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
ev = [e for e in ev if e[0]]
m = [(s, e) for k, s, e in ev if k == "merged"]
print(f"{len(m)} merged launches, {sum(1 for e in ev if e[0]=='setup')} setup-only, {sum(1 for e in ev if e[0]=='search')} search-only, {sum(1 for e in ev if e[0]=='fin')} finalize")
per = F - 2  # merged launches per pass (updates 2..F-1)
last = m[-per:]
d = [(e - s) / 1e3 for s, e in last]
gaps = [(last[i + 1][0] - last[i][1]) / 1e3 for i in range(len(last) - 1)]
def avg(x): return sum(x) / max(len(x), 1)
print(f"last pass: merged kernel avg {avg(d):.2f} us, gap between launches avg {avg(gaps):.2f} us")
for a, b in ((0, 19), (19, 59), (59, per)):
    print(f"  updates {a + 2}..{b + 1}: avg {avg(d[a:b]):.1f} us (min {min(d[a:b]):.1f}, max {max(d[a:b]):.1f})")
