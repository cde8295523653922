#!/usr/bin/env python3
"""Kernel timeline of a batch of B sequences under rocprofv3 --kernel-trace: how long the setup and search kernels of the stream groups run, how much of
the step they overlap, how much of the wall time no kernel is running.  usage: python tools/batch_trace.py <dir with *kernel_trace.csv> [steps to skip]"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "setup" if "seed_setup_compact" in n else "search" if "seed_search_compact" in n else None
        if k:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, int(r.get("Queue_Id", 0) or 0)))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
# one pass of 199 steps at the END of the run (the timed passes): take the last 199 * groups * 2 launches
queues = sorted(set(r[3] for r in rows))
per_step = 2 * len(queues)
rows = rows[-199 * per_step:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = {"setup": 0, "search": 0}
for a, b, k, q in rows:
    busy[k] += b - a
# union of all intervals, and time with >= 2 kernels in flight
ev = sorted([(a, 1) for a, b, k, q in rows] + [(b, -1) for a, b, k, q in rows])
depth, last, any_t, multi_t = 0, t0, 0, 0
for t, d in ev:
    if depth >= 1: any_t += t - last
    if depth >= 2: multi_t += t - last
    depth += d; last = t
span = t1 - t0
print(f"{len(rows)} launches on {len(queues)} queues over {span / 1e3:.0f} us ({span / 199 / 1e3:.1f} us per step)")
print(f"kernel time summed: setup {busy['setup'] / 199 / 1e3:.1f} us, search {busy['search'] / 199 / 1e3:.1f} us per step")
print(f"some kernel running {any_t / span * 100:.1f} % of the time, two or more {multi_t / span * 100:.1f} %")
for sel in ((0, 20), (20, 60), (60, 199)):
    rs = rows[sel[0] * per_step: sel[1] * per_step]
    sp = max(r[1] for r in rs) - rs[0][0]
    print(f"steps {sel[0] + 1}..{sel[1]}: {sp / (sel[1] - sel[0]) / 1e3:.1f} us per step; setup avg {sum(b - a for a, b, k, q in rs if k == 'setup') / max(1, sum(1 for r in rs if r[2] == 'setup')) / 1e3:.1f} us, "
          f"search avg {sum(b - a for a, b, k, q in rs if k == 'search') / max(1, sum(1 for r in rs if r[2] == 'search')) / 1e3:.1f} us per launch")
