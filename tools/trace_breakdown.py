#!/usr/bin/env python3
"""Per-kernel breakdown of the TIMED pass of bench.py from a rocprofv3 kernel trace (skips warm-up and diagnostics passes).
usage: trace_breakdown.py <prof_dir> [warmup=10] [steps=199]   (counts in update() calls)"""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]; warm = int(sys.argv[2]) if len(sys.argv) > 2 else 10; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 199
f = glob.glob(d + '/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
per = defaultdict(list)
for r in rows:
    for k in ('seed_search', 'seed_setup', 'seed_finalize', 'seed_plan'):
        if k in r['Kernel_Name']:
            per[k].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
stages = [k for k in ('seed_setup', 'seed_plan', 'seed_search') if len(per[k]) >= len(per['seed_setup']) > 0]
out = []
n_setup = len(per['seed_setup'])
for k in ('seed_setup', 'seed_plan', 'seed_search', 'seed_finalize'):
    t = per[k][warm:warm + steps] if len(per[k]) >= n_setup else per[k]
    if not t:
        continue
    dd = [(e - s) / 1e3 for s, e in t]
    note = "" if len(per[k]) >= n_setup else "  (stand-alone launches only; normally fused into the next seed_setup)"
    out.append(f"{k:14s} avg {sum(dd)/len(dd):7.2f}  min {min(dd):7.2f}  max {max(dd):7.2f} us  (n={len(dd)}){note}")
su, se = per['seed_setup'][warm:warm + steps], per['seed_search'][warm:warm + steps]
steps = min(steps, len(su), len(se))
span = [(se[i][1] - su[i][0]) / 1e3 for i in range(steps)]
gap = [(su[i + 1][0] - se[i][1]) / 1e3 for i in range(steps - 1)]
inner = [(se[i][0] - su[i][1]) / 1e3 for i in range(steps)]
out.append(f"frame span (setup start -> search end) avg {sum(span)/steps:.2f} us; setup end -> search start avg {sum(inner)/steps:.2f} us; between frames avg "
           f"{sum(gap)/max(steps-1,1):.2f} us; whole timed region {(se[steps-1][1]-su[0][0])/1e3/steps:.2f} us/update")
first = min(20, steps)
out.append(f"updates 1..{first} of each pass (every seed live): "
           + ", ".join(f"pass {p + 1}: {sum(span[p*199:p*199+first])/first:.1f} us" for p in range(max(steps // 199, 1)) if p * 199 + first <= steps))
for i in (0, 4, 9, 19, 39, 59, 99, 149, min(198, steps - 1)):
    if i < steps:
        out.append(f"  update {i+1:3d}: " + "  ".join(f"{k[5:]} {(per[k][warm+i][1]-per[k][warm+i][0])/1e3:6.1f}" for k in stages))
print("\n".join(out))
