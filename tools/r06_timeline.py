#!/usr/bin/env python3
"""Where the time between frames goes when frames come from host memory: from `rocprofv3 --kernel-trace --memory-copy-trace` CSVs of a run of
apps/bench_main (one sequence) or tools/batch_bench.py (a batch), over the LAST `n` updates of the run:
  * the period between consecutive setup-kernel starts per stream (= the rate), kernel durations, idle time of the compute queue
    (search end -> next setup start; setup end -> search start),
  * every host-to-device copy: duration, bytes / s when the size is known, and how long BEFORE the setup kernel of "its" frame it completed
    (copy i of the tail is matched with setup i of the tail: one frame copy per update; the small flag copies are listed apart).
usage: python tools/r06_timeline.py <dir> [n=150] [frame_bytes]"""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
frame_bytes = int(sys.argv[3]) if len(sys.argv) > 3 else 0


def rows_of(pattern):
    out = []
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


k_rows = rows_of("*kernel_trace.csv")
kern = defaultdict(list)  # queue -> [(start, end, kind)]
other = defaultdict(lambda: [0, 0])
for r in k_rows:
    nm = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    kind = "setup" if "seed_setup_compact" in nm else "search" if "seed_search_compact" in nm else None
    if kind:
        kern[r.get("Queue_Id", "0")].append((s, e, kind))
    else:
        o = other[nm[:60]]
        o[0] += 1; o[1] += e - s
queues = sorted(kern, key=lambda q: -len(kern[q]))
print(f"{len(k_rows)} kernel launches; update kernels on {len(queues)} queue(s)")
t_lo = None
for q in queues:
    ks = sorted(kern[q])
    setups = [k for k in ks if k[2] == "setup"][-n:]
    searches = [k for k in ks if k[2] == "search"][-n:]
    if len(setups) < 3:
        continue
    m = min(len(setups), len(searches))
    setups, searches = setups[-m:], searches[-m:]
    if t_lo is None or setups[0][0] < t_lo:
        t_lo = setups[0][0]
    period = [(setups[i + 1][0] - setups[i][0]) / 1e3 for i in range(m - 1)]
    d_setup = [(e - s) / 1e3 for s, e, _ in setups]
    d_search = [(e - s) / 1e3 for s, e, _ in searches]
    inner = [(searches[i][0] - setups[i][1]) / 1e3 for i in range(m)]
    between = [(setups[i + 1][0] - searches[i][1]) / 1e3 for i in range(m - 1)]
    avg = lambda v: sum(v) / max(len(v), 1)
    srt = sorted(period)
    print(f"queue {q}: last {m} updates: period avg {avg(period):.1f} us (median {srt[len(srt) // 2]:.1f}, p90 {srt[int(len(srt) * 0.9)]:.1f}, max {srt[-1]:.1f}); "
          f"setup avg {avg(d_setup):.1f} (max {max(d_setup):.1f}), search avg {avg(d_search):.1f} (max {max(d_search):.1f}); "
          f"setup end -> search start {avg(inner):.2f}, search end -> next setup start avg {avg(between):.2f} (max {max(between):.1f})")

c_rows = rows_of("*memory_copy_trace.csv")
copies = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "")) for r in c_rows)
h2d = [c for c in copies if "HOST_TO_DEVICE" in c[2].upper() or "H2D" in c[2].upper()]
print(f"{len(copies)} traced copies, {len(h2d)} host-to-device")
if h2d and t_lo is not None:
    tail = [c for c in h2d if c[0] >= t_lo - 2_000_000]
    dur = sorted((e - s) / 1e3 for s, e, _ in tail)
    # two populations: the frame copies (long) and the arrival flags (short)
    big = [c for c in tail if (c[1] - c[0]) / 1e3 > 0.5 * dur[-1]] if dur else []
    small = [c for c in tail if c not in big]
    for name, grp in (("frame-sized copies", big), ("other copies (arrival flags)", small)):
        if not grp:
            continue
        dd = sorted((e - s) / 1e3 for s, e, _ in grp)
        line = f"{name}: {len(grp)}; duration avg {sum(dd) / len(dd):.1f} us, median {dd[len(dd) // 2]:.1f}, p90 {dd[int(len(dd) * 0.9)]:.1f}, max {dd[-1]:.1f}"
        if frame_bytes and name.startswith("frame"):
            line += f"  = {frame_bytes / (sum(dd) / len(dd)) / 1e3:.1f} GB/s"
        busy = sum(e - s for s, e, _ in grp)
        span = max(e for _, e, _ in grp) - min(s for s, _, _ in grp)
        # do consecutive copies overlap in time (two copy streams -> two engines)?
        ev = sorted([(s, 1) for s, e, _ in grp] + [(e, -1) for s, e, _ in grp])
        depth, last, two = 0, ev[0][0], 0
        for t, dd_ in ev:
            if depth >= 2: two += t - last
            depth += dd_; last = t
        line += f"; engine busy {busy / max(span, 1) * 100:.0f} % of their span, two in flight {two / max(span, 1) * 100:.0f} %"
        print(line)
    # lead of each frame copy over the setup kernel that consumes it (matched from the end of the run)
    q0 = queues[0]
    setups = [k for k in sorted(kern[q0]) if k[2] == "setup"]
    m = min(len(big), len(setups), n)
    if m > 3:
        lead = sorted((setups[-m + i][0] - big[-m + i][1]) / 1e3 for i in range(m))
        print(f"copy end -> start of the setup kernel of the same index (last {m}): median {lead[m // 2]:.1f} us, p10 {lead[m // 10]:.1f}, min {lead[0]:.1f} "
              f"(negative: the kernel started before its frame had arrived)")
if other:
    print("other kernels:", "; ".join(f"{k} x{v[0]} {v[1] / 1e3 / max(v[0], 1):.1f} us" for k, v in sorted(other.items(), key=lambda kv: -kv[1][1])[:6]))
