#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel trace / stats / counter collection) into small text summaries.
usage: summarize_rocprof.py <raw_dir> <summary_dir> <tag>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(raw, sub, pattern):
    return sorted(glob.glob(os.path.join(raw, sub, "**", pattern), recursive=True))


def short(name):
    """kernel name without return type, namespace and argument list; template arguments kept"""
    name = name.replace("void ", "").replace("rmdk::", "")
    depth, out = 0, []
    for ch in name:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out)[:60]


def main():
    raw, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(out, exist_ok=True)
    lines = []
    # --- kernel stats
    for f in find(raw, "trace", "*kernel_stats.csv"):
        lines.append(f"# rocprofv3 --kernel-trace --stats ({os.path.basename(f)})")
        with open(f) as fh:
            rows = list(csv.DictReader(fh))
        lines.append(f"{'kernel':60s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
        for r in rows:
            name = short(r.get("Name", r.get("KernelName", "?")))
            calls = r.get("Calls", "0")
            tot = float(r.get("TotalDurationNs", 0)) / 1e6
            avg = float(r.get("AverageNs", 0)) / 1e3
            mn = float(r.get("MinNs", 0)) / 1e3
            mx = float(r.get("MaxNs", 0)) / 1e3
            lines.append(f"{name:60s} {calls:>7s} {tot:10.3f} {avg:10.2f} {mn:10.2f} {mx:10.2f} {r.get('Percentage', ''):>6s}")
    # --- per-dispatch trace: duration by kernel + resources
    for f in find(raw, "trace", "*kernel_trace.csv"):
        with open(f) as fh:
            rows = list(csv.DictReader(fh))
        agg = defaultdict(list)
        res = {}
        for r in rows:
            n = short(r.get("Kernel_Name", "?"))
            agg[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            res[n] = (r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")), r.get("Accum_VGPR_Count", ""), r.get("SGPR_Count", ""),
                      r.get("LDS_Block_Size", ""), r.get("Scratch_Size", ""), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")),
                      r.get("Grid_Size", r.get("Grid_Size_X", "")))
        lines.append("")
        lines.append(f"# per-kernel dispatch durations from the kernel trace ({os.path.basename(f)})")
        lines.append(f"{'kernel':60s} {'n':>6s} {'avg_us':>10s} {'p50_us':>10s} {'max_us':>10s}  vgpr/agpr/sgpr/lds/scratch/wg/grid")
        for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            v2 = sorted(v)
            lines.append(f"{n:60s} {len(v):6d} {sum(v) / len(v):10.2f} {v2[len(v2) // 2]:10.2f} {v2[-1]:10.2f}  {'/'.join(str(x) for x in res[n])}")
    # --- counters, one section PER PASS (sub-directory of the raw output): passes of different jobs (e.g. the 640x480 sequence and the
    # 1920x1080 denoiser run) must never be merged into one table
    counters = {}
    for sub in sorted(os.listdir(raw)):
        per_pass = {}
        for f in find(raw, sub, "*counter_collection.csv"):
            with open(f) as fh:
                rows = list(csv.DictReader(fh))
            agg = defaultdict(lambda: defaultdict(float))
            cnt = defaultdict(set)
            for r in rows:
                n = short(r.get("Kernel_Name", "?"))
                agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[n].add(r.get("Dispatch_Id", ""))
            for n in agg:
                per_pass.setdefault(n, {})
                for c, v in agg[n].items():
                    per_pass[n][c] = v / max(1, len(cnt[n]))
                per_pass[n]["_dispatches"] = len(cnt[n])
        if per_pass:
            counters[sub] = per_pass
    if counters:
        cmdfile = os.path.join(raw, "commands.txt")
        cmds = dict(l.rstrip("\n").split("\t", 1) for l in open(cmdfile)) if os.path.exists(cmdfile) else {}
        for sub, per_pass in counters.items():
            lines.append("")
            lines.append(f"# PMC counters of pass '{sub}', mean per dispatch" + (f"  ({cmds[sub]})" if sub in cmds else ""))
            for n, cs in per_pass.items():
                lines.append(f"[{n}]")
                for c, v in sorted(cs.items()):
                    lines.append(f"    {c:28s} {v:18.1f}")
                fs, ws = cs.get("FETCH_SIZE"), cs.get("WRITE_SIZE")
                if fs is not None:
                    # rocprofv3 reports KiB; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts 64 B per 128 B request for wide
                    # coalesced reads -> doubled figure given as the upper estimate
                    lines.append(f"    -> HBM read bytes/dispatch: {fs * 1024:.0f} (raw), {2 * fs * 1024:.0f} (x2 gfx950 correction)")
                if ws is not None:
                    lines.append(f"    -> HBM write bytes/dispatch: {ws * 1024:.0f}")
        json.dump(counters, open(os.path.join(out, f"{tag}_counters.json"), "w"), indent=1)
    open(os.path.join(out, f"{tag}_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
