#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/profile_round.sh: HBM-side bytes and VALU wave-instructions per update()
(means over the update() calls of ONE complete pass of the 200-frame sequence), TV-L1 bytes per launch.
usage: make_traffic.py <raw_dir> <out.json>"""
import csv, glob, json, os, sys
from collections import defaultdict

raw, out = sys.argv[1], sys.argv[2]


KEYS = ("seed_setup", "seed_plan", "seed_search", "seed_finalize", "seed_init", "tv_iterate", "tv_prepare")


def per_kernel(sub, first=None):
    """{kernel short name: {counter: sum over dispatches}}, {kernel: number of dispatches}; first = n: only the first n dispatches of
    every kernel (in dispatch order)"""
    rows = []
    for f in glob.glob(os.path.join(raw, sub, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    agg = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(dict)
    for r in rows:
        n = r["Kernel_Name"]
        for key in KEYS:
            if key in n:
                ids = disp[key]
                if r["Dispatch_Id"] not in ids:
                    if first is not None and len(ids) >= first:
                        break
                    ids[r["Dispatch_Id"]] = True
                agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
                break
    return agg, {k: len(v) for k, v in disp.items()}


def kernel_source_sha256():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_hash", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.kernel_source_sha256()


W, H, UPDATES = 640, 480, 199
res = {"source": "rocprofv3 --pmc passes of `python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-extras` (tools/profile_round.sh): one complete pass "
                 "of configs[1] (setReferenceImage + 199 update() calls, 8-bit frames from host memory) followed by the TV-L1 denoise; sums over the pass "
                 "divided by 199",
       "kernel_source_sha256": kernel_source_sha256(),
       "unit_note": "FETCH_SIZE / WRITE_SIZE are KiB at the L2<->fabric interface (Infinity-Cache hits included). MI355X_MICROARCH.md: on gfx950 "
                    "FETCH_SIZE tallies 64 B per 128 B request for wide coalesced reads; these kernels mostly load dwords, for which the counter is "
                    "uncalibrated, so read bytes are given raw and doubled (the doubled figure is what bench.py reports as roofline.traffic)",
       "algorithmic_bytes_per_update": 52 * W * H}
insts, n_i = per_kernel("pmc_insts")
fetch, n_f = per_kernel("pmc_fetch")
write, n_w = per_kernel("pmc_write")
seed_k = ("seed_setup", "seed_plan", "seed_search", "seed_finalize")
if insts:
    res["valu_wave_instructions_per_update"] = {k: round(insts[k]["SQ_INSTS_VALU"] / UPDATES) for k in seed_k if k in insts}
    res["lds_wave_instructions_per_update"] = {k: round(insts[k]["SQ_INSTS_LDS"] / UPDATES) for k in seed_k if k in insts}
    res["valu_active_quad_cycles_per_update"] = {k: round(insts[k]["SQ_ACTIVE_INST_VALU"] / UPDATES) for k in seed_k if k in insts}
    res["wave_quad_cycles_per_update"] = {k: round(insts[k]["SQ_WAVE_CYCLES"] / UPDATES) for k in seed_k if k in insts}
    res["lds_bank_conflict_cycles_per_update"] = {k: round(insts[k]["SQ_LDS_BANK_CONFLICT"] / UPDATES) for k in seed_k if k in insts}
    res["waves_per_update"] = {k: round(insts[k]["SQ_WAVES"] / UPDATES, 1) for k in seed_k if k in insts}
    res["dispatches_in_the_pass"] = n_i
    i20, _ = per_kernel("pmc_insts", first=20)  # updates 1..20 of the pass: every seed live (bench.py's heavy_prefix)
    res["valu_wave_instructions_first20_per_update"] = {k: round(i20[k]["SQ_INSTS_VALU"] / 20) for k in ("seed_setup", "seed_search") if k in i20}
    res["lds_wave_instructions_first20_per_update"] = {k: round(i20[k]["SQ_INSTS_LDS"] / 20) for k in ("seed_setup", "seed_search") if k in i20}
    res["lds_bank_conflict_cycles_first20_per_update"] = {k: round(i20[k]["SQ_LDS_BANK_CONFLICT"] / 20) for k in ("seed_setup", "seed_search") if k in i20}
if fetch and write:
    fk = {k: fetch[k]["FETCH_SIZE"] / UPDATES for k in seed_k if k in fetch}
    wk = {k: write[k]["WRITE_SIZE"] / UPDATES for k in seed_k if k in write}
    res["fetch_KiB_per_update"] = {k: round(v, 1) for k, v in fk.items()}
    res["write_KiB_per_update"] = {k: round(v, 1) for k, v in wk.items()}
    fb, wb = sum(fk.values()) * 1024, sum(wk.values()) * 1024
    res["fetch_bytes_per_update_raw"] = round(fb)
    res["fetch_bytes_per_update_x2"] = round(2 * fb)
    res["write_bytes_per_update"] = round(wb)
    res["seed_update_bytes_per_launch"] = round(2 * fb + wb)
    res["seed_update_bytes_per_launch_raw"] = round(fb + wb)
    tv = {}
    if "tv_iterate" in fetch and "tv_iterate" in write:
        n = max(n_f.get("tv_iterate", 1), 1)
        tv["640x480"] = round((2 * fetch["tv_iterate"]["FETCH_SIZE"] + write["tv_iterate"]["WRITE_SIZE"]) * 1024 / n)
        res["tv_launches_640x480"] = n
    f2, n2f = per_kernel("tv1080_fetch")
    w2, n2w = per_kernel("tv1080_write")
    if "tv_iterate" in f2 and "tv_iterate" in w2:
        n = max(n2f.get("tv_iterate", 1), 1)
        tv["1920x1080"] = round((2 * f2["tv_iterate"]["FETCH_SIZE"] + w2["tv_iterate"]["WRITE_SIZE"]) * 1024 / n)
        res["tv_launches_1920x1080"] = n
        res["tv_fetch_KiB_per_launch_1920x1080_raw"] = round(f2["tv_iterate"]["FETCH_SIZE"] / n, 1)
        res["tv_write_KiB_per_launch_1920x1080"] = round(w2["tv_iterate"]["WRITE_SIZE"] / n, 1)
    res["tv_bytes_per_launch"] = tv
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
