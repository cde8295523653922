#!/usr/bin/env python3
"""Benchmark of the depth-filter hot path (BASELINE.json metric: Mpix/s of depth-filter updates).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step = one rmd::SeedMatrix::update() (fused seed_check + epipolar_match + seed_update) on one frame.
At N = 1 the workload is BASELINE.json configs[1]: the 640x480 synthetic "over-table" sequence, 1 reference +
199 updates, NCC half-patch 4 (RMD_CORR_PATCH_SIDE = 9), followed by one TV-L1 denoise (lambda 0.5, 200 iterations)
reported separately.  All frames are resident in HBM before the timed region; the timed region is bracketed by a
barrier + device synchronisation on both sides and the MAX over ranks is taken.  For N > 1 every rank runs its own
independent sequence on its own GPU (weak scaling, no data-path collective) and rank 0 prints the aggregate.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel, algorithmic bytes / HIP-event time)
and "cpu_baseline" (the reference's own kernels on the host cores over a bounded sample of the same frames).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
FUSED_BYTES_PER_PIXEL = 52  # fused check+match+update, compulsory traffic (R 32 + W 20)
TV_BYTES_PER_PIXEL_ITER = 40
WIDTH, HEIGHT, FRAMES, SIDE = 640, 480, 200, 9
TV_LAMBDA, TV_ITERS = 0.5, 200


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=FRAMES - 1)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample (0 disables)")
    ap.add_argument("--matcher", type=int, default=1, help="0 per-pixel kernel, 1 tile kernel")
    ap.add_argument("--window", type=int, default=0, help="search LDS window: 0 / 2 large (default), 1 small")
    ap.add_argument("--size", default=f"{WIDTH}x{HEIGHT}", help="frame size WxH; anything but the default 640x480 is one of "
                    "BASELINE's other configs (1280x960, 1920x1080) and is labelled as such, not the headline metric")
    ap.add_argument("--tv-iters", type=int, default=TV_ITERS)
    return ap.parse_args()


VALU_PEAK_GINST_S = 256 * 4 * 2.4 / 4.0  # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles at 2.4 GHz = 614.4 G/s


def valu_roofline(avg_launch_s, traffic_path):
    """The roof that actually binds the seed update: VALU issue.  Instruction counts per update() come from the
    committed PMC passes (profiles/traffic.json, SQ_INSTS_VALU over the whole job), the launch time from this run."""
    try:
        t = json.load(open(traffic_path))
        n = float(sum(t["valu_wave_instructions_per_update"].values()))
        achieved = n / avg_launch_s / 1e9
        return {"bound": "valu", "kernel": "seed_update", "achieved": round(achieved, 1), "peak": round(VALU_PEAK_GINST_S, 1),
                "unit": "G wave-instructions/s", "frac": round(achieved / VALU_PEAK_GINST_S, 4),
                "wave_instructions_per_launch": int(n), "source": "profiles/traffic.json (rocprofv3 --pmc SQ_INSTS_VALU)"}
    except Exception:
        return None


def cpu_baseline(seq, budget_s, gpu_sample_fn):
    """The reference's own kernels (oracle/_ref, built from /root/reference for the host) over the first frames
    of the same sequence, all host cores, until `budget_s` is used up."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # idle OpenMP workers must not spin while the GPU sample is timed
    import oracles as O
    kind = "reference" if O.available("ref", SIDE) else "port"
    olib = O.OracleLib("ref" if kind == "reference" else "port", SIDE)
    cores = olib.lib.ref_max_threads() if kind == "reference" else olib.lib.orc_max_threads()
    s = O.Seeds(olib, seq.width, seq.height, seq.K)
    s.set_reference(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    t0 = time.perf_counter()
    n = 0
    while n < seq.n_frames - 1 and (time.perf_counter() - t0) < budget_s:
        n += 1
        s.update(seq.images[n], seq.T_curr_world[n])
    dt = time.perf_counter() - t0
    mpix = seq.width * seq.height * n / dt / 1e6
    out = {"value": round(mpix, 4), "unit": "Mpix/s", "cores": int(cores), "kind": kind,
           "sample": f"updates 1..{n} of the same {seq.width}x{seq.height} sequence (patch side {SIDE}), {dt:.1f} s"}
    time.sleep(0.3)
    gpu_same = gpu_sample_fn(n)
    if gpu_same:
        out["gpu_same_sample"] = round(gpu_same, 2)
    return out


def main():
    global WIDTH, HEIGHT, TV_ITERS
    args = parse()
    WIDTH, HEIGHT = (int(v) for v in args.size.lower().split("x"))
    TV_ITERS = args.tv_iters
    headline = (WIDTH, HEIGHT) == (640, 480)
    rank = int(os.environ.get("RANK", "0"))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world_env == 1:
        # plain `python bench.py --gpus N`: re-launch one rank per GPU
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__),
               "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--cpu-seconds", str(args.cpu_seconds), "--matcher", str(args.matcher), "--window", str(args.window),
               "--size", args.size, "--tv-iters", str(args.tv_iters)]
        sys.exit(subprocess.call(cmd))

    import torch
    from rpg_open_remode_amd import api, batch, synth

    rank, local_rank, world = batch.init("nccl")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback of the product path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if not api.checkCudaDevice(local_rank):
        raise SystemExit("no usable HIP device")

    # one independent sequence per rank (scene / trajectory seed = rank), rendered on the host, then made resident
    seq = synth.Sequence(WIDTH, HEIGHT, FRAMES, seed=rank)
    # device-resident frames, allocated through the library's own rmd::DeviceImage (not torch: PyTorch-ROCm ships a
    # private HIP runtime; torch is used here for torch.distributed only)
    frames = []
    for im in seq.images:
        d = api.DeviceImage(WIDTH, HEIGHT, np.float32)
        d.setDevData(im)
        frames.append(d)

    def new_seeds():
        s = api.SeedMatrix(WIDTH, HEIGHT, api.PinholeCamera(*seq.K), patch_side=SIDE)
        s.setOption(api.OPT_MATCHER, args.matcher)
        s.setOption(api.OPT_WINDOW, args.window)
        return s

    def set_ref(s):
        s.setReferenceImageDevice(frames[0].data, frames[0].stride, seq.T_curr_world[0], seq.min_depth, seq.max_depth)

    def run_updates(s, first, count):
        """`count` consecutive steps starting at step index `first` of the endless loop ref, 1..199, ref, 1..199, ..."""
        per_pass = FRAMES - 1
        for i in range(first, first + count):
            k = i % per_pass
            if k == 0 and i != 0:
                set_ref(s)
            s.updateDevice(frames[k + 1].data, frames[k + 1].stride, seq.T_curr_world[k + 1])

    # warm-up: W untimed steps on a scratch instance (clocks, code objects, allocator)
    scratch = new_seeds()
    set_ref(scratch)
    run_updates(scratch, 0, args.warmup)
    scratch.sync()

    seeds = new_seeds()
    seeds.setOption(api.OPT_TIMING, 2)  # one HIP event pair around the timed region, on the stream the kernels run on
    set_ref(seeds)
    seeds.sync()
    seeds.timingReset()

    batch.barrier(device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_updates(seeds, 0, args.steps)
    seeds.sync()
    torch.cuda.synchronize()
    batch.barrier(device)
    elapsed = time.perf_counter() - t0

    kernel_ms, kernel_launches = seeds.timing(api.STAGE_UPDATE)
    converged = seeds.getConvergedCount()
    units = float(WIDTH * HEIGHT * args.steps)
    max_elapsed, total_units, per_rank = batch.gather_throughput(elapsed, units, device)

    # ---- the rest is reporting on rank 0; other ranks idle at the final barrier
    result = None
    if rank == 0:
        # denoiser of configs[1] (reported beside the metric, not inside it)
        den = api.DepthmapDenoiser(WIDTH, HEIGHT)
        den.setLargeSigmaSq(seq.max_depth - seq.min_depth)
        den.setOption(api.DENOISE_OPT_TIMING, 1)
        den.denoise(seeds.getMu(), seeds.getSigmaSq(), seeds.getA(), seeds.getB(), TV_LAMBDA, 10, download=True)  # warm
        td = time.perf_counter()
        den.denoise(seeds.getMu(), seeds.getSigmaSq(), seeds.getA(), seeds.getB(), TV_LAMBDA, TV_ITERS, download=True)
        denoise_wall_ms = (time.perf_counter() - td) * 1e3
        tv_ms, tv_launches = den.timing()

        avg_kernel_s = kernel_ms / max(kernel_launches, 1) / 1e3
        achieved = FUSED_BYTES_PER_PIXEL * WIDTH * HEIGHT / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if headline and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("seed_update_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "seed_update (fused seed_check+epipolar_match+seed_update)",
                    "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "avg_launch_us": round(avg_kernel_s * 1e6, 2), "launches": kernel_launches,
                    "algorithmic_bytes_per_launch": FUSED_BYTES_PER_PIXEL * WIDTH * HEIGHT,
                    "note": "epipolar_match is VALU/LDS-bound (hundreds of flop per compulsory byte); see DESIGN.md"}
        tv_avg_s = tv_ms / max(tv_launches, 1) / 1e3
        tv_iters_per_launch = TV_ITERS / max(tv_launches, 1)
        tv_achieved = TV_BYTES_PER_PIXEL_ITER * WIDTH * HEIGHT * tv_iters_per_launch / tv_avg_s / 1e9 if tv_avg_s > 0 else 0.0
        roofline_tv = {"bound": "hbm", "kernel": "tv_iterate", "achieved": round(tv_achieved, 2), "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": round(tv_achieved / HBM_PEAK_GBS, 5), "traffic": None,
                       "avg_launch_us": round(tv_avg_s * 1e6, 2), "launches": tv_launches,
                       "denoise_wall_ms": round(denoise_wall_ms, 3), "iterations": TV_ITERS}

        def gpu_sample(n):
            s2 = new_seeds()
            set_ref(s2)
            s2.sync()
            ts = time.perf_counter()
            run_updates(s2, 0, n)
            s2.sync()
            return WIDTH * HEIGHT * n / (time.perf_counter() - ts) / 1e6

        # search statistics of the timed workload (separate pass, diagnostics counters on)
        s3 = new_seeds()
        s3.setOption(api.OPT_COLLECT_STATS, 1)
        set_ref(s3)
        tot = {"live_seeds": 0, "steps": 0, "ncc_evals": 0}
        for i in range(min(args.steps, FRAMES - 1)):
            s3.updateDevice(frames[i + 1].data, frames[i + 1].stride, seq.T_curr_world[i + 1])
            st = s3.lastStats()
            for key in tot:
                tot[key] += st[key]
        n_st = max(1, min(args.steps, FRAMES - 1))
        search_stats = {k: round(v / n_st, 1) for k, v in tot.items()}

        # PCIe-inclusive rates (frames start in pageable host memory): float frames through update(), 8-bit frames through
        # update_u8() (pinned double buffering + conversion on the device).  Reported beside `value`, never as `value`.
        def host_path(use_u8):
            s4 = new_seeds()
            n = min(args.steps, FRAMES - 1)
            if use_u8:
                s4.setReferenceImageU8(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
            else:
                s4.setReferenceImage(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
            s4.sync()
            ts = time.perf_counter()
            for i in range(n):
                if use_u8:
                    s4.updateU8(seq.gray[i + 1], seq.T_curr_world[i + 1])
                else:
                    s4.update(seq.images[i + 1], seq.T_curr_world[i + 1])
            s4.sync()
            return WIDTH * HEIGHT * n / (time.perf_counter() - ts) / 1e6
        pcie = {"float_frames_update_mpix_s": round(host_path(False), 1), "u8_frames_update_u8_mpix_s": round(host_path(True), 1)}

        cpu = None
        if args.cpu_seconds > 0 and world == 1:  # the CPU baseline is an N = 1 figure; at N > 1 the other ranks would spin beside it
            try:
                cpu = cpu_baseline(seq, args.cpu_seconds, gpu_sample)
            except Exception as e:  # the bench line must survive a missing oracle
                cpu = {"value": None, "unit": "Mpix/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}

        value = total_units / max_elapsed / 1e6
        result = {
            "metric": f"Mpix/s depth-filter updates ({WIDTH}x{HEIGHT}, 200 frames)", "value": round(value, 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(max_elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'configs[1]' if headline else 'non-headline size'}: {WIDTH}x{HEIGHT} synthetic over-table sequence, 1 reference + {FRAMES - 1} "
                                   f"updates per pass, NCC patch side {SIDE} (half-patch 4), max epipolar extent 100 px; "
                                   f"one independent sequence per GPU",
                       "frames_resident_in_hbm": True, "matcher": "tile" if args.matcher else "pixel",
                       "converged_seeds_at_end": converged, "mean_per_update": search_stats},
            "roofline": roofline, "roofline_valu": valu_roofline(avg_kernel_s, tpath) if headline and avg_kernel_s > 0 else None,
            "roofline_denoiser": roofline_tv, "cpu_baseline": cpu, "pcie_inclusive": pcie,
            "per_rank": [{"elapsed_s": round(e, 6), "mpix": u / 1e6} for e, u in per_rank],
        }
    batch.barrier(device)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
