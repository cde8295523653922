#!/usr/bin/env python3
"""Benchmark of the depth-filter hot path (BASELINE.json metric: Mpix/s of depth-filter updates).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size WxH] [--frames F] [--tv-iters I]

A STEP = ONE FULL PASS of the workload: rmd::SeedMatrix::setReferenceImage on frame 0 followed by F-1
rmd::SeedMatrix::update() calls (fused seed_check + epipolar_match + triangulation + seed_update) on frames 1..F-1.
The cost of an update is strongly data-dependent (it falls by an order of magnitude once seeds converge), so the unit of
timing is the whole sequence: whatever --steps / --warmup are passed, the timed region always covers complete passes.
At N = 1 the default workload is BASELINE.json configs[1]: the 640x480 synthetic "over-table" sequence, 200 frames
(1 reference + 199 updates), NCC half-patch 4 (RMD_CORR_PATCH_SIDE = 9), followed by one TV-L1 denoise (lambda 0.5,
200 iterations) that is reported separately.  `--size 1280x960 --frames 500` is configs[2], `--size 1920x1080 --frames 1000
--tv-iters 500` is configs[4]; they are labelled as such and are not the headline metric.

value = W * H * (F-1) * K / elapsed: pixels of update() calls over the wall time of the K passes (setReferenceImage is
inside the timed region, its pixels are not counted).  All frames are resident in HBM before the timed region starts; the
timed region is bracketed by barrier + device synchronisation on both sides and the MAX over ranks is taken.  For N > 1
every rank runs its own independent sequence on its own GPU (weak scaling, no data-path collective) and rank 0 prints the
aggregate.  The rate with the per-frame host-to-device copy included (test/dataset_main.cpp:101-103) is
reported in the same line as `h2d_inclusive`.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel, algorithmic bytes / HIP-event time on the
stream the kernels run on), "roofline_valu" / "roofline_flops" (the roofs that actually bind the NCC search),
"roofline_denoiser", "cpu_baseline" (the reference's own kernels on the host cores over a bounded sample of the same frames).
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
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector
# MI355X_MICROARCH.md: a wave64 VALU instruction issues in 2 cycles on a SIMD-32; 256 CUs x 4 SIMDs x 2.4 GHz / 2
VALU_PEAK_GINST_S = 256 * 4 * 2.4 / 2.0
FUSED_BYTES_PER_PIXEL = 52  # fused check+match+update, compulsory traffic (R 32 + W 20)
TV_BYTES_PER_PIXEL_ITER = 40
NCC_FLOP_PER_TAP = 14       # 3 FMA + 1 bilinear fetch per tap
WIDTH, HEIGHT, FRAMES, SIDE = 640, 480, 200, 9
TV_LAMBDA, TV_ITERS = 0.5, 200
KNOWN_CONFIGS = {(640, 480, 200): "configs[1]", (1280, 960, 500): "configs[2]", (1920, 1080, 1000): "configs[4]"}
PARITY_NOTE = ("bit-identical (all state planes, convergence masks, TV-L1 output) to the reference's own kernels compiled for the "
               "CPU with IEEE fp32, no contraction and the shared expf/sinf/acosf of csrc/rmd_math.h (tests/: golden fixtures "
               "generated from /root/reference); that build differs from the reference linked against glibc's libm by RMSE "
               "3e-4 m (DESIGN.md 2)")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5, help="timed passes over the sequence (one step = setReference + F-1 updates)")
    ap.add_argument("--warmup", type=int, default=1, help="untimed passes before the timed region")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample (0 disables)")
    ap.add_argument("--matcher", type=int, default=-1, help="A/B: 0 per-pixel kernel, 1 round-1 tile pipeline, 2 one-launch frame kernel, "
                    "3 tile pipeline with the compact search kernel (the library's default, used when the flag is absent)")
    ap.add_argument("--unit-target", type=int, default=1, help="tile pipeline: work units aimed at per frame, in multiples of the resident search workgroups (experiments)")
    ap.add_argument("--size", default=f"{WIDTH}x{HEIGHT}", help="frame size WxH; 640x480 with 200 frames is the headline metric")
    ap.add_argument("--frames", type=int, default=0, help="frames per pass incl. the reference (default: 200; 500 at 1280x960, "
                    "1000 at 1920x1080, as BASELINE.json configures them)")
    ap.add_argument("--tv-iters", type=int, default=0, help="TV-L1 iterations of the separately reported denoise (default 200; 500 at 1920x1080)")
    ap.add_argument("--dist", action="store_true", help="create the torch.distributed (RCCL) group even for a single rank")
    ap.add_argument("--no-extras", action="store_true", help="skip the H2D / statistics / CPU passes (profiling runs)")
    ap.add_argument("--rendezvous-only", action="store_true", help="launch-path check, no measurement: join the process group (gloo when "
                    "there is no GPU), run the barriers and the throughput gather with zero work, print {\"rendezvous\": ...} and exit")
    return ap.parse_args()


def resolve_workload(args):
    w, h = (int(v) for v in args.size.lower().split("x"))
    frames = args.frames or {(1280, 960): 500, (1920, 1080): 1000}.get((w, h), FRAMES)
    tv_iters = args.tv_iters or (500 if (w, h) == (1920, 1080) else TV_ITERS)
    return w, h, frames, tv_iters


def load_counters(path):
    try:
        return json.load(open(path))
    except Exception:
        return None


def valu_roofline(avg_launch_s, traffic_path):
    """The roof that actually binds the seed update: VALU issue.  Wave-instruction counts per update() come from
    the committed PMC pass over the same complete passes (profiles/traffic.json, SQ_INSTS_VALU), the launch time from this run."""
    t = load_counters(traffic_path)
    if not t or avg_launch_s <= 0:
        return None
    try:
        n = float(sum(t["valu_wave_instructions_per_update"].values()))
    except Exception:
        return None
    achieved = n / avg_launch_s / 1e9
    return {"bound": "valu", "kernel": "seed_update", "achieved": round(achieved, 1), "peak": round(VALU_PEAK_GINST_S, 1),
            "unit": "G wave-instructions/s", "frac": round(achieved / VALU_PEAK_GINST_S, 4),
            "wave_instructions_per_launch": int(n), "source": "profiles/traffic.json (rocprofv3 --pmc SQ_INSTS_VALU, whole passes)"}


def flops_roofline(avg_launch_s, ncc_evals_per_update, side):
    if avg_launch_s <= 0 or not ncc_evals_per_update:
        return None
    flop = NCC_FLOP_PER_TAP * side * side * float(ncc_evals_per_update)
    achieved = flop / avg_launch_s / 1e12
    return {"bound": "fp32", "kernel": "seed_update", "achieved": round(achieved, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_PEAK_TFLOPS, 4), "flop_per_launch": int(flop),
            "definition": f"{NCC_FLOP_PER_TAP} flop x {side * side} taps x NCC evaluations (SURVEY.md 8d), evaluations counted by this run"}


def cpu_baseline(frame_fn, width, height, K, n_frames, side, min_depth, max_depth, budget_s, gpu_sample_fn):
    """The reference's own kernels (oracle/_ref, built from /root/reference for the host) over the first frames
    of the same sequence, all host cores, until `budget_s` is used up."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # idle OpenMP workers must not spin while the GPU sample is timed
    import oracles as O
    kind = "reference" if O.available("ref", side) else "port"
    olib = O.OracleLib("ref" if kind == "reference" else "port", side)
    cores = olib.lib.ref_max_threads() if kind == "reference" else olib.lib.orc_max_threads()
    s = O.Seeds(olib, width, height, K)
    img0, T0 = frame_fn(0)
    s.set_reference(img0, T0, min_depth, max_depth)
    t0 = time.perf_counter()
    n = 0
    while n < n_frames - 1 and (time.perf_counter() - t0) < budget_s:
        n += 1
        img, T = frame_fn(n)
        s.update(img, T)
    dt = time.perf_counter() - t0
    mpix = width * height * n / dt / 1e6
    out = {"value": round(mpix, 4), "unit": "Mpix/s", "cores": int(cores), "kind": kind,
           "sample": f"updates 1..{n} of the same {width}x{height} sequence (patch side {side}), {dt:.1f} s"}
    time.sleep(0.3)
    gpu_same = gpu_sample_fn(n)
    if gpu_same:
        out["gpu_same_sample"] = round(gpu_same, 2)
    return out


def main():
    args = parse()
    W, H, F, tv_iters = resolve_workload(args)
    cfg_name = KNOWN_CONFIGS.get((W, H, F))
    headline = cfg_name == "configs[1]"
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world_env == 1:
        # plain `python bench.py --gpus N`: re-launch one rank per GPU
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    from rpg_open_remode_amd import api, batch, synth

    rank, local_rank, world = batch.init("nccl", force=args.dist)
    if args.rendezvous_only:
        # the control plane of a --gpus N launch and nothing else (tests/test_bench_cpu.py runs it under torch.distributed.run
        # with two ranks on the CPU): same barrier / gather / rank-0-prints sequence as the measurement below
        batch.barrier()
        max_e, total_u, per_rank = batch.gather_throughput(0.001 * (rank + 1), float(rank + 1), None, extra=(float(local_rank), 0.0))
        batch.barrier()
        if rank == 0:
            print(json.dumps({"rendezvous": "ok", "n_gpus": world, "control_plane": batch.backend_name(), "max_elapsed_s": max_e,
                              "total_units": total_u, "per_rank": [list(r) for r in per_rank]}), flush=True)
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback of the product path)")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % n_dev  # more ranks than devices (a 1-GPU lease driven with --gpus 2) share devices
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if not api.checkCudaDevice(dev_index):
        raise SystemExit("no usable HIP device")

    # one independent sequence per rank (scene / trajectory seed = rank), rendered on the host frame by frame and made
    # resident through the library's own rmd::DeviceImage (not torch: torch is here for torch.distributed only)
    K = synth.intrinsics(W, H)
    keep_host = headline or F <= 200  # the 8-bit frames stay on the host for the H2D-inclusive pass
    frames, poses, gray = [], [], []
    range0 = None
    for k in range(F):
        T = synth.pose(k, rank)
        g, rng = synth.render(W, H, T, rank, want_range=(k == 0), K=K)
        if k == 0:
            range0 = rng
        d = api.DeviceImage(W, H, np.float32)
        d.setDevData(synth.to_float_image(g))
        frames.append(d)
        poses.append(np.ascontiguousarray(synth.invert_pose(T).astype(np.float32).reshape(12)))
        if keep_host:
            gray.append(g)
    min_depth, max_depth = float(range0.min()), float(range0.max())

    def new_seeds():
        s = api.SeedMatrix(W, H, api.PinholeCamera(*K), patch_side=SIDE)
        if args.matcher >= 0:
            s.setOption(api.OPT_MATCHER, args.matcher)
            s.setOption(api.OPT_UNIT_TARGET, args.unit_target)
        return s

    def set_ref(s):
        s.setReferenceImageDevice(frames[0].data, frames[0].stride, poses[0], min_depth, max_depth)

    def run_pass(s, n_updates=None):
        """one step: the reference frame, then updates 1..F-1 (or the first n_updates of them)"""
        set_ref(s)
        for k in range(1, (F if n_updates is None else n_updates + 1)):
            s.updateDevice(frames[k].data, frames[k].stride, poses[k])

    seeds = new_seeds()
    for _ in range(args.warmup):  # W untimed passes (clocks, code objects, allocator)
        run_pass(seeds)
    seeds.sync()
    seeds.setOption(api.OPT_TIMING, 2)  # one HIP event pair around the timed region, on the stream the kernels run on

    batch.barrier(device)
    torch.cuda.synchronize()
    seeds.timingReset()  # records the region's start event on the (idle) stream: nothing but the timed launches follows it
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_pass(seeds)
    seeds.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    batch.barrier(device)
    kernel_ms, kernel_updates = seeds.timing(api.STAGE_UPDATE)  # device time of the region / update() calls in it
    converged = seeds.getConvergedCount()
    n_updates = (F - 1) * args.steps
    units = float(W * H * n_updates)
    max_elapsed, total_units, per_rank = batch.gather_throughput(elapsed, units, device, extra=(float(n_updates), float(converged)))

    # ---- the rest is reporting on rank 0; other ranks idle at the final barrier
    result = None
    if rank == 0:
        # denoiser of the config (reported beside the metric, not inside it)
        den = api.DepthmapDenoiser(W, H)
        den.setLargeSigmaSq(max_depth - min_depth)
        den.setOption(api.DENOISE_OPT_TIMING, 1)
        den.denoise(seeds.getMu(), seeds.getSigmaSq(), seeds.getA(), seeds.getB(), TV_LAMBDA, 10, download=True)  # warm
        td = time.perf_counter()
        den.denoise(seeds.getMu(), seeds.getSigmaSq(), seeds.getA(), seeds.getB(), TV_LAMBDA, tv_iters, download=True)
        denoise_wall_ms = (time.perf_counter() - td) * 1e3
        tv_ms, tv_launches = den.timing()

        counters = load_counters(os.path.join(ROOT, "profiles", "traffic.json")) if headline else None
        avg_kernel_s = kernel_ms / max(kernel_updates, 1) / 1e3
        achieved = FUSED_BYTES_PER_PIXEL * W * H / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        roofline = {"bound": "hbm", "kernel": "seed_update (fused seed_check+epipolar_match+triangulation+seed_update)",
                    "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": counters.get("seed_update_bytes_per_launch") if counters else None,
                    "avg_launch_us": round(avg_kernel_s * 1e6, 2), "launches": kernel_updates,
                    "algorithmic_bytes_per_launch": FUSED_BYTES_PER_PIXEL * W * H,
                    "note": "device time of the timed region (one HIP event pair on the kernels' stream, seed_init of each pass "
                            "included) / update() calls; epipolar_match is VALU/LDS-bound (hundreds of flop per compulsory "
                            "byte), see roofline_valu / roofline_flops and DESIGN.md"}
        tv_avg_s = tv_ms / max(tv_launches, 1) / 1e3
        tv_iters_per_launch = tv_iters / max(tv_launches, 1)
        tv_achieved = TV_BYTES_PER_PIXEL_ITER * W * H * tv_iters_per_launch / tv_avg_s / 1e9 if tv_avg_s > 0 else 0.0
        tv_traffic = None
        if counters:
            tv_traffic = (counters.get("tv_bytes_per_launch") or {}).get(f"{W}x{H}")
        roofline_tv = {"bound": "hbm", "kernel": "tv_iterate", "achieved": round(tv_achieved, 2), "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": round(tv_achieved / HBM_PEAK_GBS, 5), "traffic": tv_traffic,
                       "avg_launch_us": round(tv_avg_s * 1e6, 2), "launches": tv_launches,
                       "iterations_per_launch": round(tv_iters_per_launch, 2),
                       "denoise_wall_ms": round(denoise_wall_ms, 3), "iterations": tv_iters}

        search_stats, h2d, cpu = None, None, None
        if not args.no_extras:
            # search statistics of the timed workload (separate pass over the same sequence, diagnostics counters on)
            s3 = new_seeds()
            s3.setOption(api.OPT_COLLECT_STATS, 1)
            set_ref(s3)
            tot = {"live_seeds": 0, "steps": 0, "ncc_evals": 0}
            for k in range(1, F):
                s3.updateDevice(frames[k].data, frames[k].stride, poses[k])
                st = s3.lastStats()
                for key in tot:
                    tot[key] += st[key]
            search_stats = {k: round(v / (F - 1), 1) for k, v in tot.items()}
            del s3

            # H2D-inclusive rate (SURVEY.md 8d / test/dataset_main.cpp:101-103: the frame upload is inside the timed calls):
            # frames start in pageable host memory; float frames through update(), 8-bit frames through update_u8()
            if keep_host:
                def host_pass(use_u8):
                    s4 = new_seeds()
                    imgs = gray if use_u8 else [synth.to_float_image(g) for g in gray]
                    def one():
                        if use_u8:
                            s4.setReferenceImageU8(imgs[0], poses[0], min_depth, max_depth)
                            for k in range(1, F):
                                s4.updateU8(imgs[k], poses[k])
                        else:
                            s4.setReferenceImage(imgs[0], poses[0], min_depth, max_depth)
                            for k in range(1, F):
                                s4.update(imgs[k], poses[k])
                    for _ in range(2):  # untimed: staging buffers, copy engine clocks
                        one()
                    s4.sync()
                    ts = time.perf_counter()
                    for _ in range(3):
                        one()
                    s4.sync()
                    return W * H * (F - 1) * 3 / (time.perf_counter() - ts) / 1e6
                u8_rate, f32_rate = host_pass(True), host_pass(False)
                resident = total_units / max_elapsed / 1e6 / world
                h2d = {"value": round(u8_rate, 1), "unit": "Mpix/s",
                       "path": "rmd_hip_seeds_update_u8: 8-bit frames in pageable host memory -> pinned ring -> copy engine -> staging buffer in "
                               "HBM; the update's own setup kernel waits for the copy's sequence number and applies x(1/255) (what "
                               "Depthmap::inputImage feeds, depthmap.cpp:95-106); no event or wait between the copy and compute streams",
                       "float_frames_update_mpix_s": round(f32_rate, 1), "frac_of_resident": round(u8_rate / resident, 3),
                       "bound": "PCIe Gen5 x16 63 GB/s = 205 000 Mpix/s of 8-bit frames: not the limit; the update kernels are (float frames: the host's "
                                "1.2 MB copy into pinned memory, ~50 us per frame)"}

            def gpu_sample(n):
                s2 = new_seeds()
                run_pass(s2, n)
                s2.sync()
                ts = time.perf_counter()
                run_pass(s2, n)
                s2.sync()
                return W * H * n / (time.perf_counter() - ts) / 1e6

            if args.cpu_seconds > 0 and world == 1 and keep_host:  # an N = 1 figure; at N > 1 the other ranks would spin beside it
                try:
                    cpu = cpu_baseline(lambda k: (synth.to_float_image(gray[k]), poses[k]), W, H, K, F, SIDE, min_depth, max_depth,
                                       args.cpu_seconds, gpu_sample)
                except Exception as e:  # the bench line must survive a missing oracle
                    cpu = {"value": None, "unit": "Mpix/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}

        value = total_units / max_elapsed / 1e6
        label = cfg_name or "non-BASELINE workload"
        result = {
            "metric": f"Mpix/s depth-filter updates ({W}x{H}, {F} frames)", "value": round(value, 2), "unit": "Mpix/s",
            "n_gpus": world, "control_plane": batch.backend_name(), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(max_elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{label}: {W}x{H} synthetic over-table sequence; one step = setReferenceImage(frame 0) + update() on "
                                   f"frames 1..{F - 1} ({F - 1} updates); NCC patch side {SIDE} (half-patch 4), max epipolar extent "
                                   f"100 px; one independent sequence per GPU; timed region = {args.steps} complete passes",
                       "frames_per_pass": F, "updates_timed": n_updates, "frames_resident_in_hbm": True,
                       "matcher": {-1: "library default (tile pipeline, compact search kernel)", 0: "per-pixel kernel", 1: "round-1 tile pipeline",
                                   2: "one-launch frame kernel", 3: "tile pipeline, compact search kernel"}.get(args.matcher, str(args.matcher)), "converged_seeds_at_end": converged,
                       "mean_per_update": search_stats, "us_per_update_wall": round(max_elapsed / n_updates * 1e6, 3)},
            "roofline": roofline,
            "roofline_valu": valu_roofline(avg_kernel_s, os.path.join(ROOT, "profiles", "traffic.json")) if headline else None,
            "roofline_flops": flops_roofline(avg_kernel_s, search_stats["ncc_evals"] if search_stats else None, SIDE),
            "roofline_denoiser": roofline_tv, "cpu_baseline": cpu, "h2d_inclusive": h2d, "parity": PARITY_NOTE,
            "per_rank": [{"elapsed_s": round(r[0], 6), "mpix": r[1] / 1e6, "updates": int(r[2]), "converged": int(r[3])} for r in per_rank],
        }
    batch.barrier(device)
    if rank == 0:
        print(json.dumps(result), flush=True)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
