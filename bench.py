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

value = W * H * (F-1) * K / elapsed, measured the way SURVEY.md 8(d) / test/dataset_main.cpp:101-103 define the metric: the
frame upload is INSIDE update().  Every frame starts as an 8-bit gray image in pageable host memory (what the camera / the
dataset reader delivers and what Depthmap::inputImage receives, depthmap.cpp:95-106) and is handed to rmd_hip_seeds_update_u8;
the timed region is bracketed by barrier + device synchronisation on both sides and the MAX over ranks is taken.  The same passes
with the frames already resident in HBM (`resident`) and with float frames through update(float*), the reference's own signature
(`float_frames`), are measured right after and reported in the same line.  For N > 1 every rank runs its own independent sequence
on its own GPU (weak scaling, no data-path collective) and rank 0 prints the aggregate.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel, algorithmic bytes / HIP-event time on the stream the
kernels run on), "roofline_valu" / "roofline_flops" (the roofs that actually bind the NCC search), "roofline_denoiser",
"heavy_prefix" (updates 1..20 of a pass, every seed live), "batched_per_gpu" (B independent sequences stepped by one launch pair
per step, rmd_hip_batch_*), "cpu_baseline" (the reference's own kernels on the host cores) and "parity_vs_glibc_reference" (how
far the result is from the untouched reference, measured on the same run).
"""
import argparse
import resource
import glob
import hashlib
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
# What pure fp32 instruction streams SUSTAIN on this part at four waves per SIMD (tools/ubench/valu_rate.hip, profiles/r03_lds_ceiling.txt):
# v_fma_f32 834-852, v_add_f32 / v_mul_f32 917-968, v_mov_b32 1052 G wave-instructions/s.  The NCC block is 1/4 fma, 3/4 add / sub / mul.
VALU_SUSTAINED_GINST_S = 0.25 * 843.0 + 0.75 * 940.0
FUSED_BYTES_PER_PIXEL = 52  # fused check+match+update, compulsory traffic (R 32 + W 20)
TV_BYTES_PER_PIXEL_ITER = 40
NCC_FLOP_PER_TAP = 14       # 3 FMA + 1 bilinear fetch per tap
WIDTH, HEIGHT, FRAMES, SIDE = 640, 480, 200, 9
TV_LAMBDA, TV_ITERS = 0.5, 200
KNOWN_CONFIGS = {(640, 480, 200): "configs[1]", (1280, 960, 500): "configs[2]", (1920, 1080, 1000): "configs[4]"}
PARITY_NOTE = ("bit-identical (all state planes, convergence masks, TV-L1 output) to the reference's own kernels compiled UNMODIFIED for the "
               "CPU (IEEE fp32, no contraction) and linked against the system's libm (glibc 2.35): the expf/sinf/acosf of csrc/rmd_math.h "
               "restate glibc's routines and equal them for all 2^32 arguments (oracle/libm_exhaustive.cpp); golden fixtures generated from "
               "/root/reference in tests/golden; measured on this run: parity_vs_glibc_reference")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5, help="timed passes over the sequence (one step = setReference + F-1 updates)")
    ap.add_argument("--warmup", type=int, default=3, help="untimed passes before the timed region (with one, the first timed pass is still 3-5 % slower than the rest: pass_ms.first)")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="budget of the cpu_baseline run (0 disables it and the glibc parity figures)")
    ap.add_argument("--matcher", type=int, default=-1, help="A/B: 0 per-pixel kernel, 3 tile pipeline (the library's default, used when the flag "
                    "is absent); 1 / 2 (retired variants) only with an A/B build of the library")
    ap.add_argument("--unit-target", type=int, default=0, help="tile pipeline: work units aimed at per frame, in multiples of the resident search workgroups (experiments; "
                    "0 = the library's defaults: 2 for a single sequence, 1 for a batch)")
    ap.add_argument("--size", default=f"{WIDTH}x{HEIGHT}", help="frame size WxH; 640x480 with 200 frames is the headline metric")
    ap.add_argument("--frames", type=int, default=0, help="frames per pass incl. the reference (default: 200; 500 at 1280x960, "
                    "1000 at 1920x1080, as BASELINE.json configures them)")
    ap.add_argument("--tv-iters", type=int, default=0, help="TV-L1 iterations of the separately reported denoise (default 200; 500 at 1920x1080)")
    ap.add_argument("--resident", action="store_true", help="timed region with the frames already resident in HBM (rmd_hip_seeds_update_device) instead "
                    "of 8-bit frames from host memory; the line says so")
    ap.add_argument("--batch", default="2,4,8,16", help="batch sizes of the batched_per_gpu section (empty: skip it)")
    ap.add_argument("--batch-per-gpu", type=int, default=1, help="B independent sequences per rank, stepped as ONE batch (rmd_hip_batch_*): rank r runs scenes "
                    "r*B .. r*B+B-1; the headline stays whole-job pixels / max elapsed.  Default 1: one sequence per GPU (the driver's BENCH / SCALE lines)")
    ap.add_argument("--dist", action="store_true", help="create the torch.distributed (RCCL) group even for a single rank")
    ap.add_argument("--no-extras", action="store_true", help="the timed region and the denoise only (profiling runs)")
    ap.add_argument("--configs", default="configs[2],configs[4]", help="the other single-GPU configurations of BASELINE.json measured after the headline (one pass each; "
                    "empty: skip them); only with the default workload")
    ap.add_argument("--no-live", dest="live", action="store_false", help="skip the live-use figure (the node's state machine over the headline's frames)")
    ap.add_argument("--rendezvous-only", action="store_true", help="launch-path check, no measurement: join the process group (gloo when "
                    "there is no GPU), run the barriers and the throughput gather with zero work, print {\"rendezvous\": ...} and exit")
    return ap.parse_args()


def resolve_workload(args):
    w, h = (int(v) for v in args.size.lower().split("x"))
    frames = args.frames or {(1280, 960): 500, (1920, 1080): 1000}.get((w, h), FRAMES)
    tv_iters = args.tv_iters or (500 if (w, h) == (1920, 1080) else TV_ITERS)
    return w, h, frames, tv_iters


def _code_only(text):
    """C++ source without comments and with white space collapsed: what a compiler sees"""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":  # string / character literal: copied as it is
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


KERNEL_SOURCES = ["rmd_device.hpp", "rmd_frame.hpp", "rmd_frame_window.hpp", "rmd_frame_ingest.hpp", "rmd_frame_setup.hpp", "rmd_frame_search.hpp", "rmd_kernels.hpp", "rmd_lab.hpp", "rmd_matcher.hpp", "rmd_math.h", "rmd_tv_kernels.hpp"]  # the device code of librmd_hip.so (csrc/rmd_host.hpp and the .hip units are host code)


def kernel_source_sha256():
    """hash of the device code of librmd_hip.so (the kernel headers; the host orchestration in rmd_capi.hip is not part of it), comments
    and white space left out: committed counter files (profiles/traffic.json) carry the hash of the sources they were measured on, and a
    figure derived from them is refused when the kernels have changed since -- editing a comment does not change a kernel"""
    h = hashlib.sha256()
    for f in [os.path.join(ROOT, "rpg_open_remode_amd", "csrc", n) for n in KERNEL_SOURCES]:
        h.update(os.path.basename(f).encode())
        h.update(_code_only(open(f, "r", encoding="utf-8", errors="replace").read()).encode())
    return h.hexdigest()


def load_counters(path):
    try:
        return json.load(open(path))
    except Exception:
        return None


NCC_VALU_PER_WAVE_EVALUATION = {9: 851}  # VALU instructions of one NCC evaluation of 64 lanes (the filter + accumulate block), by patch side


def valu_roofline(avg_launch_s, counters, n_sequences=1, ncc_evals_per_update=None):
    """The roof that actually binds the seed update: VALU issue.  Wave-instruction counts per update() come from
    the committed PMC pass over the same complete passes (profiles/traffic.json, SQ_INSTS_VALU) -- accepted only if that file was
    measured on the kernel sources of this build --, the launch time from this run."""
    if not counters or avg_launch_s <= 0:
        return None
    if counters.get("kernel_source_sha256") != kernel_source_sha256():
        return {"bound": "valu", "stale": True, "note": "profiles/traffic.json was measured on other kernel sources (kernel_source_sha256 differs): "
                "re-run tools/profile_round.sh; no figure is derived from stale instruction counts"}
    try:
        n = float(sum(counters["valu_wave_instructions_per_update"].values())) * n_sequences
    except Exception:
        return None
    achieved = n / avg_launch_s / 1e9
    useful = None
    if ncc_evals_per_update and SIDE in NCC_VALU_PER_WAVE_EVALUATION:
        # how much of what the chip issues is the arithmetic the reference prescribes: NCC evaluations / 64 lanes x the block's instruction count
        # over ALL VALU wave-instructions of the update (setup + search; the rest is unit set-up, item decode, window staging, the fused
        # finalisation, partly filled waves)
        useful = round(float(ncc_evals_per_update) / 64.0 * NCC_VALU_PER_WAVE_EVALUATION[SIDE] * n_sequences / n, 4)
    return {"bound": "valu", "kernel": "seed_update", "achieved": round(achieved, 1), "peak": round(VALU_PEAK_GINST_S, 1), "useful_valu_frac": useful,
            "per_kernel": {k: int(v) for k, v in counters["valu_wave_instructions_per_update"].items()},
            "unit": "G wave-instructions/s", "frac": round(achieved / VALU_PEAK_GINST_S, 4),
            "sustained_peak": round(VALU_SUSTAINED_GINST_S, 1), "frac_of_sustained": round(achieved / VALU_SUSTAINED_GINST_S, 4),
            "sustained_peak_note": "what pure fp32 fma / add / mul streams reach on this part at four waves per SIMD, weighted by the NCC block's mix "
                                   "(tools/ubench/valu_rate.hip); the search kernel on the heaviest updates issues at 2.88 cycles per instruction "
                                   "against 2.95 for a pure v_fma_f32 stream (profiles/r03_lds_ceiling.txt)",
            "wave_instructions_per_launch": int(n), "source": "profiles/traffic.json (rocprofv3 --pmc SQ_INSTS_VALU, whole passes, same kernel sources)"}


def flops_roofline(avg_launch_s, ncc_evals_per_update, side):
    if avg_launch_s <= 0 or not ncc_evals_per_update:
        return None
    flop = NCC_FLOP_PER_TAP * side * side * float(ncc_evals_per_update)
    achieved = flop / avg_launch_s / 1e12
    return {"bound": "fp32", "kernel": "seed_update", "achieved": round(achieved, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_PEAK_TFLOPS, 4), "flop_per_launch": int(flop),
            "definition": f"{NCC_FLOP_PER_TAP} flop x {side * side} taps x NCC evaluations (SURVEY.md 8d), evaluations counted by this run"}


def cpu_reference_run(frame_fn, width, height, K, n_frames, side, min_depth, max_depth, budget_s, tv_iters):
    """The reference's own kernels (oracle/_ref, built from /root/reference for the host, glibc libm: the UNTOUCHED reference) over the
    same sequence, all host cores, until `budget_s` is used up (the whole sequence fits on the GPU box's cores).  Returns the
    cpu_baseline object, the number of updates run, the oracle's state after its last update, its TV-L1 result on that state (None if
    the run was cut short) and the oracle kind."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # idle OpenMP workers must not spin while the GPU is timed
    import oracles as O
    kind = "reference" if O.available("ref", side) else "port"
    olib = O.OracleLib("ref" if kind == "reference" else "port", side)
    # as many OpenMP threads as the process may really use: a container that sees 256 cores but has a CPU quota of 16 is throttled for
    # most of every scheduling period when all visible cores spin up (the figure would be the throttle's, not the cores')
    from rpg_open_remode_amd import synth as _synth
    visible = olib.lib.ref_max_threads() if kind == "reference" else olib.lib.orc_max_threads()
    cores = max(1, min(int(visible), _synth.effective_cpus()))
    (olib.lib.ref_set_num_threads if kind == "reference" else olib.lib.orc_set_num_threads)(cores)
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
    what = ("the reference's own seed_matrix.cu kernels compiled for the host (oracle/Makefile), glibc libm" if kind == "reference"
            else "CPU restatement of the reference (oracle/remode_oracle.cpp)")
    out = {"value": round(mpix, 4), "unit": "Mpix/s", "cores": int(cores), "kind": kind,
           "sample": f"updates 1..{n} of the same {width}x{height} sequence (patch side {side}), {dt:.1f} s; {what}; {cores} OpenMP threads = the CPUs this "
                     f"process may use (CPU quota / affinity; {int(visible)} cores visible)"}
    state = s.state()
    den = None
    if n == n_frames - 1:
        d = O.Denoiser(olib, width, height)
        d.set_large_sigma_sq(max_depth - min_depth)
        den = d.denoise(s, TV_LAMBDA, tv_iters)

    # SURVEY 8(d) "CPU baseline timing": (i) one thread and (ii) all cores, for the reference's own kernels (Oracle A, kind "reference") AND for
    # the restatement (Oracle B, kind "port").  The one-thread figures are taken on updates 1..SHORT (a thread needs seconds per update there;
    # every seed is live on them, so they are the sequence's most expensive updates) with the all-cores figure on the SAME updates beside them.
    SHORT = max(1, min(10, max(3, 800000 // (width * height)), n_frames - 1))

    def rate(okind, threads, n_updates):
        ol = O.OracleLib(okind, side)
        (ol.lib.ref_set_num_threads if okind == "ref" else ol.lib.orc_set_num_threads)(threads)
        sd = O.Seeds(ol, width, height, K)
        sd.set_reference(img0, T0, min_depth, max_depth)
        frames = [frame_fn(k) for k in range(1, n_updates + 1)]
        t = time.perf_counter()
        for img, T in frames:
            sd.update(img, T)
        dt_ = time.perf_counter() - t
        sd.close()
        return {"value": round(width * height * n_updates / dt_ / 1e6, 4), "unit": "Mpix/s", "cores": int(threads), "sample": f"updates 1..{n_updates}, {dt_:.1f} s"}

    if budget_s >= 10.0:
        kinds = (["ref"] if kind == "reference" else []) + (["port"] if O.available("port", side) else [])
        for okind in kinds:
            entry = {"one_thread": rate(okind, 1, SHORT), "all_cores_same_updates": rate(okind, cores, SHORT)}
            if okind == "ref" or kind == "port":  # the object's own kind: its whole-sequence figure is the headline above
                out.update(entry)
            if okind == "port":
                full = rate("port", cores, n) if kind == "reference" else {k: out[k] for k in ("value", "unit", "cores")}
                port = {"kind": "port", "what": "CPU restatement of the reference (oracle/remode_oracle.cpp), bit-identical to the reference's kernels", **full, **entry}
                if kind == "reference":
                    out["port"] = port
        (olib.lib.ref_set_num_threads if kind == "reference" else olib.lib.orc_set_num_threads)(cores)
    return out, n, state, den, kind


def cuda_build_tolerance():
    """north_star: "depth-map equality vs. the reference CUDA path ... within a stated float tolerance".  No CUDA device exists here; the distance to
    such a build (8-bit texture weights, -use_fast_math) is bounded with a MODEL of it -- Oracle B's "cudalike" build, tests/cudalike_tolerance.py,
    CPU only -- and the table it wrote is quoted here, not re-measured: profiles/r06_cudalike_tolerance.txt.  The model's switches are of two kinds:
    SPECIFIED behaviour of that build (texture weights with 8 fractional bits -- the CUDA programming guide's linear-filtering formula --, flush to
    zero, FMA contraction) and GUESSES (the approximate division / square roots and the fast or other-library expf / sinf / acosf, which the guide
    gives as error BOUNDS only); the guessed ones are therefore run three ways -- the model's fixed forms, "correctly rounded moved by up to 1 ulp",
    "moved by up to the documented maximum" -- and the figures below are the RANGE over the three "all switches" rows."""
    path = os.path.join(ROOT, "profiles", "r06_cudalike_tolerance.txt")

    def row(line):
        f = [c.strip() for c in line.split("|")]
        d = f[3].split()
        return {"converged_mask_mismatches": int(f[1].split()[0]), "convergence_states_differing": int(f[2]), "depth_rmse_m": float(d[0]),
                "depth_rmse_best_99pct_m": float(d[1]), "depth_median_abs_diff_m": float(d[2]), "depth_p99_abs_diff_m": float(d[3]), "depth_max_abs_diff_m": float(d[4]),
                "seeds_off_by_more_than_1cm": int(d[5]), "denoised_rmse_m": float(f[5].split()[0])}
    try:
        rows = {}
        for line in open(path):
            if line.startswith("ALL OF THE ABOVE"):
                rows["guessed operations in the model's fixed forms"] = row(line)
            elif line.startswith("[range] ALL switches"):
                rows["guessed operations moved by up to " + line.split("at ", 1)[1].split("|")[0].strip()] = row(line)
        if not rows:
            return None
        keys = next(iter(rows.values())).keys()
        rng = {k: [min(r[k] for r in rows.values()), max(r[k] for r in rows.values())] for k in keys}
        px = 640 * 480
        return {"source": "profiles/r06_cudalike_tolerance.txt (tests/cudalike_tolerance.py --sensitivity: configs[1] + TV-L1 on the CPU, plain Oracle A semantics against a model of the "
                          "reference's nvcc -use_fast_math build reading images through the texture unit, every switch on)",
                "specified_switches": "texture filter weights with 8 fractional bits (rounded; truncated also run), flush-to-zero, FMA contraction",
                "guessed_switches": "x / y, sqrtf / rsqrtf, expf, sinf, acosf / atan2f: known as error bounds only; run as fixed forms, at 1 ulp and at their documented maxima",
                "all_switches_rows": rows, "range": rng,
                "reading": f"against a physical CUDA run expect {100.0 * rng['converged_mask_mismatches'][0] / px:.2f}-{100.0 * rng['converged_mask_mismatches'][1] / px:.2f} % of the convergence "
                           f"mask to differ, a median depth difference of {rng['depth_median_abs_diff_m'][0]:.1e}-{rng['depth_median_abs_diff_m'][1]:.1e} m, a depth RMSE of "
                           f"{rng['depth_rmse_m'][0]:.1e}-{rng['depth_rmse_m'][1]:.1e} m (a handful of seeds lock onto another NCC peak) and {rng['denoised_rmse_m'][0]:.1e}-"
                           f"{rng['denoised_rmse_m'][1]:.1e} m on the denoised map -- an ESTIMATE from a model validated against no CUDA output; what it shows robustly is that a "
                           "last-ulp change of any single operation already gives 1e-3: north_star's 1e-4 holds -- as exact equality -- against the reference's executable "
                           "semantics (parity_vs_glibc_reference), and cannot hold against any build that rounds differently"}
    except Exception as e:  # the bench line must survive a missing file
        return {"source": path, "unavailable": str(e)}


def scenes_of_rank(rank, batch_per_gpu):
    """the independent sequences (scene / trajectory seeds) rank `rank` runs: B consecutive ones, stepped together as one batch"""
    return [rank * batch_per_gpu + i for i in range(batch_per_gpu)]


def bind_rank_device(local_rank, n_dev, set_torch_device, check_device):
    """Rank r of a --gpus N launch works on device local_rank % n_dev (more ranks than devices -- a one-GPU lease driven with --gpus 8 -- share
    devices): torch's current device first (the RCCL group's), then the library's (api.checkCudaDevice = rmd_hip_set_device,
    check_cuda_device.cu:60-71,109).  Called BEFORE the rank creates its first handle: handles bind to the device that is current when they
    are created (tests/test_bench_cpu.py holds bench.py to that order)."""
    if n_dev < 1:
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback of the product path)")
    dev_index = local_rank % n_dev
    set_torch_device(dev_index)
    if not check_device(dev_index):
        raise SystemExit("no usable HIP device")
    return dev_index


def rank_record(r):
    """one entry of `per_rank` from the gathered 7 x f64 record (elapsed, pixels, updates, converged, sequences, host CPU seconds, submit seconds)"""
    return {"elapsed_s": round(r[0], 6), "mpix": r[1] / 1e6, "updates": int(r[2]), "converged": int(r[3]), "sequences": int(r[4]),
            "host_cpu_s": round(r[5], 4), "host_cores_busy": round(r[5] / r[0], 3), "host_submit_us_per_update": round(r[6] / max(r[2] / max(r[4], 1.0), 1.0) * 1e6, 2)}


def measure_config(api, synth, label, Wc, Hc, Fc, tv_iters, passes=1):
    """One of the other single-GPU configurations of BASELINE.json (configs[2], configs[4]) measured like the headline -- complete passes, 8-bit host
    frames inside update() (`value`) and frames resident in HBM (`resident`), one HIP event pair on the kernels' stream, TV-L1 of the config --
    in a few seconds: `passes` timed passes after one warm-up pass each.  The driver's line carries them so that they are no longer figures only
    the builder has seen (profiles/r06_bench_config*.json are the same runs through `--size`)."""
    t_r = time.perf_counter()
    Kc = synth.intrinsics(Wc, Hc)
    gray, poses, dev = [], [], []
    rng0 = None
    for k in range(Fc):
        T = synth.pose(k, 0)
        g, rng = synth.render(Wc, Hc, T, 0, want_range=(k == 0), K=Kc)
        if k == 0:
            rng0 = rng
        gray.append(g)
        poses.append(np.ascontiguousarray(synth.invert_pose(T).astype(np.float32).reshape(12)))
        d = api.DeviceImage(Wc, Hc, np.float32)
        d.setDevData(synth.to_float_image(g))
        dev.append(d)
    lo, hi = float(rng0.min()), float(rng0.max())
    render_s = time.perf_counter() - t_r
    s = api.SeedMatrix(Wc, Hc, api.PinholeCamera(*Kc), patch_side=SIDE)

    def pass_u8():
        s.setReferenceImageU8(gray[0], poses[0], lo, hi)
        for k in range(1, Fc):
            s.updateU8(gray[k], poses[k])

    def pass_res():
        s.setReferenceImageDevice(dev[0].data, dev[0].stride, poses[0], lo, hi)
        for k in range(1, Fc):
            s.updateDevice(dev[k].data, dev[k].stride, poses[k])

    def timed(one_pass):
        one_pass()  # warm-up (staging buffers, clocks)
        s.sync()
        s.setOption(api.OPT_TIMING, 2)
        s.timingReset()
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        for _ in range(passes):
            one_pass()
        s.sync()
        dt = time.perf_counter() - t0
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        ms, n = s.timing(api.STAGE_UPDATE)
        s.setOption(api.OPT_TIMING, 0)
        cpu = (ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime)
        return {"value": round(Wc * Hc * (Fc - 1) * passes / dt / 1e6, 1), "unit": "Mpix/s", "us_per_update_wall": round(dt / n * 1e6, 2),
                "us_per_update_device": round(ms / n * 1e3, 2), "host_cores_busy": round(cpu / dt, 3), "passes": passes}
    u8 = timed(pass_u8)
    u8["frames_on_copy_engines_addressed_directly"], u8["frames_on_the_copy_stream"] = s.stagedFrames()
    res = timed(pass_res)
    pinned = None
    if hasattr(api, "PinnedFrames"):  # frames the caller keeps in pinned host memory (rmd_hip_seeds_update_u8_pinned): no memcpy into the ring
        pf = api.PinnedFrames(Fc, Hc, Wc)
        for k in range(Fc):
            pf.frames[k][:] = gray[k]

        def pass_pinned():
            s.setReferenceImageU8(gray[0], poses[0], lo, hi)
            for k in range(1, Fc):
                s.updateU8Pinned(pf.frames[k], poses[k])
        pinned = timed(pass_pinned)
        s.sync()
        pf.close()
    avg_s = u8["us_per_update_device"] / 1e6
    ach = FUSED_BYTES_PER_PIXEL * Wc * Hc / avg_s / 1e9
    den = api.DepthmapDenoiser(Wc, Hc)
    den.setLargeSigmaSq(hi - lo)
    den.setOption(api.DENOISE_OPT_TIMING, 1)
    den.denoise(s.getMu(), s.getSigmaSq(), s.getA(), s.getB(), TV_LAMBDA, 10, download=True)
    td = time.perf_counter()
    den.denoise(s.getMu(), s.getSigmaSq(), s.getA(), s.getB(), TV_LAMBDA, tv_iters, download=True)
    wall_ms = (time.perf_counter() - td) * 1e3
    tv_ms, tv_launches = den.timing()
    tv_bw = TV_BYTES_PER_PIXEL_ITER * Wc * Hc * tv_iters / (tv_ms / 1e3) / 1e9 if tv_ms > 0 else 0.0
    out = {"workload": f"{label}: {Wc}x{Hc}, {Fc} frames ({Fc - 1} updates per pass), patch side {SIDE}, TV-L1 {tv_iters} iterations; 8-bit host frames inside update()",
           "value": u8["value"], "unit": "Mpix/s", "u8_host_frames": u8, "resident": res, "u8_over_resident": round(u8["value"] / res["value"], 4),
           "caller_pinned_frames": pinned, "caller_pinned_over_resident": round(pinned["value"] / res["value"], 4) if pinned else None,
           "converged_seeds_at_end": s.getConvergedCount(),
           "roofline": {"bound": "hbm", "kernel": "seed_update (fused)", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                        "frac_resident": round(FUSED_BYTES_PER_PIXEL * Wc * Hc / (res["us_per_update_device"] / 1e6) / 1e9 / HBM_PEAK_GBS, 5),
                        "avg_launch_us": u8["us_per_update_device"], "algorithmic_bytes_per_launch": FUSED_BYTES_PER_PIXEL * Wc * Hc, "traffic": None},
           "roofline_denoiser": {"bound": "hbm", "kernel": "tv_iterate", "achieved": round(tv_bw, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(tv_bw / HBM_PEAK_GBS, 5),
                                 "avg_launch_us": round(tv_ms / max(tv_launches, 1) * 1e3, 2), "launches": tv_launches, "iterations": tv_iters,
                                 "denoise_wall_ms": round(wall_ms, 3), "traffic": None},
           "host_render_and_upload_s": round(render_s, 1)}
    del den, s, dev
    return out


class BatchAsSeeds:
    """the few SeedMatrix methods the timed region uses, on a SeedMatrixBatch (--batch-per-gpu)"""

    def __init__(self, bm):
        self.bm = bm

    def sync(self): self.bm.sync()
    def setOption(self, option, value): self.bm.setOption(option, value)
    def timingReset(self): self.bm.timingReset()
    def timing(self, stage): return self.bm.timing()  # (device ms of the region, steps in it)
    def getConvergedCount(self): return sum(self.bm[i].getConvergedCount() for i in range(self.bm.n))


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
        Bq = max(1, args.batch_per_gpu)
        max_e, total_u, per_rank = batch.gather_throughput(0.001 * (rank + 1), float(rank + 1), None, extra=(float(local_rank), 0.0, float(Bq)))
        batch.barrier()
        if rank == 0:
            print(json.dumps({"rendezvous": "ok", "n_gpus": world, "control_plane": batch.backend_name(), "max_elapsed_s": max_e,
                              "total_units": total_u, "per_rank": [list(r) for r in per_rank], "batch_per_gpu": Bq,
                              "scenes_of_rank": [scenes_of_rank(r, Bq) for r in range(world)]}), flush=True)
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback of the product path)")
    dev_index = bind_rank_device(local_rank, torch.cuda.device_count(), torch.cuda.set_device, api.checkCudaDevice)  # before the rank's first handle
    device = torch.device("cuda", dev_index)

    # one independent sequence per rank (scene / trajectory seed = rank), rendered on the host frame by frame: the 8-bit frames stay
    # in pageable host memory (the timed region hands them over like a camera driver would); a float copy of every frame is made
    # resident through the library's own rmd::DeviceImage for the `resident` figure (torch is here for torch.distributed only)
    K = synth.intrinsics(W, H)

    def render_scene(scene, want_resident):
        gray, poses, dev = [], [], []
        rng0 = None
        for k in range(F):
            T = synth.pose(k, scene)
            g, rng = synth.render(W, H, T, scene, want_range=(k == 0), K=K)
            if k == 0:
                rng0 = rng
            gray.append(g)
            poses.append(np.ascontiguousarray(synth.invert_pose(T).astype(np.float32).reshape(12)))
            if want_resident:
                d = api.DeviceImage(W, H, np.float32)
                d.setDevData(synth.to_float_image(g))
                dev.append(d)
        return {"gray": gray, "poses": poses, "dev": dev, "min": float(rng0.min()), "max": float(rng0.max())}

    B = max(1, args.batch_per_gpu)
    if B > api.MAX_BATCH:
        raise SystemExit(f"--batch-per-gpu {B}: a batch holds at most {api.MAX_BATCH} sequences")
    my_scenes = scenes_of_rank(rank, B)
    t_render = time.perf_counter()
    seq = render_scene(my_scenes[0], True)
    batch_scenes = [seq] + [render_scene(sc, args.resident) for sc in my_scenes[1:]]
    render_s = time.perf_counter() - t_render
    gray, poses, frames = seq["gray"], seq["poses"], seq["dev"]
    min_depth, max_depth = seq["min"], seq["max"]

    def new_seeds():
        s = api.SeedMatrix(W, H, api.PinholeCamera(*K), patch_side=SIDE)
        if args.matcher >= 0:
            s.setOption(api.OPT_MATCHER, args.matcher)
        if args.unit_target > 0:
            s.setOption(api.OPT_UNIT_TARGET, args.unit_target)
        return s

    def pass_resident(s, n_updates=None):
        """one step: the reference frame, then updates 1..F-1 (or the first n_updates of them), frames resident in HBM"""
        s.setReferenceImageDevice(frames[0].data, frames[0].stride, poses[0], min_depth, max_depth)
        for k in range(1, (F if n_updates is None else n_updates + 1)):
            s.updateDevice(frames[k].data, frames[k].stride, poses[k])

    def pass_u8(s, n_updates=None):
        """the same step with every frame handed over as an 8-bit image in pageable host memory"""
        s.setReferenceImageU8(gray[0], poses[0], min_depth, max_depth)
        for k in range(1, (F if n_updates is None else n_updates + 1)):
            s.updateU8(gray[k], poses[k])

    run_pass = pass_resident if args.resident else pass_u8
    bm = None
    if B > 1:  # the rank's B sequences as ONE batch: one setup + one search launch per stream group and step
        bm = api.SeedMatrixBatch(B, W, H, api.PinholeCamera(*K), patch_side=SIDE)
        if args.unit_target > 0:
            bm.setOption(api.OPT_UNIT_TARGET, args.unit_target)

        def run_pass(_s, n_updates=None):
            sc = batch_scenes
            for i in range(B):
                if args.resident:
                    bm[i].setReferenceImageDevice(sc[i]["dev"][0].data, sc[i]["dev"][0].stride, sc[i]["poses"][0], sc[i]["min"], sc[i]["max"])
                else:
                    bm[i].setReferenceImageU8(sc[i]["gray"][0], sc[i]["poses"][0], sc[i]["min"], sc[i]["max"])
            for k in range(1, (F if n_updates is None else n_updates + 1)):
                p = [sc[i]["poses"][k] for i in range(B)]
                if args.resident:
                    bm.updateDevice([sc[i]["dev"][k].data for i in range(B)], [sc[i]["dev"][k].stride for i in range(B)], p)
                else:
                    bm.updateU8([sc[i]["gray"][k] for i in range(B)], p)

    def timed(s, one_pass, passes):
        """(wall seconds, device ms of the region, update() calls in it) of `passes` complete passes"""
        s.sync()
        s.setOption(api.OPT_TIMING, 2)  # one HIP event pair around the timed region, on the stream the kernels run on
        s.timingReset()  # records the region's start event on the (idle) stream: nothing but the timed launches follows it
        t0 = time.perf_counter()
        for _ in range(passes):
            one_pass(s)
        s.sync()
        dt = time.perf_counter() - t0
        ms, n = s.timing(api.STAGE_UPDATE)
        s.setOption(api.OPT_TIMING, 0)
        return dt, ms, n

    seeds = BatchAsSeeds(bm) if bm is not None else new_seeds()
    for _ in range(args.warmup):  # W untimed passes (clocks, code objects, allocator, staging buffers)
        run_pass(seeds)
    seeds.sync()
    seeds.setOption(api.OPT_TIMING, 2)

    batch.barrier(device)
    torch.cuda.synchronize()
    seeds.timingReset()
    ru0 = resource.getrusage(resource.RUSAGE_SELF)  # CPU time of ALL threads of this process (the library's copy helpers included)
    t0 = time.perf_counter()
    pass_marks = [t0]
    for _ in range(args.steps):
        run_pass(seeds)
        pass_marks.append(time.perf_counter())  # (the caller is at most three frames ahead of the device: a pass's submit time is its device time +- 0.1 ms)
    t_submitted = pass_marks[-1]  # every update() has returned (frames copied, launches queued); the device is still busy
    seeds.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    host_cpu_s = (ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime)
    host_submit_s = t_submitted - t0
    batch.barrier(device)
    kernel_ms, kernel_updates = seeds.timing(api.STAGE_UPDATE)  # device time of the region / update() calls in it
    seeds.setOption(api.OPT_TIMING, 0)
    converged = seeds.getConvergedCount()
    # which way the host frames of this handle went (warm-up + timed passes): the library falls back to its copy stream by itself where it cannot
    # address the copy engines (csrc/rmd_engines.hip)
    staged_route = None
    if hasattr(seeds, "stagedFrames") and not args.resident:
        by_engines, by_stream = seeds.stagedFrames()
        staged_route = {"RMD_HIP_TUNE_COPY_ENGINES": api.getTunable(api.TUNE_COPY_ENGINES), "frames_on_copy_engines_addressed_directly": by_engines,
                        "frames_on_the_copy_stream": by_stream}
    n_updates = (F - 1) * args.steps          # update() calls of ONE sequence; a batch steps B sequences per call
    units = float(W * H * n_updates * B)
    max_elapsed, total_units, per_rank = batch.gather_throughput(elapsed, units, device, extra=(float(n_updates * B), float(converged), float(B), host_cpu_s, host_submit_s))

    # ---- the rest is reporting on rank 0; other ranks idle at the final barrier
    result = None
    if rank == 0:
        # denoiser of the config (reported beside the metric, not inside it)
        if bm is not None:  # the B depth maps of the batch in one launch sequence (rmd_hip_batch_denoise)
            ranges = [sc["max"] - sc["min"] for sc in batch_scenes]
            bm.denoise(ranges, TV_LAMBDA, 10)  # warm
            td = time.perf_counter()
            hip_denoised = bm.denoise(ranges, TV_LAMBDA, tv_iters)[0]
            denoise_wall_ms = (time.perf_counter() - td) * 1e3
            tv_ms, tv_launches = bm.denoiseTiming()
        else:
            den = api.DepthmapDenoiser(W, H)
            den.setLargeSigmaSq(max_depth - min_depth)
            den.setOption(api.DENOISE_OPT_TIMING, 1)
            den.denoise(seeds.getMu(), seeds.getSigmaSq(), seeds.getA(), seeds.getB(), TV_LAMBDA, 10, download=True)  # warm
            td = time.perf_counter()
            hip_denoised = den.denoise(seeds.getMu(), seeds.getSigmaSq(), seeds.getA(), seeds.getB(), TV_LAMBDA, tv_iters, download=True)
            denoise_wall_ms = (time.perf_counter() - td) * 1e3
            tv_ms, tv_launches = den.timing()
        # the state the TIMED handle ended in (after its last complete pass of 8-bit host frames, or resident frames with --resident): what
        # parity_vs_glibc_reference compares with the reference's run over the same sequence
        timed_state = {p: seeds.download(p) for p in range(5)} if (args.cpu_seconds > 0 and world == 1 and F <= 500 and not args.no_extras and bm is None) else None

        counters = load_counters(os.path.join(ROOT, "profiles", "traffic.json")) if headline else None
        fresh = bool(counters) and counters.get("kernel_source_sha256") == kernel_source_sha256()
        avg_kernel_s = kernel_ms / max(kernel_updates, 1) / 1e3
        achieved = FUSED_BYTES_PER_PIXEL * W * H * B / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0  # (a batch step updates B sequences)
        roofline = {"bound": "hbm", "kernel": "seed_update (fused seed_check+epipolar_match+triangulation+seed_update)",
                    "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": counters.get("seed_update_bytes_per_launch") if fresh else None,
                    "avg_launch_us": round(avg_kernel_s * 1e6, 2), "launches": kernel_updates,
                    "algorithmic_bytes_per_launch": FUSED_BYTES_PER_PIXEL * W * H * B, "sequences_per_launch": B,
                    "note": "device time of the timed region (one HIP event pair on the kernels' stream, seed_init of each pass "
                            "included) / update() calls; epipolar_match is VALU/LDS-bound (hundreds of flop per compulsory "
                            "byte), see roofline_valu / roofline_flops and DESIGN.md"}
        tv_avg_s = tv_ms / max(tv_launches, 1) / 1e3
        tv_iters_per_launch = tv_iters / max(tv_launches, 1)
        tv_achieved = TV_BYTES_PER_PIXEL_ITER * W * H * B * tv_iters_per_launch / tv_avg_s / 1e9 if tv_avg_s > 0 else 0.0
        tv_traffic = None
        if fresh:
            tv_traffic = (counters.get("tv_bytes_per_launch") or {}).get(f"{W}x{H}")
        roofline_tv = {"bound": "hbm", "kernel": "tv_iterate", "achieved": round(tv_achieved, 2), "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": round(tv_achieved / HBM_PEAK_GBS, 5), "traffic": tv_traffic,
                       "avg_launch_us": round(tv_avg_s * 1e6, 2), "launches": tv_launches,
                       "iterations_per_launch": round(tv_iters_per_launch, 2),
                       "denoise_wall_ms": round(denoise_wall_ms, 3), "iterations": tv_iters, "depth_maps_per_launch": B}

        search_stats, cpu, glibc, resident, floats, heavy, batched, other_path = None, None, None, None, None, None, None, None
        floats_other, other_configs, live, pinned = None, None, None, None
        extra_passes = max(1, min(args.steps, 3))
        if not args.no_extras and bm is None:
            # search statistics of the timed workload (separate pass over the same sequence, diagnostics counters on)
            s3 = new_seeds()
            s3.setOption(api.OPT_COLLECT_STATS, 1)
            s3.setReferenceImageDevice(frames[0].data, frames[0].stride, poses[0], min_depth, max_depth)
            tot = {"live_seeds": 0, "steps": 0, "ncc_evals": 0}
            evals_first20 = 0
            for k in range(1, F):
                s3.updateDevice(frames[k].data, frames[k].stride, poses[k])
                st = s3.lastStats()
                for key in tot:
                    tot[key] += st[key]
                if k <= 20:
                    evals_first20 += st["ncc_evals"]
            search_stats = {k: round(v / (F - 1), 1) for k, v in tot.items()}
            del s3

            # the other frame sources, measured like the timed region (complete passes, one event pair, synchronised on both sides)
            def rate(one_pass):
                s4 = new_seeds()
                one_pass(s4)  # untimed: staging buffers, copy engine clocks
                dt, ms, n = timed(s4, one_pass, extra_passes)
                return {"value": round(W * H * (F - 1) * extra_passes / dt / 1e6, 1), "unit": "Mpix/s", "us_per_update_wall": round(dt / n * 1e6, 2),
                        "us_per_update_device": round(ms / n * 1e3, 2), "passes": extra_passes}
            if args.resident:
                other_path = dict(rate(pass_u8), path="rmd_hip_seeds_update_u8: 8-bit frames in pageable host memory (SURVEY.md 8d: the upload inside update())")
            else:
                resident = dict(rate(pass_resident), path="rmd_hip_seeds_update_device: frames already resident in HBM, read in place")

            if hasattr(api, "PinnedFrames"):
                # frames the CALLER keeps in pinned host memory (rmd_hip_seeds_update_u8_pinned, not in the reference): the copy engine reads them where
                # they lie -- update() without the memcpy into the library's ring
                pf = api.PinnedFrames(F, H, W)
                for k in range(F):
                    pf.frames[k][:] = gray[k]

                def pass_pinned(s):
                    s.setReferenceImageU8(gray[0], poses[0], min_depth, max_depth)
                    for k in range(1, F):
                        s.updateU8Pinned(pf.frames[k], poses[k])
                ru_a = resource.getrusage(resource.RUSAGE_SELF)
                t_a = time.perf_counter()
                pinned = rate(pass_pinned)
                t_b = time.perf_counter()
                ru_b = resource.getrusage(resource.RUSAGE_SELF)
                pinned = dict(pinned, path="rmd_hip_seeds_update_u8_pinned: 8-bit frames the caller keeps in pinned host memory, read by the copy engine where they lie "
                              "(an extension: the reference's update() takes pageable memory)",
                              host_cores_busy_incl_warmup_pass=round(((ru_b.ru_utime + ru_b.ru_stime) - (ru_a.ru_utime + ru_a.ru_stime)) / (t_b - t_a), 3))
                pf.close()

            if F <= 500:
                fimgs = [synth.to_float_image(g) for g in gray]

                def pass_float(s):
                    s.setReferenceImage(fimgs[0], poses[0], min_depth, max_depth)
                    for k in range(1, F):
                        s.update(fimgs[k], poses[k])
                floats = dict(rate(pass_float), path="rmd_hip_seeds_update: float frames in pageable host memory, the reference's own signature "
                              "(seed_matrix.cu:120-128), frames as the reference's host path produces them (convertTo(CV_32F, 1.0f / 255.0f), depthmap.cpp:105): the library "
                              "finds every pixel to be an 8-bit level, bit pattern by bit pattern, and sends the bytes (DESIGN.md 4.6); the examination is split across a few host threads")
                # ... and float frames that are NOT 8-bit levels (every pixel scaled by 0.999): nothing to pack, 4 bytes per pixel cross the host link
                fimgs = [np.ascontiguousarray(f * np.float32(0.999)) for f in fimgs]
                floats_other = dict(rate(pass_float), path="rmd_hip_seeds_update with float frames that are not 8-bit levels (x 0.999): sent as floats, 4 bytes per pixel")
                del fimgs

            # updates 1..20 of a pass: every seed is live and searches its full range (the heaviest twentieth of the job)
            s5 = new_seeds()
            run_pass(s5, 20)
            s5.sync()
            s5.setOption(api.OPT_TIMING, 2)
            hp_ms, hp_n = 0.0, 0
            for _ in range(3):
                if args.resident:
                    s5.setReferenceImageDevice(frames[0].data, frames[0].stride, poses[0], min_depth, max_depth)
                else:
                    s5.setReferenceImageU8(gray[0], poses[0], min_depth, max_depth)
                s5.sync()
                s5.timingReset()
                for k in range(1, 21):
                    if args.resident:
                        s5.updateDevice(frames[k].data, frames[k].stride, poses[k])
                    else:
                        s5.updateU8(gray[k], poses[k])
                ms, n = s5.timing(api.STAGE_UPDATE)
                hp_ms += ms
                hp_n += n
            heavy = {"updates": "1..20 of a pass (3 repetitions)", "us_per_update_device": round(hp_ms / max(hp_n, 1) * 1e3, 2)}
            if fresh and counters.get("valu_wave_instructions_first20_per_update"):
                n_inst = float(sum(counters["valu_wave_instructions_first20_per_update"].values()))
                heavy["valu_frac"] = round(n_inst / (hp_ms / max(hp_n, 1) / 1e3) / 1e9 / VALU_PEAK_GINST_S, 4)
                if SIDE in NCC_VALU_PER_WAVE_EVALUATION and F > 20:
                    heavy["useful_valu_frac"] = round(evals_first20 / 20.0 / 64.0 * NCC_VALU_PER_WAVE_EVALUATION[SIDE] / n_inst, 4)
            del s5

            # batched mode: B independent sequences (scenes 0..B-1) stepped by ONE launch pair per step (rmd_hip_batch_*)
            sizes = [int(v) for v in args.batch.split(",") if v.strip()] if (args.batch and world == 1 and W * H <= 1280 * 960) else []
            if sizes:
                t_render = time.perf_counter()
                scenes = {rank: seq}
                for sc in range(max(sizes)):
                    if sc not in scenes:
                        scenes[sc] = render_scene(sc, True)
                batched = {"what": "B independent sequences of this workload (scenes 0..B-1) on ONE GPU, stepped together: one setup + one search launch "
                                   "per step for all of them (rmd_hip_batch_update_*); aggregate Mpix/s over all B sequences; every member is bit-identical "
                                   "to the same sequence run alone (tests/test_batch.py)",
                           "host_render_s": round(time.perf_counter() - t_render, 1)}
                for Bq in sizes:
                    sc = [scenes[i] for i in range(Bq)]
                    bmq = api.SeedMatrixBatch(Bq, W, H, api.PinholeCamera(*K), patch_side=SIDE)
                    if args.unit_target > 0:
                        bmq.setOption(api.OPT_UNIT_TARGET, args.unit_target)

                    def bpass(use_u8):
                        for i in range(Bq):
                            if use_u8:
                                bmq[i].setReferenceImageU8(sc[i]["gray"][0], sc[i]["poses"][0], sc[i]["min"], sc[i]["max"])
                            else:
                                bmq[i].setReferenceImageDevice(sc[i]["dev"][0].data, sc[i]["dev"][0].stride, sc[i]["poses"][0], sc[i]["min"], sc[i]["max"])
                        for k in range(1, F):
                            p = [sc[i]["poses"][k] for i in range(Bq)]
                            if use_u8:
                                bmq.updateU8([sc[i]["gray"][k] for i in range(Bq)], p)
                            else:
                                bmq.updateDevice([sc[i]["dev"][k].data for i in range(Bq)], [sc[i]["dev"][k].stride for i in range(Bq)], p)
                    entry = {}
                    for use_u8, name in ((False, "resident"), (True, "u8_host_frames")):
                        bpass(use_u8)
                        bmq.sync()
                        bmq.setOption(api.OPT_TIMING, 2)
                        bmq.timingReset()
                        tb = time.perf_counter()
                        for _ in range(extra_passes):
                            bpass(use_u8)
                        bmq.sync()
                        dtb = time.perf_counter() - tb
                        msb, nb = bmq.timing()
                        bmq.setOption(api.OPT_TIMING, 0)
                        entry[name] = {"value": round(W * H * (F - 1) * Bq * extra_passes / dtb / 1e6, 1), "unit": "Mpix/s",
                                       "us_per_step_wall": round(dtb / nb * 1e6, 2), "us_per_step_device": round(msb / nb * 1e3, 2),
                                       "us_per_sequence_update": round(dtb / nb / Bq * 1e6, 2)}
                    rv = valu_roofline(entry["resident"]["us_per_step_device"] / 1e6, counters, n_sequences=Bq)
                    if rv and not rv.get("stale"):
                        entry["roofline_valu_frac"] = rv["frac"]
                        # (the instruction counts are per sequence: a batch executes B times the single sequence's, the useful fraction is the same)
                        entry["useful_valu_frac"] = valu_roofline(1.0, counters, ncc_evals_per_update=search_stats["ncc_evals"])["useful_valu_frac"] if search_stats else None
                    entry["converged_seeds_at_end"] = [bmq[i].getConvergedCount() for i in range(Bq)]
                    # TV-L1 of the Bq depth maps in ONE launch sequence (rmd_hip_batch_denoise: grid z = member), device time of the iteration launches
                    ranges_q = [sc[i]["max"] - sc[i]["min"] for i in range(Bq)]
                    bmq.denoise(ranges_q, TV_LAMBDA, 8, download=False)
                    tdq = time.perf_counter()
                    bmq.denoise(ranges_q, TV_LAMBDA, tv_iters, download=False)
                    dn_wall = (time.perf_counter() - tdq) * 1e3
                    dn_ms, dn_launches = bmq.denoiseTiming()
                    dn_bw = TV_BYTES_PER_PIXEL_ITER * W * H * Bq * tv_iters / (dn_ms / 1e3) / 1e9 if dn_ms > 0 else 0.0
                    entry["denoise_all_members"] = {"iterations": tv_iters, "launches": dn_launches, "device_ms": round(dn_ms, 3), "wall_ms": round(dn_wall, 3),
                                                    "ms_per_depth_map": round(dn_ms / Bq, 4), "achieved_GBs_algorithmic": round(dn_bw, 1),
                                                    "frac_of_hbm_peak": round(dn_bw / HBM_PEAK_GBS, 4)}
                    batched[f"B={Bq}"] = entry
                    del bmq

            # the other single-GPU configurations of BASELINE.json and live use (the node's state machine), after everything of the headline
            if headline and world == 1 and not args.resident and args.configs:
                other_configs = {}
                for name, (Wc, Hc, Fc, tvc) in (("configs[2]", (1280, 960, 500, 200)), ("configs[4]", (1920, 1080, 1000, 500))):
                    if name in args.configs:
                        try:
                            other_configs[name] = measure_config(api, synth, name, Wc, Hc, Fc, tvc)
                        except Exception as e:  # the headline line must survive
                            other_configs[name] = {"unavailable": str(e)}
            if headline and world == 1 and args.live:
                try:
                    from rpg_open_remode_amd import live as live_mod
                    lseq = synth.Sequence(W, H, F)
                    live = {"what": "rmd::DepthmapNode's state machine over this workload's frames (depthmap_node.cpp:125-173): 8-bit host frames, a converged count after every "
                                    "update, a new reference when 10 % have converged or the camera has moved 0.5, and at every reference change TV-L1 (0.5, 200), the "
                                    "point cloud and -- every 11 messages -- the coloured convergence map; Mpix/s = frame pixels x messages / wall time",
                            "publication_in_the_callback": live_mod.run_live(W, H, F, SIDE, async_publish=False, passes=3, seq=lseq),
                            "publication_off_the_update_stream": live_mod.run_live(W, H, F, SIDE, async_publish=True, passes=3, breakdown=False, seq=lseq)}
                    live["value"] = live["publication_off_the_update_stream"]["value"]
                    live["unit"] = "Mpix/s"
                    del lseq
                except Exception as e:
                    live = {"unavailable": str(e)}

            # CPU baseline = the untouched reference on the host cores; the same run gives the distance of the GPU result from it
            if args.cpu_seconds > 0 and world == 1 and F <= 500:
                try:
                    cpu, n_cpu, ref_state, ref_den, kind = cpu_reference_run(lambda k: (synth.to_float_image(gray[k]), poses[k]), W, H, K, F, SIDE,
                                                                             min_depth, max_depth, args.cpu_seconds, tv_iters)
                    time.sleep(0.3)
                    s2 = new_seeds()
                    pass_resident(s2, n_cpu)
                    s2.sync()
                    ts = time.perf_counter()
                    pass_resident(s2, n_cpu)
                    s2.sync()
                    cpu["gpu_same_sample"] = round(W * H * n_cpu / (time.perf_counter() - ts) / 1e6, 2)
                    if kind == "reference":
                        sys.path.insert(0, os.path.join(ROOT, "tests"))
                        import glibc_parity
                        full = ref_den is not None and n_cpu == F - 1
                        # the whole sequence fitted the CPU budget: the planes of the handle that ran the TIMED region are compared (and its
                        # denoised map); a CPU run that was cut short is compared with a second handle stopped at the same update
                        hip_state = timed_state if (full and timed_state is not None) else {p: s2.download(p) for p in range(5)}
                        glibc = glibc_parity.compare(ref_state, hip_state, ref_den if full else None, hip_denoised if full else None)
                        glibc["after_updates"] = n_cpu
                        glibc["hip_side"] = ("the handle of the timed region after its last pass (frame source as timed), planes and TV-L1 map" if full and timed_state is not None
                                             else "a second handle stopped after the same number of updates (frames resident in HBM)")
                        glibc["reference"] = ("the reference's seed_matrix.cu / depthmap_denoiser.cu compiled unmodified for the host against glibc's libm "
                                              "(Oracle A); the HIP path equals that build with expf/sinf/acosf from csrc/rmd_math.h bit for bit")
                        glibc["asserted_in"] = "tests/test_parity_glibc.py"
                    del s2
                except Exception as e:  # the bench line must survive a missing oracle
                    cpu = {"value": None, "unit": "Mpix/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}

        value = total_units / max_elapsed / 1e6
        label = cfg_name or "non-BASELINE workload"
        source = ("frames already resident in HBM (rmd_hip_seeds_update_device)" if args.resident else
                  "every frame an 8-bit image in pageable host memory handed to rmd_hip_seeds_update_u8: the H2D copy and the x(1/255) conversion are "
                  "inside the timed update() calls (SURVEY.md 8d, test/dataset_main.cpp:101-103)")
        result = {
            "metric": f"Mpix/s depth-filter updates ({W}x{H}, {F} frames)", "value": round(value, 2), "unit": "Mpix/s",
            "n_gpus": world, "control_plane": batch.backend_name(), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(max_elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{label}: {W}x{H} synthetic over-table sequence; one step = setReferenceImage(frame 0) + update() on "
                                   f"frames 1..{F - 1} ({F - 1} updates); NCC patch side {SIDE} (half-patch 4), max epipolar extent "
                                   f"100 px; " + ("one independent sequence per GPU" if B == 1 else f"{B} independent sequences per GPU (scenes r*{B} .. r*{B}+{B - 1} on rank r) stepped "
                                   f"together as one batch, one launch pair per stream group and step") + f"; timed region = {args.steps} complete passes; frame source: {source}",
                       "batch_per_gpu": B,
                       "frames_per_pass": F, "updates_timed": n_updates, "frames_resident_in_hbm": bool(args.resident), "h2d_inclusive": not args.resident,
                       "matcher": {-1: "library default (two-launch tile pipeline)", 0: "per-pixel kernel", 3: "two-launch tile pipeline"}.get(args.matcher, str(args.matcher)),
                       "converged_seeds_at_end": converged, "mean_per_update": search_stats,
                       "us_per_update_wall": round(max_elapsed / n_updates * 1e6, 3), "host_render_s": round(render_s, 1)},
            "roofline": roofline,
            "roofline_valu": valu_roofline(avg_kernel_s, counters, n_sequences=B, ncc_evals_per_update=search_stats["ncc_evals"] if search_stats else None) if headline else None,
            "roofline_flops": flops_roofline(avg_kernel_s, search_stats["ncc_evals"] if search_stats else None, SIDE),
            "roofline_denoiser": roofline_tv, "cpu_baseline": cpu,
            "resident": resident, "h2d_inclusive": other_path, "caller_pinned_frames": pinned, "float_frames": floats, "float_frames_not_8bit_levels": floats_other, "heavy_prefix": heavy, "batched_per_gpu": batched,
            "configs": other_configs, "live": live,
            "parity": PARITY_NOTE, "parity_vs_glibc_reference": glibc, "tolerance_vs_cuda_build_model": cuda_build_tolerance() if headline else None,
            "per_rank": [dict(rank_record(r), device=i % max(torch.cuda.device_count(), 1)) for i, r in enumerate(per_rank)],
            # host side of the timed region on rank 0: CPU seconds of all of the process's threads (getrusage), the same as cores kept busy, and
            # the wall time per update() call until the call returned (frame copied into the pinned ring, two launches queued) -- the device
            # time per update is roofline.avg_launch_us; a host that needs longer than that paces the run
            "host_cpu_s": round(host_cpu_s, 4), "host_cores_busy": round(host_cpu_s / elapsed, 3), "host_submit_us_per_update": round(host_submit_s / n_updates * 1e6, 2),
            "host_frame_route": staged_route,
            # spread inside the timed region (rank 0): the K passes one by one, as the caller sees them -- the time between the returns of the last
            # update() of consecutive passes; the caller runs at most three frames ahead of the device, so a pass is its device time +- 0.1 ms
            "pass_ms": ({"first": round((pass_marks[1] - pass_marks[0]) * 1e3, 4), "min": round(min(b - a for a, b in zip(pass_marks, pass_marks[1:])) * 1e3, 4),
                         "max": round(max(b - a for a, b in zip(pass_marks, pass_marks[1:])) * 1e3, 4), "passes": len(pass_marks) - 1} if (len(pass_marks) > 1 and not args.resident) else None),  # (resident frames: nothing holds the caller back, its marks say nothing)
        }
    batch.barrier(device)
    if rank == 0:
        print(json.dumps(result), flush=True)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
