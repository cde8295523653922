#!/usr/bin/env python3
"""How far can a run of the reference's REAL build -- nvcc -use_fast_math, images read through the texture unit -- be from the arithmetic
contract this repository is bit-identical to (IEEE fp32, fp32 filter weights; Oracle A = the reference's sources on the CPU)?  No CUDA
device exists here, so the distance is BOUNDED with a model: Oracle B's "cudalike" build (oracle/remode_oracle.cpp; equal to Oracle A bit
for bit with every switch off, tests/test_oracle_pin.py) evaluates, switch by switch, the operations such a build evaluates differently:
8-bit texture weights, approximate division / square roots, __expf, __sinf, another acosf, flush-to-zero, FMA contraction.

Runs BASELINE configs[1] (640x480, 199 updates, patch side 9) + TV-L1 (0.5, 200) once per switch and once with all of them, against the
plain run, and prints / writes the table DESIGN.md 2 quotes.  Test infrastructure (CPU only).

usage: PYTHONPATH=. python tests/cudalike_tolerance.py [--frames 200] [--size 640x480] [--out profiles/r05_cudalike_tolerance.txt]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracles as O  # noqa: E402
from rpg_open_remode_amd import synth  # noqa: E402

SIDE = 9


def run(kind, flags, seq, lam=0.5, iters=200, amplitude=0):
    olib = O.OracleLib(kind, SIDE)
    assert olib.set_cudalike(flags, amplitude)
    s = O.Seeds(olib, seq.width, seq.height, seq.K)
    s.set_reference(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    for n in range(1, seq.n_frames):
        s.update(seq.images[n], seq.T_curr_world[n])
    d = O.Denoiser(olib, seq.width, seq.height)
    d.set_large_sigma_sq(seq.max_depth - seq.min_depth)
    out = {"mu": s.download(O.PLANE_MU), "conv": s.download(O.PLANE_CONV), "den": d.denoise(s, lam, iters)}
    olib.set_cudalike(0, 0)
    return out


def compare(ref, got):
    both = (ref["conv"] == O.CONVERGED) & (got["conv"] == O.CONVERGED)
    dm = (got["mu"].astype(np.float64) - ref["mu"].astype(np.float64))[both]
    dd = got["den"].astype(np.float64) - ref["den"].astype(np.float64)
    return {
        "mask_mismatch": int(np.count_nonzero((ref["conv"] == O.CONVERGED) != (got["conv"] == O.CONVERGED))),
        "state_mismatch": int(np.count_nonzero(ref["conv"] != got["conv"])),
        "converged": int(np.count_nonzero(got["conv"] == O.CONVERGED)),
        "depth_rmse": float(np.sqrt(np.mean(dm * dm))) if dm.size else 0.0,
        "depth_median": float(np.median(np.abs(dm))) if dm.size else 0.0,
        "depth_p99": float(np.percentile(np.abs(dm), 99)) if dm.size else 0.0,
        "depth_max": float(np.abs(dm).max()) if dm.size else 0.0,
        "depth_rmse_99": float(np.sqrt(np.mean(np.sort(dm * dm)[: max(1, int(0.99 * dm.size))]))) if dm.size else 0.0,  # without the worst 1 %
        "over_1cm": int(np.count_nonzero(np.abs(dm) > 0.01)),
        "bit_identical": float(np.mean(got["mu"][both].view(np.uint32) == ref["mu"][both].view(np.uint32))) if dm.size else 1.0,
        "denoised_rmse": float(np.sqrt(np.mean(dd * dd))),
        "denoised_max": float(np.abs(dd).max()),
    }


def switches():
    L = O.OracleLib
    return [("[specified] texture weights: 8 fractional bits, rounded; 4-tap form", "cudalike", L.TEX8),
            ("[specified] texture weights: 8 fractional bits, truncated", "cudalike", L.TEX8_TRUNC),
            ("[guess] x / y -> x * (1 / y)", "cudalike", L.DIV),
            ("[guess] sqrtf -> x * rsqrt(x), rsqrtf correctly rounded", "cudalike", L.SQRT),
            ("[guess] expf -> exp2(x * log2 e)", "cudalike", L.EXP),
            ("[guess] sinf: another <= 2.5-ulp sine", "cudalike", L.SIN),
            ("[guess] acosf / atan2f: another <= 2.5-ulp implementation", "cudalike", L.ACOS),
            ("[specified] flush to zero", "cudalike", L.FTZ),
            ("[specified] FMA contraction (-fmad=true)", "cudalike_fma", 0),
            ("ALL OF THE ABOVE (rounded weights)", "cudalike_fma", L.TEX8 | L.DIV | L.SQRT | L.EXP | L.SIN | L.ACOS | L.FTZ),
            ("all, without the texture weights (-use_fast_math alone)", "cudalike_fma", L.DIV | L.SQRT | L.EXP | L.SIN | L.ACOS | L.FTZ),
            ("(upper bracket) sinf on the documented ABSOLUTE bound of __sinf: 2^-21 grid", "cudalike", L.SIN_ABS)]


def sensitivity():
    """the GUESSED switches one by one and all together, each at 1 ulp and at its documented maximum (orc_set_cudalike_amplitude): how much of the
    table above is the shape this model happens to give an operation whose CUDA form is only known as an error bound"""
    L = O.OracleLib
    guessed = L.DIV | L.SQRT | L.EXP | L.SIN | L.ACOS
    out = []
    for name, flags in (("x / y", L.DIV), ("sqrtf / rsqrtf", L.SQRT), ("expf", L.EXP), ("sinf", L.SIN), ("acosf / atan2f", L.ACOS)):
        for amp, label in ((1, "1 ulp"), (-1, "documented maximum")):
            out.append((f"[guess] {name}: correctly rounded, moved by up to {label}", "cudalike", flags, amp))
    for amp, label in ((1, "1 ulp"), (-1, "documented maximum")):
        out.append((f"[range] ALL switches, the guessed ones at {label}", "cudalike_fma", L.TEX8 | guessed | L.FTZ, amp))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sensitivity", action="store_true", help="also the guessed switches at 1 ulp and at their documented maxima (adds 12 runs)")
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--size", default="640x480")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    w, h = (int(v) for v in a.size.split("x"))
    seq = synth.Sequence(w, h, a.frames, 0)
    t0 = time.time()
    ref = run("cudalike", 0, seq)
    n_conv = int(np.count_nonzero(ref["conv"] == O.CONVERGED))
    lines = [f"# tests/cudalike_tolerance.py: {w}x{h}, {a.frames - 1} updates, patch side {SIDE}, TV-L1 (0.5, 200); plain run: {n_conv} of {w * h} seeds converged "
             f"(depth range {seq.min_depth:.3f} .. {seq.max_depth:.3f} m).  Depth statistics over the seeds CONVERGED in both runs, metres.",
             f"{'switch':78s} | conv-mask mismatches | states differing | depth RMSE   (best 99 %) median    p99       max       > 1 cm | bit-identical | denoised RMSE   max"]
    print("\n".join(lines), flush=True)
    rows = [(n, k, f, 0) for n, k, f in switches()] + (sensitivity() if a.sensitivity else [])
    for name, kind, flags, amp in rows:
        c = compare(ref, run(kind, flags, seq, amplitude=amp))
        line = (f"{name:78s} | {c['mask_mismatch']:8d} ({100.0 * c['mask_mismatch'] / (w * h):6.3f} %) | {c['state_mismatch']:16d} | {c['depth_rmse']:.3e} {c['depth_rmse_99']:.3e} {c['depth_median']:.3e} "
                f"{c['depth_p99']:.3e} {c['depth_max']:.3e} {c['over_1cm']:6d} | {100.0 * c['bit_identical']:11.1f} % | {c['denoised_rmse']:.3e} {c['denoised_max']:.3e}")
        print(line, flush=True)
        lines.append(line)
    lines.append(f"# {time.time() - t0:.0f} s on {synth.effective_cpus()} CPUs")
    print(lines[-1])
    if a.out:
        open(a.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
