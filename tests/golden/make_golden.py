#!/usr/bin/env python3
"""Generates tests/golden/remode_golden_s<side>.npz from the REFERENCE ITSELF (Oracle A: /root/reference sources compiled
for the CPU through oracle/cuda_shim, see oracle/Makefile).  Run in the build container (needs oracle/_ref):

    python tests/golden/make_golden.py

Each fixture holds the inputs (8-bit frames, poses, intrinsics, depth range) and the reference's outputs on them:
  ref_*      reference sources + glibc libm            (pins oracle B built with -DRMD_ORACLE_LIBM)
  refrmd_*   reference sources + csrc/rmd_math.h libm   (pins oracle B default build AND the HIP path, bit for bit)
for every plane after the last update, the convergence plane after every update, and the TV-L1 output
(reference kernel with emulated barriers) for (lambda 0.5, 25 iterations).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracles as O  # noqa: E402
from rpg_open_remode_amd import synth  # noqa: E402

W, H, FRAMES, SEED = 96, 72, 8, 7


def main():
    frames_u8, poses = [], []
    K = synth.intrinsics(W, H)
    rng0 = None
    for k in range(FRAMES):
        T = synth.pose(k, SEED)
        gray, rng = synth.render(W, H, T, SEED, want_range=(k == 0), K=K)
        if k == 0:
            rng0 = rng
        frames_u8.append(gray)
        poses.append(synth.invert_pose(T).astype(np.float32).reshape(12))
    images = [synth.to_float_image(g) for g in frames_u8]
    dmin, dmax = np.float32(rng0.min()), np.float32(rng0.max())
    for side in (5, 9):
        out = {"width": W, "height": H, "side": side, "K": np.asarray(K, np.float32), "depth_range": np.asarray([dmin, dmax], np.float32),
               "frames_u8": np.stack(frames_u8), "T_curr_world": np.stack(poses), "tv_lambda": np.float32(0.5), "tv_iterations": 25}
        for kind, tag in (("ref", "ref"), ("ref_rmd", "refrmd")):
            lib = O.OracleLib(kind, side)
            s = O.Seeds(lib, W, H, K)
            s.set_reference(images[0], poses[0], dmin, dmax)
            out[f"{tag}_sum_templ"] = s.download(O.PLANE_SUM_TEMPL)
            out[f"{tag}_denom"] = s.download(O.PLANE_DENOM)
            convs = []
            for k in range(1, FRAMES):
                s.update(images[k], poses[k])
                convs.append(s.download(O.PLANE_CONV).astype(np.int8))
            out[f"{tag}_conv_per_update"] = np.stack(convs)
            for p, name in ((O.PLANE_MU, "mu"), (O.PLANE_SIGMA_SQ, "sigma_sq"), (O.PLANE_A, "a"), (O.PLANE_B, "b"), (O.PLANE_MATCH, "match")):
                out[f"{tag}_{name}"] = s.download(p)
            out[f"{tag}_converged_count"] = s.converged_count()
            out[f"{tag}_dist_from_ref"] = np.float32(s.dist_from_ref())
            d = O.Denoiser(lib, W, H)
            d.set_large_sigma_sq(dmax - dmin)
            out[f"{tag}_denoised"] = d.denoise(s, 0.5, 25)
        path = os.path.join(HERE, f"remode_golden_s{side}.npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
