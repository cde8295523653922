#!/usr/bin/env python3
"""Experiment: device time of rmd_hip_batch_denoise (8 members, 640x480, 200 iterations) -- run once per A/B library (RMD_HIP_LIB)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rpg_open_remode_amd import api, synth
W, H, F, N = 640, 480, 12, 8
seqs = [synth.Sequence(W, H, F, s) for s in range(N)]
B = api.SeedMatrixBatch(N, W, H, api.PinholeCamera(*seqs[0].K), patch_side=9)
for i in range(N):
    B[i].setReferenceImage(seqs[i].images[0], seqs[i].T_curr_world[0], seqs[i].min_depth, seqs[i].max_depth)
for k in range(1, F):
    B.updateU8([s.gray[k] for s in seqs], [s.T_curr_world[k] for s in seqs])
dr = [s.max_depth - s.min_depth for s in seqs]
outs = B.denoise(dr, 0.5, 200)
ms = []
for _ in range(10):
    B.denoise(dr, 0.5, 200, download=False)
    ms.append(B.denoiseTiming()[0])
import hashlib
print(f"{os.path.basename(os.environ.get('RMD_HIP_LIB', 'product'))}: 8 x 640x480 x 200 iterations: device ms median {np.median(ms):.3f} min {min(ms):.3f}; launches {B.denoiseTiming()[1]}; "
      f"result sha {hashlib.sha256(b''.join(o.tobytes() for o in outs)).hexdigest()[:12]}")
