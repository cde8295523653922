#!/usr/bin/env python3
"""Fast GPU-only regression: the tile kernel against the per-pixel kernel (itself pinned to the oracle by
tests/test_hip_parity.py) on a full-size sequence, every plane bit for bit.  usage: quick_parity.py [W H frames side]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpg_open_remode_amd import api, synth

w, h, n, side = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (640, 480, 60, 9)))
seq = synth.Sequence(w, h, n, seed=0)
cam = api.PinholeCamera(*seq.K)
a, b = api.SeedMatrix(w, h, cam, patch_side=side), api.SeedMatrix(w, h, cam, patch_side=side)
a.setOption(api.OPT_MATCHER, 0); b.setOption(api.OPT_MATCHER, 1)
b.setOption(api.OPT_WINDOW, int(os.environ.get("RMD_WINDOW", "0")))
for s in (a, b):
    s.setOption(api.OPT_TIMING, 1)
    s.setReferenceImage(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
bad = 0
for k in range(1, n):
    a.update(seq.images[k], seq.T_curr_world[k]); b.update(seq.images[k], seq.T_curr_world[k])
    if k % 10 == 0 or k == n - 1:
        sa, sb = a.state(), b.state()
        for p in range(8):
            x, y = sa[p], sb[p]
            m = int(np.count_nonzero(~((x == y) | (np.isnan(x) & np.isnan(y))))) if x.dtype.kind == 'f' else int((x != y).sum())
            if m:
                bad += m; print(f"frame {k} plane {p}: {m} mismatches")
ta, na = a.timing(api.STAGE_UPDATE); tb, nb = b.timing(api.STAGE_UPDATE)
print(f"{w}x{h} side {side} {n-1} updates: pixel kernel {ta/na*1e3:.1f} us/frame, tile kernel {tb/nb*1e3:.1f} us/frame, mismatches {bad}")
sys.exit(1 if bad else 0)
