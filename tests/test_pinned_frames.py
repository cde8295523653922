"""rmd_hip_seeds_update_u8_pinned (not in the reference): 8-bit frames the caller keeps in pinned host memory are read by the copy engine where
they lie -- the same bits as frames that are resident / copied by update_u8, tickets that say when a frame may be written again, and the
copying fall-back wherever the engine cannot take the frame from the caller's buffer."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from rpg_open_remode_amd import api, synth
from rpg_open_remode_amd._lib import RmdHipError
w, h, side, n = int(sys.argv[3]), int(sys.argv[4]), 7, 14
seq = synth.Sequence(w, h, n, 0)
cam = api.PinholeCamera(*seq.K)
def bits(st): return [np.ascontiguousarray(st[p]).view(np.uint32).copy() for p in sorted(st)]
def same(a, b): return all(np.array_equal(x, y) for x, y in zip(a, b))
s = api.SeedMatrix(w, h, cam, patch_side=side)
s.setReferenceImageU8(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
for k in range(1, n): s.updateU8(seq.gray[k], seq.T_curr_world[k])
want = bits(s.state())
direct = sys.argv[2] == "direct"

# (a) every frame in its own place of one pinned block
pf = api.PinnedFrames(n, h, w)
for k in range(n): pf.frames[k][:] = seq.gray[k]
p = api.SeedMatrix(w, h, cam, patch_side=side)
p.setReferenceImageU8(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
tickets = [p.updateU8Pinned(pf.frames[k], seq.T_curr_world[k]) for k in range(1, n)]
assert tickets == list(range(1, n)), tickets
p.sync()
assert p.pinnedFramesDone() == n - 1
assert same(want, bits(p.state())), "frames read from the caller's pinned block"
by_engines, by_stream = p.stagedFrames()
assert (by_engines == n - 1 and by_stream == 0) if direct else True, (by_engines, by_stream)

# (b) TWO pinned frames written in turn, each only when its ticket is done: what a producer with a double buffer does
two = api.PinnedFrames(2, h, w)
q = api.SeedMatrix(w, h, cam, patch_side=side)
q.setReferenceImageU8(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
last = [0, 0]
for k in range(1, n):
    b = k & 1
    while q.pinnedFramesDone() < last[b]: pass
    two.frames[b][:] = seq.gray[k]
    last[b] = q.updateU8Pinned(two.frames[b], seq.T_curr_world[k])
assert same(want, bits(q.state())), "double buffer driven by the tickets"

# (c) pageable memory is refused, and the handle goes on
try:
    q.updateU8Pinned(np.ascontiguousarray(seq.gray[1]), seq.T_curr_world[1]); raise SystemExit("pageable frame accepted")
except RmdHipError as e:
    assert "pinned" in str(e), str(e)
q.updateU8(seq.gray[1], seq.T_curr_world[1])
print("PINNED-OK")
'''


@pytest.mark.gpu
@pytest.mark.parametrize("env_extra, how, size", [({}, "direct", (320, 240)), ({"RMD_HIP_COPY_ENGINES": "0"}, "copied", (320, 240)),
                                                  ({"RMD_HIP_HOST_FRAMES": "inplace"}, "copied", (320, 240)), ({"RMD_HIP_COPY_ENGINES": "1"}, "direct", (320, 240)),
                                                  ({}, "copied", (322, 242))])  # (rows that are no multiple of 4 bytes: the ring's rows are padded, the frame is copied row by row)
def test_frames_kept_in_pinned_memory(env_extra, how, size):
    env = {k: v for k, v in os.environ.items() if not k.startswith("RMD_HIP_") or k == "RMD_HIP_LIB"}
    env.update(env_extra)
    res = subprocess.run([sys.executable, "-c", CHILD, ROOT, how, str(size[0]), str(size[1])], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0 and "PINNED-OK" in res.stdout, res.stdout[-2500:]
