"""Every way a host frame can reach the current image (RMD_HIP_HOST_FRAMES, read once per process, hence one child process per mode;
DESIGN.md 4.6) gives the state of the resident-frame path bit for bit: single sequences with 8-bit frames, float frames of 8-bit
levels (sent as bytes) and other float frames, and a batch of three with 8-bit frames."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from rpg_open_remode_amd import api, synth
w, h, side, n = 320, 240, 7, 12
seq = synth.Sequence(w, h, n, 0)
cam = api.PinholeCamera(*seq.K)
other = [(im * np.float32(0.999)).astype(np.float32) for im in seq.images]

def planes_of(images):
    out = []
    for im in images:
        d = api.DeviceImage(w, h, np.float32); d.setDevData(im); out.append(d)
    return out

def run(kind):
    s = api.SeedMatrix(w, h, cam, patch_side=side)
    if kind == "resident" or kind == "resident-other":
        pl = planes_of(seq.images if kind == "resident" else other)
        s.setReferenceImageDevice(pl[0].data, pl[0].stride, seq.T_curr_world[0], seq.min_depth, seq.max_depth)
        for k in range(1, n): s.updateDevice(pl[k].data, pl[k].stride, seq.T_curr_world[k])
    elif kind == "u8":
        s.setReferenceImageU8(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
        for k in range(1, n): s.updateU8(seq.gray[k], seq.T_curr_world[k])
    else:
        src = seq.images if kind == "f32" else other
        s.setReferenceImage(src[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
        for k in range(1, n): s.update(src[k], seq.T_curr_world[k])
    st = s.state()
    return [np.ascontiguousarray(st[p]).view(np.uint32).copy() for p in sorted(st)]

def same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))

want, want_other = run("resident"), run("resident-other")
assert same(want, run("u8")), "8-bit host frames"
assert same(want, run("f32")), "float host frames of 8-bit levels"
assert same(want_other, run("f32-other")), "other float host frames"
b = api.SeedMatrixBatch(3, w, h, cam, patch_side=side)
for m in b.members: m.setReferenceImageU8(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
for k in range(1, n): b.updateU8([seq.gray[k]] * 3, [seq.T_curr_world[k]] * 3)
for i in range(3):
    st = b[i].state()
    assert same(want, [np.ascontiguousarray(st[p]).view(np.uint32).copy() for p in sorted(st)]), f"batch member {i}"
print("MODES-OK")
'''


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["staged", "staged_ahead", "inplace", "inplace_ahead"])
def test_every_host_frame_mode_equals_the_resident_path(mode):
    env = dict(os.environ, RMD_HIP_HOST_FRAMES=mode, RMD_HIP_PACK_BACKOFF="0")
    res = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0 and "MODES-OK" in res.stdout, res.stdout[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["staged", "staged_ahead"])
@pytest.mark.parametrize("engines", [0, 1, 2, 3, 4])
def test_staged_frames_by_every_copy_route_equal_the_resident_path(engines, mode):
    """RMD_HIP_COPY_ENGINES: staged frames on copy engines addressed directly (one to four engines in rotation) or on the copy stream -- the same bits,
    and the route asked for is the route taken (the library falls back to the copy stream by itself where the engines cannot be addressed: on this box they can)."""
    env = dict(os.environ, RMD_HIP_HOST_FRAMES=mode, RMD_HIP_PACK_BACKOFF="0", RMD_HIP_COPY_ENGINES=str(engines), RMD_HIP_INGEST_PROFILE="1")
    res = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0 and "MODES-OK" in res.stdout, res.stdout[-2000:]
    import re
    routes = [(int(a), int(b)) for a, b in re.findall(r"staged frames: (\d+) on copy engines addressed directly \(route \d\), (\d+) on the copy stream", res.stdout)]
    assert len(routes) >= 3, res.stdout[-2000:]  # (the three single-sequence runs with host frames)
    for by_engines, by_stream in routes:
        assert by_engines + by_stream == 11 and (by_stream == 0 if engines else by_engines == 0), (engines, routes, res.stdout[-1500:])


@pytest.mark.gpu
def test_the_default_route_of_host_frames_is_the_copy_engines():
    """no environment: a SeedMatrix's 8-bit host frames travel on copy engines addressed directly (rmd_hip_seeds_staged_frames says so) -- a silent fall-back to
    the copy stream on a box where the engines can be addressed would be a regression nobody sees in the parity tests"""
    code = ("import sys; sys.path.insert(0, sys.argv[1])\n"
            "from rpg_open_remode_amd import api, synth\n"
            "seq = synth.Sequence(320, 240, 8, 0); s = api.SeedMatrix(320, 240, api.PinholeCamera(*seq.K), patch_side=5)\n"
            "s.setReferenceImageU8(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)\n"
            "for k in range(1, 8): s.updateU8(seq.gray[k], seq.T_curr_world[k])\n"
            "s.sync(); print('ROUTE', s.stagedFrames(), api.getTunable(api.TUNE_COPY_ENGINES), api.getTunable(api.TUNE_HOST_FRAMES))\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith("RMD_HIP_") or k == "RMD_HIP_LIB"}
    env["RMD_HIP_INGEST_PROFILE"] = "1"
    res = subprocess.run([sys.executable, "-c", code, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0 and "ROUTE (7, 0) -1 -1" in res.stdout, res.stdout[-2000:]
    # the device's HSA agent is found by its PCI address, not by being the only one: on a node of eight that is what keeps a rank's frames off
    # the engines of another rank's device
    assert "matched by address" in res.stdout, res.stdout[-2000:]


MIXED_CHILD = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from rpg_open_remode_amd import api, synth
w, h, side, n, B = 320, 240, 7, 26, 3
seq = synth.Sequence(w, h, n, 0)
cam = api.PinholeCamera(*seq.K)
def bits(st): return [np.ascontiguousarray(st[p]).view(np.uint32).copy() for p in sorted(st)]
s = api.SeedMatrix(w, h, cam, patch_side=side)
pl = []
for im in seq.images:
    d = api.DeviceImage(w, h, np.float32); d.setDevData(im); pl.append(d)
s.setReferenceImageDevice(pl[0].data, pl[0].stride, seq.T_curr_world[0], seq.min_depth, seq.max_depth)
for k in range(1, n): s.updateDevice(pl[k].data, pl[k].stride, seq.T_curr_world[k])
want = bits(s.state())
b = api.SeedMatrixBatch(B, w, h, cam, patch_side=side)
for m in b.members: m.setReferenceImageU8(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
# the way a step's frames travel changes from step to step, in runs of 1, 2, 3 ... steps: every transition between "kernels read the pinned
# block" and "an engine copies it into one of eight staging buffers" in both directions, at every phase of the two rings
modes, k = [api.getTunable(api.TUNE_HOST_FRAMES)], 1
run = 1
while k < n:
    mode = 2 if (run % 2) else 0   # inplace / staged
    for _ in range(run):
        if k >= n: break
        api.setTunable(api.TUNE_HOST_FRAMES, mode)
        b.updateU8([seq.gray[k]] * B, [seq.T_curr_world[k]] * B)
        k += 1
    run += 1
api.setTunable(api.TUNE_HOST_FRAMES, -1)
for i in range(B):
    got = bits(b[i].state())
    assert all(np.array_equal(x, y) for x, y in zip(want, got)), f"batch member {i}"
print("MIXED-OK")
'''


@pytest.mark.gpu
def test_a_batch_that_changes_between_in_place_and_staged_steps():
    """steps read in place by the kernels and steps staged by a copy engine into the deep staging ring, alternating in runs of 1, 2, 3 ... steps:
    the pinned blocks are protected by the kernels' progress words in one mode and by the engine's signals in the other"""
    res = subprocess.run([sys.executable, "-c", MIXED_CHILD, ROOT], env=dict(os.environ), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0 and "MIXED-OK" in res.stdout, res.stdout[-2000:]


GROUPS_CHILD = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from rpg_open_remode_amd import api, synth
w, h, side, n, B = 256, 192, 5, 10, 5
seqs = [synth.Sequence(w, h, n, s) for s in range(B)]
cam = api.PinholeCamera(*seqs[0].K)
def bits(st): return [np.ascontiguousarray(st[p]).view(np.uint32).copy() for p in sorted(st)]
alone = []
for q in seqs:
    s = api.SeedMatrix(w, h, cam, patch_side=side)
    s.setReferenceImage(q.images[0], q.T_curr_world[0], q.min_depth, q.max_depth)
    for k in range(1, n): s.update(q.images[k], q.T_curr_world[k])
    alone.append(bits(s.state()))
b = api.SeedMatrixBatch(B, w, h, cam, patch_side=side)
for m, q in zip(b.members, seqs): m.setReferenceImage(q.images[0], q.T_curr_world[0], q.min_depth, q.max_depth)
for k in range(1, n):
    frames = [q.gray[k] for q in seqs]
    if k == 4: frames[2] = None  # member 2 sits step 4 out ...
    b.updateU8(frames, [q.T_curr_world[k] for q in seqs])
    if k == 4:  # ... and catches up on its own
        b.updateU8([None, None, seqs[2].gray[4], None, None], [q.T_curr_world[4] for q in seqs])
for i in range(B):
    got = bits(b[i].state())
    assert all(np.array_equal(x, y) for x, y in zip(alone[i], got)), f"member {i}"
print("GROUPS-OK")
'''


@pytest.mark.gpu
@pytest.mark.parametrize("groups", [1, 2, 4])
def test_batch_with_other_numbers_of_stream_groups(groups):
    """RMD_HIP_BATCH_GROUPS (A/B switch; default three groups, tests/test_batch.py): five different scenes stepped together, one member sitting
    a step out and catching up alone, equal the sequences stepped on their own bit for bit"""
    env = dict(os.environ, RMD_HIP_BATCH_GROUPS=str(groups))
    res = subprocess.run([sys.executable, "-c", GROUPS_CHILD, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0 and "GROUPS-OK" in res.stdout, res.stdout[-2000:]
