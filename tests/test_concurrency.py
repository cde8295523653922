"""Handles are independent (SURVEY 8(b) "Threading": one handle is used from one thread at a time; distinct handles share nothing -- no texture
references, no __constant__ symbols, no default stream -- unlike the reference, whose globals allow ONE live SeedMatrix per process,
texture_memory.cuh:27-42).  Four host threads drive four different handles at the same time -- two single sequences with 8-bit host frames,
one batch of three with 8-bit host frames, one denoiser -- at 640x480 over 60 frames, and every one of them must produce, bit for bit, what
it produces when it runs alone.  The same with the handles created on device 1 where the box has two devices (one host thread
per device)."""
import threading

import numpy as np
import pytest

import oracles as O
from common import assert_states_equal
from rpg_open_remode_amd import api, synth

pytestmark = pytest.mark.gpu

W, H, F = 640, 480, 61
TWO_DEVICES_REASON = "needs two HIP devices (the measurement lease has one)"


def _device_count():
    import ctypes
    from rpg_open_remode_amd import _lib
    n = ctypes.c_int(0)
    _lib.check(_lib.lib().rmd_hip_device_count(ctypes.byref(n)))
    return n.value


def _device_of(ptr):
    """the device a device pointer lives on (hipPointerGetAttributes, straight from the HIP runtime: test infrastructure)"""
    import ctypes

    class Attr(ctypes.Structure):  # hipPointerAttribute_t
        _fields_ = [("type", ctypes.c_int), ("device", ctypes.c_int), ("devicePointer", ctypes.c_void_p), ("hostPointer", ctypes.c_void_p),
                    ("isManaged", ctypes.c_int), ("allocationFlags", ctypes.c_uint)]
    hip = ctypes.CDLL("libamdhip64.so")
    a = Attr()
    rc = hip.hipPointerGetAttributes(ctypes.byref(a), ctypes.c_void_p(ptr))
    assert rc == 0, f"hipPointerGetAttributes: {rc}"
    return a.device


def _run_all(device, concurrent, seqs):
    import ctypes
    from rpg_open_remode_amd import _lib
    cam = api.PinholeCamera(*seqs[0].K)
    results, errors = {}, []

    def on_device(fn):
        def wrapped():
            try:
                _lib.check(_lib.lib().rmd_hip_set_device(device))  # the current device is per host thread: handles bind to it at creation
                fn()
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))
        return wrapped

    def single(scene):
        def work():
            seq = seqs[scene]
            s = api.SeedMatrix(W, H, cam, patch_side=9)
            s.setReferenceImageU8(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
            counts = []
            for k in range(1, F):
                s.updateU8(seq.gray[k], seq.T_curr_world[k])
                if k % 20 == 0:
                    counts.append(s.getConvergedCount())
            results[("single", scene)] = (s.state(), counts)
            results[("placed", "single", scene)] = [_device_of(s.getMu().data), _device_of(s._plane(api.PLANE_REF_IMG).data)]
        return work

    def batch3():
        scenes = [2, 3, 4]
        b = api.SeedMatrixBatch(3, W, H, cam, patch_side=9)
        for i, sc in enumerate(scenes):
            b[i].setReferenceImageU8(seqs[sc].gray[0], seqs[sc].T_curr_world[0], seqs[sc].min_depth, seqs[sc].max_depth)
        for k in range(1, F):
            b.updateU8([seqs[sc].gray[k] for sc in scenes], [seqs[sc].T_curr_world[k] for sc in scenes])
        results[("batch",)] = [b[i].state() for i in range(3)]
        results[("placed", "batch")] = [_device_of(b[i].getMu().data) for i in range(3)]

    def denoiser():
        seq = seqs[5]
        s = api.SeedMatrix(W, H, cam, patch_side=9)
        s.setReferenceImage(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
        for k in range(1, 25):
            s.update(seq.images[k], seq.T_curr_world[k])
        d = api.DepthmapDenoiser(W, H)
        d.setLargeSigmaSq(seq.max_depth - seq.min_depth)
        outs = [d.denoise(s.getMu(), s.getSigmaSq(), s.getA(), s.getB(), 0.5, 200) for _ in range(6)]
        results[("denoise",)] = outs
        results[("placed", "denoise")] = [_device_of(d.result().data), _device_of(s.getSigmaSq().data)]

    jobs = [on_device(single(0)), on_device(single(1)), on_device(batch3), on_device(denoiser)]
    if concurrent:
        threads = [threading.Thread(target=j) for j in jobs]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    else:
        for j in jobs:
            j()
    assert not errors, errors
    return results


def _check(device):
    seqs = {s: synth.Sequence(W, H, F, s) for s in range(6)}  # (rendered once: 6 x 61 frames)
    alone = _run_all(device, False, seqs)
    together = _run_all(device, True, seqs)
    # placement, not only equality: every handle created by a thread whose current device was `device` keeps its planes THERE (the library's
    # handles bind to the creating thread's device, check_cuda_device.cu:109 / rmd_hip_set_device)
    for run in (alone, together):
        placed = {k: v for k, v in run.items() if k[0] == "placed"}
        assert len(placed) == 4 and all(d == device for v in placed.values() for d in v), (device, placed)
    for scene in (0, 1):
        assert_states_equal(alone[("single", scene)][0], together[("single", scene)][0], f"single sequence, scene {scene}: alone vs among three other handles")
        assert alone[("single", scene)][1] == together[("single", scene)][1]
    for i in range(3):
        assert_states_equal(alone[("batch",)][i], together[("batch",)][i], f"batch member {i}: alone vs among three other handles")
    for a, b in zip(alone[("denoise",)], together[("denoise",)]):
        assert O.planes_equal(a, b)
    assert all(O.planes_equal(alone[("denoise",)][0], x) for x in alone[("denoise",)][1:])
    # and the stand-alone results are the oracle's (scene 0, every plane)
    seq = seqs[0]
    orc = O.Seeds(O.OracleLib("port", 9), W, H, seq.K)
    orc.set_reference(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    for k in range(1, F):
        orc.update(seq.images[k], seq.T_curr_world[k])
    assert_states_equal(orc.state(), together[("single", 0)][0], "scene 0 among three other handles vs Oracle B")


def test_four_handles_on_four_host_threads_equal_their_stand_alone_results():
    _check(0)


def test_the_same_with_handles_created_on_a_second_device():
    if _device_count() < 2:
        pytest.skip(TWO_DEVICES_REASON)
    _check(1)
