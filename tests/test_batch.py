"""GPU parity of the batched mode (rmd_hip_batch_*, BASELINE configs[3]: independent 640x480 sequences): every member of a batch is
bit-identical, on every state plane and after every checked step, to the same sequence stepped alone -- by the oracle (Oracle B,
the CPU restatement pinned to the reference's own kernels) and by a stand-alone SeedMatrix on the GPU."""
import os

import numpy as np
import pytest

import oracles as O
from common import assert_states_equal, random_state, sequence
from rpg_open_remode_amd import api

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _more_oracle_threads():
    lib = O.OracleLib("port", 9)
    before = lib.fn("max_threads")()
    lib.fn("set_num_threads")(max(1, min(48, os.cpu_count() or 1)))
    yield
    lib.fn("set_num_threads")(before)


def _oracle(seq, side):
    o = O.Seeds(O.OracleLib("port", side), seq.width, seq.height, seq.K)
    o.set_reference(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    return o


def _batch(seqs, side):
    s0 = seqs[0]
    b = api.SeedMatrixBatch(len(seqs), s0.width, s0.height, api.PinholeCamera(*s0.K), patch_side=side)
    for m, seq in zip(b.members, seqs):
        m.setReferenceImage(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    return b


def _device_frames(seq):
    out = []
    for im in seq.images:
        d = api.DeviceImage(seq.width, seq.height, np.float32)
        d.setDevData(im)
        out.append(d)
    return out


def test_config3_eight_scenes_30_updates_each_vs_oracle():
    """configs[3]: the eight 640x480 scenes (seeds 0..7), patch side 9, stepped TOGETHER through 30 updates as one batch of 8; every
    member against Oracle B on every plane at six steps, and its converged count after every step (no device synchronisation)."""
    n_frames = 31
    seqs = [sequence(640, 480, n_frames, scene) for scene in range(8)]
    b = _batch(seqs, 9)
    orcs = [_oracle(seq, 9) for seq in seqs]
    dev = [_device_frames(seq) for seq in seqs]
    for k in range(1, n_frames):
        b.updateDevice([d[k].data for d in dev], [d[k].stride for d in dev], [seq.T_curr_world[k] for seq in seqs])
        for o, seq in zip(orcs, seqs):
            o.update(seq.images[k], seq.T_curr_world[k])
        for i in range(8):
            assert b[i].getConvergedCount() == orcs[i].converged_count(), f"scene {i} update {k}: converged count"
        if k in (1, 2, 3, 10, 20, 30):
            for i in range(8):
                assert_states_equal(orcs[i].state(), b[i].state(), f"batch of 8, scene {i}, update {k}")
    assert sum(b[i].getConvergedCount() for i in range(8)) > 0


@pytest.mark.parametrize("n", [1, 2, 3, 5])
@pytest.mark.parametrize("side", [5, 9])
def test_batch_equals_standalone(n, side):
    """batches of other sizes (1 takes the single-sequence kernels), ragged image size, host frames of both kinds"""
    seqs = [sequence(203, 131, 9, scene) for scene in range(n)]
    b = _batch(seqs, side)
    alone = []
    for seq in seqs:
        s = api.SeedMatrix(seq.width, seq.height, api.PinholeCamera(*seq.K), patch_side=side)
        s.setReferenceImage(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
        alone.append(s)
    for k in range(1, 9):
        poses = [seq.T_curr_world[k] for seq in seqs]
        if k % 2:
            b.updateU8([seq.gray[k] for seq in seqs], poses)
        else:
            b.update([seq.images[k] for seq in seqs], poses)
        for s, seq in zip(alone, seqs):
            s.update(seq.images[k], seq.T_curr_world[k])
        if k in (1, 4, 8):
            for i in range(n):
                assert_states_equal(alone[i].state(), b[i].state(), f"batch of {n}, member {i}, update {k}, side {side}")
                assert b[i].getConvergedCount() == alone[i].getConvergedCount()


def test_members_step_independently():
    """a member may sit a step out (no frame), take a new reference frame or be observed between steps: the others are not disturbed"""
    side = 5
    seqs = [sequence(160, 120, 12, scene) for scene in range(3)]
    b = _batch(seqs, side)
    orcs = [_oracle(seq, side) for seq in seqs]
    rng = np.random.default_rng(7)
    # adversarial start state for member 1 (NaN / Inf / negative planes), through the member's own upload
    st = random_state(160, 120, seqs[1], rng, side)
    for p in range(4):
        b[1].upload(p, st[p])
        orcs[1].upload(p, st[p])
    for k in range(1, 12):
        frames = [seq.images[k] for seq in seqs]
        poses = [seq.T_curr_world[k] for seq in seqs]
        if k in (3, 4):  # member 0 has no frame in these steps
            frames[0] = None
        if k == 6:       # member 2 restarts from a new reference frame (its own frame 5), then continues
            b[2].setReferenceImage(seqs[2].images[5], seqs[2].T_curr_world[5], seqs[2].min_depth, seqs[2].max_depth)
            orcs[2].set_reference(seqs[2].images[5], seqs[2].T_curr_world[5], seqs[2].min_depth, seqs[2].max_depth)
        b.update(frames, poses)
        for i in range(3):
            if frames[i] is not None:
                orcs[i].update(frames[i], poses[i])
        if k in (2, 7):  # an observer between steps forces one member's deferred finalisation only
            assert_states_equal(orcs[1].state(), b[1].state(), f"member 1 observed after step {k}")
    for i in range(3):
        assert_states_equal(orcs[i].state(), b[i].state(), f"member {i} at the end")
    with pytest.raises(api.RmdHipError):
        b[0].update(seqs[0].images[1], seqs[0].T_curr_world[1])  # a member's updates are issued by the batch


def test_batch_with_denoiser_and_point_cloud():
    """a member is a full SeedMatrix: the denoiser and the point cloud read its planes"""
    side = 5
    seqs = [sequence(192, 144, 10, scene) for scene in range(2)]
    b = _batch(seqs, side)
    orcs = [_oracle(seq, side) for seq in seqs]
    for k in range(1, 10):
        b.updateU8([seq.gray[k] for seq in seqs], [seq.T_curr_world[k] for seq in seqs])
        for o, seq in zip(orcs, seqs):
            o.update(seq.images[k], seq.T_curr_world[k])
    for i, seq in enumerate(seqs):
        den = api.DepthmapDenoiser(seq.width, seq.height)
        den.setLargeSigmaSq(seq.max_depth - seq.min_depth)
        got = den.denoise(b[i].getMu(), b[i].getSigmaSq(), b[i].getA(), b[i].getB(), 0.5, 30)
        od = O.Denoiser(orcs[i].o, seq.width, seq.height)
        od.set_large_sigma_sq(seq.max_depth - seq.min_depth)
        assert O.planes_equal(od.denoise(orcs[i], 0.5, 30), got)
        assert len(b[i].pointCloud()) == b[i].getConvergedCount() == orcs[i].converged_count()


@pytest.mark.parametrize("n", [1, 3, 8])
def test_batch_denoise_equals_every_member_denoised_alone(n):
    """rmd_hip_batch_denoise: TV-L1 for all members in one launch sequence (grid z = member) -- per member the oracle's result and the
    stand-alone DepthmapDenoiser's, bit for bit, for iteration counts that are and are not multiples of the blocking depth, members with
    different depth ranges (large_sigma_sq per member), and the device-resident results feed the point cloud"""
    side = 5
    seqs = [sequence(203, 131, 8, scene) for scene in range(n)]
    b = _batch(seqs, side)
    orcs = [_oracle(seq, side) for seq in seqs]
    for k in range(1, 8):
        b.updateU8([seq.gray[k] for seq in seqs], [seq.T_curr_world[k] for seq in seqs])
        for o, seq in zip(orcs, seqs):
            o.update(seq.images[k], seq.T_curr_world[k])
    ranges = [(seq.max_depth - seq.min_depth) * (1.0 + 0.25 * i) for i, seq in enumerate(seqs)]
    for lam, iters in ((0.5, 40), (0.3, 7), (0.5, 1), (0.5, 0)):
        got = b.denoise(ranges, lam, iters)
        ms, launches = b.denoiseTiming()
        assert launches == (iters + 3) // 4
        for i, seq in enumerate(seqs):
            od = O.Denoiser(orcs[i].o, seq.width, seq.height)
            od.set_large_sigma_sq(ranges[i])
            want = od.denoise(orcs[i], lam, iters)
            assert O.planes_equal(want, got[i]), f"batch of {n}, member {i}, lambda {lam}, {iters} iterations: {O.count_mismatch(want, got[i])} pixels differ"
            den = api.DepthmapDenoiser(seq.width, seq.height)
            den.setLargeSigmaSq(ranges[i])
            alone = den.denoise(b[i].getMu(), b[i].getSigmaSq(), b[i].getA(), b[i].getB(), lam, iters)
            assert O.planes_equal(alone, got[i])
            if iters == 40:
                assert O.count_mismatch(b[i].pointCloud(den.result()), b[i].pointCloud(b.denoiseResult(i))) == 0
    assert b.denoise(ranges, 0.5, 5, download=False) is None
    assert O.planes_equal(b.denoiseResult(n - 1).getDevData(), b.denoise(ranges, 0.5, 5)[n - 1])


@pytest.mark.parametrize("n", [11, 16, 24])
def test_batches_beyond_eight_members_run_as_three_groups(n):
    """A launch pair carries at most eight sequences (their parameter blocks are kernel arguments), a batch up to three stream groups of them:
    11, 16 and 24 members (groups of 4+4+3, 6+5+5, 8+8+8), 8-bit host frames and resident frames alternating, a member sitting a step out --
    every member against Oracle B on every plane, and TV-L1 for all of them in one launch sequence against the stand-alone denoiser."""
    side = 5
    seqs = [sequence(171, 113, 9, scene) for scene in range(n)]
    b = _batch(seqs, side)
    orcs = [_oracle(seq, side) for seq in seqs]
    dev = [_device_frames(seq) for seq in seqs]
    for k in range(1, 9):
        skip = k % n if k in (3, 6) else -1  # this member has no frame in this step
        if k % 2:
            b.updateU8([None if i == skip else seq.gray[k] for i, seq in enumerate(seqs)], [seq.T_curr_world[k] for seq in seqs])
        else:
            b.updateDevice([None if i == skip else d[k].data for i, d in enumerate(dev)], [d[k].stride for d in dev], [seq.T_curr_world[k] for seq in seqs])
        for i, (o, seq) in enumerate(zip(orcs, seqs)):
            if i != skip:
                o.update(seq.images[k], seq.T_curr_world[k])
        if k in (1, 4, 8):
            for i in range(n):
                assert_states_equal(orcs[i].state(), b[i].state(), f"batch of {n}, member {i}, step {k}")
                assert b[i].getConvergedCount() == orcs[i].converged_count()
    ranges = [seq.max_depth - seq.min_depth for seq in seqs]
    got = b.denoise(ranges, 0.5, 21)
    for i in (0, n // 2, n - 1):
        den = api.DepthmapDenoiser(seqs[i].width, seqs[i].height)
        den.setLargeSigmaSq(ranges[i])
        assert O.planes_equal(den.denoise(b[i].getMu(), b[i].getSigmaSq(), b[i].getA(), b[i].getB(), 0.5, 21), got[i])
    with pytest.raises(api.RmdHipError):
        api.SeedMatrixBatch(25, 64, 48, api.PinholeCamera(*seqs[0].K), patch_side=side)
