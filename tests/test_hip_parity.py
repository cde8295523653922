"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs.

Bar: BIT-EXACT on every plane (mu, sigma^2, a, b, convergence, template statistics, epipolar matches) against
  * Oracle B (oracle/remode_oracle.cpp, default build), and, where oracle/_ref travelled,
  * Oracle A "ref_rmd" = the reference's own sources with only expf/sinf/acosf taken from csrc/rmd_math.h.
NaNs compare equal to NaNs; nothing else is tolerated.  Float tolerance appears in exactly one place: the fp32
image sum (reduction_test.cpp:24-70 allows 4 ulp against a double sum; so do we).
"""
import numpy as np
import pytest

import oracles as O
from common import MATCHERS, PLANE_NAMES, apply_matcher, assert_states_equal, random_state, rmse, sequence
from rpg_open_remode_amd import api

pytestmark = pytest.mark.gpu


def _hip_seeds(seq, side, matcher):
    s = api.SeedMatrix(seq.width, seq.height, api.PinholeCamera(*seq.K), patch_side=side)
    apply_matcher(s, matcher)
    return s


def _oracle_seeds(kind, seq, side):
    return O.Seeds(O.OracleLib(kind, side), seq.width, seq.height, seq.K)


def _compare_run(seq, side, matcher, n_updates, oracle_kind="port", state0=None, check_every=1):
    hip = _hip_seeds(seq, side, matcher)
    orc = _oracle_seeds(oracle_kind, seq, side)
    hip.setReferenceImage(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    orc.set_reference(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    assert_states_equal(orc.state(), hip.state(), "after setReferenceImage")
    if state0 is not None:
        for p, arr in enumerate(state0):
            hip.upload(p, arr)
            orc.upload(p, arr)
    for k in range(1, n_updates + 1):
        hip.update(seq.images[k], seq.T_curr_world[k])
        orc.update(seq.images[k], seq.T_curr_world[k])
        if k % check_every == 0 or k == n_updates:
            assert_states_equal(orc.state(), hip.state(), f"side {side} matcher {matcher} update {k}")
    assert hip.getConvergedCount() == orc.converged_count()
    assert hip.getDistFromRef() == orc.dist_from_ref()
    return hip, orc


def test_device_is_gfx950():
    assert api.checkCudaDevice(0)
    import ctypes
    from rpg_open_remode_amd import _lib
    buf = ctypes.create_string_buffer(256)
    _lib.check(_lib.lib().rmd_hip_device_name(0, buf, 256))
    assert b"gfx950" in buf.value, buf.value


@pytest.mark.parametrize("matcher", MATCHERS)
@pytest.mark.parametrize("side", [3, 5, 7, 9])
def test_sequence_bit_exact_vs_port(side, matcher):
    _compare_run(sequence(320, 240, 13), side, matcher, 12)


@pytest.mark.parametrize("matcher", MATCHERS)
def test_sequence_bit_exact_vs_reference_sources(matcher):
    if not O.available("ref_rmd", 5):
        pytest.skip("oracle/_ref not present")
    _compare_run(sequence(160, 120, 9), 5, matcher, 8, oracle_kind="ref_rmd")
    _compare_run(sequence(160, 120, 9), 9, matcher, 8, oracle_kind="ref_rmd")


@pytest.mark.parametrize("matcher", MATCHERS)
@pytest.mark.parametrize("side", [5, 9])
def test_long_sequence_through_convergence(side, matcher):
    hip, orc = _compare_run(sequence(160, 120, 45), side, matcher, 44, check_every=11)
    assert hip.getConvergedCount() > 0.3 * 160 * 120


@pytest.mark.parametrize("matcher", MATCHERS)
@pytest.mark.parametrize("side", [5, 9])
def test_adversarial_seed_states(side, matcher):
    """NaN / Inf / negative / huge state planes, searches at the 100 px cap, depths behind the camera"""
    seq = sequence(192, 144, 4)
    rng = np.random.default_rng(4321 + side)
    st0 = random_state(seq.width, seq.height, seq, rng, side)
    hip, orc = _compare_run(seq, side, matcher, 3, state0=st0)
    conv = hip.downloadConvergence()
    for st in range(5):
        assert (conv == st).any(), f"state {st} never produced"


@pytest.mark.parametrize("matcher", MATCHERS)
@pytest.mark.parametrize("wh", [(101, 67), (64, 48), (257, 33), (40, 200)])
def test_ragged_sizes(wh, matcher):
    """widths/heights that are not multiples of any tile, pitched rows (stride != width)"""
    w, h = wh
    _compare_run(sequence(w, h, 5), 5, matcher, 4)


@pytest.mark.parametrize("matcher", MATCHERS)
def test_tiny_images_are_all_border(matcher):
    for (w, h) in ((10, 10), (11, 30), (18, 18)):
        seq = sequence(w, h, 3)
        hip, _ = _compare_run(seq, 5, matcher, 2)
        conv = hip.downloadConvergence()
        if w <= 10 or h <= 10:
            assert np.all(conv == api.ConvergenceStates.BORDER)


@pytest.mark.parametrize("matcher", MATCHERS)
def test_identity_pose_reference_kat(matcher):
    """test/epipolar_test.cpp:206-220 on the GPU: zero-length segments -> NO_MATCH, the rest match themselves"""
    seq = sequence(320, 240, 2)
    hip = _hip_seeds(seq, 5, matcher)
    orc = _oracle_seeds("port", seq, 5)
    for s in (hip,):
        s.setReferenceImage(seq.images[0], seq.T_curr_world[0], 0.4, 1.8)
        s.update(seq.images[0], seq.T_curr_world[0])
    orc.set_reference(seq.images[0], seq.T_curr_world[0], 0.4, 1.8)
    orc.update(seq.images[0], seq.T_curr_world[0])
    assert_states_equal(orc.state(), hip.state(), "identity pose")
    conv, match = hip.downloadConvergence(), hip.downloadEpipolarMatches()
    upd = conv == api.ConvergenceStates.UPDATE
    yy, xx = np.mgrid[0:seq.height, 0:seq.width]
    assert upd.sum() > 100 and (conv == api.ConvergenceStates.NO_MATCH).sum() > 100
    assert np.max(np.abs(match[..., 0][upd] - xx[upd])) < 0.01 and np.max(np.abs(match[..., 1][upd] - yy[upd])) < 0.01


@pytest.mark.parametrize("matcher", MATCHERS)
def test_search_statistics_equal_the_oracle(matcher):
    """live seeds, search-loop iterations and NCC evaluations per update: same work as the reference does"""
    seq = sequence(192, 144, 8)
    hip, orc = _hip_seeds(seq, 5, matcher), _oracle_seeds("port", seq, 5)
    hip.setOption(api.OPT_COLLECT_STATS, 1)
    hip.setReferenceImage(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    orc.set_reference(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    for k in range(1, 8):
        hip.update(seq.images[k], seq.T_curr_world[k])
        orc.update(seq.images[k], seq.T_curr_world[k])
        assert hip.lastStats() == orc.last_stats(), f"update {k}"


def test_reference_kat_seed_matrix_init_and_check_on_gpu():
    """test/seed_matrix_test.cpp:99-110 and :219-241"""
    seq = sequence(320, 240, 21)
    s = _hip_seeds(seq, 5, 3)
    s.setReferenceImage(seq.images[0], seq.T_curr_world[0], 0.4, 1.8)
    avg = np.float32((np.float32(0.4) + np.float32(1.8)) / np.float32(2.0))
    assert np.all(s.downloadDepthmap() == avg) and np.all(s.downloadA() == 10.0) and np.all(s.downloadB() == 10.0)
    s.update(seq.images[0], seq.T_curr_world[20])
    conv = s.downloadConvergence()
    h, w = conv.shape
    yy, xx = np.mgrid[0:h, 0:w]
    border = (yy > h - 5 - 1) | (yy < 5) | (xx > w - 5 - 1) | (xx < 5)
    assert np.all(conv[border] == api.ConvergenceStates.BORDER)
    assert np.all(np.isin(conv[~border], [0, 1, 3, 4, 5]))


def test_two_instances_are_independent():
    """the reference's global texture references allow ONE live SeedMatrix per process (texture_memory.cuh:27-42)"""
    sa, sb = sequence(160, 120, 5, seed=0), sequence(128, 96, 5, seed=3)
    ha, hb = _hip_seeds(sa, 5, 3), _hip_seeds(sb, 9, 3)
    oa, ob = _oracle_seeds("port", sa, 5), _oracle_seeds("port", sb, 9)
    for h, o, s in ((ha, oa, sa), (hb, ob, sb)):
        h.setReferenceImage(s.images[0], s.T_curr_world[0], s.min_depth, s.max_depth)
        o.set_reference(s.images[0], s.T_curr_world[0], s.min_depth, s.max_depth)
    for k in range(1, 5):  # interleaved
        ha.update(sa.images[k], sa.T_curr_world[k])
        hb.update(sb.images[k], sb.T_curr_world[k])
        oa.update(sa.images[k], sa.T_curr_world[k])
        ob.update(sb.images[k], sb.T_curr_world[k])
    assert_states_equal(oa.state(), ha.state(), "instance A")
    assert_states_equal(ob.state(), hb.state(), "instance B")


def test_deferred_finalisation_is_unobservable():
    """the last kernel of an update is deferred and fused into the next one unless somebody looks at the state first:
    every interleaving of updates and observers must give the same bits as the eager path"""
    seq = sequence(160, 120, 14)
    lazy, eager = _hip_seeds(seq, 5, 3), _hip_seeds(seq, 5, 3)
    eager.setOption(api.OPT_LAZY_FINALIZE, 0)
    d = api.DepthmapDenoiser(seq.width, seq.height)
    d.setLargeSigmaSq(seq.max_depth - seq.min_depth)
    for s in (lazy, eager):
        s.setReferenceImage(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    observers = {3: lambda s: s.getConvergedCount(), 5: lambda s: s.downloadConvergence().sum(), 6: lambda s: s.sync(),
                 8: lambda s: float(d.denoise(s.getMu(), s.getSigmaSq(), s.getA(), s.getB(), 0.5, 3).sum()),
                 9: lambda s: api.ImageReducer().countEqual(s.getConvergence(), 0), 11: lambda s: s.downloadEpipolarMatches().sum()}
    for k in range(1, 14):
        for s in (lazy, eager):
            s.update(seq.images[k], seq.T_curr_world[k])
        if k in observers:
            a, b = observers[k](lazy), observers[k](eager)
            assert a == b or (a != a and b != b), f"observer after update {k}"
        if k == 7:  # re-reference in the middle of a stream, with a finalisation pending
            for s in (lazy, eager):
                s.setReferenceImage(seq.images[7], seq.T_curr_world[7], seq.min_depth, seq.max_depth)
    assert_states_equal(eager.state(), lazy.state(), "lazy vs eager")
    orc = _oracle_seeds("port", seq, 5)
    orc.set_reference(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    for k in range(1, 14):
        orc.update(seq.images[k], seq.T_curr_world[k])
        if k == 7:
            orc.set_reference(seq.images[7], seq.T_curr_world[7], seq.min_depth, seq.max_depth)
    assert_states_equal(orc.state(), lazy.state(), "lazy vs oracle")


def test_device_resident_frames_equal_host_frames():
    seq = sequence(160, 120, 6)
    h1, h2 = _hip_seeds(seq, 5, 3), _hip_seeds(seq, 5, 3)
    frames = []
    for im in seq.images:
        d = api.DeviceImage(seq.width, seq.height, np.float32)
        d.setDevData(im)
        frames.append(d)
    h1.setReferenceImage(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    h2.setReferenceImageDevice(frames[0].data, frames[0].stride, seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    for k in range(1, 6):
        h1.update(seq.images[k], seq.T_curr_world[k])
        h2.updateDevice(frames[k].data, frames[k].stride, seq.T_curr_world[k])
    assert_states_equal(h1.state(), h2.state(), "device-resident input")


def test_u8_ingest_and_depthmap_facade_equal_float_path():
    """8-bit frames through pinned double buffers + device conversion (Depthmap::inputImage on the GPU) == float frames"""
    seq = sequence(101, 67, 9)
    hf, hu = _hip_seeds(seq, 5, 3), _hip_seeds(seq, 5, 3)
    hf.setReferenceImage(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    hu.setReferenceImageU8(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    assert np.array_equal(hu.download(api.PLANE_REF_IMG), seq.images[0])
    for k in range(1, 9):
        hf.update(seq.images[k], seq.T_curr_world[k])
        hu.updateU8(seq.gray[k], seq.T_curr_world[k])  # returns while the frame is still in flight; buffers alternate
    assert np.array_equal(hu.download(api.PLANE_CURR_IMG), seq.images[8])
    assert_states_equal(hf.state(), hu.state(), "u8 ingest")
    # rmd::Depthmap mirror (constructor order fx, cx, fy, cy as depthmap.h:37-43)
    fx, fy, cx, cy = seq.K
    dm = api.Depthmap(seq.width, seq.height, fx, cx, fy, cy)
    dm.setReferenceImage(seq.gray[0], api.SE3(seq.T_curr_world[0]), seq.min_depth, seq.max_depth)
    for k in range(1, 9):
        dm.update(seq.gray[k], api.SE3(seq.T_curr_world[k]))
    dm.downloadDepthmap()
    assert O.planes_equal(dm.getDepthmap(), hf.downloadDepthmap())
    dm.downloadConvergenceMap()
    assert np.array_equal(dm.getConvergenceMap(), hf.downloadConvergence())
    assert dm.getConvergedCount() == hf.getConvergedCount()
    dm.downloadDenoisedDepthmap(0.5, 10)
    assert dm.getDepthmap().shape == (seq.height, seq.width) and np.isfinite(dm.getDepthmap()).all()


def test_error_paths():
    seq = sequence(64, 48, 2)
    s = _hip_seeds(seq, 5, 3)
    with pytest.raises(api.RmdHipError) as e:
        s.update(seq.images[1], seq.T_curr_world[1])  # no reference yet
    assert e.value.code == -3
    d = api.DepthmapDenoiser(seq.width, seq.height)
    s.setReferenceImage(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    with pytest.raises(api.RmdHipError) as e:
        d.denoise(s.getMu(), s.getSigmaSq(), s.getA(), s.getB(), 0.5, 3)  # setLargeSigmaSq not called
    assert e.value.code == -3
    with pytest.raises(api.RmdHipError):
        api.SeedMatrix(64, 48, api.PinholeCamera(*seq.K), patch_side=4)


# ---------------------------------------------------------------------------------------------- denoiser
@pytest.mark.parametrize("wh", [(160, 120), (101, 67), (64, 5), (3, 70)])
@pytest.mark.parametrize("iters_per_launch", [0, 1, 2, 4])
def test_denoiser_bit_exact(wh, iters_per_launch):
    w, h = wh
    seq = sequence(max(w, 24), max(h, 24), 8)
    rng = np.random.default_rng(w * 1000 + h)
    lo, hi = seq.min_depth, seq.max_depth
    mu = rng.uniform(lo, hi, (h, w)).astype(np.float32)
    sig = (10.0 ** rng.uniform(-6, -1, (h, w))).astype(np.float32)
    a = rng.uniform(1, 30, (h, w)).astype(np.float32)
    b = rng.uniform(1, 30, (h, w)).astype(np.float32)
    imgs = []
    for arr in (mu, sig, a, b):
        im = api.DeviceImage(w, h, np.float32)
        im.setDevData(arr)
        imgs.append(im)
    d = api.DepthmapDenoiser(w, h)
    d.setOption(api.DENOISE_OPT_ITERS_PER_LAUNCH, iters_per_launch)
    d.setLargeSigmaSq(hi - lo)
    ol = O.OracleLib("port", 5)
    od = O.Denoiser(ol, w, h)
    od.set_large_sigma_sq(hi - lo)
    assert np.array_equal(d.constants(), od.constants())
    for lam, iters in ((0.5, 1), (0.5, 2), (0.5, 37), (0.2, 200)):
        got = d.denoise(*imgs, lam, iters)
        exp = np.empty((h, w), np.float32)
        rc = ol.lib.orc_denoiser_denoise_planes(od.ptr, mu.ctypes.data, sig.ctypes.data, a.ctypes.data, b.ctypes.data,
                                                exp.ctypes.data, lam, iters)
        assert rc == 0
        assert O.planes_equal(exp, got), f"{w}x{h} lambda {lam} iters {iters}: {O.count_mismatch(exp, got)} pixels differ"


@pytest.mark.parametrize("iters_per_launch", [0, 3])
def test_denoiser_bit_exact_large_image_geometry(iters_per_launch):
    """>= 1 Mpixel selects the 64x16 / 4-iterations-per-launch geometry"""
    w, h = 1290, 821
    rng = np.random.default_rng(77)
    mu = rng.uniform(1.0, 2.0, (h, w)).astype(np.float32)
    sig = (10.0 ** rng.uniform(-6, -1, (h, w))).astype(np.float32)
    a = rng.uniform(1, 30, (h, w)).astype(np.float32)
    b = rng.uniform(1, 30, (h, w)).astype(np.float32)
    imgs = []
    for arr in (mu, sig, a, b):
        im = api.DeviceImage(w, h, np.float32)
        im.setDevData(arr)
        imgs.append(im)
    d = api.DepthmapDenoiser(w, h)
    d.setOption(api.DENOISE_OPT_ITERS_PER_LAUNCH, iters_per_launch)
    d.setLargeSigmaSq(1.0)
    ol = O.OracleLib("port", 5)
    od = O.Denoiser(ol, w, h)
    od.set_large_sigma_sq(1.0)
    for lam, iters in ((0.5, 1), (0.5, 7), (0.3, 26)):
        got = d.denoise(*imgs, lam, iters)
        exp = np.empty((h, w), np.float32)
        assert ol.lib.orc_denoiser_denoise_planes(od.ptr, mu.ctypes.data, sig.ctypes.data, a.ctypes.data, b.ctypes.data,
                                                  exp.ctypes.data, lam, iters) == 0
        assert O.planes_equal(exp, got), f"lambda {lam} iters {iters}: {O.count_mismatch(exp, got)} pixels differ"


def test_denoiser_after_sequence_matches_oracle_and_reference_kernel():
    seq = sequence(160, 120, 25)
    hip, orc = _compare_run(seq, 5, 3, 24, check_every=24)  # the production pipeline (matcher 3)
    d = api.DepthmapDenoiser(seq.width, seq.height)
    d.setLargeSigmaSq(seq.max_depth - seq.min_depth)
    got = d.denoise(hip.getMu(), hip.getSigmaSq(), hip.getA(), hip.getB(), 0.5, 60)
    od = O.Denoiser(orc.o, seq.width, seq.height)
    od.set_large_sigma_sq(seq.max_depth - seq.min_depth)
    assert O.planes_equal(od.denoise(orc, 0.5, 60), got)
    if O.available("ref_rmd", 5):  # the reference's own TV kernel, barriers emulated with fibres
        ref = _oracle_seeds("ref_rmd", seq, 5)
        ref.set_reference(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
        for k in range(1, 25):
            ref.update(seq.images[k], seq.T_curr_world[k])
        rd = O.Denoiser(ref.o, seq.width, seq.height)
        rd.set_large_sigma_sq(seq.max_depth - seq.min_depth)
        assert O.planes_equal(rd.denoise(ref, 0.5, 60), got)


# ---------------------------------------------------------------------------------------------- reductions, images
def test_reducer_raw_pointer_forms_and_int_sum():
    """reduction.cuh:33-47: sum / countEqual on (device pointer, stride, width, height), and ImageReducer<int>::sum
    (instantiated by the reference, reduction.cu:186): equal to the DeviceImage forms and to numpy, on a pitched image"""
    rng = np.random.default_rng(5)
    w, h = 333, 77  # pitched: stride != width
    ints = rng.integers(-5, 6, (h, w)).astype(np.int32)
    flts = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    ii, ff = api.DeviceImage(w, h, np.int32), api.DeviceImage(w, h, np.float32)
    ii.setDevData(ints); ff.setDevData(flts)
    assert ii.stride != w
    red = api.ImageReducer()
    assert red.sum(ii) == int(ints.sum()) == red.sumIntRaw(ii.data, ii.stride, w, h)
    assert red.countEqual(ii.data, 3, ii.stride, w, h) == red.countEqual(ii, 3) == int((ints == 3).sum())
    assert red.sum(ff.data, ff.stride, w, h) == red.sum(ff)
    e = np.float32(flts.astype(np.float64).sum())
    assert abs(red.sum(ff) - e) <= 4 * np.spacing(np.float32(np.abs(flts).astype(np.float64).sum()))
    big = np.full((h, w), 2**30, np.int32)  # the int sum wraps like the reference's int accumulation
    ii.setDevData(big)
    assert red.sum(ii) == int(np.int32((int(big.astype(np.int64).sum()) + 2**31) % 2**32 - 2**31))
    # a sub-rectangle through the raw form: rows 10.., columns 7..
    ii.setDevData(ints)
    assert red.countEqual(ii.data + 4 * (10 * ii.stride + 7), -2, ii.stride, 100, 40) == int((ints[10:50, 7:107] == -2).sum())


def test_reductions_reference_kat():
    """test/reduction_test.cpp:24-122 at its own size (752x480)"""
    rng = np.random.default_rng(11)
    w, h = 752, 480
    img = rng.random((h, w), dtype=np.float32)
    di = api.DeviceImage(w, h, np.float32)
    di.setDevData(img)
    red = api.ImageReducer()
    expect = np.float32(img.astype(np.float64).sum())
    assert abs(red.sum(di) - expect) <= 4 * np.spacing(expect)
    ints = rng.integers(0, 256, (h, w), dtype=np.int32)
    ii = api.DeviceImage(w, h, np.int32)
    ii.setDevData(ints)
    assert red.countEqual(ii, 2) == int((ints == 2).sum())
    for (ww, hh) in ((1, 1), (63, 1), (1, 300), (1000, 3)):
        a = rng.integers(0, 3, (hh, ww), dtype=np.int32)
        im = api.DeviceImage(ww, hh, np.int32)
        im.setDevData(a)
        assert red.countEqual(im, 1) == int((a == 1).sum())
        f = rng.random((hh, ww), dtype=np.float32)
        fm = api.DeviceImage(ww, hh, np.float32)
        fm.setDevData(f)
        e = np.float32(f.astype(np.float64).sum())
        assert abs(red.sum(fm) - e) <= 4 * np.spacing(e)


def test_device_image_round_trips():
    """test/device_image_test.cpp:27-156: upload/download float, float2; device-to-device copy; zero"""
    rng = np.random.default_rng(5)
    for (w, h) in ((640, 480), (101, 7)):
        f = rng.random((h, w), dtype=np.float32)
        a = api.DeviceImage(w, h, np.float32)
        a.setDevData(f)
        assert np.array_equal(a.getDevData(), f)
        f2 = rng.random((h, w, 2), dtype=np.float32)
        b = api.DeviceImage(w, h, "float2")
        b.setDevData(f2)
        assert np.array_equal(b.getDevData(), f2)
        c = api.DeviceImage(w, h, np.float32)
        c.assign(a)
        assert np.array_equal(c.getDevData(), f)
        c.zero()
        assert not c.getDevData().any()
        assert a.pitch % 256 == 0 and a.stride * 4 == a.pitch


# ---------------------------------------------------------------------------------------------- arithmetic contract
def test_device_math_equals_host_math_bit_for_bit():
    import ctypes
    from rpg_open_remode_amd import _lib
    ol = O.OracleLib("port", 5).lib
    rng = np.random.default_rng(2024)
    n = 1 << 18

    def dev(op, x, y=None, z=None):
        out = np.empty_like(x)
        _lib.check(_lib.lib().rmd_hip_math_eval(op, x.ctypes.data, None if y is None else y.ctypes.data,
                                                None if z is None else z.ctypes.data, out.ctypes.data, x.size))
        return out

    def host(op, x, y=None, z=None):
        out = np.empty_like(x)
        ol.orc_math_eval_array(op, x.ctypes.data, None if y is None else y.ctypes.data, None if z is None else z.ctypes.data,
                               out.ctypes.data, x.size)
        return out

    specials = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, np.inf, -np.inf, np.nan, 1e-38, 1e-45, 3.4e38, 88.7, -103.9, -87.3],
                        np.float32)
    cases = {
        0: np.concatenate([rng.uniform(-110, 90, n).astype(np.float32), specials]),
        1: np.concatenate([rng.uniform(-8, 8, n).astype(np.float32), rng.uniform(-1e5, 1e5, 4096).astype(np.float32), specials]),
        2: np.concatenate([rng.uniform(-1.001, 1.001, n).astype(np.float32), specials]),
        3: np.concatenate([(10.0 ** rng.uniform(-40, 38, n)).astype(np.float32), specials]),
    }
    for op, x in cases.items():
        x = np.ascontiguousarray(x)
        got, exp = dev(op, x), host(op, x)
        bad = ~((got == exp) | (np.isnan(got) & np.isnan(exp)))
        assert not bad.any(), f"op {op}: {bad.sum()} mismatches, e.g. x={x[bad][:4]} dev={got[bad][:4]} host={exp[bad][:4]}"
    # IEEE sqrt and divide on the device (the contract relies on them being correctly rounded)
    x = np.concatenate([(10.0 ** rng.uniform(-44, 38, n)).astype(np.float32), specials])
    y = np.concatenate([(10.0 ** rng.uniform(-20, 20, n)).astype(np.float32) * rng.choice([-1, 1], n).astype(np.float32), specials[::-1]])
    with np.errstate(all="ignore"):
        assert O.planes_equal(dev(4, x), np.sqrt(x))
        assert O.planes_equal(dev(5, x, y), x / y)
    t = rng.random(60000).astype(np.float32)
    a, b = rng.random(60000).astype(np.float32), rng.random(60000).astype(np.float32)
    assert O.planes_equal(dev(6, t, a, b), host(6, t, a, b))


# ---------------------------------------------------------------------------------------------- point cloud
def test_point_cloud_equals_the_publisher_loop():
    seq = sequence(160, 120, 41)
    dm = api.Depthmap(160, 120, seq.K[0], seq.K[2], seq.K[1], seq.K[3], patch_side=5)
    with pytest.raises(api.RmdHipError):
        dm.seeds_.pointCloud()  # no reference yet
    dm.setReferenceImage(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    assert dm.seeds_.pointCloud().shape == (0, 4)  # nothing has converged yet
    for k in range(1, 41):
        dm.update(seq.gray[k], seq.T_curr_world[k])
    T_world_ref = dm.getT_world_ref().data
    dm.downloadConvergenceMap()
    conv = dm.getConvergenceMap()
    assert (conv == 1).sum() > 2000
    # raw depth estimate
    dm.downloadDepthmap()
    want = O.point_cloud(dm.getDepthmap(), conv, seq.gray[0], seq.K, T_world_ref)
    got = dm.downloadPointCloud(denoised=False)
    assert got.shape == want.shape and O.count_mismatch(want, got) == 0
    # denoised depth, straight from the denoiser's device buffer
    dm.downloadDenoisedDepthmap(0.5, 30)
    want = O.point_cloud(dm.getDepthmap(), conv, seq.gray[0], seq.K, T_world_ref)
    got = dm.downloadPointCloud()
    assert O.count_mismatch(want, got) == 0
    # every point lies on the scene: back-projected depth along the ray reproduces the analytic range within the filter's accuracy
    centre = np.asarray(seq.T_world_cam[0], np.float64)[:, 3]
    rng = np.linalg.norm(got[:, :3].astype(np.float64) - centre, axis=1)
    assert np.median(np.abs(rng - seq.range0[conv == 1])) < 5e-3
    # a caller buffer smaller than the cloud: the count is still reported, the first points are written
    import ctypes
    from rpg_open_remode_amd import _lib
    small = np.zeros((100, 4), np.float32)
    n = ctypes.c_size_t()
    _lib.check(_lib.lib().rmd_hip_seeds_point_cloud(dm.seeds_.ptr, None, small.ctypes.data, 100, ctypes.byref(n)))
    assert n.value == (conv == 1).sum()
    dm.downloadDepthmap()
    assert O.count_mismatch(O.point_cloud(dm.getDepthmap(), conv, seq.gray[0], seq.K, T_world_ref)[:100], small) == 0


def test_point_cloud_ragged_size_and_intensity_round_trip():
    seq = sequence(101, 67, 30)
    s = api.SeedMatrix(101, 67, api.PinholeCamera(*seq.K), patch_side=3)
    ramp = (np.arange(101 * 67) % 256).astype(np.uint8).reshape(67, 101)  # every 8-bit value as reference intensity
    s.setReferenceImageU8(ramp, seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    rngs = np.random.default_rng(5).uniform(0.5, 3.0, (67, 101)).astype(np.float32)
    s.upload(api.PLANE_MU, rngs)
    s.upload(api.PLANE_SIGMA_SQ, np.full((67, 101), 1e-9, np.float32))
    s.upload(api.PLANE_A, np.full((67, 101), 50.0, np.float32))
    s.upload(api.PLANE_B, np.full((67, 101), 1.0, np.float32))
    s.update(seq.images[1], seq.T_curr_world[1])  # seed_check marks all interior seeds CONVERGED, they keep their depth
    conv = s.downloadConvergence()
    assert (conv == 1).sum() == (101 - 6) * (67 - 6)
    T_world_ref = api.SE3(seq.T_curr_world[0]).inv().data
    want = O.point_cloud(s.downloadDepthmap(), conv, ramp, seq.K, T_world_ref)
    got = s.pointCloud()
    assert O.count_mismatch(want, got) == 0
    assert set(np.unique(got[:, 3]).astype(int)) == set(range(256))


def _publisher_colouring(ref_u8, conv):
    """Publisher::publishConvergenceMap (publisher.cpp:112-147): cv::cvtColor(ref_img, GRAY2BGR), then [0] = 255 for CONVERGED, [2] = 255 for
    DIVERGED.  (The reference's own publisher.cpp, compiled unmodified, is compared with the node in test_reference_host_sources.py.)"""
    out = np.repeat(ref_u8[:, :, None], 3, axis=2)
    out[..., 0][conv == api.ConvergenceStates.CONVERGED] = 255
    out[..., 2][conv == api.ConvergenceStates.DIVERGED] = 255
    return out


@pytest.mark.parametrize("wh", [(101, 67), (203, 131), (640, 480), (33, 19)])
def test_convergence_colouring_on_the_device_equals_the_publisher_loop(wh):
    """rmd_hip_seeds_convergence_bgr8: every state present (adversarial planes: CONVERGED, DIVERGED, UPDATE, NO_MATCH, BORDER), every 8-bit
    level as reference intensity, widths that are not multiples of four (the packed W x 3 rows are not dword-aligned), with the update's
    finalisation still deferred and after an observer has forced it"""
    w, h = wh
    seq = sequence(w, h, 4)
    s = api.SeedMatrix(w, h, api.PinholeCamera(*seq.K), patch_side=3)
    with pytest.raises(api.RmdHipError):
        s.convergenceBGR8()  # no reference yet
    ramp = ((np.arange(w * h) * 7) % 256).astype(np.uint8).reshape(h, w)
    s.setReferenceImageU8(ramp, seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    st = random_state(w, h, seq, np.random.default_rng(99), 3)
    for p in range(4):
        s.upload(p, st[p])
    s.updateU8(seq.gray[1], seq.T_curr_world[1])
    got_deferred = s.convergenceBGR8()  # the finalisation of this update is still pending: UPDATE / NO_MATCH have no colour either way
    conv = s.downloadConvergence()      # (forces it)
    got = s.convergenceBGR8()
    want = _publisher_colouring(ramp, conv)
    assert got.shape == (h, w, 3) and got.dtype == np.uint8
    assert np.array_equal(want, got) and np.array_equal(want, got_deferred)
    if w * h > 2000:
        assert (conv == api.ConvergenceStates.CONVERGED).any() and (conv == api.ConvergenceStates.DIVERGED).any()
    s.updateU8(seq.gray[2], seq.T_curr_world[2])
    assert np.array_equal(_publisher_colouring(ramp, s.downloadConvergence()), s.convergenceBGR8())


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(101, 67, 3), (640, 480, 9), (1280, 960, 9)])
def test_host_frames_arriving_on_an_idle_device(size):
    """Frames handed over in host memory (8-bit and float) while the device is idle: the setup kernel starts before the staging copy
    has finished and its ingest workgroups have to wait for it -- without keeping the copy from running (at 1280x960 every tile
    workgroup waiting filled the device and starved the blit kernel that carries the copy out).  Both paths must give the state of
    the resident-frame path, bit for bit."""
    import time
    w, h, side = size
    seq = sequence(w, h, 6)
    cam = api.PinholeCamera(*seq.K)
    u8, f32, res = (api.SeedMatrix(w, h, cam, patch_side=side) for _ in range(3))
    u8.setReferenceImageU8(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    f32.setReferenceImage(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    res.setReferenceImage(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    planes = []
    for im in seq.images:
        d = api.DeviceImage(w, h, np.float32)
        d.setDevData(im)
        planes.append(d)
    t0 = time.perf_counter()
    for k in range(1, 6):
        time.sleep(0.003)
        u8.updateU8(seq.gray[k], seq.T_curr_world[k])
        f32.update(seq.images[k], seq.T_curr_world[k])
        res.updateDevice(planes[k].data, planes[k].stride, seq.T_curr_world[k])
    want = res.state()
    assert_states_equal(want, u8.state(), "8-bit host frames")
    assert_states_equal(want, f32.state(), "float host frames")
    assert time.perf_counter() - t0 < 5.0  # a wait that ran into its bound would take far longer


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(101, 67, 5), (640, 480, 9)])
def test_float_frames_of_8bit_levels_travel_as_bytes_and_mix_with_other_floats(size):
    """update(float*) examines the frame: 8-bit levels (what the reference's host path produces) are sent as bytes, anything else as floats
    -- NaN, -0.0, values outside [0, 1] and levels that are one ulp off included.  The kinds alternate here without a pause, so frames
    of one kind are converted one step ahead while the other kind's staging buffers and arrival flags are in use; the current image and the
    state must equal those of the resident-frame path bit for bit after every kind of frame."""
    w, h, side = size
    n = 14
    seq = sequence(w, h, n)
    cam = api.PinholeCamera(*seq.K)
    rng = np.random.default_rng(5)
    frames = []
    for k in range(n):
        im = seq.images[k].copy()
        kind = k % 4
        if kind == 1:      # not levels: every pixel scaled
            im = (im * np.float32(0.999)).astype(np.float32)
        elif kind == 2:    # levels but for a handful of awkward pixels
            ys, xs = rng.integers(0, h, 6), rng.integers(0, w, 6)
            im[ys[0], xs[0]] = np.float32(-0.0)
            im[ys[1], xs[1]] = np.nextafter(im[ys[1], xs[1]], np.float32(2.0))
            im[ys[2], xs[2]] = np.float32(1.5)
            im[ys[3], xs[3]] = np.float32(-0.25)
            im[ys[4], xs[4]] = np.float32(np.nan)
        frames.append(np.ascontiguousarray(im))
    import os
    os.environ["RMD_HIP_PACK_BACKOFF"] = "0"  # examine every float frame (read when a handle sees its first host frame)
    try:
        host, res = (api.SeedMatrix(w, h, cam, patch_side=side) for _ in range(2))
        host.setReferenceImage(frames[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    finally:
        del os.environ["RMD_HIP_PACK_BACKOFF"]
    planes = []
    for im in frames:
        d = api.DeviceImage(w, h, np.float32)
        d.setDevData(im)
        planes.append(d)
    res.setReferenceImageDevice(planes[0].data, planes[0].stride, seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    for k in range(1, n):
        host.update(frames[k], seq.T_curr_world[k])
        res.updateDevice(planes[k].data, planes[k].stride, seq.T_curr_world[k])
        if k in (3, 6, 13):  # after a frame of bytes, of awkward floats, of scaled floats
            got = host.download(api.PLANE_CURR_IMG)
            assert np.array_equal(got.view(np.uint32), frames[k].view(np.uint32)), f"current image after frame {k}"
    assert_states_equal(res.state(), host.state(), "float host frames of mixed kinds")


@pytest.mark.parametrize("extent", [150, 178])
@pytest.mark.parametrize("matcher", MATCHERS)
def test_max_extent_equals_the_reference_built_with_that_extent(matcher, extent):
    """RMD_MAX_EXTENT_EPIPOLAR_SEARCH is a compile-time constant of the reference that its CMakeLists leaves to the user (CMakeLists.txt:52-53);
    here it is a constructor argument (1..178).  With the prior variance inflated (3 sigma beyond the depth range on both sides) every search is
    capped: 150 pixels = 215 steps per seed, 178 pixels = 255 steps -- the LIMIT: the step index and the step count fill their 8-bit fields
    (rmd_matcher.hpp), a tile of 256 such seeds is 255 rounds of work, its units fill the unit writer -- against Oracle B and the reference's own
    sources, both built with -DRMD_MAX_EXTENT_EPIPOLAR_SEARCH=<extent>.  A batch of two members takes the same argument."""
    n_steps = {150: 215, 178: 255}[extent]
    full = sequence(320, 240, 41)
    rng = full.max_depth - full.min_depth
    wide = np.full((full.height, full.width), 4.0 * rng * rng, np.float32)
    kinds = [f"port_e{extent}"] + ([f"ref_rmd_e{extent}"] if O.available(f"ref_rmd_e{extent}", 9) else [])
    for kind in kinds:
        hip = api.SeedMatrix(full.width, full.height, api.PinholeCamera(*full.K), patch_side=9, max_extent=extent)
        apply_matcher(hip, matcher)
        orc = _oracle_seeds(kind, full, 9)
        hip.setReferenceImage(full.images[0], full.T_curr_world[0], full.min_depth, full.max_depth)
        orc.set_reference(full.images[0], full.T_curr_world[0], full.min_depth, full.max_depth)
        hip.upload(O.PLANE_SIGMA_SQ, wide)
        orc.upload(O.PLANE_SIGMA_SQ, wide)
        for n, k in enumerate((4, 8, 12, 16), 1):
            hip.update(full.images[k], full.T_curr_world[k])
            orc.update(full.images[k], full.T_curr_world[k])
            assert_states_equal(orc.state(), hip.state(), f"max_extent {extent}, {kind}, matcher {matcher}, update {n}")
            if kind.startswith("port"):
                st = orc.last_stats()
                assert st["steps"] == n_steps * st["live_seeds"] and st["ncc_evals"] > extent * st["live_seeds"], st  # every search capped, most of it in the image
                if n == 1:  # the arg-max really lands on the last step indices somewhere (step numbers up to n_steps - 1 are decoded)
                    assert st["live_seeds"] > 10000
    b = api.SeedMatrixBatch(2, full.width, full.height, api.PinholeCamera(*full.K), patch_side=9, max_extent=extent)
    orcs = [_oracle_seeds(f"port_e{extent}", full, 9) for _ in range(2)]
    for i in range(2):
        b[i].setReferenceImage(full.images[4 * i], full.T_curr_world[4 * i], full.min_depth, full.max_depth)
        orcs[i].set_reference(full.images[4 * i], full.T_curr_world[4 * i], full.min_depth, full.max_depth)
        b[i].upload(O.PLANE_SIGMA_SQ, wide)
        orcs[i].upload(O.PLANE_SIGMA_SQ, wide)
    for k in (8, 12, 16):
        b.update([full.images[k], full.images[k + 4]], [full.T_curr_world[k], full.T_curr_world[k + 4]])
        orcs[0].update(full.images[k], full.T_curr_world[k])
        orcs[1].update(full.images[k + 4], full.T_curr_world[k + 4])
    for i in range(2):
        assert_states_equal(orcs[i].state(), b[i].state(), f"batch member {i}, max_extent {extent}")
    with pytest.raises(api.RmdHipError):
        api.SeedMatrix(64, 48, api.PinholeCamera(*full.K), patch_side=9, max_extent=179)
