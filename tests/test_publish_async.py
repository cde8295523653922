"""Publication off the update stream (rmd_hip_seeds_publish_async / _collect; DepthmapNode(async_publish=True)): what is published equals
the synchronous calls on the state at the time of the request, bit for bit, whatever the handle goes on to do while the publication is in
flight (depthmap_node.cpp:165-182, publisher.cpp:54-147)."""
import numpy as np
import pytest

import oracles as O
from common import sequence
from rpg_open_remode_amd import api

pytestmark = pytest.mark.gpu
ALL = api.PUBLISH_DEPTH | api.PUBLISH_CLOUD | api.PUBLISH_CONVERGENCE_BGR | api.PUBLISH_CONVERGENCE


def _twin_products(seq, side, n_updates, first, lam, iters):
    """the synchronous products of a handle that took frame `first` as its reference and saw the next n_updates frames"""
    s = api.SeedMatrix(seq.width, seq.height, api.PinholeCamera(*seq.K), patch_side=side)
    s.setReferenceImageU8(seq.gray[first], seq.T_curr_world[first], seq.min_depth, seq.max_depth)
    for k in range(first + 1, first + 1 + n_updates):
        s.updateU8(seq.gray[k], seq.T_curr_world[k])
    den = api.DepthmapDenoiser(seq.width, seq.height)
    den.setLargeSigmaSq(seq.max_depth - seq.min_depth)
    depth = den.denoise(s.getMu(), s.getSigmaSq(), s.getA(), s.getB(), lam, iters)
    return {"depth": depth, "points": s.pointCloud(den.result()), "bgr": s.convergenceBGR8(), "convergence": s.downloadConvergence()}


@pytest.mark.parametrize("size", [(192, 144), (640, 480)])
def test_async_publication_equals_the_synchronous_products_while_the_handle_moves_on(size):
    w, h = size
    seq = sequence(w, h, 72)
    side, lam, iters = 5, 0.5, 60
    s = api.SeedMatrix(w, h, api.PinholeCamera(*seq.K), patch_side=side)
    rng = float(np.float32(seq.max_depth) - np.float32(seq.min_depth))
    want = []
    # three publications in flight at once, each followed at once by a new reference and more updates: the snapshots must hold
    for first, n_up in ((0, 30), (31, 6), (38, 31)):
        s.setReferenceImageU8(seq.gray[first], seq.T_curr_world[first], seq.min_depth, seq.max_depth)
        for k in range(first + 1, first + 1 + n_up):
            s.updateU8(seq.gray[k], seq.T_curr_world[k])
        ticket = s.publishAsync(ALL, rng, lam, iters)
        want.append((ticket, first, n_up))
    with pytest.raises(api.RmdHipError):  # the ring is full
        s.publishAsync(api.PUBLISH_CONVERGENCE_BGR, rng, lam, iters)
    s.setReferenceImageU8(seq.gray[70], seq.T_curr_world[70], seq.min_depth, seq.max_depth)  # the handle moves on before anything is collected
    s.updateU8(seq.gray[71], seq.T_curr_world[71])
    n_points = []
    for ticket, first, n_up in want:
        got = s.collectPublication(wait=True)
        assert got["ticket"] == ticket and got["what"] == ALL
        exp = _twin_products(seq, side, n_up, first, lam, iters)
        assert O.count_mismatch(exp["depth"], got["depth"]) == 0, (first, "depth")
        assert exp["points"].shape == got["points"].shape and O.count_mismatch(exp["points"], got["points"]) == 0, (first, "cloud")
        n_points.append(len(got["points"]))
        assert np.array_equal(exp["bgr"], got["bgr"]), (first, "bgr")
        assert np.array_equal(exp["convergence"], got["convergence"]), (first, "convergence")
    assert max(n_points) > 200, n_points  # (seeds converge after ~20 updates: the long runs publish real clouds)
    with pytest.raises(api.RmdHipError):  # nothing left
        s.collectPublication(wait=True)
    # the handle itself was not disturbed: its state equals a handle that never published
    t = api.SeedMatrix(w, h, api.PinholeCamera(*seq.K), patch_side=side)
    t.setReferenceImageU8(seq.gray[70], seq.T_curr_world[70], seq.min_depth, seq.max_depth)
    t.updateU8(seq.gray[71], seq.T_curr_world[71])
    for p in range(5):
        assert O.count_mismatch(t.download(p), s.download(p)) == 0, p


def test_single_products_and_polling():
    seq = sequence(160, 120, 8)
    s = api.SeedMatrix(160, 120, api.PinholeCamera(*seq.K), patch_side=5)
    with pytest.raises(api.RmdHipError):  # no reference yet
        s.publishAsync(api.PUBLISH_DEPTH, 1.0)
    s.setReferenceImageU8(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    for k in range(1, 6):
        s.updateU8(seq.gray[k], seq.T_curr_world[k])
    with pytest.raises(api.RmdHipError):
        s.publishAsync(0, 1.0)
    with pytest.raises(api.RmdHipError):
        s.publishAsync(64, 1.0)
    t1 = s.publishAsync(api.PUBLISH_CONVERGENCE_BGR)
    t2 = s.publishAsync(api.PUBLISH_CLOUD, float(seq.max_depth - seq.min_depth), 0.5, 20)  # implies the depth map
    got = None
    for _ in range(100000):  # polling: BUSY until the device is done, then the product
        got = s.collectPublication(wait=False)
        if got is not None:
            break
    assert got is not None and got["ticket"] == t1 and got["what"] == api.PUBLISH_CONVERGENCE_BGR and got["depth"] is None and got["points"] is None
    assert np.array_equal(got["bgr"], s.convergenceBGR8())
    got = s.collectPublication(wait=True)
    assert got["ticket"] == t2 and got["what"] == (api.PUBLISH_CLOUD | api.PUBLISH_DEPTH) and got["bgr"] is None
    exp = _twin_products(seq, 5, 5, 0, 0.5, 20)
    assert O.count_mismatch(exp["depth"], got["depth"]) == 0 and O.count_mismatch(exp["points"], got["points"]) == 0


def test_peek_hands_out_views_until_release():
    seq = sequence(160, 120, 36)
    s = api.SeedMatrix(160, 120, api.PinholeCamera(*seq.K), patch_side=5)
    s.setReferenceImageU8(seq.gray[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    for k in range(1, 34):
        s.updateU8(seq.gray[k], seq.T_curr_world[k])
    s.publishAsync(ALL, float(seq.max_depth - seq.min_depth), 0.5, 30)
    s.publishAsync(api.PUBLISH_CONVERGENCE_BGR)
    v = s.peekPublication(wait=True)
    exp = _twin_products(seq, 5, 33, 0, 0.5, 30)
    assert len(v["points"]) == len(exp["points"]) > 100
    for key in ("depth", "points", "bgr", "convergence"):
        assert not v[key].flags.owndata  # views of the library's pinned buffers
        assert np.array_equal(exp[key].view(np.uint8), v[key].view(np.uint8)), key
    again = s.peekPublication(wait=True)  # peeking does not consume
    assert again["ticket"] == v["ticket"]
    s.releasePublication()
    nxt = s.collectPublication(wait=True)
    assert nxt["what"] == api.PUBLISH_CONVERGENCE_BGR and nxt["ticket"] == v["ticket"] + 1
    with pytest.raises(api.RmdHipError):
        s.releasePublication()


def test_node_with_publication_off_the_update_stream_publishes_the_same_things_in_the_same_order():
    """DepthmapNode(async_publish=True) against the default node over the same 70 messages: per topic the same payloads in the same order, none
    earlier than the synchronous node delivers it, everything delivered after flush()."""
    from rpg_open_remode_amd.depthmap_node import DepthmapNode
    seq = sequence(160, 120, 70)

    def run(async_publish):
        events, payload, box = [], {}, {"k": 0}

        def on(topic):
            def cb(x):
                i = sum(1 for e in events if e[1] == topic)
                events.append((box["k"], topic, i))
                payload[(topic, i)] = np.array(x, copy=True)
            return cb
        node = DepthmapNode(160, 120, *seq.K, ref_compl_perc=10.0, max_dist_from_ref=0.5, publish_conv_every_n=10, patch_side=5, denoise_iterations=200,
                            on_depthmap=on("depth"), on_pointcloud=on("cloud"), on_convergence=on("conv"), async_publish=async_publish)
        for k in range(70):
            box["k"] = k
            node.denseInput(seq.gray[k], api.SE3(seq.T_curr_world[k]).inv(), seq.min_depth, seq.max_depth)
        box["k"] = 70
        flushed = node.flush()
        return events, payload, node, flushed

    ev_s, pay_s, node_s, flushed_s = run(False)
    ev_a, pay_a, node_a, flushed_a = run(True)
    assert flushed_s == 0 and not node_a.in_flight_
    assert node_a.references_taken == node_s.references_taken >= 3
    assert sorted(pay_s) == sorted(pay_a) and sum(1 for e in ev_s if e[1] == "cloud") >= 2
    for key in pay_s:
        assert np.array_equal(pay_s[key], pay_a[key]), key
    when_s = {(t, i): k for k, t, i in ev_s}
    when_a = {(t, i): k for k, t, i in ev_a}
    assert all(when_a[key] >= when_s[key] for key in when_s)
    assert any(when_a[key] > when_s[key] for key in when_s)  # it really was deferred somewhere
    # depth and cloud of one publication stay together and in the reference's order (publisher.cpp:106-110)
    order = [(t, i) for _, t, i in ev_a if t != "conv"]
    assert order == [(t, i) for _, t, i in ev_s if t != "conv"]
    # the host mirrors the reference's getters return hold the last publication
    assert np.array_equal(node_a.depthmap_.getDepthmap(), node_s.depthmap_.getDepthmap())
    # (getConvergenceMap(): refreshed by the results publications only in this mode -- the synchronous node also refreshes it for every coloured map)
