"""Throughput in LIVE use: rmd::DepthmapNode's state machine (depthmap_node.py; depthmap_node.cpp:125-173) over a synthetic sequence --
8-bit frames from host memory, a converged-seed count after every update (the node's re-reference test needs it), and at every reference change a
TV-L1 denoise (0.5, 200), the point cloud and -- every `publish_conv_every_n` messages -- the coloured convergence map.  Used by bench.py (the
`live` object of its line) and tools/live_bench.py."""
import time

import numpy as np

from . import api, synth
from .depthmap_node import DepthmapNode, State


def run_live(width=640, height=480, frames=200, side=9, ref_compl_perc=10.0, max_dist_from_ref=0.5, async_publish=False, passes=2, breakdown=False, seq=None):
    """`passes` passes of ONE node over the sequence (the first pays for every buffer the library allocates on first use); the figures of the
    last pass.  async_publish: publication off the update stream (DepthmapNode(async_publish=True)); the pass ends when everything requested
    has been delivered (flush)."""
    if seq is None:
        seq = synth.Sequence(width, height, frames)
    poses = [api.SE3(T).inv() for T in seq.T_curr_world]
    published = {"depth": 0, "points": 0, "conv": 0}

    def on_depth(d): published["depth"] += 1
    def on_pc(p): published["points"] = len(p)
    def on_conv(c): published["conv"] += 1
    node = DepthmapNode(width, height, *seq.K, ref_compl_perc=ref_compl_perc, max_dist_from_ref=max_dist_from_ref, patch_side=side,
                        on_depthmap=on_depth, on_pointcloud=on_pc, on_convergence=on_conv, async_publish=async_publish)
    spent = {}
    if breakdown:
        def timed(obj, name, key):
            f = getattr(obj, name)

            def g(*args, **kw):
                t = time.perf_counter()
                try:
                    return f(*args, **kw)
                finally:
                    spent[key] = spent.get(key, 0.0) + time.perf_counter() - t
            setattr(obj, name, g)
        dm = node.depthmap_
        timed(dm, "setReferenceImage", "setReferenceImage (upload + seed_init)")
        timed(dm, "update", "update (8-bit host frame)")
        timed(dm, "getConvergedPercentage", "getConvergedPercentage (waits for the update's setup kernel)")
        timed(dm, "downloadDenoisedDepthmap", "TV-L1 denoise + download")
        timed(dm, "downloadConvergenceMap", "convergence map download")
        timed(dm, "downloadPointCloud", "point cloud (device compaction + download)")
        timed(dm, "publishAsync", "publishAsync (snapshot + queueing the publication)")
        timed(dm, "collectPublication", "collectPublication (poll / wait + copies out of pinned memory)")
        timed(node.publisher_, "publishConvergenceMap", "coloured convergence map (device kernel + 3 B/pixel download, or delivery of a collected one)")
        timed(node.publisher_, "publishPointCloud", "point cloud publication incl. the growing host-side concatenation")
    dt = 0.0
    for _ in range(max(1, passes)):
        node.state_ = State.TAKE_REFERENCE_FRAME
        node.num_msgs_ = 0
        node.references_taken = node.updates_done = 0
        node.publisher_.pc_ = np.zeros((0, 4), np.float32)
        for key in published:
            published[key] = 0
        spent.clear()
        t0 = time.perf_counter()
        for k in range(frames):
            node.denseInput(seq.gray[k], poses[k], seq.min_depth, seq.max_depth)
        node.flush()
        node.depthmap_.seeds_.sync()
        dt = time.perf_counter() - t0
    return {"value": round(width * height * frames / dt / 1e6, 1), "unit": "Mpix/s", "messages": frames, "ms": round(dt * 1e3, 3), "us_per_message": round(dt / frames * 1e6, 2),
            "publication": "off the update stream (rmd_hip_seeds_publish_async)" if async_publish else "inside the callback, as the reference's blocking std::async (depthmap_node.cpp:165-182)",
            "reference_frames": node.references_taken, "updates": node.updates_done, "depth_maps_and_clouds": published["depth"], "points_accumulated": published["points"],
            "convergence_maps": published["conv"],
            "host_ms_by_call": {k: round(v * 1e3, 3) for k, v in sorted(spent.items(), key=lambda kv: -kv[1])} if breakdown else None}
