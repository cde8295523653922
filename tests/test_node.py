"""rmd::DepthmapNode / rmd::Publisher as a library against the same state machine driven by the CPU oracle."""
import numpy as np
import pytest

import oracles as O
from common import sequence
from rpg_open_remode_amd import api

pytestmark = pytest.mark.gpu


class OracleNode:
    """depthmap_node.cpp:125-157 + publisher.cpp:54-104 on Oracle B."""

    def __init__(self, seq, side, ref_compl_perc, max_dist_from_ref, lam, iters):
        self.seq, self.side = seq, side
        self.olib = O.OracleLib("port", side)
        self.seeds = O.Seeds(self.olib, seq.width, seq.height, seq.K)
        self.den = O.Denoiser(self.olib, seq.width, seq.height)
        self.taking_reference = True
        self.perc, self.dist, self.lam, self.iters = np.float32(ref_compl_perc), np.float32(max_dist_from_ref), lam, iters
        self.ref_gray, self.T_world_ref = None, None
        self.depthmaps, self.clouds, self.reference_frames = [], [], []
        self.cloud = np.zeros((0, 4), np.float32)

    def dense_input(self, k):
        seq = self.seq
        # the node receives T_world_curr and inverts it (depthmap_node.cpp:130,144): the same fp32 round trip here
        T_curr_world = api.SE3(seq.T_curr_world[k]).inv().inv().data
        if self.taking_reference:
            self.seeds.set_reference(seq.images[k], T_curr_world, seq.min_depth, seq.max_depth)
            self.den.set_large_sigma_sq(seq.max_depth - seq.min_depth)
            self.ref_gray, self.T_world_ref = seq.gray[k], api.SE3(T_curr_world).inv().data
            self.taking_reference = False
            self.reference_frames.append(k)
            return
        self.seeds.update(seq.images[k], T_curr_world)
        perc = np.float32(self.seeds.converged_count()) / np.float32(seq.width * seq.height) * np.float32(100.0)
        if perc > self.perc or np.float32(self.seeds.dist_from_ref()) > self.dist:
            self.taking_reference = True
            depth = self.den.denoise(self.seeds, self.lam, self.iters)
            self.depthmaps.append(depth)
            pts = O.point_cloud(depth, self.seeds.download(4), self.ref_gray, seq.K, self.T_world_ref)
            self.cloud = np.concatenate([self.cloud, pts], axis=0)
            self.clouds.append(self.cloud)


def test_node_cycles_references_like_the_oracle_state_machine():
    from rpg_open_remode_amd.depthmap_node import DepthmapNode, State
    seq = sequence(160, 120, 70)
    got_depth, got_clouds, got_conv = [], [], []
    node = DepthmapNode(160, 120, *seq.K, ref_compl_perc=10.0, max_dist_from_ref=0.5, publish_conv_every_n=10, patch_side=5,
                        denoise_iterations=40, on_depthmap=lambda d: got_depth.append(d.copy()),
                        on_pointcloud=lambda p: got_clouds.append(p.copy()), on_convergence=lambda c: got_conv.append(c.copy()))
    orc = OracleNode(seq, 5, 10.0, 0.5, 0.5, 40)
    states = []
    for k in range(70):
        T_world_curr = api.SE3(seq.T_curr_world[k]).inv()
        states.append(node.denseInput(seq.gray[k], T_world_curr, seq.min_depth, seq.max_depth))
        orc.dense_input(k)
        assert (states[-1] == State.TAKE_REFERENCE_FRAME) == orc.taking_reference, f"frame {k}"
    assert len(orc.reference_frames) >= 3 and node.references_taken == len(orc.reference_frames)
    assert len(got_depth) == len(orc.depthmaps) >= 2
    for a, b in zip(orc.depthmaps, got_depth):
        assert O.count_mismatch(a, b) == 0
    assert len(got_clouds) == len(orc.clouds)
    for a, b in zip(orc.clouds, got_clouds):
        assert a.shape == b.shape and O.count_mismatch(a, b) == 0
    assert len(got_clouds[-1]) > len(got_clouds[0])  # the cloud accumulates over publications (publisher.cpp:83)
    assert len(got_conv) == 70 // 11  # every time num_msgs_ exceeds publish_conv_every_n (depthmap_node.cpp:158-162)
    c = got_conv[0]
    assert c.shape == (120, 160, 3) and c.dtype == np.uint8 and (c[..., 0] == 255).any()
