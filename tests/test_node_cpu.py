"""Host logic of rmd::DepthmapNode / rmd::Publisher (rpg_open_remode_amd/depthmap_node.py) on the CPU: the node drives an
oracle-backed stand-in for rmd::Depthmap, so the state machine, the publication schedule and the convergence colouring are
covered without a GPU (tests/test_node.py runs the same node on the device and compares the published data bit for bit)."""
import numpy as np

import oracles as O
from common import OracleDepthmap, sequence
from rpg_open_remode_amd import api
from rpg_open_remode_amd.depthmap_node import DepthmapNode, State


def test_node_state_machine_and_publication_schedule():
    seq = sequence(96, 72, 64)
    events = []
    node = DepthmapNode(96, 72, *seq.K, ref_compl_perc=10.0, max_dist_from_ref=0.5, publish_conv_every_n=10, denoise_iterations=10,
                        depthmap=OracleDepthmap(seq, 5), on_depthmap=lambda d: events.append(("depth", d.shape)),
                        on_pointcloud=lambda p: events.append(("cloud", len(p))), on_convergence=lambda c: events.append(("conv", c.copy())))
    assert node.state_ == State.TAKE_REFERENCE_FRAME
    states = []
    for k in range(64):
        T_world_curr = api.SE3(seq.T_curr_world[k]).inv()
        states.append(node.denseInput(seq.gray[k], T_world_curr if k % 2 else T_world_curr.data, seq.min_depth, seq.max_depth))
    assert states[0] == State.UPDATE  # the first message becomes the reference (depthmap_node.cpp:127-139)
    switches = [k for k in range(1, 64) if states[k] == State.TAKE_REFERENCE_FRAME]
    assert len(switches) >= 1 and node.references_taken == len(switches) + (1 if states[-1] == State.UPDATE else 0)
    for k in switches:  # the message after a switch is taken as the next reference, not as an update
        if k + 1 < 64:
            assert states[k + 1] == State.UPDATE
    assert node.updates_done + node.references_taken == 64
    kinds = [e[0] for e in events]
    assert kinds.count("depth") == kinds.count("cloud") == len(switches)  # denoiseAndPublishResults per switch (:150-154,165-173)
    assert kinds.count("conv") == 64 // 11                                # every 11th message (:158-162)
    clouds = [e[1] for e in events if e[0] == "cloud"]
    assert all(b > a for a, b in zip(clouds, clouds[1:]))                 # the cloud accumulates (publisher.cpp:83)
    conv_img = [e[1] for e in events if e[0] == "conv"][-1]
    conv = node.depthmap_.getConvergenceMap()
    gray = node.depthmap_.getReferenceImage()
    assert np.array_equal(conv_img[..., 1], gray)                                          # green channel untouched
    assert (conv_img[..., 0][conv == 1] == 255).all() and np.array_equal(conv_img[..., 0][conv != 1], gray[conv != 1])  # CONVERGED -> blue
    assert (conv_img[..., 2][conv == 3] == 255).all() and np.array_equal(conv_img[..., 2][conv != 3], gray[conv != 3])  # DIVERGED -> red


def test_dist_from_ref_alone_triggers_a_new_reference():
    seq = sequence(96, 72, 12)
    node = DepthmapNode(96, 72, *seq.K, ref_compl_perc=100.0, max_dist_from_ref=0.05, denoise_iterations=2, depthmap=OracleDepthmap(seq, 3))
    states = [node.denseInput(seq.gray[k], api.SE3(seq.T_curr_world[k]).inv(), seq.min_depth, seq.max_depth) for k in range(12)]
    first = states.index(State.TAKE_REFERENCE_FRAME)
    centre = lambda k: np.asarray(seq.T_world_cam[k], np.float64)[:, 3]
    assert np.linalg.norm(centre(first) - centre(0)) > 0.05 >= np.linalg.norm(centre(first - 1) - centre(0)) - 1e-6
