#!/usr/bin/env python3
"""Generates tests/golden/host_steps_golden.npz: inputs and outputs of the host-side steps around the path (point cloud, lens
undistortion) as computed by oracle/host_steps.py.  Unlike remode_golden_s*.npz these vectors do NOT come from the reference
itself (its publisher needs ROS/PCL, the undistortion is OpenCV's): they freeze the numpy restatement, so that the library, the
oracle and later revisions of both are held to one fixed answer.

    python tests/golden/make_golden_host_steps.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracles as O  # noqa: E402

W, H = 64, 48


def main():
    rng = np.random.default_rng(20260925)
    K = np.asarray([57.25, -56.5, 31.5, 23.5], np.float32)
    D = np.asarray([-0.31, 0.12, 1.5e-3, -8e-4], np.float32)
    map1, map2 = O.undistort_maps(W, H, K, D)
    gray = rng.integers(0, 256, (H, W), dtype=np.uint8)
    remapped = O.remap_u8(gray, map1, map2)
    depth = rng.uniform(0.8, 2.5, (H, W)).astype(np.float32)
    conv = rng.integers(0, 5, (H, W)).astype(np.int32)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    from rpg_open_remode_amd import api
    T_world_ref = np.asarray(api.SE3(q[0], q[1], q[2], q[3], 0.3, -1.2, 2.0).data, np.float32)
    cloud = O.point_cloud(depth, conv, gray, K, T_world_ref)
    np.savez_compressed(os.path.join(HERE, "host_steps_golden.npz"), width=W, height=H, K=K, D=D, map1=map1, map2=map2, gray=gray,
                        remapped=remapped, depth=depth, convergence=conv, T_world_ref=T_world_ref, cloud=cloud)
    print("wrote host_steps_golden.npz:", cloud.shape[0], "points,", int((remapped != gray).sum()), "pixels moved by the undistortion")


if __name__ == "__main__":
    main()
