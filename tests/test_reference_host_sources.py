"""The reference's HOST sources for the path, compiled UNMODIFIED against this repository's drop-in headers:

  /root/reference/src/depthmap.cpp        rmd::Depthmap (cv::Mat facade over SeedMatrix + DepthmapDenoiser)
  /root/reference/test/dataset.cpp        rmd::test::Dataset
  /root/reference/test/dataset_main.cpp   the stand-alone experiment (BASELINE configs[0])

Third-party headers (Eigen, Boost.Filesystem, OpenCV) are replaced by the test-only stand-ins of tests/cpp/stubs; include/rmd/
comes from this repository (the reference's include/ is on the path only for depthmap.h).  rpg_open_remode_amd.build compiles
them into oracle/_ref/ wherever /root/reference exists; the GPU box runs those prebuilt programs (no /root/reference there).
CPU: the sources compile and link; without a GPU the experiment stops at checkCudaDevice like the reference.
GPU: what the programs produce equals the CPU oracle on the same frames, bit for bit."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracles as O
from common import sequence
from rpg_open_remode_amd import build as B
from rpg_open_remode_amd import dataset as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
HAVE_REFERENCE = os.path.isdir(os.path.join(B.REFERENCE, "src"))
K_DATASET_MAIN = (481.2, -480.0, 319.5, 239.5)  # dataset_main.cpp:37


def _program(name):
    path = os.path.join(REF_DIR, name)
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (python -m rpg_open_remode_amd.build, where /root/reference exists)")
    return path


def scale_and_colour(depth):
    """Depthmap::scaleMat (depthmap.cpp:158-170) with the stub's element arithmetic: float32 (d - min) * 1 / (max - min),
    x 255 rounded half to even and saturated, grey replicated into three channels."""
    d = np.asarray(depth, np.float32)
    lo, hi = float(d.min()), float(d.max())
    s = (d - np.float32(lo)) * np.float32(1.0) / np.float32(hi - lo)
    g = np.clip(np.rint(s.astype(np.float64) * 255.0), 0, 255).astype(np.uint8)
    return np.repeat(g[..., None], 3, axis=2)


def read_imshow(path):
    raw = open(path, "rb").read()
    rows, cols, typ = struct.unpack_from("3i", raw)
    ch = (typ >> 3) + 1
    assert typ & 7 == 0
    return np.frombuffer(raw, np.uint8, rows * cols * ch, 12).reshape(rows, cols, ch)


@pytest.mark.skipif(not HAVE_REFERENCE, reason="/root/reference not present")
def test_reference_host_sources_compile_unmodified_on_the_drop_in_headers(tmp_path):
    for name, cmd in B.reference_host_program_cmds(str(tmp_path)).items():
        assert [a for a in cmd if a.startswith(B.REFERENCE) and a.endswith(".cpp")], "the sources are the reference's own files"
        res = subprocess.run(cmd + ["-Wall", "-Wno-reorder", "-Wno-range-loop-construct"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert res.returncode == 0, res.stdout
        assert os.path.exists(tmp_path / name)
    # the header set the reference sources see: ours for everything the library replaces
    for hdr in ("seed_matrix.cuh", "depthmap_denoiser.cuh", "se3.cuh", "pinhole_camera.cuh", "check_cuda_device.cuh", "device_image.cuh"):
        assert os.path.exists(os.path.join(ROOT, "include", "rmd", hdr)), hdr
    assert not os.path.exists(os.path.join(ROOT, "include", "rmd", "depthmap.h"))  # host class: stays the reference's file


@pytest.mark.gpu
def test_reference_dataset_main_unmodified_equals_the_oracle(tmp_path):
    """test/dataset_main.cpp end to end on a 640x480 sequence in the reference's dataset layout (12 frames, ground-truth .depth
    for every frame, as the reference reads one per frame): the two images it shows -- scaleMat of the raw and of the denoised
    depth map (lambda 0.5, 200 iterations) -- against the oracle run on the same files."""
    exe = _program("dataset_main_ref")
    n = 12
    root, shown = str(tmp_path / "over_table"), str(tmp_path / "shown")
    os.makedirs(shown)
    D.export_synthetic(root, 640, 480, n, image_ext="pgm", depth_every=1)
    env = dict(os.environ, RMD_TEST_DATA_PATH=root, RMD_STUB_IMSHOW_DIR=shown)
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    assert res.stdout.count("UPDATE execution time") == n - 1 and "RUN EXPERIMENT: inputting image scene_000.pgm" in res.stdout
    assert "MEAN update time" in res.stdout and "STDDEV" in res.stdout and "ERROR" not in res.stdout
    ds = D.Dataset(root)
    assert ds.readDataSequence(0, 200) and len(list(ds)) == n
    orc = O.Seeds(O.OracleLib("port", 5), 640, 480, K_DATASET_MAIN)
    scale = np.float32(1.0 / 255.0)
    for k, e in enumerate(ds):
        img = ds.readImage(e).astype(np.float32) * scale  # Depthmap::inputImage, depthmap.cpp:105
        T = np.asarray(ds.readCameraPose(e).inv().data, np.float32)
        if k == 0:
            gt = ds.readDepthmap(e, 640, 480)
            orc.set_reference(img, T, np.float32(gt.min()), np.float32(gt.max()))
            depth_range = np.float32(gt.max()) - np.float32(gt.min())
        else:
            orc.update(img, T)
    raw = orc.download(O.PLANE_MU)
    assert np.array_equal(read_imshow(os.path.join(shown, "result.bin")), scale_and_colour(raw))
    od = O.Denoiser(orc.o, 640, 480)
    od.set_large_sigma_sq(depth_range)
    den = od.denoise(orc, 0.5, 200)
    assert np.array_equal(read_imshow(os.path.join(shown, "denoised_result.bin")), scale_and_colour(den))
    assert (scale_and_colour(den) != scale_and_colour(raw)).any()


@pytest.mark.gpu
@pytest.mark.parametrize("distortion", [None, (-0.25, 0.08, 1e-3, -7e-4)])
def test_reference_depthmap_class_unmodified_equals_the_oracle(tmp_path, distortion):
    """src/depthmap.cpp's rmd::Depthmap driven by tests/cpp/depthmap_check.cpp: every getter, raw values."""
    exe = _program("depthmap_check_ref")
    seq = sequence(160, 120, 36)
    w, h, n, iters, lam = seq.width, seq.height, seq.n_frames, 40, np.float32(0.4)
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("5i", w, h, n, int(distortion is not None), iters))
        f.write(np.asarray(seq.K, np.float32).tobytes())
        f.write(np.asarray(distortion or (0, 0, 0, 0), np.float32).tobytes())
        f.write(np.asarray([seq.min_depth, seq.max_depth], np.float32).tobytes())
        f.write(lam.tobytes())
        for g, T in zip(seq.gray, seq.T_curr_world):
            f.write(np.ascontiguousarray(g, np.uint8).tobytes())
            f.write(np.asarray(T, np.float32).tobytes())
    res = subprocess.run([exe, inp, outp], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    raw = open(outp, "rb").read()
    px = w * h
    depth = np.frombuffer(raw, np.float32, px, 0).reshape(h, w)
    conv = np.frombuffer(raw, np.int32, px, 4 * px).reshape(h, w)
    den = np.frombuffer(raw, np.float32, px, 8 * px).reshape(h, w)
    ref_img = np.frombuffer(raw, np.uint8, px, 12 * px).reshape(h, w)
    (count,) = struct.unpack_from("Q", raw, 13 * px)
    pct, dist = struct.unpack_from("2f", raw, 13 * px + 8)
    coloured = np.frombuffer(raw, np.uint8, 3 * px, 13 * px + 16).reshape(h, w, 3)
    gray = seq.gray
    if distortion is not None:
        m1, m2 = O.undistort_maps(w, h, seq.K, distortion)
        gray = [O.remap_u8(g, m1, m2) for g in seq.gray]
        assert any((u != g).any() for u, g in zip(gray, seq.gray))
    orc = O.Seeds(O.OracleLib("port", 5), w, h, seq.K)
    scale = np.float32(1.0 / 255.0)
    orc.set_reference(gray[0].astype(np.float32) * scale, seq.T_curr_world[0], np.float32(seq.min_depth), np.float32(seq.max_depth))
    for k in range(1, n):
        orc.update(gray[k].astype(np.float32) * scale, seq.T_curr_world[k])
    assert np.array_equal(ref_img, gray[0])
    assert O.planes_equal(orc.download(O.PLANE_MU), depth)
    assert O.planes_equal(orc.download(O.PLANE_CONV), conv)
    od = O.Denoiser(orc.o, w, h)
    od.set_large_sigma_sq(np.float32(seq.max_depth) - np.float32(seq.min_depth))
    assert O.planes_equal(od.denoise(orc, lam, iters), den)
    assert count == orc.converged_count() > 100
    assert pct == np.float32(np.float32(count) / np.float32(px) * np.float32(100.0))
    assert dist == np.float32(orc.dist_from_ref())
    assert np.array_equal(coloured, scale_and_colour(depth))


def _write_bag(path, seq, quats):
    with open(path, "wb") as f:
        f.write(struct.pack("3i", seq.n_frames, seq.width, seq.height))
        for k in range(seq.n_frames):
            f.write(np.asarray(quats[k], np.float64).tobytes())  # position xyz, orientation wxyz
            f.write(np.asarray([seq.min_depth, seq.max_depth], np.float32).tobytes())
            f.write(np.ascontiguousarray(seq.gray[k], np.uint8).tobytes())


@pytest.mark.gpu
def test_reference_ros_node_unmodified_equals_the_oracle_driven_state_machine(tmp_path):
    """src/main_ros.cpp + depthmap_node.cpp + publisher.cpp + depthmap.cpp, unmodified, on file-backed stand-ins for ROS: fed 70
    DenseInput messages, everything the node publishes (remode/depth, remode/pointcloud, remode/convergence) and WHEN it publishes it
    must equal the Python restatement of the node (rpg_open_remode_amd/depthmap_node.py) driving the CPU oracle, whose point cloud is
    oracle/host_steps.py::point_cloud -- this pins both restatements to the reference's own code -- and the
    library's own node (device point cloud kernel) on the same messages."""
    from common import OracleDepthmap
    from rpg_open_remode_amd import api
    from rpg_open_remode_amd.depthmap_node import DepthmapNode
    exe = _program("remode_node_ref")
    seq = sequence(160, 120, 70)
    w, h, n = seq.width, seq.height, seq.n_frames
    poses7 = []  # the message carries position + quaternion (float64); the node builds SE3<float>(qw, qx, qy, qz, tx, ty, tz) from them
    for k in range(n):
        T_world_curr = api.SE3(seq.T_curr_world[k]).inv().data.reshape(3, 4).astype(np.float64)
        q = D.quaternion_from_rotation(T_world_curr[:, :3])  # x, y, z, w
        poses7.append([T_world_curr[0, 3], T_world_curr[1, 3], T_world_curr[2, 3], q[3], q[0], q[1], q[2]])
    bag, params, topics = str(tmp_path / "bag.bin"), str(tmp_path / "params.txt"), str(tmp_path / "topics")
    os.makedirs(topics)
    _write_bag(bag, seq, poses7)
    with open(params, "w") as f:
        for name, v in (("cam_width", w), ("cam_height", h), ("cam_fx", seq.K[0]), ("cam_fy", seq.K[1]), ("cam_cx", seq.K[2]), ("cam_cy", seq.K[3]),
                        ("ref_compl_perc", 10.0), ("max_dist_from_ref", 0.5), ("publish_conv_every_n", 10)):
            f.write(f"remode/{name} {v}\n" if isinstance(v, int) else f"remode/{name} {float(np.float32(v))!r}\n")
    env = dict(os.environ, RMD_STUB_BAG=bag, RMD_STUB_PARAMS=params, RMD_STUB_TOPIC_DIR=topics)
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    assert res.stdout.count("DEPTHMAP NODE: received image 160x120") == n and "ROS_ERROR" not in res.stdout
    got_events = [tuple(l.split()) for l in open(os.path.join(topics, "events.txt")).read().splitlines()]

    def run_python_node(depthmap):
        events, payload = [], {}
        box = {"k": 0}
        def on(topic):
            def cb(x):
                i = sum(1 for e in events if e[1] == topic)
                events.append((str(box["k"]), topic, str(i)))
                payload[(topic, i)] = np.array(x, copy=True)
            return cb
        node = DepthmapNode(w, h, *seq.K, ref_compl_perc=10.0, max_dist_from_ref=0.5, publish_conv_every_n=10, patch_side=5, denoise_lambda=0.5,
                            denoise_iterations=200, depthmap=depthmap, on_depthmap=on("remode_depth"), on_pointcloud=on("remode_pointcloud"),
                            on_convergence=on("remode_convergence"))
        for k in range(n):
            box["k"] = k + 1  # messages delivered when the callback runs
            p = poses7[k]
            node.denseInput(seq.gray[k], (p[3], p[4], p[5], p[6], p[0], p[1], p[2]), np.float32(seq.min_depth), np.float32(seq.max_depth))
        return events, payload, node

    want_events, want, onode = run_python_node(OracleDepthmap(seq, 5))
    assert got_events == want_events, (got_events, want_events)
    assert onode.references_taken >= 3 and sum(1 for e in want_events if e[1] == "remode_pointcloud") >= 2

    def read_topic(topic, i):
        raw = open(os.path.join(topics, f"{topic}.{i}.bin"), "rb").read()
        if topic == "remode_pointcloud":
            (cnt,) = struct.unpack_from("i", raw)
            return np.frombuffer(raw, np.float32, cnt * 4, 4).reshape(cnt, 4)
        rows, cols, eb = struct.unpack_from("3i", raw)
        if topic == "remode_depth":
            assert eb == 4
            return np.frombuffer(raw, np.float32, rows * cols, 12).reshape(rows, cols)
        assert eb == 3
        return np.frombuffer(raw, np.uint8, rows * cols * 3, 12).reshape(rows, cols, 3)

    for (_, topic, i) in want_events:
        got, exp = read_topic(topic, int(i)), want[(topic, int(i))]
        assert got.shape == exp.shape, (topic, i)
        assert O.count_mismatch(exp, got) == 0 if got.dtype == np.float32 else np.array_equal(exp, got), (topic, i)
    clouds = [want[k] for k in sorted(want) if k[0] == "remode_pointcloud"]
    assert len(clouds[-1]) > len(clouds[0]) > 100  # the reference's cloud accumulates over publications (publisher.cpp:83)

    # the library's own node (api.Depthmap, device point-cloud kernel) on the same messages
    hip_events, hip, _ = run_python_node(None)
    assert hip_events == got_events
    for (_, topic, i) in got_events:
        got, mine = read_topic(topic, int(i)), hip[(topic, int(i))]
        assert O.count_mismatch(got, mine) == 0 if got.dtype == np.float32 else np.array_equal(got, mine), (topic, i)


@pytest.mark.gpu
def test_reference_gtest_suite_unmodified_passes_on_the_library(tmp_path):
    """test/seed_matrix_test.cpp, epipolar_test.cpp, reduction_test.cpp with test/main_test.cpp -- the reference's own tests of the
    path, compiled unmodified with -DRMD_BUILD_TESTS like its CMakeLists.txt:47 does -- run on a 640x480 sequence
    in the dataset layout (they read entries 1, 20 and 199 and the ground-truth depth of entry 1): every one of them must pass."""
    exe = _program("rmd_gtests_ref")
    root = str(tmp_path / "over_table")
    D.export_synthetic(root, 640, 480, 200, image_ext="pgm", only_frames={1, 20, 199})
    env = dict(os.environ, RMD_TEST_DATA_PATH=root)
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    tail = res.stdout[-4000:]
    assert res.returncode == 0, tail
    for name in ("RMDCuTests.seedMatrixInit", "RMDCuTests.seedMatrixCheck", "RMDCuTests.epipolarTest", "RMDCuTests.epipolarMatchTest",
                 "deviceImageReduction.sum", "deviceImageReduction.countEqual"):
        assert f"[       OK ] {name}" in res.stdout, tail
    assert "[  PASSED  ] 6 tests." in res.stdout and "FAILED" not in res.stdout and "Failure" not in res.stdout, tail


@pytest.mark.skipif(not HAVE_REFERENCE, reason="/root/reference not present")
def test_reference_dataset_reader_equals_dataset_py(tmp_path):
    """rmd::test::Dataset (test/dataset.cpp, unmodified) on a directory written by rpg_open_remode_amd/dataset.py: sequence file, file
    names, poses (the fp32 quaternion constructor of se3.cuh), 8-bit images and the centimetre .depth files -- no device involved, so this
    pins the Python reader / writer to the reference's own reader on the CPU."""
    exe = str(tmp_path / "dataset_check_ref")
    res = subprocess.run(B.reference_host_program_cmds(str(tmp_path))["dataset_check_ref"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    root = str(tmp_path / "over_table")
    w, h, n = 96, 72, 7
    D.export_synthetic(root, w, h, n, image_ext="pgm", depth_every=3)
    res = subprocess.run([exe, root, D.DEFAULT_SEQUENCE_FILE, str(w), str(h)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    lines = res.stdout.strip().splitlines()
    ds = D.Dataset(root)
    assert ds.readDataSequence(0, 0)
    entries = list(ds)
    assert len(lines) == len(entries) == n
    with_depth = 0
    for line, e in zip(lines, entries):
        tok = line.split()
        assert tok[0] == e.getImageFileName() and tok[1] == e.getDepthmapFileName()
        pose = np.array([int(t, 16) for t in tok[2:14]], np.uint32).view(np.float32)
        assert np.array_equal(pose, np.asarray(ds.readCameraPose(e).data, np.float32))
        img = ds.readImage(e)
        assert tok[14:18] == ["img", str(w), str(h), str(int(img.astype(np.uint64).sum()))]
        path = os.path.join(root, "depthmaps", e.getDepthmapFileName())
        if os.path.exists(path):
            dm = ds.readDepthmap(e, w, h)
            assert tok[18] == "depth" and int(tok[19]) == w * h
            assert float(tok[20]) == float(np.cumsum(dm.astype(np.float64).ravel())[-1])  # sequential, like the driver's loop
            assert int(tok[21], 16) == int(dm.reshape(-1)[:1].view(np.uint32)[0]) and int(tok[22], 16) == int(dm.reshape(-1)[-1:].view(np.uint32)[0])
            with_depth += 1
        else:
            assert tok[18:] == ["depth", "-"]
    assert with_depth == 3  # frames 0, 3, 6
