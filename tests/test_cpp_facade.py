"""The C++ drop-in surface: include/rmd/*.cuh must compile with a plain host compiler (no GPU toolchain, no CUDA
headers) and behave like the reference classes.  CPU: compile + link + device-independent probe.
GPU: run the reference-style flow through the headers and compare with the oracle bit for bit."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracles as O
from common import sequence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "facade_check.cpp")
PKG = os.path.join(ROOT, "rpg_open_remode_amd")


def _build(tmp_path, side):
    exe = str(tmp_path / f"facade_check_s{side}")
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-Werror", f"-DRMD_CORR_PATCH_SIDE={side}", "-DRMD_BUILD_TESTS=1",
           "-I" + os.path.join(ROOT, "include"), SRC, "-L" + PKG, "-lrmd_hip", "-Wl,-rpath," + PKG, "-o", exe]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    return exe


def test_facade_headers_compile_with_plain_gxx_and_probe(tmp_path):
    exe = _build(tmp_path, 5)
    res = subprocess.run([exe, "--probe"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    assert "patch_side 5" in res.stdout and "caught: CudaException: Image: unable to allocate pitched memory." in res.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("side", [5, 9])
def test_reference_style_cpp_flow_equals_oracle(tmp_path, side):
    exe = _build(tmp_path, side)
    seq = sequence(160, 120, 36)
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("3i", seq.width, seq.height, seq.n_frames))
        f.write(np.asarray(seq.K, np.float32).tobytes())
        f.write(np.asarray([seq.min_depth, seq.max_depth], np.float32).tobytes())
        for im, T in zip(seq.images, seq.T_curr_world):
            f.write(im.tobytes())
            f.write(T.tobytes())
    res = subprocess.run([exe, inp, outp], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    px = seq.width * seq.height
    raw = open(outp, "rb").read()
    planes = [np.frombuffer(raw, np.float32 if i != 4 else np.int32, px, i * px * 4).reshape(seq.height, seq.width) for i in range(6)]
    n_conv, n_conv2 = struct.unpack_from("2Q", raw, 6 * px * 4)
    mu_sum, dist = struct.unpack_from("2f", raw, 6 * px * 4 + 16)
    orc = O.Seeds(O.OracleLib("port", side), seq.width, seq.height, seq.K)
    orc.set_reference(seq.images[0], seq.T_curr_world[0], np.float32(seq.min_depth), np.float32(seq.max_depth))
    for k in range(1, seq.n_frames):
        orc.update(seq.images[k], seq.T_curr_world[k])
    for got, plane in zip(planes[:5], (O.PLANE_MU, O.PLANE_SIGMA_SQ, O.PLANE_A, O.PLANE_B, O.PLANE_CONV)):
        assert O.planes_equal(orc.download(plane), got)
    od = O.Denoiser(orc.o, seq.width, seq.height)
    od.set_large_sigma_sq(np.float32(seq.max_depth) - np.float32(seq.min_depth))
    assert O.planes_equal(od.denoise(orc, 0.5, 30), planes[5])
    assert n_conv == n_conv2 == orc.converged_count()
    assert dist == np.float32(orc.dist_from_ref())
    expect = np.float32(orc.download(O.PLANE_MU).astype(np.float64).sum())
    assert abs(mu_sum - expect) <= 4 * np.spacing(expect)
    (n_points,) = struct.unpack_from("Q", raw, 6 * px * 4 + 24)
    cloud = np.frombuffer(raw, np.float32, n_points * 4, 6 * px * 4 + 32).reshape(-1, 4)
    from rpg_open_remode_amd import api
    want = O.point_cloud(planes[5], planes[4], seq.gray[0], seq.K, api.SE3(seq.T_curr_world[0]).inv().data)
    assert n_points == n_conv > 100 and O.count_mismatch(want, cloud) == 0


def _build_batch_check(tmp_path, side):
    exe = str(tmp_path / f"batch_check_s{side}")
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-Werror", f"-DRMD_CORR_PATCH_SIDE={side}", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "batch_check.cpp"), "-L" + PKG, "-lrmd_hip", "-Wl,-rpath," + PKG, "-o", exe]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    return exe


def test_batch_header_compiles_with_plain_gxx(tmp_path):
    _build_batch_check(tmp_path, 5)


@pytest.mark.gpu
def test_cpp_batch_equals_standalone_objects(tmp_path):
    """rmd::SeedMatrixBatch (include/rmd/seed_matrix_batch.cuh): three sequences stepped together through the C++ headers equal three
    stand-alone rmd::SeedMatrix objects, plane for plane, and feed rmd::DepthmapDenoiser like them"""
    exe = _build_batch_check(tmp_path, 9)
    seqs = [sequence(160, 120, 12, scene) for scene in range(3)]
    inp = str(tmp_path / "in.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("4i", 3, 160, 120, 12))
        f.write(np.asarray(seqs[0].K, np.float32).tobytes())
        f.write(np.asarray([min(s.min_depth for s in seqs), max(s.max_depth for s in seqs)], np.float32).tobytes())
        for s in seqs:
            for im, T in zip(s.images, s.T_curr_world):
                f.write(im.tobytes())
                f.write(T.tobytes())
    res = subprocess.run([exe, inp], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0 and "batch == stand-alone" in res.stdout, res.stdout
