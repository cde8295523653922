"""Dataset layout of the reference (test/dataset.cpp) and the dataset_main experiment (BASELINE configs[0]).

CPU: codecs, sequence-file / pose / depth-map round trips, and configs[0] itself -- the first 30 frames of the 640x480
trajectory read back from disk and run through the CPU oracle (plumbing, no GPU).
GPU: the dataset_main program end to end on an exported sequence, against the direct API on the same frames.
"""
import os
import struct
import zlib

import numpy as np
import pytest

import oracles as O
from rpg_open_remode_amd import dataset as D
from rpg_open_remode_amd import synth


def test_pgm_and_png_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    for name in ("a.pgm", "a.png", "A.PNG"):
        p = str(tmp_path / name)
        D.write_gray_image(p, img)
        assert np.array_equal(D.read_gray_image(p), img)
    # the pure-Python PNG decoder as well (Pillow, when installed, is only a faster inflate + unfilter)
    assert np.array_equal(D._read_png(open(str(tmp_path / "a.png"), "rb").read()), img)
    # ASCII PGM with comments, 16-bit binary PGM
    with open(str(tmp_path / "b.pgm"), "w") as f:
        f.write("P2\n# a comment\n3 2\n255\n0 1 2\n253 254 255\n")
    assert np.array_equal(D.read_gray_image(str(tmp_path / "b.pgm")), np.array([[0, 1, 2], [253, 254, 255]], np.uint8))
    with open(str(tmp_path / "c.pgm"), "wb") as f:
        f.write(b"P5\n2 1\n65535\n" + struct.pack(">HH", 0x1234, 0xff00))
    assert np.array_equal(D.read_gray_image(str(tmp_path / "c.pgm")), np.array([[0x12, 0xff]], np.uint8))


def _png_bytes(pixels, color_type, filters, palette=None):
    """Hand-made PNG with a chosen filter type per row (exercises Sub / Up / Average / Paeth of the decoder)."""
    h, w, ch = pixels.shape
    bpp = ch
    rows, prev = [], np.zeros(w * ch, np.int32)
    for y in range(h):
        cur = pixels[y].reshape(-1).astype(np.int32)
        ft = filters[y % len(filters)]
        out = np.zeros_like(cur)
        for i in range(cur.size):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0: pred = 0
            elif ft == 1: pred = a
            elif ft == 2: pred = b
            elif ft == 3: pred = (a + b) >> 1
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[i] = (cur[i] - pred) & 255
        rows.append(bytes([ft]) + out.astype(np.uint8).tobytes())
        prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    body = chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0))
    if palette is not None:
        body += chunk(b"PLTE", palette.astype(np.uint8).tobytes())
    return b"\x89PNG\r\n\x1a\n" + body + chunk(b"IDAT", zlib.compress(b"".join(rows))) + chunk(b"IEND", b"")


def test_png_filters_and_colour_types(tmp_path):
    rng = np.random.default_rng(2)
    gray = rng.integers(0, 256, (9, 11, 1), dtype=np.uint8)
    buf = _png_bytes(gray, 0, [0, 1, 2, 3, 4])
    assert np.array_equal(D._read_png(buf), gray[..., 0])
    rgb = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    want = ((rgb[..., 0].astype(np.uint32) * 4899 + rgb[..., 1].astype(np.uint32) * 9617 + rgb[..., 2].astype(np.uint32) * 1868 + 8192)
            >> 14).astype(np.uint8)  # OpenCV's fixed-point BGR2GRAY
    for ct, px in ((2, rgb), (6, np.concatenate([rgb, rng.integers(0, 256, (7, 5, 1), dtype=np.uint8)], axis=2))):
        buf = _png_bytes(px, ct, [4, 3, 1, 2])
        assert np.array_equal(D._read_png(buf), want)
        p = str(tmp_path / f"c{ct}.png")
        open(p, "wb").write(buf)
        assert np.array_equal(D.read_gray_image(p), want)  # through Pillow when it is there
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    idx = rng.integers(0, 16, (6, 6, 1), dtype=np.uint8)
    got = D._read_png(_png_bytes(idx, 3, [0, 2], palette=pal))
    p3 = pal[idx[..., 0]].astype(np.uint32)
    assert np.array_equal(got, ((p3[..., 0] * 4899 + p3[..., 1] * 9617 + p3[..., 2] * 1868 + 8192) >> 14).astype(np.uint8))


def test_depth_file_is_centimetres_ascii(tmp_path):
    rng = np.random.default_rng(3)
    depth = rng.uniform(0.5, 3.0, (5, 7)).astype(np.float32)
    p = str(tmp_path / "x.depth")
    D.write_depth_file(p, depth)
    first = float(open(p).read().split()[0])
    assert abs(first - float(depth[0, 0]) * 100.0) < 1e-3
    got = D.read_depth_file(p, 7, 5)
    want = ((depth * np.float32(100.0)).astype(np.float32) / np.float32(100.0)).astype(np.float32)  # z / 100.0f, dataset.cpp:178
    assert np.array_equal(got, want)
    with pytest.raises(ValueError):
        D.read_depth_file(p, 8, 5)


def test_sequence_file_round_trip(tmp_path, monkeypatch):
    root = str(tmp_path / "seq")
    K = D.export_synthetic(root, 64, 48, 6, seed=3, image_ext="pgm", depth_every=2)
    assert sorted(os.listdir(os.path.join(root, "depthmaps"))) == ["scene_000.depth", "scene_002.depth", "scene_004.depth"]
    ds = D.Dataset()
    assert not ds.readDataSequence()  # no path yet (dataset.cpp:83-86)
    monkeypatch.setenv(D.DATA_PATH_ENV_VAR, root)
    assert ds.loadPathFromEnv() and ds.readDataSequence(1, 5) and len(ds) == 4
    assert ds.readDataSequence() and len(ds) == 6
    for k, e in enumerate(ds):
        assert e.getImageFileName() == f"scene_{k:03d}.pgm" and e.getDepthmapFileName() == f"scene_{k:03d}.depth"
        gray, rng_img = synth.render(64, 48, synth.pose(k, 3), 3, want_range=True, K=K)
        assert np.array_equal(ds.readImage(e), gray)
        T = np.asarray(ds.readCameraPose(e).data, np.float64).reshape(3, 4)
        assert np.abs(T - synth.pose(k, 3)).max() < 2e-6  # through 9-digit text and the fp32 quaternion constructor
        dm = ds.readDepthmap(e, 64, 48)
        if k % 2 == 0:
            assert np.abs(dm - rng_img).max() < 1e-6
        else:
            assert dm is None
    assert ds.readImage("nope.png") is None
    monkeypatch.delenv(D.DATA_PATH_ENV_VAR)
    assert not D.Dataset().loadPathFromEnv()


def test_config0_first_30_frames_on_the_cpu_oracle(tmp_path):
    """BASELINE configs[0]: the dataset_main trajectory, 640x480, first 30 frames, CPU path (plumbing, no GPU)."""
    root = str(tmp_path / "over_table")
    D.export_synthetic(root, 640, 480, 30)
    ds = D.Dataset(root)
    assert ds.readDataSequence(0, 30) and len(ds) == 30
    fx, fy, cx, cy = 481.2, -480.0, 319.5, 239.5  # dataset_main.cpp:37
    seeds = None
    gt = None
    for k, e in enumerate(ds):
        img = ds.readImage(e).astype(np.float32) * np.float32(1.0 / 255.0)  # depthmap.cpp:105
        T_curr_world = ds.readCameraPose(e).inv().data
        if k == 0:
            gt = ds.readDepthmap(e, 640, 480)
            seeds = O.Seeds(O.OracleLib("port", 5), 640, 480, (fx, fy, cx, cy))
            seeds.set_reference(img, T_curr_world, float(gt.min()), float(gt.max()))
        else:
            seeds.update(img, T_curr_world)
    conv = seeds.download(4) == 1
    assert conv.mean() > 0.5
    err = np.abs(seeds.download(0)[conv] - gt[conv])
    assert np.median(err) < 3e-3 and np.mean(err < 0.03) > 0.95


def test_scale_mat_is_the_reference_s():
    from rpg_open_remode_amd.dataset_main import parse, scale_mat
    d = np.array([[1.0, 1.5], [2.0, 1.25]], np.float32)
    assert np.array_equal(scale_mat(d), np.array([[0, 128], [255, 64]], np.uint8))  # (d - min) / (max - min) * 255, rounded
    a = parse(["--size", "320x240", "--end", "10"])
    assert (a.size, a.end, a.patch_side, a.lam, a.iterations) == ("320x240", 10, 5, 0.5, 200)


@pytest.mark.gpu
def test_dataset_main_end_to_end_equals_the_direct_api(tmp_path, capsys):
    from rpg_open_remode_amd import api, dataset_main
    root, out = str(tmp_path / "data"), str(tmp_path / "out")
    rc = dataset_main.main(["--export-synthetic", root, "--frames", "25", "--size", "320x240", "--out", out, "--iterations", "50"])
    assert rc == 0
    text = capsys.readouterr().out
    assert "RUN EXPERIMENT: inputting image scene_000.png" in text and text.count("UPDATE execution time") == 24
    assert "MEAN update time" in text and "STDDEV" in text
    depth, den, conv = (np.load(os.path.join(out, n)) for n in ("depth.npy", "denoised.npy", "convergence.npy"))
    assert D.read_gray_image(os.path.join(out, "result.png")).shape == (240, 320)
    # the same frames through the API directly
    ds = D.Dataset(root)
    assert ds.readDataSequence(0, 25)
    K = synth.intrinsics(320, 240)
    dm = api.Depthmap(320, 240, K[0], K[2], K[1], K[3])
    for k, e in enumerate(ds):
        T = ds.readCameraPose(e).inv()
        if k == 0:
            gt = ds.readDepthmap(e, 320, 240)
            dm.setReferenceImage(ds.readImage(e), T, float(gt.min()), float(gt.max()))
        else:
            dm.update(ds.readImage(e), T)
    dm.downloadDepthmap()
    assert O.count_mismatch(dm.getDepthmap(), depth) == 0
    dm.downloadDenoisedDepthmap(0.5, 50)
    assert O.count_mismatch(dm.getDepthmap(), den) == 0
    dm.downloadConvergenceMap()
    assert np.array_equal(dm.getConvergenceMap(), conv)
