"""The reference's test dataset layout (test/dataset.h, test/dataset.cpp): reader, writer and exporter.

    <root>/<sequence file>              one line per frame:  <image name> tx ty tz qx qy qz qw      (dataset.cpp:100-113)
    <root>/images/<image name>          8-bit gray image (the reference reads it with cv::imread(..., GRAYSCALE), :134-143)
    <root>/depthmaps/<stem>.depth       W*H ASCII floats, row-major, centimetres along the pixel ray  (:163-186, /100 at :178)

The pose of a line is T_world_cam as (translation, quaternion) and becomes rmd::SE3<float>(qw, qx, qy, qz, tx, ty, tz)
(dataset.cpp:150-161); callers hand its inverse to the depth filter (dataset_main.cpp:89,102).  <stem> is the image name up
to and including its first '.', plus "depth" (dataset.cpp:104).  The data path comes from $RMD_TEST_DATA_PATH (:198-211).

OpenCV is not part of this repository's environment, so images are decoded here: binary/ASCII PGM (P5/P2) and PNG (8-bit
gray, gray+alpha, RGB, RGBA, palette; non-interlaced), colour converted to gray with OpenCV's BGR2GRAY weights and
rounding (0.299 R + 0.587 G + 0.114 B in 14-bit fixed point), so that the bytes match what cv::imread(GRAYSCALE) yields.
"""
import os
import struct
import zlib

import numpy as np

from .api import SE3

DATA_PATH_ENV_VAR = "RMD_TEST_DATA_PATH"  # dataset.h:82
DEFAULT_SEQUENCE_FILE = "first_200_frames_traj_over_table_input_sequence.txt"  # dataset_main.cpp:39


class DatasetEntry:
    """One line of the sequence file (dataset.h:32-49)."""

    def __init__(self, image_file_name, translation, quaternion_xyzw):
        self.image_file_name = image_file_name
        self.depthmap_file_name = image_file_name[:image_file_name.find(".") + 1] + "depth"
        self.translation = np.asarray(translation, np.float32)
        self.quaternion = np.asarray(quaternion_xyzw, np.float32)  # x, y, z, w as in the file

    def getImageFileName(self): return self.image_file_name
    def getDepthmapFileName(self): return self.depthmap_file_name
    def getTranslation(self): return self.translation
    def getQuaternion(self): return self.quaternion


# ---------------------------------------------------------------------------------------------------------- image codecs
def _gray_from_rgb(rgb):
    """OpenCV's 8-bit RGB->gray: (R*4899 + G*9617 + B*1868 + 8192) >> 14."""
    r, g, b = (rgb[..., k].astype(np.uint32) for k in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def _read_pnm(buf):
    tokens, pos, n = [], 0, len(buf)
    magic = buf[:2]
    if magic not in (b"P5", b"P2"):
        raise ValueError("not a PGM file")
    pos = 2
    while len(tokens) < 3:  # width, height, maxval, with '#' comments
        while pos < n and buf[pos:pos + 1].isspace():
            pos += 1
        if buf[pos:pos + 1] == b"#":
            while pos < n and buf[pos:pos + 1] != b"\n":
                pos += 1
            continue
        start = pos
        while pos < n and not buf[pos:pos + 1].isspace():
            pos += 1
        tokens.append(int(buf[start:pos]))
    w, h, maxval = tokens
    if magic == b"P5":
        pos += 1  # the single whitespace after maxval
        if maxval < 256:
            img = np.frombuffer(buf, np.uint8, w * h, pos).reshape(h, w)
        else:
            img = (np.frombuffer(buf, ">u2", w * h, pos).reshape(h, w) >> 8).astype(np.uint8)
    else:
        vals = np.array(buf[pos:].split(), np.int64)[:w * h].reshape(h, w)
        img = (vals if maxval < 256 else vals >> 8).astype(np.uint8)
    if maxval not in (255, 65535):
        img = np.rint(img.astype(np.float64) * (255.0 / maxval)).astype(np.uint8)
    return np.ascontiguousarray(img)


def _png_unfilter(raw, h, stride, bpp):
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    pos = 0
    for y in range(h):
        ft = raw[pos]
        line = np.frombuffer(raw, np.uint8, stride, pos + 1).astype(np.int32)
        pos += stride + 1
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:
            cur = line.copy()
            for off in range(bpp, stride, bpp):  # running sum per channel, bpp bytes at a time
                cur[off:off + bpp] = (cur[off:off + bpp] + cur[off - bpp:off]) & 255
        elif ft in (3, 4):
            cur = line.copy()
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                if ft == 3:
                    cur[i] = (cur[i] + ((a + b) >> 1)) & 255
                else:
                    c = prev[i - bpp] if i >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                    cur[i] = (cur[i] + pred) & 255
        else:
            raise ValueError("PNG: bad filter type")
        out[y] = cur
        prev = cur
    return out


def _read_png(buf):
    if buf[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, palette, hdr = 8, [], None, None
    while pos < len(buf):
        length, ctype = struct.unpack(">I4s", buf[pos:pos + 8])
        data = buf[pos + 8:pos + 8 + length]
        pos += 12 + length
        if ctype == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        elif ctype == b"PLTE":
            palette = np.frombuffer(data, np.uint8).reshape(-1, 3)
        elif ctype == b"IDAT":
            idat.append(data)
        elif ctype == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    if interlace != 0 or depth not in (8, 16):
        raise ValueError("PNG: only non-interlaced 8/16-bit images are supported")
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = channels * depth // 8
    px = _png_unfilter(zlib.decompress(b"".join(idat)), h, w * bpp, bpp)
    if depth == 16:
        px = px[:, 0::2]  # high bytes
    px = px.reshape(h, w, channels)
    if ctype == 3:
        return _gray_from_rgb(palette[px[..., 0]])
    if ctype in (2, 6):
        return _gray_from_rgb(px[..., :3])
    return np.ascontiguousarray(px[..., 0])


def _read_png_pillow(path):
    """Same result as _read_png, with Pillow doing the inflate + unfilter when it is installed (much faster on Paeth rows)."""
    from PIL import Image
    im = Image.open(path)
    if im.mode in ("L", "LA"):
        return np.ascontiguousarray(np.asarray(im.convert("L") if im.mode == "L" else im.getchannel(0), np.uint8))
    if im.mode in ("P", "RGB", "RGBA"):
        return _gray_from_rgb(np.asarray(im.convert("RGB"), np.uint8))
    raise ValueError("unsupported PNG mode")


def read_gray_image(path):
    """8-bit gray image from a .pgm or .png file (what cv::imread(path, CV_LOAD_IMAGE_GRAYSCALE) returns)."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:2] in (b"P5", b"P2"):
        return _read_pnm(buf)
    try:
        return _read_png_pillow(path)
    except (ImportError, ValueError):
        return _read_png(buf)


def write_pgm(path, img):
    img = np.ascontiguousarray(img, np.uint8)
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def write_png(path, img):
    """8-bit gray PNG (filter type 0, zlib level 6)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape

    def chunk(ctype, data):
        return struct.pack(">I", len(data)) + ctype + data + struct.pack(">I", zlib.crc32(ctype + data) & 0xffffffff)

    raw = np.zeros((h, w + 1), np.uint8)
    raw[:, 1:] = img
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)) + chunk(b"IEND", b""))


def write_gray_image(path, img):
    (write_png if path.lower().endswith(".png") else write_pgm)(path, img)


# -------------------------------------------------------------------------------------------------------------- depth maps
def read_depth_file(path, width, height):
    """ASCII centimetres -> float32 metres, z / 100.0f as dataset.cpp:178."""
    vals = np.fromfile(path, dtype=np.float32, sep=" ", count=width * height)
    if vals.size != width * height:
        raise ValueError(f"{path}: expected {width * height} depth values, found {vals.size}")
    return (vals / np.float32(100.0)).astype(np.float32).reshape(height, width)


def write_depth_file(path, depth_m):
    """float metres -> ASCII centimetres, 9 significant digits (round-trips float32)."""
    cm = np.asarray(depth_m, np.float32) * np.float32(100.0)
    with open(path, "w") as f:
        for row in cm:
            f.write(" ".join(f"{v:.9g}" for v in row))
            f.write("\n")


# ------------------------------------------------------------------------------------------------------------------ dataset
class Dataset:
    """rmd::test::Dataset (dataset.h:51-85)."""

    def __init__(self, dataset_path="", sequence_file=DEFAULT_SEQUENCE_FILE):
        self.dataset_path_ = dataset_path or ""
        self.sequence_file_ = sequence_file
        self.dataset_ = []

    @staticmethod
    def getDataPathEnvVar():
        return DATA_PATH_ENV_VAR

    def loadPathFromEnv(self):  # dataset.cpp:198-207
        p = os.environ.get(DATA_PATH_ENV_VAR)
        if p is None:
            return False
        self.dataset_path_ = p
        return True

    def readDataSequence(self, start=0, end=0):  # dataset.cpp:81-127: lines [start, end), end == 0 = to the end of the file
        if not self.dataset_path_ or not self.sequence_file_:
            return False
        self.dataset_ = []
        try:
            f = open(os.path.join(self.dataset_path_, self.sequence_file_), "r")
        except OSError:
            return False
        with f:
            for line_cnt, line in enumerate(f):
                if line_cnt < start or (end != 0 and line_cnt >= end):
                    continue
                tok = line.split()
                if len(tok) < 8:
                    continue
                v = [float(t) for t in tok[1:8]]
                self.dataset_.append(DatasetEntry(tok[0], v[0:3], v[3:7]))
        return True

    def readImage(self, entry_or_name):  # dataset.cpp:129-148; None when the file cannot be read
        name = entry_or_name if isinstance(entry_or_name, str) else entry_or_name.image_file_name
        try:
            return read_gray_image(os.path.join(self.dataset_path_, "images", name))
        except (OSError, ValueError, KeyError, struct.error, zlib.error):
            return None

    @staticmethod
    def readCameraPose(entry):  # dataset.cpp:150-161 -> T_world_curr
        q, t = entry.quaternion, entry.translation
        return SE3(q[3], q[0], q[1], q[2], t[0], t[1], t[2])

    def readDepthmap(self, entry, width, height):  # dataset.cpp:163-186; None when the file cannot be read
        try:
            return read_depth_file(os.path.join(self.dataset_path_, "depthmaps", entry.depthmap_file_name), width, height)
        except (OSError, ValueError):
            return None

    def __iter__(self): return iter(self.dataset_)
    def __len__(self): return len(self.dataset_)
    def __call__(self, index): return self.dataset_[index]


def quaternion_from_rotation(R):
    """(x, y, z, w) of a 3x3 rotation matrix, float64 (Shepperd's method); inverse of the quaternion constructor of se3.cuh:38-66."""
    R = np.asarray(R, np.float64)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s, (R[2, 1] - R[1, 2]) / s]
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s, (R[0, 2] - R[2, 0]) / s]
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s, (R[1, 0] - R[0, 1]) / s]
    q = np.asarray(q, np.float64)
    return q / np.linalg.norm(q)


def export_synthetic(root, width=640, height=480, n_frames=200, seed=0, image_ext="png", sequence_file=DEFAULT_SEQUENCE_FILE,
                     depth_every=0, only_frames=None):
    """Writes the synthetic over-table sequence (synth.py) in the reference's dataset layout, so that everything written
    against that layout -- the reference's dataset_main and gtest sources included -- runs without the original download.
    Frame 0 always gets its ground-truth .depth file (3 MB of ASCII at 640x480), every `depth_every`-th frame too if > 0.
    `only_frames`: render just these frames (image + .depth each); the sequence file still lists all n_frames."""
    from . import synth
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    os.makedirs(os.path.join(root, "depthmaps"), exist_ok=True)
    K = synth.intrinsics(width, height)
    lines = []
    for k in range(n_frames):
        T = synth.pose(k, seed)
        name = f"scene_{k:03d}.{image_ext}"
        if only_frames is None or k in only_frames:
            want_depth = only_frames is not None or k == 0 or (depth_every > 0 and k % depth_every == 0)
            gray, rng = synth.render(width, height, T, seed, want_range=want_depth, K=K)
            write_gray_image(os.path.join(root, "images", name), gray)
            if want_depth:
                write_depth_file(os.path.join(root, "depthmaps", f"scene_{k:03d}.depth"), rng)
        q = quaternion_from_rotation(T[:, :3])
        lines.append(f"{name} {T[0, 3]:.9g} {T[1, 3]:.9g} {T[2, 3]:.9g} {q[0]:.9g} {q[1]:.9g} {q[2]:.9g} {q[3]:.9g}")
    with open(os.path.join(root, sequence_file), "w") as f:
        f.write("\n".join(lines) + "\n")
    return K
