"""bench.py pieces that do not need a GPU: argument plumbing, the JSON helpers, the committed traffic figures."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(argv):
    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod, mod.parse()
    finally:
        sys.argv = old


def test_defaults_are_the_headline_workload():
    b, a = _bench([])
    assert (a.gpus, a.steps, a.size, a.tv_iters, a.matcher) == (1, 199, "640x480", 200, 1)
    assert (b.WIDTH, b.HEIGHT, b.FRAMES, b.SIDE) == (640, 480, 200, 9)  # BASELINE.json configs[1]
    assert b.FUSED_BYTES_PER_PIXEL == 52 and b.TV_BYTES_PER_PIXEL_ITER == 40


def test_contract_flags_parse():
    _, a = _bench(["--gpus", "8", "--steps", "50", "--warmup", "5", "--size", "1920x1080", "--tv-iters", "500", "--cpu-seconds", "0"])
    assert (a.gpus, a.steps, a.warmup, a.size, a.tv_iters, a.cpu_seconds) == (8, 50, 5, "1920x1080", 500, 0.0)


def test_committed_traffic_file_feeds_both_rooflines():
    b, _ = _bench([])
    path = os.path.join(ROOT, "profiles", "traffic.json")
    t = json.load(open(path))
    assert t["algorithmic_bytes_per_update"] == 52 * 640 * 480
    assert t["seed_update_bytes_per_launch"] >= t["algorithmic_bytes_per_update"]  # measured traffic cannot be below the compulsory bytes
    r = b.valu_roofline(60e-6, path)
    assert r["bound"] == "valu" and 0.0 < r["frac"] < 1.0 and r["peak"] == 614.4
    assert b.valu_roofline(60e-6, os.path.join(ROOT, "no_such_file.json")) is None
