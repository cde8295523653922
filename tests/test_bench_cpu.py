"""bench.py pieces that do not need a GPU: argument plumbing, workload resolution, the JSON helpers, the committed counters."""
import importlib.util
import json
import os
import sys

import pytest

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
    assert (a.gpus, a.size, a.matcher) == (1, "640x480", -1)  # matcher: the library's default
    assert b.resolve_workload(a) == (640, 480, 200, 200)  # BASELINE.json configs[1]: 200 frames, TV-L1 200 iterations
    assert b.KNOWN_CONFIGS[(640, 480, 200)] == "configs[1]"
    assert (b.WIDTH, b.HEIGHT, b.FRAMES, b.SIDE) == (640, 480, 200, 9)
    assert b.FUSED_BYTES_PER_PIXEL == 52 and b.TV_BYTES_PER_PIXEL_ITER == 40


def test_steps_do_not_change_the_workload():
    """Whatever --steps / --warmup the driver passes, a step is a complete pass over the configured sequence."""
    b, a = _bench(["--steps", "20", "--warmup", "5"])
    assert (a.steps, a.warmup) == (20, 5)
    assert b.resolve_workload(a) == (640, 480, 200, 200)


def test_other_baseline_configs_run_at_their_configured_length():
    b, a = _bench(["--size", "1280x960"])
    assert b.resolve_workload(a) == (1280, 960, 500, 200) and b.KNOWN_CONFIGS[(1280, 960, 500)] == "configs[2]"
    b, a = _bench(["--size", "1920x1080"])
    assert b.resolve_workload(a) == (1920, 1080, 1000, 500) and b.KNOWN_CONFIGS[(1920, 1080, 1000)] == "configs[4]"
    b, a = _bench(["--gpus", "8", "--steps", "50", "--warmup", "5", "--size", "1920x1080", "--frames", "30", "--tv-iters", "20", "--cpu-seconds", "0"])
    assert (a.gpus, a.steps, a.warmup, a.cpu_seconds) == (8, 50, 5, 0.0)
    assert b.resolve_workload(a) == (1920, 1080, 30, 20)


def test_rooflines():
    b, _ = _bench([])
    assert b.VALU_PEAK_GINST_S == 1228.8  # wave64 VALU = 2 cycles on a SIMD-32 (MI355X_MICROARCH.md), 1024 SIMDs, 2.4 GHz
    path = os.path.join(ROOT, "profiles", "traffic.json")
    t = json.load(open(path))
    assert t["algorithmic_bytes_per_update"] == 52 * 640 * 480
    assert t["seed_update_bytes_per_launch"] >= t["algorithmic_bytes_per_update"]  # measured traffic cannot be below the compulsory bytes
    # instruction counts are only used with the kernel sources they were measured on
    fresh = dict(t, kernel_source_sha256=b.kernel_source_sha256())
    r = b.valu_roofline(60e-6, fresh)
    assert r["bound"] == "valu" and 0.0 < r["frac"] < 1.0 and r["peak"] == 1228.8
    r4 = b.valu_roofline(4 * 60e-6, fresh, n_sequences=4)  # a batch of 4 sequences in 4x the time: the same fraction
    assert abs(r4["frac"] - r["frac"]) < 1e-3
    stale = b.valu_roofline(60e-6, dict(t, kernel_source_sha256="0" * 64))
    assert stale["stale"] is True and "frac" not in stale
    assert b.valu_roofline(60e-6, None) is None
    assert b.load_counters(os.path.join(ROOT, "no_such_file.json")) is None
    f = b.flops_roofline(60e-6, 489000.0, 9)
    assert f["flop_per_launch"] == 14 * 81 * 489000 and f["peak"] == 157.3
    assert abs(f["achieved"] - 14 * 81 * 489000 / 60e-6 / 1e12) < 0.01
    assert b.flops_roofline(60e-6, None, 9) is None


def test_committed_counters_belong_to_the_committed_kernels():
    """profiles/traffic.json carries the hash of the kernel sources it was measured on; a commit that changes the kernels without
    re-measuring makes bench.py drop roofline_valu (stale) instead of reporting a figure for code that no longer exists"""
    b, _ = _bench([])
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    if t.get("kernel_source_sha256") != b.kernel_source_sha256():
        pytest.skip("kernel sources changed since the last PMC pass (tools/profile_round.sh): bench.py reports roofline_valu as stale")
    assert b.valu_roofline(46e-6, t)["frac"] > 0.1


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("launcher", ["driver", "self"])
def test_two_rank_launch_path_of_bench_py(launcher):
    """The --gpus 2 launch exactly as the driver issues it (python -m torch.distributed.run ... bench.py --gpus 2 ...), and
    bench.py's own re-launch from a plain `python bench.py --gpus 2`: rendezvous on 127.0.0.1, barrier, gather (MAX of elapsed,
    SUM of units), rank 0 alone prints one JSON line, the group is torn down.  No GPU here: the control plane falls to gloo."""
    import json
    import subprocess
    import sys
    port = str(_free_port())
    bench_py = os.path.join(ROOT, "bench.py")
    tail = ["--gpus", "2", "--steps", "1", "--warmup", "0", "--rendezvous-only"]
    if launcher == "driver":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", port, bench_py] + tail
        env = dict(os.environ)
    else:
        cmd = [sys.executable, bench_py] + tail
        env = dict(os.environ, MASTER_PORT=port)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["rendezvous"] == "ok" and out["n_gpus"] == 2 and out["control_plane"] == "gloo"
    assert out["max_elapsed_s"] == pytest.approx(0.002) and out["total_units"] == 3.0
    assert [r[2] for r in out["per_rank"]] == [0.0, 1.0]  # LOCAL_RANK of each rank, in rank order
    assert out["batch_per_gpu"] == 1 and out["scenes_of_rank"] == [[0], [1]]


def test_two_ranks_times_four_sequences_launch_path():
    """--gpus 2 --batch-per-gpu 4 (N GPUs x B sequences, BASELINE configs[3] composed with the batched mode): rank r owns scenes 4r .. 4r+3, the
    gathered record carries B for every rank"""
    import subprocess
    port = str(_free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch-per-gpu", "4", "--steps", "1", "--warmup", "0", "--rendezvous-only"]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert out["batch_per_gpu"] == 4 and out["scenes_of_rank"] == [[0, 1, 2, 3], [4, 5, 6, 7]]
    assert [r[4] for r in out["per_rank"]] == [4.0, 4.0]


def test_scene_assignment_and_useful_valu_fraction():
    b, a = _bench(["--batch-per-gpu", "8"])
    assert a.batch_per_gpu == 8 and b.scenes_of_rank(0, 8) == list(range(8)) and b.scenes_of_rank(3, 2) == [6, 7] and b.scenes_of_rank(5, 1) == [5]
    t = dict(json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))), kernel_source_sha256=b.kernel_source_sha256())
    n = sum(t["valu_wave_instructions_per_update"].values())
    r = b.valu_roofline(46e-6, t, ncc_evals_per_update=64.0 * 1000)
    assert r["useful_valu_frac"] == round(1000 * 851 / n, 4) and set(r["per_kernel"]) == set(t["valu_wave_instructions_per_update"])
    r8 = b.valu_roofline(8 * 46e-6, t, n_sequences=8, ncc_evals_per_update=64.0 * 1000)
    assert r8["useful_valu_frac"] == r["useful_valu_frac"]
    assert b.valu_roofline(46e-6, t)["useful_valu_frac"] is None


def test_kernel_source_hash_ignores_comments_and_white_space_only():
    b, _ = _bench([])
    a = 'int f(int x) { // add one\n  return x + 1; /* here */ }\nconst char* s = "// not a comment";\n'
    same = 'int f(int x) {\n\n   return x + 1;   }   // something else entirely\nconst char* s = "// not a comment";'
    other = 'int f(int x) { return x + 2; }\nconst char* s = "// not a comment";'
    assert b._code_only(a) == b._code_only(same)
    assert b._code_only(a) != b._code_only(other)
    assert '"// not a comment"' in b._code_only(a)
