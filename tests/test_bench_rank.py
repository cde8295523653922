"""bench.py's multi-rank plumbing, on the CPU (check_cuda_device.cu:60-71,109): which device a rank binds to and WHEN, and what a
per-rank record carries."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod_rank", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_rank_binds_its_device_before_anything_else():
    b = _bench()
    for n_dev, world in ((8, 8), (1, 8), (4, 8), (2, 2)):
        for local_rank in range(world):
            calls = []
            dev = b.bind_rank_device(local_rank, n_dev, lambda d: calls.append(("torch", d)), lambda d: calls.append(("rmd_hip_set_device", d)) or True)
            assert dev == local_rank % n_dev
            assert calls == [("torch", dev), ("rmd_hip_set_device", dev)]  # torch's current device (the RCCL group's), then the library's
    with pytest.raises(SystemExit):
        b.bind_rank_device(0, 0, lambda d: None, lambda d: True)  # no device: no CPU fallback
    with pytest.raises(SystemExit):
        b.bind_rank_device(1, 2, lambda d: None, lambda d: False)  # the library refuses the device
    # ... and main() binds before it creates its first handle of any kind (handles bind to the device that is current at their creation)
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    bind = main.index("bind_rank_device(")
    first_handle = min(m.start() for m in re.finditer(r"api\.(SeedMatrix|SeedMatrixBatch|DeviceImage|DepthmapDenoiser)\(", main))
    assert bind < first_handle
    assert "render_scene(" in main and main.index("def render_scene") > bind  # the resident frames (DeviceImage) are created after it too


def test_per_rank_record_carries_host_cpu_and_submit_time():
    b = _bench()
    # (elapsed s, pixels, updates, converged, sequences, host CPU s, submit s) as batch.gather_throughput returns them
    rec = b.rank_record((0.5, 199 * 307200 * 2.0, 398.0, 123456.0, 2.0, 0.31, 0.42))
    assert rec["host_cpu_s"] == 0.31 and rec["host_cores_busy"] == 0.62 and rec["sequences"] == 2 and rec["updates"] == 398
    assert rec["host_submit_us_per_update"] == round(0.42 / 199 * 1e6, 2)  # per update() CALL: a batch steps its sequences together
    assert set(rec) >= {"elapsed_s", "mpix", "updates", "converged", "sequences", "host_cpu_s", "host_cores_busy", "host_submit_us_per_update"}
