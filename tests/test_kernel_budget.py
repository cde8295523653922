"""The compiled update kernels stay inside the budgets their performance depends on (CPU test: reads the gfx950 code objects inside
librmd_hip.so with the LLVM tools of the ROCm image; nothing runs).

DESIGN.md 4.1 measured each of these as a first-order quantity: the search kernel must fit four workgroups per CU and leave room for a setup wave beside them (<= 112 VGPRs, no scratch),
its code must stay resident in the instruction cache two CUs share (a build of 10 500 instructions ran 17 % slower), scalar spills are vector
instructions at every workgroup's entry, and the LDS window is staged LDS-direct (global_load_lds_dword: one memory round trip per window)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from rpg_open_remode_amd import _lib

LLVM = "/opt/rocm/lib/llvm/bin"
SEARCH = "_ZN4rmdk26seed_search_compact_kernelILi9ELi1EEE"
SEARCH_BATCH = "_ZN4rmdk26seed_search_compact_kernelILi9ELi8EEE"
SETUP = "_ZN4rmdk25seed_setup_compact_kernelILi9ELi1EEE"


@pytest.fixture(scope="module")
def code_object():
    """(path of the gfx950 code object that holds the update kernels, its directory)"""
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("the LLVM binary tools of the ROCm image are not installed")
    d = tempfile.mkdtemp(prefix="rmd_co_")
    lib = os.path.join(d, "lib.so")
    shutil.copy(_lib.LIB_PATH, lib)
    subprocess.run([objdump, "--offloading", lib], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    found = None
    for f in sorted(os.listdir(d)):
        if "gfx950" in f:
            syms = subprocess.run([readelf, "-s", "--wide", os.path.join(d, f)], capture_output=True, text=True).stdout
            if SEARCH in syms:
                found = os.path.join(d, f)
    assert found, "no gfx950 code object with the search kernel inside librmd_hip.so"
    yield found
    shutil.rmtree(d, ignore_errors=True)


def _metadata(co, prefix):
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    # one YAML map per kernel; the fields of a kernel lie between its '- .agpr_count' (the first key, alphabetically) and the next one
    for block in re.split(r"\n\s*- \.agpr_count:", notes):
        m = re.search(r"\.name:\s+(\S+)", block)
        if m and m.group(1).startswith(prefix):
            return {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)\s*$", block, flags=re.M)}
    raise AssertionError(f"kernel {prefix} not in the code object's metadata")


def _symbol_size(co, prefix):
    syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", co], capture_output=True, text=True, check=True).stdout
    for line in syms.splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == "FUNC" and f[7].startswith(prefix):
            return int(f[2])
    raise AssertionError(f"no function symbol {prefix}")


@pytest.mark.parametrize("kernel", [SEARCH, SEARCH_BATCH])
def test_search_kernel_fits_four_workgroups_per_cu(code_object, kernel):
    md = _metadata(code_object, kernel)
    assert md["vgpr_count"] <= 112, md          # 512 VGPRs per SIMD / 4 waves = 128; at <= 112 ALLOCATED registers (granule 8) four search waves leave 64 for a
                                                # setup wave (56) of another stream group of a batch: at 116 (120 allocated) a batch of 8 lost 5 % (LAB.md, round 5)
    assert md["vgpr_spill_count"] == 0 and md["private_segment_fixed_size"] == 0, md  # no scratch
    assert md["sgpr_spill_count"] <= 64, md     # every spilled scalar is a v_writelane / v_readlane pair somewhere hot (138 at the start of round 4)
    assert md["group_segment_fixed_size"] == 0  # the 36.3 KB window + descriptors are dynamic LDS (4 x 36.3 KB <= 160 KB per CU)


def test_setup_kernel_keeps_every_tile_resident(code_object):
    md = _metadata(code_object, SETUP)
    assert md["vgpr_count"] <= 64 and md["vgpr_spill_count"] == 0 and md["private_segment_fixed_size"] == 0, md  # 8 waves per SIMD: all 1 200 tiles of a 640x480 frame at once
    assert md["group_segment_fixed_size"] <= 1024, md


def test_update_kernels_fit_the_instruction_cache(code_object):
    search, setup = _symbol_size(code_object, SEARCH), _symbol_size(code_object, SETUP)
    assert search <= 32 * 1024, search  # 64 KB of instruction cache per pair of CUs, workgroups in every phase of the kernel at once
    assert setup <= 32 * 1024, setup


def test_search_kernel_stages_its_window_lds_direct(code_object):
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f"--disassemble-symbols={SEARCH}vNS_9BatchArgsIXT0_EEENS_11MatcherArgsE", code_object],
                         capture_output=True, text=True, check=True).stdout
    assert dis.count("global_load_lds_dword") >= 3, "window (two call sites) and patch halo are expected to be staged with global_load_lds_dword"
    assert "scratch_" not in dis


def test_lds_direct_transfers_are_drained_before_the_barrier_that_publishes_them(code_object):
    """global_load_lds_dword writes LDS behind the compiler's back: a wave reads window / halo rows that OTHER waves transferred, and nothing but
    an explicit s_waitcnt vmcnt(0) in EVERY wave before the workgroup barrier orders that (a workgroup-scope release guarantees lgkmcnt(0)
    only).  Layout-order scan of the search kernels: between an LDS-direct load and the next s_barrier there must be a vmcnt(0) wait."""
    for kernel in (SEARCH, SEARCH_BATCH):
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f"--disassemble-symbols={kernel}vNS_9BatchArgsIXT0_EEENS_11MatcherArgsE", code_object],
                             capture_output=True, text=True, check=True).stdout
        pending, n_loads, n_barriers = False, 0, 0
        for line in dis.splitlines():
            ins = line.strip()
            if ins.startswith("global_load_lds_dword"):
                pending, n_loads = True, n_loads + 1
            elif ins.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", ins):
                pending = False
            elif ins.startswith("s_barrier"):
                n_barriers += 1
                assert not pending, f"{kernel}: an s_barrier follows a global_load_lds_dword without s_waitcnt vmcnt(0) in between"
        assert n_loads >= 3 and n_barriers >= 4, (n_loads, n_barriers)


def test_every_kernel_has_one_home_code_object():
    """each translation unit of librmd_hip.so is a code object of its own; a `static __global__` kernel in a header every unit includes would be
    compiled into all of them (round 4: count_eq_kernel, convergence_bgr8_kernel and a dozen others sat in five code objects each)"""
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("the LLVM binary tools of the ROCm image are not installed")
    d = tempfile.mkdtemp(prefix="rmd_co_")
    try:
        lib = os.path.join(d, "lib.so")
        shutil.copy(_lib.LIB_PATH, lib)
        subprocess.run([objdump, "--offloading", lib], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        homes = {}
        for f in sorted(os.listdir(d)):
            if "gfx950" not in f:
                continue
            syms = subprocess.run([readelf, "--dyn-syms", "--wide", os.path.join(d, f)], capture_output=True, text=True, check=True).stdout
            for line in syms.splitlines():
                c = line.split()
                if len(c) >= 8 and c[3] == "FUNC" and "kernel" in c[7]:
                    homes.setdefault(c[7], []).append(f)
        assert len(homes) >= 40, sorted(homes)
        twice = {k: v for k, v in homes.items() if len(v) != 1}
        assert not twice, twice
    finally:
        shutil.rmtree(d, ignore_errors=True)
