"""Builds every native artefact of the repository, in-tree.

  rpg_open_remode_amd/librmd_hip.so     HIP kernels + C ABI, hipcc --offload-arch=gfx950 (cross-compiles without a GPU)
  rpg_open_remode_amd/librmd_synth.so   synthetic sequence generator (host only)
  oracle/libremode_oracle*_s{3,5,7,9}.so  CPU oracle B (test infrastructure)
  oracle/_ref/libremode_ref_s{3,5,7,9}.so CPU oracle A, only where /root/reference exists
  oracle/_ref/{dataset_main,dataset_check,depthmap_check,remode_node,rmd_gtests}_ref  the reference's host sources, unmodified, on include/rmd/ (test infrastructure)

Run as `python -m rpg_open_remode_amd.build` or through __graft_entry__.build().
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # arithmetic contract (csrc/rmd_math.h): no contraction, IEEE fp32 divide / sqrt
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
    # gfx950 issues v_pk_{add,mul,fma}_f32 at half the rate of the scalar forms (tools/ubench/valu_rate.hip: 4.6 vs 2.4 cycles
    # per wave64 instruction with >= 2 waves per SIMD), so SLP-packing two fp32 operations into one gains nothing and costs the
    # v_mov's that assemble the register pairs: -22 % issue slots in the NCC inner block without it
    "-fno-slp-vectorize",
]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, cwd=None, verbose=False):
    if verbose:
        print("+", " ".join(cmd), flush=True)
    res = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"build step failed ({' '.join(cmd)}):\n{res.stdout}")
    if verbose and res.stdout.strip():
        print(res.stdout)
    return res.stdout


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


HIP_UNITS = ["rmd_capi", "rmd_update", "rmd_ingest", "rmd_batch", "rmd_denoise", "rmd_reduce", "rmd_publish", "rmd_engines"]  # translation units of librmd_hip.so (csrc/rmd_host.hpp says who owns what)


def build_hip(force=False, verbose=False, extra_flags=(), out=None):
    """librmd_hip.so from its translation units, compiled in parallel (the unit that instantiates the seed kernels dominates: ~10 s) and linked
    by hipcc.  out: another file name (A/B variants, tools/ab_make.sh; selected at run time with RMD_HIP_LIB).
    Returns the library's path; build_hip.last_report says what was compiled and what was reused."""
    product = out is None
    out = out or os.path.join(HERE, "librmd_hip.so")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))] + [os.path.join(ROOT, "include", "rmd_hip.h")]
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags) + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    units = HIP_UNITS
    # objects are kept per output file and flag set: a product build and an A/B variant never share them (a stable digest: Python's hash()
    # of a string is salted per interpreter)
    tag = "" if product and not extra_flags else "_" + hashlib.sha1(" ".join([os.path.basename(out), *extra_flags]).encode()).hexdigest()[:8]
    obj_dir = os.path.join(HERE, "build")
    os.makedirs(obj_dir, exist_ok=True)
    jobs, compiled, reused = [], [], []
    for u in units:
        src, obj = os.path.join(CSRC, u + ".hip"), os.path.join(obj_dir, u + tag + ".o")
        deps = [src] + headers
        if force or _newer(obj, deps):
            cmd = [hipcc_path(), *flags, "-c", src, "-o", obj]
            if verbose:
                print("+", " ".join(cmd), flush=True)
            jobs.append((u, cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
            compiled.append(u)
        else:
            reused.append(u)
    for u, cmd, proc in jobs:
        log, _ = proc.communicate()
        if proc.returncode != 0:
            raise RuntimeError(f"build step failed ({' '.join(cmd)}):\n{log}")
    objs = [os.path.join(obj_dir, u + tag + ".o") for u in units]
    linked = False
    if force or compiled or _newer(out, objs):
        # (libhsa-runtime64: the runtime libamdhip64 itself sits on -- rmd_engines.hip addresses the copy engines through it)
        _run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out, "-lhsa-runtime64"], verbose=verbose)
        linked = True
    build_hip.last_report = {"library": out, "compiled": compiled, "reused": reused, "linked": linked}
    return out


build_hip.last_report = None


def build_synth(force=False, verbose=False):
    out = os.path.join(HERE, "librmd_synth.so")
    src = os.path.join(CSRC, "synth.cpp")
    if force or _newer(out, [src]):
        _run(["g++", "-O2", "-fPIC", "-shared", "-fopenmp", src, "-o", out], verbose=verbose)
    return out


def build_oracles(force=False, verbose=False):
    odir = os.path.join(ROOT, "oracle")
    if force:
        _run(["make", "-C", odir, "clean"], verbose=verbose)
    _run(["make", "-C", odir, "port"], verbose=verbose)
    if os.path.isdir("/root/reference/src"):
        _run(["make", "-C", odir, "ref"], verbose=verbose)


REFERENCE = "/root/reference"


def reference_host_program_cmds(out_dir):
    """The reference's HOST sources around the path -- src/depthmap.cpp (rmd::Depthmap), test/dataset.cpp, test/dataset_main.cpp,
    src/depthmap_node.cpp, src/publisher.cpp, src/main_ros.cpp -- compiled UNMODIFIED, where they lie under /root/reference, against include/rmd/ (this repository's drop-in headers first,
    the reference's include/ only for depthmap.h, which has no counterpart here) and the test-only Eigen / Boost / OpenCV stand-ins
    of tests/cpp/stubs.  Test infrastructure: {program: command}."""
    stubs = os.path.join(ROOT, "tests", "cpp", "stubs")
    common = ["g++", "-std=c++11", "-O1", "-DRMD_CORR_PATCH_SIDE=5", "-DRMD_MAX_EXTENT_EPIPOLAR_SEARCH=100", "-I" + stubs,
              "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(REFERENCE, "include")]
    link = ["-L" + HERE, "-lrmd_hip", "-Wl,-rpath," + HERE, "-Wl,-rpath,$ORIGIN/../../rpg_open_remode_amd"]
    ref = lambda *p: os.path.join(REFERENCE, *p)
    return {
        "dataset_main_ref": common + [ref("src", "depthmap.cpp"), ref("test", "dataset.cpp"), ref("test", "dataset_main.cpp")] + link +
        ["-o", os.path.join(out_dir, "dataset_main_ref")],
        "depthmap_check_ref": common + [ref("src", "depthmap.cpp"), os.path.join(ROOT, "tests", "cpp", "depthmap_check.cpp")] + link +
        ["-o", os.path.join(out_dir, "depthmap_check_ref")],
        # rmd::test::Dataset alone (no device needed): the reference's reader on a directory written by dataset.py
        "dataset_check_ref": common + ["-I" + ref("test"), ref("test", "dataset.cpp"), os.path.join(ROOT, "tests", "cpp", "dataset_check.cpp")] + link +
        ["-o", os.path.join(out_dir, "dataset_check_ref")],
        # the reference's own googletest suite for the path (seedMatrixInit, seedMatrixCheck, epipolarTest, epipolarMatchTest,
        # deviceImageReduction.sum / .countEqual) with its main; device_image_test.cpp needs the test-only CUDA kernels copy.cu / sobel.cu
        # (outside the path) and is left out
        "rmd_gtests_ref": common + ["-pthread", "-DRMD_BUILD_TESTS=1", ref("src", "depthmap.cpp"), ref("test", "dataset.cpp"), ref("test", "seed_matrix_test.cpp"),
                                    ref("test", "epipolar_test.cpp"), ref("test", "reduction_test.cpp"), ref("test", "main_test.cpp")] + link +
        ["-o", os.path.join(out_dir, "rmd_gtests_ref")],
        # the live system: main_ros.cpp + DepthmapNode + Publisher + Depthmap on stand-ins for roscpp / image_transport / pcl / cv_bridge /
        # svo_msgs / vikit whose "topics" and "bag" are files (tests/cpp/stubs/ros/ros.h)
        "remode_node_ref": common + ["-pthread", ref("src", "depthmap.cpp"), ref("src", "publisher.cpp"), ref("src", "depthmap_node.cpp"),
                                     ref("src", "main_ros.cpp")] + link + ["-o", os.path.join(out_dir, "remode_node_ref")],
    }


def build_reference_host_programs(force=False, verbose=False):
    """oracle/_ref/{dataset_main,depthmap_check,remode_node,rmd_gtests}_ref; only where /root/reference exists (the GPU box uses the prebuilt files)."""
    if not os.path.isdir(os.path.join(REFERENCE, "src")):
        return
    out_dir = os.path.join(ROOT, "oracle", "_ref")
    os.makedirs(out_dir, exist_ok=True)
    deps = [os.path.join(HERE, "librmd_hip.so"), os.path.join(ROOT, "tests", "cpp", "depthmap_check.cpp"), os.path.join(ROOT, "tests", "cpp", "dataset_check.cpp")]
    deps += [os.path.join(ROOT, "include", "rmd", f) for f in os.listdir(os.path.join(ROOT, "include", "rmd"))]
    for d, _, files in os.walk(os.path.join(ROOT, "tests", "cpp", "stubs")):
        deps += [os.path.join(d, f) for f in files]
    for name, cmd in reference_host_program_cmds(out_dir).items():
        if force or _newer(os.path.join(out_dir, name), deps):
            _run(cmd, verbose=verbose)


def build_apps(force=False, verbose=False):
    """apps/bench_main: the C++ timed driver over include/rmd/ (the Python bench's timed region without the interpreter)"""
    src, out = os.path.join(ROOT, "apps", "bench_main.cpp"), os.path.join(ROOT, "apps", "bench_main")
    deps = [src, os.path.join(HERE, "librmd_hip.so"), os.path.join(HERE, "librmd_synth.so")] + [os.path.join(ROOT, "include", "rmd", f) for f in os.listdir(os.path.join(ROOT, "include", "rmd"))]
    if force or _newer(out, deps):
        _run(["g++", "-std=c++11", "-O2", "-DRMD_CORR_PATCH_SIDE=9", "-I" + os.path.join(ROOT, "include"), src, "-L" + HERE, "-lrmd_hip", "-lrmd_synth",
              "-Wl,-rpath," + HERE, "-Wl,-rpath,$ORIGIN/../rpg_open_remode_amd", "-o", out], verbose=verbose)
    return out


def build_all(force=False, verbose=False, report=True):
    """everything; prints one line saying what was compiled and what was found up to date (the driver's build check reads it)"""
    build_hip(force, verbose)
    build_synth(force, verbose)
    build_oracles(force, verbose)
    build_reference_host_programs(force, verbose)
    build_apps(force, verbose)
    if report:
        r = build_hip.last_report
        print(f"[rpg_open_remode_amd.build] librmd_hip.so: compiled {r['compiled'] or 'nothing'} for gfx950, reused {r['reused'] or 'nothing'}, "
              f"{'linked' if r['linked'] else 'up to date'}; synth / oracles / reference host programs: make-style (rebuilt where sources are newer)", flush=True)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
    print("build OK")
