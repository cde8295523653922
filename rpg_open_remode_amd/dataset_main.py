#!/usr/bin/env python3
"""The reference's stand-alone experiment (test/dataset_main.cpp:32-138) on this library.

    RMD_TEST_DATA_PATH=/data/remode_test_data python -m rpg_open_remode_amd.dataset_main
    python -m rpg_open_remode_amd.dataset_main --export-synthetic /tmp/over_table --frames 60      # no download at hand

Reads the sequence file, feeds frame 0 as the reference view (scene range = min / max of its ground-truth depth map,
dataset_main.cpp:76-89) and every further frame as an update with T_world_curr.inv(), prints the per-update wall time and
its mean / variance / standard deviation (:106-135), downloads the depth map and the TV-L1 denoised depth map
(lambda 0.5, 200 iterations, :110-118).  Instead of cv::imshow the results are written to --out (float32 .npy and the
[0,1]-scaled 8-bit picture of Depthmap::scaleMat, depthmap.cpp:158-169).

Differences from the reference program, both to make it usable without its 600 MB of ASCII depth maps: the ground-truth
depth map is needed for the reference frame only (the reference skips every frame whose .depth file is missing, :70-75), and
intrinsics / image size / patch size are options with the reference's values as defaults.
"""
import argparse
import math
import os
import sys
import time

import numpy as np


def scale_mat(depthmap):
    """Depthmap::scaleMat (depthmap.cpp:158-169) without the gray->BGR replication: [min, max] -> [0, 255], saturate_cast rounding."""
    d = np.asarray(depthmap, np.float64)
    lo, hi = float(np.nanmin(d)), float(np.nanmax(d))
    s = (d - lo) * 1.0 / (hi - lo) if hi > lo else np.zeros_like(d)
    return np.clip(np.rint(s * 255.0), 0, 255).astype(np.uint8)


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data", default=None, help="dataset root (default: $RMD_TEST_DATA_PATH)")
    ap.add_argument("--sequence", default=None, help="sequence file inside the root")
    ap.add_argument("--start", type=int, default=0)
    ap.add_argument("--end", type=int, default=200, help="one past the last line of the sequence file to use (dataset_main.cpp:45)")
    ap.add_argument("--size", default="640x480")
    ap.add_argument("--cam", type=float, nargs=4, default=None, metavar=("FX", "FY", "CX", "CY"),
                    help="default 481.2 -480 319.5 239.5 (dataset_main.cpp:37), scaled with --size")
    ap.add_argument("--patch-side", type=int, default=5, help="RMD_CORR_PATCH_SIDE (CMakeLists.txt:51)")
    ap.add_argument("--max-extent", type=int, default=100, help="RMD_MAX_EXTENT_EPIPOLAR_SEARCH (CMakeLists.txt:52)")
    ap.add_argument("--lambda", dest="lam", type=float, default=0.5)
    ap.add_argument("--iterations", type=int, default=200)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", default=None, help="directory for depth.npy, denoised.npy, convergence.npy and the scaled pictures")
    ap.add_argument("--export-synthetic", default=None, metavar="DIR",
                    help="first write the synthetic over-table sequence in the dataset layout to DIR and run on it")
    ap.add_argument("--frames", type=int, default=200, help="frames to export with --export-synthetic")
    ap.add_argument("--quiet", action="store_true", help="no per-frame output")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse(argv)
    from . import api
    from .dataset import DATA_PATH_ENV_VAR, DEFAULT_SEQUENCE_FILE, Dataset, export_synthetic, write_gray_image

    width, height = (int(v) for v in args.size.lower().split("x"))
    if args.cam is None:
        fx, fy, cx, cy = 481.2 * width / 640.0, -480.0 * height / 480.0, (width - 1) / 2.0, (height - 1) / 2.0
    else:
        fx, fy, cx, cy = args.cam

    if not api.checkCudaDevice(args.device, verbose=not args.quiet):  # dataset_main.cpp:34-35
        return 1

    if args.export_synthetic:
        export_synthetic(args.export_synthetic, width, height, args.frames, depth_every=0)
        args.data = args.export_synthetic
        args.end = min(args.end, args.frames) if args.end else args.frames

    dataset = Dataset(args.data or "", args.sequence or DEFAULT_SEQUENCE_FILE)
    if not args.data and not dataset.loadPathFromEnv():
        print(f"ERROR: could not retrieve dataset path from the environment variable '{DATA_PATH_ENV_VAR}'", file=sys.stderr)
    if not dataset.readDataSequence(args.start, args.end):
        print("ERROR: could not read dataset", file=sys.stderr)
        return 1

    depthmap = api.Depthmap(width, height, fx, cx, fy, cy, patch_side=args.patch_side, max_extent=args.max_extent)
    first_img = True
    update_time = []
    gt_ref = None
    for data in dataset:
        img = dataset.readImage(data)
        if img is None or img.shape != (height, width):
            print(f"ERROR: could not read image {data.getImageFileName()}", file=sys.stderr)
            continue
        T_world_curr = dataset.readCameraPose(data)
        if not args.quiet:
            print(f"RUN EXPERIMENT: inputting image {data.getImageFileName()}")
            print("T_world_curr:")
            print(T_world_curr)
        if first_img:
            depth_32fc1 = dataset.readDepthmap(data, width, height)
            if depth_32fc1 is None:
                print(f"ERROR: could not read depthmap {data.getDepthmapFileName()}", file=sys.stderr)
                continue
            min_depth, max_depth = float(depth_32fc1.min()), float(depth_32fc1.max())
            if depthmap.setReferenceImage(img, T_world_curr.inv(), min_depth, max_depth):
                first_img = False
                gt_ref = depth_32fc1
            else:
                print("ERROR: could not set reference image", file=sys.stderr)
                return 1
        else:
            t = time.perf_counter()
            depthmap.update(img, T_world_curr.inv())
            t = time.perf_counter() - t
            if not args.quiet:
                print(f"\nUPDATE execution time: {t:f} seconds.")
            update_time.append(t)
    if first_img:
        print("ERROR: no reference frame could be set", file=sys.stderr)
        return 1

    t_all = time.perf_counter()
    depthmap.downloadDepthmap()
    result = depthmap.getDepthmap().copy()
    depthmap.downloadDenoisedDepthmap(args.lam, args.iterations)
    denoised_result = depthmap.getDepthmap().copy()
    depthmap.downloadConvergenceMap()
    convergence = depthmap.getConvergenceMap()
    t_all = time.perf_counter() - t_all

    if args.out:
        os.makedirs(args.out, exist_ok=True)
        np.save(os.path.join(args.out, "depth.npy"), result)
        np.save(os.path.join(args.out, "denoised.npy"), denoised_result)
        np.save(os.path.join(args.out, "convergence.npy"), convergence)
        write_gray_image(os.path.join(args.out, "result.png"), scale_mat(result))
        write_gray_image(os.path.join(args.out, "denoised_result.png"), scale_mat(denoised_result))

    n = max(len(update_time), 1)
    time_mean = sum(update_time) / n
    time_var = sum((t - time_mean) ** 2 for t in update_time) / n
    print("\n")
    print(f"MEAN update time: {time_mean:g}")
    print(f"VAR  update time: {time_var:g}")
    print(f"(STDDEV: {math.sqrt(time_var):g})")
    # not in the reference program: what the map looks like against the ground truth of the reference view
    conv = convergence == api.ConvergenceStates.CONVERGED
    print(f"updates: {len(update_time)}   converged: {depthmap.getConvergedPercentage():.1f} %   "
          f"download + denoise({args.lam}, {args.iterations}): {t_all * 1e3:.2f} ms")
    if gt_ref is not None and conv.any():
        err = np.abs(result[conv] - gt_ref[conv])
        err_d = np.abs(denoised_result[conv] - gt_ref[conv])
        print(f"converged seeds vs ground truth of the reference view: median |error| {np.median(err) * 1e3:.2f} mm, "
              f"RMSE {math.sqrt(float(np.mean(err.astype(np.float64) ** 2))) * 1e3:.2f} mm; "
              f"denoised: median {np.median(err_d) * 1e3:.2f} mm")
    return 0


if __name__ == "__main__":
    sys.exit(main())
