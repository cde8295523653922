#!/usr/bin/env python3
"""Offline model of the search kernel's LDS-window policy (test infrastructure; not a test, nothing in the product depends on it).

Drives Oracle B over the benchmark sequence (640x480, side 9) and, at selected updates, rebuilds per seed what the setup kernel computes
(epipolar segment, run of in-image steps, sample box) in numpy, forms the tiles' work units like the kernels do, and counts for several
window policies how many NCC evaluations of a unit would miss the LDS window (and be served from L2):

  centre     the round-4 policy: box of the unit's samples; if it does not fit, cut around the BOX'S centre
  centroid   cut around the work-weighted centroid of the samples instead
  two        centroid-cut main window + a second small window around the centroid of what the first one misses

usage: python tests/sim_window_policy.py [--frames 24,30,...] [--second 1024]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracles as O  # noqa: E402
from rpg_open_remode_amd import synth  # noqa: E402

W, H, SIDE, HALF = 640, 480, 9, 4
TILE = 16
WIN_CAP = 5632


def seed_runs(mu, sig2, a, b, K, T_curr_ref, eps, max_extent=100.0):
    """per pixel: live mask, n_valid, first sample, last sample (float64 model of epipolar_match.cu:59-97)"""
    fx, fy, cx, cy = K
    ys, xs = np.mgrid[0:H, 0:W]
    border = (xs < SIDE) | (ys < SIDE) | (xs > W - SIDE - 1) | (ys > H - SIDE - 1)
    conv = (a / (a + b) > 0.7) & (sig2 < eps)
    div = (a - 1.0) / (a + b - 2.0) < 0.05
    live = ~border & ~conv & ~div
    f = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones_like(xs, float)], -1)
    f /= np.linalg.norm(f, axis=-1, keepdims=True)
    R, t = T_curr_ref[:, :3], T_curr_ref[:, 3]

    def proj(d):
        p = (f * d[..., None]) @ R.T + t
        return np.stack([fx * p[..., 0] / p[..., 2] + cx, fy * p[..., 1] / p[..., 2] + cy], -1)

    sig = np.sqrt(sig2)
    pm, p0, p1 = proj(mu), proj(np.maximum(mu - 3 * sig, 0.01)), proj(mu + 3 * sig)
    line = p1 - p0
    ln = np.linalg.norm(line, axis=-1)
    d = line / np.maximum(ln, 1e-30)[..., None]
    half = 0.5 * np.minimum(ln, max_extent)
    n_steps = np.floor(2 * half / 0.7).astype(int) + 1
    return live, pm, d, half, n_steps


def unit_samples(pm, d, half, n_steps, seeds):
    """all in-image samples of the given seeds (flat indices): arrays (seed index in `seeds`, x, y)"""
    out_s, out_x, out_y = [], [], []
    for k, s in enumerate(seeds):
        y, x = divmod(s, W)
        l = -half[y, x] + 0.7 * np.arange(n_steps[y, x])
        px = pm[y, x, 0] + l * d[y, x, 0]
        py = pm[y, x, 1] + l * d[y, x, 1]
        ok = (px >= SIDE) & (py >= SIDE) & (px < W - SIDE) & (py < H - SIDE)
        out_s.append(np.full(ok.sum(), k)); out_x.append(px[ok]); out_y.append(py[ok])
    return np.concatenate(out_s), np.concatenate(out_x), np.concatenate(out_y)


def fits(x0, y0, x1, y1, cap=WIN_CAP):
    return x1 >= x0 and y1 >= y0 and ((x1 - x0 + 1) | 1) * (y1 - y0 + 1) <= cap


def box_of(px, py):
    return (max(int(np.floor(px.min())) - HALF - 1, 0), max(int(np.floor(py.min())) - HALF - 1, 0),
            min(int(np.floor(px.max())) + HALF + 2, W - 1), min(int(np.floor(py.max())) + HALF + 2, H - 1))


def clamp_around(box, cx, cy, cap=WIN_CAP):
    x0, y0, x1, y1 = box
    ww, wh = x1 - x0 + 1, y1 - y0 + 1
    flat = cap // 129
    nw = min(ww, 64) if wh > flat else min(ww, (cap // wh - 1) | 1)
    nh = min(wh, cap // (nw | 1))
    nx0 = int(min(max(round(cx - nw / 2), x0), x1 - nw + 1))
    ny0 = int(min(max(round(cy - nh / 2), y0), y1 - nh + 1))
    return nx0, ny0, nx0 + nw - 1, ny0 + nh - 1


def inside(box, px, py):
    x0, y0, x1, y1 = box
    ix, iy = np.floor(px).astype(int) - HALF, np.floor(py).astype(int) - HALF
    return (ix >= x0) & (iy >= y0) & (ix + SIDE + 1 <= x1 + 1) & (iy + SIDE + 1 <= y1 + 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="24,28,32,36,40,45,50,55,70,90,100,110,130,160,190")
    ap.add_argument("--second", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    want = sorted(int(v) for v in a.frames.split(","))
    seq = synth.Sequence(W, H, max(want) + 1, a.seed)
    olib = O.OracleLib("port", SIDE)
    s = O.Seeds(olib, W, H, seq.K)
    s.set_reference(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    eps = (seq.max_depth - seq.min_depth) / 1000.0
    T_world_ref = seq.T_world_cam[0]
    print("update | units unboxed | miss%: centre centroid two | units with a miss: centre centroid two | worst unit misses: centre centroid two | 2nd-window texels p50 p90")
    for n in range(1, max(want) + 1):
        if n in want:
            mu, sig2, aa, bb = (s.download(p).astype(np.float64) for p in (O.PLANE_MU, O.PLANE_SIGMA_SQ, O.PLANE_A, O.PLANE_B))
            Tcw = np.vstack([synth.invert_pose(seq.T_world_cam[n]), [0, 0, 0, 1]])
            Twr = np.vstack([T_world_ref, [0, 0, 0, 1]])
            T_curr_ref = (Tcw @ Twr)[:3]
            live, pm, d, half, n_steps = seed_runs(mu, sig2, aa, bb, seq.K, T_curr_ref, eps)
            stat = {k: [0, 0, 0] for k in ("centre", "centroid", "two")}  # misses, units with a miss, worst
            n_units = n_unboxed = n_items = 0
            second_sizes = []
            for ty in range(H // TILE):
                for tx in range(W // TILE):
                    ys, xs = np.mgrid[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE]
                    m = live[ys, xs]
                    if not m.any():
                        continue
                    seeds = (ys[m] * W + xs[m]).ravel()
                    sid, px, py = unit_samples(pm, d, half, n_steps, seeds)
                    if len(px) == 0:
                        continue
                    # units: light frames -> one round each (256 items in seed order)
                    for u0 in range(0, len(px), 256):
                        ux, uy = px[u0:u0 + 256], py[u0:u0 + 256]
                        n_units += 1; n_items += len(ux)
                        tb = box_of(px, py)
                        if fits(*tb):
                            continue
                        ub = box_of(ux, uy)
                        if fits(*ub):
                            n_unboxed += 1
                            continue
                        n_unboxed += 1
                        # centre policy
                        wb = clamp_around(ub, (ub[0] + ub[2] + 1) / 2, (ub[1] + ub[3] + 1) / 2)
                        miss = ~inside(wb, ux, uy)
                        st = stat["centre"]; st[0] += miss.sum(); st[1] += bool(miss.any()); st[2] = max(st[2], miss.sum())
                        # centroid policy
                        wb = clamp_around(ub, ux.mean(), uy.mean())
                        miss = ~inside(wb, ux, uy)
                        st = stat["centroid"]; st[0] += miss.sum(); st[1] += bool(miss.any()); st[2] = max(st[2], miss.sum())
                        # two windows: main cut to WIN_CAP - second, second around the centroid of the misses
                        wb = clamp_around(ub, ux.mean(), uy.mean(), WIN_CAP - a.second)
                        miss = ~inside(wb, ux, uy)
                        if miss.any():
                            mb = box_of(ux[miss], uy[miss])
                            second_sizes.append(((mb[2] - mb[0] + 1) | 1) * (mb[3] - mb[1] + 1))
                            if not fits(*mb, cap=a.second):
                                mb = clamp_around(mb, ux[miss].mean(), uy[miss].mean(), a.second)
                            miss = miss & ~inside(mb, ux, uy)
                        st = stat["two"]; st[0] += miss.sum(); st[1] += bool(miss.any()); st[2] = max(st[2], miss.sum())
            ss = np.array(second_sizes) if second_sizes else np.array([0])
            print(f"{n:6d} | {n_units:5d} {n_unboxed:5d}   | " + " ".join(f"{100.0 * stat[k][0] / max(n_items, 1):6.2f}" for k in stat) + "   | " +
                  " ".join(f"{stat[k][1]:5d}" for k in stat) + "   | " + " ".join(f"{stat[k][2]:4d}" for k in stat) +
                  f"   | {int(np.percentile(ss, 50))} {int(np.percentile(ss, 90))}", flush=True)
        s.update(seq.images[n], seq.T_curr_world[n])


if __name__ == "__main__":
    main()
