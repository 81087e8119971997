#!/usr/bin/env python3
"""What predicts the length of a tile of the union kernel before anything has been rendered?
step 1, GPU box, measuring build:   VR_LIB_PATH=volumerenderer_amd/tune/libvr_items.so python tools/first_call_study.py dump [volume]
        -> gpurun_out/ray_rounds_<volume>.npz: per ray of the list its pixel and the rounds of each of its items (cut: 3 items per lane)
step 2, CPU:                        python tools/first_call_study.py eval [volume]
        -> candidate predictors computed per ray from the camera and the volume alone, per tile of 64 rays the largest value, against the
           tile's measured length: rank correlation, where the longest 5 % of the tiles end up in the predicted order, and the makespan
           of a greedy schedule of the tiles' items in that order (the order's cost in launch time, in rounds)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
W = 1024


def load_volume(volume):
    import volumerenderer_amd as vr
    from volumerenderer_amd import datasets
    if volume == "skull":
        return vr.Data.loadUCD(os.path.join(ROOT, "tests", "data", "Skull.vol.gz"))
    if volume == "c60":
        return vr.Data.loadUCD(os.path.join(ROOT, "tests", "data", "C60.vol"))
    return vr.Data(datasets.bonsai_standin())


def dump(volume):
    import item_lengths
    d = item_lengths.run_cut(volume, 3, "/tmp/vr_items_fc.bin")
    out = os.path.join(ROOT, "gpurun_out", "ray_rounds_%s.npz" % volume)
    np.savez_compressed(out, pix=d["pix"].astype(np.uint32), rounds=d["rounds"].astype(np.uint16), span_ms=d["span_ms"])
    print("%s: %d rays, %d blocks per ray, launch %.3f ms -> %s" % (volume, d["rounds"].shape[0], d["rounds"].shape[1], d["span_ms"], out))


def rays(data, pix):
    """direction and entry t of the rays, as vr_make_ray computes them (float64 here: good enough for a predictor)"""
    import volumerenderer_amd as vr
    pos, forw, up = vr.main_camera(data.resulution)
    pos, forw, up = (np.array(v, dtype=np.float64) for v in (pos, forw, up))
    right = np.cross(forw, up)
    xi, yi = (pix % W).astype(np.float64), (pix // W).astype(np.float64)
    ps = 2.0 / W
    d = forw[None, :] + up[None, :] * ((yi - W // 2) * ps)[:, None] + right[None, :] * ((xi - W // 2) * ps)[:, None]
    d /= np.linalg.norm(d, axis=1)[:, None]
    dim = np.array(data.resulution, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t_in = np.where(d >= 0, (0.0 - pos) / d, (dim - pos) / d)
        t_out = np.where(d >= 0, (dim - pos) / d, (0.0 - pos) / d)
    t0 = np.nanmax(np.where(np.isfinite(t_in), t_in, -np.inf), axis=1) + 0.001
    t1 = np.nanmin(np.where(np.isfinite(t_out), t_out, np.inf), axis=1)
    return pos, d, t0, np.maximum(t1, t0)


def coarse_grid(vox, cell):
    """min and max of the volume per cell of cell^3 voxels (corner values of the cells along the border included)"""
    nx, ny, nz = vox.shape
    gx, gy, gz = (nx + cell - 1) // cell, (ny + cell - 1) // cell, (nz + cell - 1) // cell
    pad = np.pad(vox, ((0, gx * cell - nx + 1), (0, gy * cell - ny + 1), (0, gz * cell - nz + 1)), mode="edge")
    lo = np.full((gx, gy, gz), 255, dtype=np.uint8)
    hi = np.zeros((gx, gy, gz), dtype=np.uint8)
    for dx in range(cell + 1):
        for dy in range(cell + 1):
            for dz in range(cell + 1):
                v = pad[dx:dx + gx * cell:cell, dy:dy + gy * cell:cell, dz:dz + gz * cell:cell]
                lo = np.minimum(lo, v)
                hi = np.maximum(hi, v)
    return lo, hi


def schedule(lengths_by_tile, order, n_blocks_per_tile, n_slots=4096):
    """greedy: the tiles' items (one per block of iso values, each as long as given) go to the slot that is free first, in `order`"""
    import heapq
    free = [0.0] * n_slots
    heapq.heapify(free)
    end = 0.0
    for t in order:
        for b in range(n_blocks_per_tile):
            s = heapq.heappop(free)
            e = s + float(lengths_by_tile[t, b])
            end = max(end, e)
            heapq.heappush(free, e)
    return end


def evaluate(volume):
    from scipy.stats import spearmanr
    z = np.load(os.path.join(ROOT, "gpurun_out", "ray_rounds_%s.npz" % volume))
    pix, rounds = z["pix"], z["rounds"].astype(np.float64)
    data = load_volume(volume)
    vox = np.asarray(data.data).reshape(data.resulution)
    pos, d, t0, t1 = rays(data, pix)
    n = len(pix)
    n_tiles = (n + 63) // 64
    pad = n_tiles * 64 - n
    tile_len = np.pad(rounds, ((0, pad), (0, 0))).reshape(n_tiles, 64, -1).max(axis=1)      # [tile, block]
    tile_max = tile_len.max(axis=1)
    print("%s: %d rays, %d tiles, longest item %d rounds, launch %.3f ms" % (volume, n, n_tiles, tile_max.max(), float(z["span_ms"])))
    preds = {"chord": t1 - t0}
    n_s = 32
    ts = t0[:, None] + (t1 - t0)[:, None] * ((np.arange(n_s) + 0.5) / n_s)[None, :]
    for cell in (4, 8, 16):
        lo, hi = coarse_grid(vox, cell)
        g = np.array(lo.shape)
        width = (hi.astype(np.int32) - lo.astype(np.int32)).astype(np.float64)
        acc_w = np.zeros(n)
        acc_n = np.zeros(n)
        acc_sq = np.zeros(n)
        for s in range(n_s):
            p = pos[None, :] + d * ts[:, s][:, None]
            c = np.clip((p // cell).astype(np.int64), 0, g - 1)
            w = width[c[:, 0], c[:, 1], c[:, 2]]
            acc_w += w
            acc_n += w > 0
            acc_sq += np.sqrt(w)
        seg = (t1 - t0) / n_s
        preds["cell %2d: chord x mean range width" % cell] = acc_w * seg
        preds["cell %2d: chord inside cells with any range" % cell] = acc_n * seg
        preds["cell %2d: chord x mean sqrt(width)" % cell] = acc_sq * seg
    dim = np.array(vox.shape)
    vals = np.empty((n, n_s), dtype=np.int32)  # the voxel under each sample point: how much the density varies along the ray
    for s in range(n_s):
        c = np.clip((pos[None, :] + d * ts[:, s][:, None]).astype(np.int64), 0, dim - 1)
        vals[:, s] = vox[c[:, 0], c[:, 1], c[:, 2]]
    preds["total variation of the density along the ray"] = np.abs(np.diff(vals, axis=1)).sum(axis=1).astype(np.float64)
    preds["chord x (1 + share of samples where it changes)"] = (t1 - t0) * (1.0 + (np.diff(vals, axis=1) != 0).sum(axis=1) / n_s)
    top = np.argsort(tile_max)[::-1][: max(1, n_tiles // 20)]
    ideal = schedule(tile_len, np.argsort(tile_max)[::-1], tile_len.shape[1])
    listed = schedule(tile_len, np.arange(n_tiles), tile_len.shape[1])
    print("  schedule of the measured items on 4096 waves: longest first %.0f rounds, list order %.0f" % (ideal, listed))
    for name, p in preds.items():
        tp = np.pad(p, (0, pad)).reshape(n_tiles, 64).max(axis=1)
        order = np.argsort(tp)[::-1]
        rank = np.empty(n_tiles, dtype=np.int64)
        rank[order] = np.arange(n_tiles)
        rho = spearmanr(tp, tile_max).correlation
        print("  %-44s rank correlation %.3f; the longest 5 %% of the tiles lie at positions up to %4d (mean %4.0f) of %d; schedule %.0f rounds" % (
            name, rho, rank[top].max(), rank[top].mean(), n_tiles, schedule(tile_len, order, tile_len.shape[1])))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "eval"
    vols = sys.argv[2:] or ["skull", "c60", "standin256"]
    for v in vols:
        (dump if what == "dump" else evaluate)(v)
