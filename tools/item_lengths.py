#!/usr/bin/env python3
"""GPU box, measuring build (tools/tune_build.sh items -DVR_MEASURE -DVR_SWEEP_MAX_BLOCKS=256): how long is every
(ray, block of iso values) item of the union kernel, for several cuts of the iso range down to one iso value per item?

    VR_LIB_PATH=volumerenderer_amd/tune/libvr_items.so python tools/item_lengths.py [volume] [out.txt]

For every cut it renders the 256-iso sweep at 1024^2 three times in a process of its own (the cut is read from the
environment once per process; the third launch runs with the tile order of the second) and reads the kernel's per-item
table: rounds and clock ticks from the item's start to the end of its last front, and when it started.  Prints, per cut,
the launch span, the work (sum of rounds), the longest items, and for the hardest rays of the coarsest cut how their
longest item shrinks as the cut gets finer -- down to the single (ray, iso value) walk, which is the floor of any cut."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = 1024
INNER = r"""
import ctypes as C, os, sys
sys.path.insert(0, %(root)r)
import torch
import volumerenderer_amd as vr
from volumerenderer_amd import capi, datasets
volume = %(volume)r
data = {"skull": lambda: vr.Data.loadUCD(os.path.join(%(root)r, "tests", "data", "Skull.vol.gz")),
        "c60": lambda: vr.Data.loadUCD(os.path.join(%(root)r, "tests", "data", "C60.vol")),
        "standin256": lambda: vr.Data(datasets.bonsai_standin())}[volume]()
pos, forw, up = vr.main_camera(data.resulution)
L = capi.lib()
r = vr.Renderer(0); r.upload_volume(data); r.build_kdt("AlwaysCenter")
view = vr.raytrace._view(pos, forw, up, %(w)d, %(w)d, 2.0)
rgb = torch.empty((256, %(w)d, %(w)d, 3), dtype=torch.uint8, device="cuda")
stats = torch.zeros((256, 7), dtype=torch.int64, device="cuda")
for i in range(3):
    if i == 2: os.environ["VR_MEASURE_DUMP"] = %(dump)r
    capi.check(L.vr_render_device(r.handle, C.byref(view), 0, 256, 0, %(w)d, rgb.data_ptr(), stats.data_ptr(), None, None), "render")
    capi.check(L.vr_sync(r.handle), "sync")
si = capi.SweepInfo(); capi.check(L.vr_sweep_info_read(r.handle, C.byref(si)), "info")
print("SPAN %%.4f LPT %%d ERR %%d" %% ((si.tick_end - si.tick_begin) / 1e5, si.tiles_longest_first, int(stats[:, 6].sum().item())))
"""


def run_cut(volume, ipl, dump):
    env = dict(os.environ, VR_SWEEP_ITEMS=str(ipl))
    env.pop("VR_MEASURE_DUMP", None)
    out = subprocess.run([sys.executable, "-c", INNER % {"root": ROOT, "volume": volume, "w": W, "dump": dump}], env=env,
                         capture_output=True, text=True, timeout=280)
    if out.returncode != 0:
        raise SystemExit("cut %d failed:\n%s" % (ipl, out.stderr[-2000:]))
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("SPAN")][0].split()
    raw = np.fromfile(dump, dtype=np.uint32)
    os.remove(dump)
    n_rays, n_blocks, block_isos, iso_begin = (int(x) for x in raw[:4])
    n = n_rays * n_blocks
    rounds = raw[4:4 + n].reshape(n_rays, n_blocks)
    ticks = raw[4 + n:4 + 2 * n].reshape(n_rays, n_blocks)
    begin = raw[4 + 2 * n:4 + 3 * n].reshape(n_rays, n_blocks)
    pix = raw[4 + 3 * n:4 + 3 * n + n_rays]
    return {"span_ms": float(line[1]), "lpt": int(line[3]), "errors": int(line[5]), "n_blocks": n_blocks, "block_isos": block_isos,
            "rounds": rounds, "ticks": ticks, "begin": begin, "pix": pix}


def main():
    volume = sys.argv[1] if len(sys.argv) > 1 else "skull"
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "item_lengths_%s.txt" % volume)
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    lines = []

    def say(s=""):
        print(s, flush=True)
        lines.append(s)

    # items per lane -> blocks per ray = ceil(262144 * ipl / rays); Skull has 304 380 rays in the list
    cuts = [1, 3, 6, 12, 24, 48, 96, 300]
    res = {}
    say("# %s, 256 iso values at %dx%d: lengths of the union kernel's (ray, block of iso values) items per cut of the iso range" % (volume, W, W))
    say("# a round = at most 5 levels down, one MOVE, at most 7 levels up for every lane of the wave; ticks of the 100 MHz clock -> us")
    hardest = None
    for ipl in cuts:
        d = run_cut(volume, ipl, "/tmp/vr_items_%d.bin" % ipl)
        if d["n_blocks"] in [x["n_blocks"] for x in res.values()]:
            continue
        res[ipl] = d
        r, t, b = d["rounds"], d["ticks"], d["begin"]
        us = t / 100.0
        end_us = (b.astype(np.int64) - 1 + t) / 100.0
        flat = np.argsort(t, axis=None)[::-1][:6]
        say()
        say("cut: %d block(s) of %d iso values per ray (VR_SWEEP_ITEMS=%d): launch span %.3f ms (tiles longest first: %d, errors %d)" % (
            d["n_blocks"], d["block_isos"], ipl, d["span_ms"], d["lpt"], d["errors"]))
        say("  items %d, rounds: sum %.3e  mean %.1f  p50 %d  p90 %d  p99 %d  max %d;  item time us: mean %.0f p99 %.0f max %.0f;  us per round (all items) %.2f" % (
            r.size, r.sum(), r.mean(), np.percentile(r, 50), np.percentile(r, 90), np.percentile(r, 99), r.max(), us.mean(),
            np.percentile(us, 99), us.max(), us.sum() / max(1, r.sum())))
        say("  last item ends at %.0f us after the first wave's start" % end_us.max())
        for f in flat:
            ray, blk = divmod(int(f), d["n_blocks"])
            p = int(d["pix"][ray])
            say("    ray %6d pixel (%4d,%4d) iso %3d..%3d: %4d rounds, %7.1f us (%.2f us/round), started at %6.1f us" % (
                ray, p % W, p // W, blk * d["block_isos"], min(255, (blk + 1) * d["block_isos"] - 1), r[ray, blk], us[ray, blk],
                us[ray, blk] / max(1, r[ray, blk]), (b[ray, blk] - 1) / 100.0))
        tile_max = np.maximum.reduceat(r.max(axis=1), np.arange(0, r.shape[0], 64))
        say("  tiles of 64 rays by their longest item (rounds): mean %.1f; tiles within 10 %% / 25 %% / 50 %% of the longest: %d / %d / %d of %d" % (
            tile_max.mean(), (tile_max >= 0.9 * tile_max.max()).sum(), (tile_max >= 0.75 * tile_max.max()).sum(),
            (tile_max >= 0.5 * tile_max.max()).sum(), tile_max.size))
        if hardest is None:
            hardest = np.argsort(r.max(axis=1))[::-1][:8]
    say()
    say("# the hardest rays of the coarsest cut: longest item per cut (rounds / us) and the iso block it is in")
    hdr = "ray     pixel       " + "".join("%18s" % ("%d x %d iso" % (d["n_blocks"], d["block_isos"])) for d in res.values())
    say(hdr)
    for ray in hardest:
        p = int(next(iter(res.values()))["pix"][ray])
        row = "%6d (%4d,%4d) " % (ray, p % W, p // W)
        for d in res.values():
            blk = int(np.argmax(d["rounds"][ray]))
            row += "%18s" % ("%d/%.0f @%d" % (d["rounds"][ray, blk], d["ticks"][ray, blk] / 100.0, blk * d["block_isos"]))
        say(row)
    finest = list(res.values())[-1]
    if finest["block_isos"] == 1:
        say()
        say("# single (ray, iso value) fronts of the hardest ray: rounds per iso value (the floor of every cut is the largest of them)")
        ray = int(hardest[0])
        rr = finest["rounds"][ray]
        top = np.argsort(rr)[::-1][:12]
        say("ray %d: " % ray + ", ".join("iso %d: %d rounds / %.0f us" % (k, rr[k], finest["ticks"][ray, k] / 100.0) for k in top))
        say("sum over its 256 single-iso fronts: %d rounds; its one front for all 256 (coarsest cut): %d rounds" % (
            rr.sum(), next(iter(res.values()))["rounds"][ray].max()))
        allmax = finest["rounds"].max()
        say("longest single (ray, iso value) front of the whole sweep: %d rounds, %.0f us" % (allmax, finest["ticks"].max() / 100.0))
    with open(out_path, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
