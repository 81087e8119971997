#!/usr/bin/env python3
"""GPU box, measuring build (tools/tune_build.sh items -DVR_MEASURE -DVR_SWEEP_MAX_BLOCKS=256): how full is the machine over
one launch of the union kernel?  Renders the 256-iso sweep at 1024^2 until the cut has settled, dumps the per-item table of
the last launch (start and duration of every (ray, block of iso values) item) and prints, per twentieth of the launch span,
how many of the 262 144 lanes hold an item, and when the launch falls below 90 / 50 / 10 % of them.

    VR_LIB_PATH=volumerenderer_amd/tune/libvr_items.so python tools/launch_occupancy.py [skull|c60|standin256] [out.txt]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import volumerenderer_amd as vr
from volumerenderer_amd import capi, datasets

volume = sys.argv[1] if len(sys.argv) > 1 else "skull"
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "launch_occupancy_%s.txt" % volume)
W = 1024
data = {"skull": lambda: vr.Data.loadUCD(os.path.join(ROOT, "tests", "data", "Skull.vol.gz")),
        "c60": lambda: vr.Data.loadUCD(os.path.join(ROOT, "tests", "data", "C60.vol")),
        "standin256": lambda: vr.Data(datasets.bonsai_standin())}[volume]()
pos, forw, up = vr.main_camera(data.resulution)
L = capi.lib()
r = vr.Renderer(0); r.upload_volume(data); r.build_kdt("AlwaysCenter")
view = vr.raytrace._view(pos, forw, up, W, W, 2.0)
rgb = torch.empty((256, W, W, 3), dtype=torch.uint8, device="cuda")
dump = "/tmp/vr_items_occ.bin"
N = 12
for i in range(N):
    if i == N - 1: os.environ["VR_MEASURE_DUMP"] = dump
    capi.check(L.vr_render_device(r.handle, C.byref(view), 0, 256, 0, W, rgb.data_ptr(), None, None, None), "render")
    capi.check(L.vr_sync(r.handle), "sync")
si = capi.SweepInfo(); capi.check(L.vr_sweep_info_read(r.handle, C.byref(si)), "info")
raw = np.fromfile(dump, dtype=np.uint32); os.remove(dump)
n_rays, n_blocks, block_isos, iso_begin = (int(x) for x in raw[:4])
n = n_rays * n_blocks
rounds = raw[4:4 + n].astype(np.int64)
ticks = raw[4 + n:4 + 2 * n].astype(np.int64)
begin = raw[4 + 2 * n:4 + 3 * n].astype(np.int64) - 1
end = begin + ticks
span = int(end.max())
lines = []
def say(s=""):
    print(s, flush=True); lines.append(s)
say("# %s sweep 0..255 at %dx%d, launch %d of %d on the context: %d rays x %d blocks of %d iso values, items per lane %d, tiles longest first %d" % (
    volume, W, W, N, N, n_rays, n_blocks, block_isos, si.items_per_lane, si.tiles_longest_first))
say("# last item ends %.1f us after the first wave began; lanes on the machine 262144; lane time in items %.1f %% of lanes x span" % (
    span / 100.0, 100.0 * ticks.sum() / (262144.0 * span)))
# lanes in flight at time t: items begun - items ended
ts = np.linspace(0, span, 21)
b_sorted = np.sort(begin); e_sorted = np.sort(end)
for k in range(20):
    t = 0.5 * (ts[k] + ts[k + 1])
    fl = np.searchsorted(b_sorted, t, side="right") - np.searchsorted(e_sorted, t, side="right")
    say("  %5.1f %% of the span (%6.1f us): %6d lanes hold an item (%.1f %%)" % (100.0 * t / span, t / 100.0, fl, 100.0 * fl / 262144))
grid = np.arange(0, span + 1)
fl = np.searchsorted(b_sorted, grid, side="right") - np.searchsorted(e_sorted, grid, side="right")
peak_t = int(np.argmax(fl))
for frac in (0.9, 0.5, 0.1):
    below = np.nonzero((grid > peak_t) & (fl < frac * 262144))[0]
    say("below %2.0f %% of the lanes from %.1f us on (%.1f %% of the span left)" % (100 * frac, below[0] / 100.0 if below.size else span / 100.0,
        100.0 * (span - (below[0] if below.size else span)) / span))
# what is in flight at the end
late = np.argsort(end)[::-1][:8]
for f in late:
    ray, blk = divmod(int(f), n_blocks)
    say("  ends last: ray %6d block %2d  length %4d  began %6.1f us  took %6.1f us (%.2f us per unit of length)" % (
        ray, blk, rounds[f], begin[f] / 100.0, ticks[f] / 100.0, ticks[f] / 100.0 / max(1, rounds[f])))
for f in np.argsort(ticks)[::-1][:4]:
    ray, blk = divmod(int(f), n_blocks)
    say("  takes longest: ray %6d block %2d  length %4d  began %6.1f us  took %6.1f us (%.2f us per unit of length)" % (
        ray, blk, rounds[f], begin[f] / 100.0, ticks[f] / 100.0, ticks[f] / 100.0 / max(1, rounds[f])))
# what could another order of the tiles gain?  List scheduling of (tile, block) units on 4096 wave slots; a unit lasts as long as
# its longest item did in this launch.  Orders: as the launch took them, by the tile's measured time, by its longest item in rounds.
import heapq
n_tiles = (n_rays + 63) // 64
pad = n_tiles * 64 - n_rays
def per_unit(a):
    a = np.concatenate([a.reshape(n_rays, n_blocks), np.zeros((pad, n_blocks), a.dtype)]).reshape(n_tiles, 64, n_blocks)
    return a.max(axis=1)                     # [tile][block]
u_ticks, u_rounds = per_unit(ticks), per_unit(rounds)
u_begin = np.concatenate([begin.reshape(n_rays, n_blocks), np.full((pad, n_blocks), 1 << 40)]).reshape(n_tiles, 64, n_blocks).min(axis=1)
def makespan(tile_order):
    slots = [0] * 4096
    heapq.heapify(slots)
    end = 0
    for t in tile_order:
        for b in range(n_blocks):
            at = heapq.heappop(slots)
            fin = at + int(u_ticks[t, b])
            end = max(end, fin)
            heapq.heappush(slots, fin)
    return end / 100.0
say("list scheduling of the launch's (tile, block) units on 4096 waves, a unit as long as its longest item was (us):")
say("  bounds: work / 4096 = %.0f, longest unit = %.0f" % (u_ticks.sum() / 4096 / 100.0, u_ticks.max() / 100.0))
say("  in the order the launch began them: %.0f" % makespan(np.argsort(u_begin.min(axis=1), kind="stable")))
say("  tiles by their longest unit's time, longest first: %.0f" % makespan(np.argsort(-u_ticks.max(axis=1), kind="stable")))
say("  tiles by their longest item's rounds, longest first: %.0f" % makespan(np.argsort(-u_rounds.max(axis=1), kind="stable")))
say("  tiles by the sum of their units' times, longest first: %.0f" % makespan(np.argsort(-u_ticks.sum(axis=1), kind="stable")))
def makespan_units(order):
    slots = [0] * 4096
    heapq.heapify(slots)
    end = 0
    flat = u_ticks.reshape(-1)
    for u in order:
        at = heapq.heappop(slots)
        fin = at + int(flat[u])
        end = max(end, fin)
        heapq.heappush(slots, fin)
    return end / 100.0
say("  units by their own time, longest first (an order per (tile, block) instead of per tile): %.0f" % makespan_units(np.argsort(-u_ticks.reshape(-1), kind="stable")))
say("  units by their own rounds, longest first: %.0f" % makespan_units(np.argsort(-u_rounds.reshape(-1), kind="stable")))
open(out_path, "w").write("\n".join(lines) + "\n")
