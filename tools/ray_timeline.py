#!/usr/bin/env python3
"""GPU box, measuring build (tools/tune_build.sh items -DVR_MEASURE ...): where do the cycles of ONE walk go?
Finds the ray with the most node checks of a Skull frame (default 1024^2, iso 80), walks it alone -- one lane of one wave,
nothing else on the chip -- with the pipeline kernel's lane code and a cycle stamp around every step, and prints the budget:
node fetches (from the LDS copy of the top levels / from global memory), the rest of a descend level, MOVEs, popped levels.
usage: VR_LIB_PATH=volumerenderer_amd/tune/libvr_items.so python tools/ray_timeline.py [width] [iso]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import volumerenderer_amd as vr  # noqa: E402
from volumerenderer_amd import capi  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iso = int(sys.argv[2]) if len(sys.argv) > 2 else 80
data = vr.Data.loadUCD(os.path.join(ROOT, "tests", "data", "Skull.vol.gz"))
pos, forw, up = vr.main_camera(data.resulution)
L = capi.lib()
L.vr_measure_ray_timeline.restype = C.c_int
with vr.Renderer(0) as r:
    r.upload_volume(data)
    r.build_kdt("AlwaysCenter")
    out = r.renderSweep(pos, forw, up, W, W, 2.0, iso, 1, taps=True)
    checked = out["checked"][0]
    y, x = np.unravel_index(int(np.argmax(checked)), checked.shape)
    view = vr.raytrace._view(pos, forw, up, W, W, 2.0)
    res = (C.c_ulonglong * 25)()
    lat = (C.c_uint32 * 4096)()
    best = None
    for _ in range(5):  # the first run warms the caches; the shortest of the rest is reported
        capi.check(L.vr_measure_ray_timeline(r.handle, C.byref(view), C.c_uint8(iso), C.c_uint32(int(x)), C.c_uint32(int(y)), res, lat), "timeline")
        cur = (list(res), list(lat))
        if best is None or cur[0][20] < best[0][20]:
            best = cur
res, lat = best
names = ["node fetch, LDS copy of the top levels", "node fetch, global memory", "rest of a descend level (range test, child, box, push)",
         "MOVE (3 plane hits, point, containment, escape loop)", "popped level", "step into the sibling (end of a pop)",
         "loop bookkeeping between steps"]
total = res[20]
print("Skull %dx%d iso %d, pixel (%d, %d): %d node checks (tap: %d), %d MOVEs, flag %d -- one lane of one wave alone on the GPU" % (
    W, W, iso, x, y, res[21], int(checked[y, x]), res[22], res[23]))
print("whole walk: %d cycles (stamps and waits at every step included) = %.1f us at 2.4 GHz" % (total, total / 2400.0))
acc = 0
for i, n in enumerate(names):
    cyc, cnt = res[2 * i], res[2 * i + 1]
    acc += cyc
    print("  %-56s %6d x %7.1f cycles = %9d (%4.1f %%)" % (n, cnt, cyc / max(1, cnt), cyc, 100.0 * cyc / total))
print("  accounted for: %.1f %% of the stamped walk" % (100.0 * acc / total))
print("  a stamp alone (wait for everything outstanding + read of the cycle counter, two in a row): %.1f cycles -- every line above holds one" % (res[16] / 16.0))
lv_l, lv_g, rest = res[0] / max(1, res[1]), res[2] / max(1, res[3]), res[4] / max(1, res[5])
print("one descend level = node fetch + rest: %.0f cycles with the node in LDS (fetch %.0f %%), %.0f cycles with the node in global memory "
      "(fetch %.0f %%)" % (lv_l + rest, 100 * lv_l / (lv_l + rest), lv_g + rest, 100 * lv_g / (lv_g + rest)))
work = sum(res[2 * i] for i in range(6))
print("the lane code's own steps (everything but the bookkeeping of this measurement): %d cycles = %.1f us" % (work, work / 2400.0))
print("the same walk with no stamps, the lane code called as the kernels call it (descend <= 6 levels, MOVE, pop <= 6 levels per turn): "
      "%d cycles = %.1f us for %d checks -- what one lane needs for this ray with nothing else to do" % (res[14], res[14] / 2400.0, res[15]))
n_lat = int(res[24])
l = np.array(lat[:n_lat], dtype=np.uint32)
cyc, glob, lvl = l & 0xfffff, (l >> 20) & 1, l >> 24
print("node fetches by tree level (first %d): count x mean cycles" % n_lat)
for d in range(int(lvl.max()) + 1):
    m = lvl == d
    if m.any():
        print("    level %2d (%s): %4d x %6.0f" % (d, "global" if glob[m].any() else "LDS", int(m.sum()), cyc[m].mean()))
print("physical node fetches: %d for %d reference-defined checks (the resume traversal does not re-read the path above the lowest "
      "ancestor that holds the new point)" % (res[1] + res[3], res[21]))
