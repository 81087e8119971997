"""How much of a single frame's kernel time is the critical path of its longest rays?
Renders one iso value of Skull at W x W, finds the rows that hold the rays with the most node checks, and times
the traversal kernel on bands of rows around them (a band of 8 rows has ~1 ray per 32 lanes of the grid: its time
is the latency of its longest ray plus launch and LDS fill), next to the whole frame.
usage: python tools/critical_path.py [width] [iso]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import volumerenderer_amd as vr  # noqa: E402
from volumerenderer_amd import capi  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iso = int(sys.argv[2]) if len(sys.argv) > 2 else 80
data = vr.Data.loadUCD(os.path.join(ROOT, "tests/data/Skull.vol.gz"))
pos, forw, up = vr.main_camera(data.resulution)
L = capi.lib()


def timed(r, view, r0, r1, reps=20):
    d_rgb = torch.empty((1, r1 - r0, W, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        capi.check(L.vr_render_device(r.handle, C.byref(view), iso, 1, r0, r1, d_rgb.data_ptr(), None, None, None), "render")
    capi.check(L.vr_sync(r.handle), "sync")
    capi.check(L.vr_set_timing(r.handle, 1), "timing")
    for _ in range(reps):
        capi.check(L.vr_render_device(r.handle, C.byref(view), iso, 1, r0, r1, d_rgb.data_ptr(), None, None, None), "render")
    ms, n = C.c_double(), C.c_uint32()
    capi.check(L.vr_timing_read(r.handle, C.byref(ms), C.byref(n)), "read")
    capi.check(L.vr_set_timing(r.handle, 0), "timing")
    return ms.value / n.value


with vr.Renderer(0) as r:
    r.upload_volume(data)
    r.build_kdt("AlwaysCenter")
    out = r.renderSweep(pos, forw, up, W, W, 2.0, iso, 1, taps=True)
    checked = out["checked"][0]
    view = vr.raytrace._view(pos, forw, up, W, W, 2.0)
    print("Skull %dx%d iso %d: rays in volume %d, checks mean %.1f (of rays in the volume), max %d" % (
        W, W, iso, int((checked > 0).sum()), checked[checked > 0].mean(), checked.max()))
    print("whole frame: kernel %.3f ms" % timed(r, view, 0, W))
    rows = np.argsort(checked.max(axis=1))[::-1]
    for y in rows[:3]:
        for band in (1, 8, 64):
            r0 = max(0, min(W - band, int(y) - band // 2))
            print("rows %4d..%4d (max checks in band %5d, rays in volume %6d): kernel %.3f ms" % (
                r0, r0 + band, checked[r0:r0 + band].max(), int((checked[r0:r0 + band] > 0).sum()), timed(r, view, r0, r0 + band)))
    # how the per-ray work is distributed
    c = np.sort(checked[checked > 0])[::-1]
    print("share of all checks in the longest 1%% / 10%% of the rays: %.2f / %.2f" % (c[:len(c) // 100].sum() / c.sum(), c[:len(c) // 10].sum() / c.sum()))
