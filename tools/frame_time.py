"""Times single-frame renders (one iso value per call, frames resident on the device): the reference's
ms/frame figure (src/main.zig:147-161) for this library.  usage: python tools/frame_time.py [width] [kernel variant]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import volumerenderer_amd as vr  # noqa: E402
from volumerenderer_amd import capi  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
data = vr.Data.loadUCD(os.path.join(ROOT, "tests/data/Skull.vol.gz"))
pos, forw, up = vr.main_camera(data.resulution)
L = capi.lib()
with vr.Renderer(0) as r:
    r.upload_volume(data)
    r.build_kdt("AlwaysCenter")
    if len(sys.argv) > 2:
        r.set_kernel(int(sys.argv[2]))
    view = vr.raytrace._view(pos, forw, up, W, W, 2.0)
    d_rgb = torch.empty((1, W, W, 3), dtype=torch.uint8, device="cuda")
    d_stats = torch.zeros((1, 7), dtype=torch.int64, device="cuda")
    for iso in (40, 80, 160):
        for _ in range(3):
            capi.check(L.vr_render_device(r.handle, C.byref(view), iso, 1, 0, W, d_rgb.data_ptr(), d_stats.data_ptr(), None, None), "render")
        capi.check(L.vr_sync(r.handle), "sync")
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            capi.check(L.vr_render_device(r.handle, C.byref(view), iso, 1, 0, W, d_rgb.data_ptr(), d_stats.data_ptr(), None, None), "render")
        capi.check(L.vr_sync(r.handle), "sync")
        dt = (time.perf_counter() - t0) / n
        print("Skull %dx%d iso %d: %.3f ms/frame (%.0f Mrays/s), %d node checks" % (W, W, iso, dt * 1e3, W * W / dt / 1e6, int(d_stats[0, 0])))
