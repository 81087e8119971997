"""Times vr_render_device for iso blocks of different length (frames resident on the device): how the time of
a sweep grows with the number of iso values.  usage: python tools/sweep_scaling.py [width] [kernel id]"""
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
    r.set_kernel(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    view = vr.raytrace._view(pos, forw, up, W, W, 2.0)
    d_rgb = torch.empty((256, W, W, 3), dtype=torch.uint8, device="cuda")
    d_stats = torch.zeros((256, 7), dtype=torch.int64, device="cuda")
    for n in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        b = 128 - n // 2
        for _ in range(16):  # lets the library settle its cut of the iso range for this workload
            capi.check(L.vr_render_device(r.handle, C.byref(view), b, n, 0, W, d_rgb.data_ptr(), d_stats.data_ptr(), None, None), "render")
            capi.check(L.vr_sync(r.handle), "sync")
        for rep in range(2):
            capi.check(L.vr_sync(r.handle), "sync")
            t0 = time.perf_counter()
            for _ in range(5):
                capi.check(L.vr_render_device(r.handle, C.byref(view), b, n, 0, W, d_rgb.data_ptr(), d_stats.data_ptr(), None, None), "render")
            capi.check(L.vr_sync(r.handle), "sync")
            dt = (time.perf_counter() - t0) / 5
        print("Skull %dx%d, %3d iso values from %3d: %.3f ms (%.3f ms per iso value, %.1f Grays/s)" % (W, W, n, b, dt * 1e3, dt * 1e3 / n, n * W * W / dt / 1e9))
