"""GPU box: how fast does the union kernel walk its longest fronts when the rest of the GPU has nothing to do?
Renders the 256-iso Skull sweep at 1024^2 on bands of rows around a given image row (default 456: it holds the ray
whose front has the most rounds) and prints the kernel time per band height next to the whole frame's.
usage: python tools/union_band_time.py [row] [kernel id, default 0]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import volumerenderer_amd as vr  # noqa: E402
from volumerenderer_amd import capi  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = 1024
row = int(sys.argv[1]) if len(sys.argv) > 1 else 456
kernel = int(sys.argv[2]) if len(sys.argv) > 2 else 0
data = vr.Data.loadUCD(os.path.join(ROOT, "tests/data/Skull.vol.gz"))
pos, forw, up = vr.main_camera(data.resulution)
L = capi.lib()
with vr.Renderer(0) as r:
    r.upload_volume(data)
    r.build_kdt("AlwaysCenter")
    r.set_kernel(kernel)
    view = vr.raytrace._view(pos, forw, up, W, W, 2.0)
    d_rgb = torch.empty((256, W, W, 3), dtype=torch.uint8, device="cuda")
    d_stats = torch.zeros((256, 7), dtype=torch.int64, device="cuda")
    for band in (1, 2, 8, 32, 128, 1024):
        r0 = max(0, min(W - band, row - band // 2))
        times, spans = [], []
        for i in range(8):
            if i == 3:
                capi.check(L.vr_set_timing(r.handle, 1), "timing")
            capi.check(L.vr_render_device(r.handle, C.byref(view), 0, 256, r0, r0 + band, d_rgb.data_ptr(), d_stats.data_ptr(), None, None), "render")
            capi.check(L.vr_sync(r.handle), "sync")
            if i >= 3:
                ms, n = C.c_double(), C.c_uint32()
                capi.check(L.vr_timing_read(r.handle, C.byref(ms), C.byref(n)), "read")
                times.append(ms.value / max(1, n.value))
                si = capi.SweepInfo()
                if L.vr_sweep_info_read(r.handle, C.byref(si)) == capi.VR_OK:
                    spans.append((si.tick_end - si.tick_begin) / 1e5)
        capi.check(L.vr_set_timing(r.handle, 0), "timing")
        st = d_stats.cpu().numpy()
        print("rows %4d..%4d: kernel %.3f ms (min of 5; span by the GPU clock %.3f), items per lane %d, rays in the volume %d, moves %d, errors %d" % (
            r0, r0 + band, min(times), min(spans) if spans else 0.0, si.items_per_lane, band * W - int(st[0, 3]), int(st[:, 5].sum()), int(st[:, 6].sum())), flush=True)
