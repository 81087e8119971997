#!/usr/bin/env python3
"""GPU box: per-launch spans of the sweep kernel (its own report) for one workload, launches synchronised one by one and
back to back; run once per VR_SWEEP_ITEMS value to see how far single launches scatter.  usage: sweep_spans.py [volume] [n]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import volumerenderer_amd as vr  # noqa: E402
from volumerenderer_amd import capi  # noqa: E402
from volumerenderer_amd import datasets  # noqa: E402

volume = sys.argv[1] if len(sys.argv) > 1 else "skull"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
data = {"skull": lambda: vr.Data.loadUCD(os.path.join(ROOT, "tests", "data", "Skull.vol.gz")),
        "c60": lambda: vr.Data.loadUCD(os.path.join(ROOT, "tests", "data", "C60.vol")),
        "standin256": lambda: vr.Data(datasets.bonsai_standin())}[volume]()
pos, forw, up = vr.main_camera(data.resulution)
L = capi.lib()
r = vr.Renderer(0)
r.upload_volume(data)
r.build_kdt("AlwaysCenter")
view = vr.raytrace._view(pos, forw, up, 1024, 1024, 2.0)
rgb = torch.empty((256, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
stats = torch.zeros((256, 7), dtype=torch.int64, device="cuda")
stream = torch.cuda.Stream()
sp = C.c_void_p(stream.cuda_stream)


def launch():
    capi.check(L.vr_render_device(r.handle, C.byref(view), 0, 256, 0, 1024, rgb.data_ptr(), stats.data_ptr(), None, sp), "render")


spans, tails = [], []
for _ in range(n):
    launch()
    torch.cuda.synchronize()
    si = capi.SweepInfo()
    capi.check(L.vr_sweep_info_read(r.handle, C.byref(si)), "info")
    spans.append((si.tick_end - si.tick_begin) / 1e5)
    tails.append((si.tick_end - min(si.tick_queue_empty, si.tick_end)) / max(1, si.tick_end - si.tick_begin))
t0 = time.perf_counter()
for _ in range(n):
    launch()
torch.cuda.synchronize()
b2b = (time.perf_counter() - t0) / n * 1e3
s = np.array(spans[2:])
print("%s items=%s (%d, settled %d): one by one: span min %.3f median %.3f max %.3f ms, tail %.2f | back to back %.3f ms per step" % (
    volume, os.environ.get("VR_SWEEP_ITEMS", "tuned"), si.items_per_lane, si.cut_settled, s.min(), np.median(s), s.max(), float(np.median(tails)), b2b))
