#!/usr/bin/env python3
"""GPU box, measuring build (tools/tune_build.sh items -DVR_MEASURE -DVR_SWEEP_MAX_BLOCKS=256): the rounds of ONE
(ray, block) item of the union kernel's Skull sweep, one line per round: round, us since the launch began, lanes of its wave
holding a front, lanes descending, whether the traced lane is popping.
usage: VR_LIB_PATH=.../libvr_items.so VR_SWEEP_ITEMS=3 python tools/item_trace.py ray block blocks_per_ray out.txt [skull|c60|standin256]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import volumerenderer_amd as vr
from volumerenderer_amd import capi, datasets
ray, blk, nb, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
os.environ["VR_MEASURE_TRACE"] = "%d,%d,%d" % (ray, blk, nb)
os.environ["VR_MEASURE_TRACE_OUT"] = out
volume = sys.argv[5] if len(sys.argv) > 5 else "skull"
data = {"skull": lambda: vr.Data.loadUCD(os.path.join(ROOT, "tests", "data", "Skull.vol.gz")),
        "c60": lambda: vr.Data.loadUCD(os.path.join(ROOT, "tests", "data", "C60.vol")),
        "standin256": lambda: vr.Data(datasets.bonsai_standin())}[volume]()
pos, forw, up = vr.main_camera(data.resulution)
L = capi.lib()
r = vr.Renderer(0); r.upload_volume(data); r.build_kdt("AlwaysCenter")
view = vr.raytrace._view(pos, forw, up, 1024, 1024, 2.0)
rgb = torch.empty((256, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
for i in range(3):
    if i == 2: os.environ["VR_MEASURE_DUMP"] = "/tmp/vr_items_trace.bin"
    capi.check(L.vr_render_device(r.handle, C.byref(view), 0, 256, 0, 1024, rgb.data_ptr(), None, None, None), "render")
    capi.check(L.vr_sync(r.handle), "sync")
os.remove("/tmp/vr_items_trace.bin")
rows = [l.split() for l in open(out)]
print("rounds traced:", len(rows))
prev = None
for row in rows:
    t = float(row[1])
    print("round %4s  t %8.1f us  dt %6.2f  fronts in wave %2s  descending %2s  popping %s" % (row[0], t, t - prev if prev is not None else 0.0, row[2], row[3], row[4]))
    prev = t
