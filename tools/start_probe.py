#!/usr/bin/env python3
"""GPU box, probe builds (tools/tune_build.sh probe1 -DVR_START_PROBE=1, probe2 -DVR_START_PROBE=2): when, after the first
wave's start, does the first wave of a union launch have its first items, and when the last one?
usage: python tools/start_probe.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subprocess
INNER = r"""
import ctypes as C, os, sys
sys.path.insert(0, %r)
import torch
import volumerenderer_amd as vr
from volumerenderer_amd import capi
data = vr.Data.loadUCD(os.path.join(%r, "tests", "data", "Skull.vol.gz"))
pos, forw, up = vr.main_camera(data.resulution)
L = capi.lib()
r = vr.Renderer(0); r.upload_volume(data); r.build_kdt("AlwaysCenter")
view = vr.raytrace._view(pos, forw, up, 1024, 1024, 2.0)
rgb = torch.empty((256, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
for i in range(6):
    capi.check(L.vr_render_device(r.handle, C.byref(view), 0, 256, 0, 1024, rgb.data_ptr(), None, None, None), "render")
    capi.check(L.vr_sync(r.handle), "sync")
si = capi.SweepInfo(); capi.check(L.vr_sweep_info_read(r.handle, C.byref(si)), "info")
print("TICKS", si.tick_begin, si.tick_queue_empty, si.tick_end)
""" % (ROOT, ROOT)
for name in ("probe1", "probe2"):
    env = dict(os.environ, VR_LIB_PATH=os.path.join(ROOT, "volumerenderer_amd", "tune", "libvr_%s.so" % name), VR_SWEEP_ITEMS="3")
    out = subprocess.run([sys.executable, "-c", INNER], env=env, capture_output=True, text=True, timeout=200)
    line = [l for l in out.stdout.splitlines() if l.startswith("TICKS")]
    if not line:
        print(out.stderr[-1500:]); continue
    b, q, e = (int(x) for x in line[0].split()[1:])
    if name == "probe2": q = (~q) & ((1 << 64) - 1)
    print("%s: %s wave has its first items %.1f us after the first wave's start; launch span %.1f us" % (name, "first" if name == "probe1" else "last", (q - b) / 100.0, (e - b) / 100.0))
