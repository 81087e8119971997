#!/usr/bin/env python3
"""GPU box: one of the views in which tools/stress_union.py counted `errors` (a ray past the bound on its checks): which pixel
is it, and what do the first and the last recorded steps of its literal walk (vr_trace_ray, the reference's loop) look like?"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import volumerenderer_amd as vr
from volumerenderer_amd import capi
L = capi.lib()
data = vr.Data.loadUCD(os.path.join(ROOT, "tests", "data", "Skull.vol.gz"))
r = vr.Renderer(0); r.upload_volume(data); r.build_kdt("AlwaysCenter")
pos = (304.1269991564012, -27.44165395375535, 144.00327369163497)
forw = (-0.9173649640094544, 0.39454469731446484, -0.05268780341758113)
up = (0.37874756115297314, 0.8244890730930323, -0.4204379303427166)
W, H, iso = 608, 104, 217
v = vr.raytrace._view(pos, forw, up, W, H, 2.4901439500564786)
d_rgb = torch.zeros((1, H, W, 3), dtype=torch.uint8, device="cuda")
d_chk = torch.zeros((1, H, W), dtype=torch.int32, device="cuda")
d_stats = torch.zeros((1, 7), dtype=torch.int64, device="cuda")
taps = capi.Taps(None, None, d_chk.data_ptr(), None)
torch.cuda.synchronize()
capi.check(L.vr_render_device(r.handle, C.byref(v), iso, 1, 0, H, d_rgb.data_ptr(), d_stats.data_ptr(), C.byref(taps), None), "render")
capi.check(L.vr_sync(r.handle), "sync")
chk = d_chk.cpu().numpy()[0].astype(np.uint32)
y, x = np.unravel_index(int(np.argmax(chk)), chk.shape)
print("stats", d_stats.cpu().numpy()[0].tolist(), "largest per-pixel check count", int(chk.max()), "at pixel", (int(x), int(y)), flush=True)
cap = 60000
steps = (capi.RayStep * cap)(); rep = capi.RayReport()
capi.check(L.vr_trace_ray(r.handle, C.byref(v), iso, int(x), int(y), steps, cap, C.byref(rep)), "trace_ray")
print("origin", list(rep.origin), "dir", list(rep.dir), "t_entry", rep.t_entry, "t_final", rep.t_final, "flag", rep.flag, "steps", rep.n_steps, "stored", rep.n_stored)
n = rep.n_stored
for i in list(range(0, 6)) + list(range(n - 24, n)):
    s = steps[i]
    print("%6d t=%.9g p=(%.7g %.7g %.7g) node %d size %d range %d..%d action %d" % (i, s.t, s.p[0], s.p[1], s.p[2], s.node, s.size, s.min, s.max, s.action))
