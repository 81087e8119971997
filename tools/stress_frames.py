#!/usr/bin/env python3
"""GPU box: random views, 1 .. 7 iso values per call, the pipeline kernel with one traversal per iso value (ray list, queues, the waves'
first chunks by rank) against the per-pixel kernel (no list, no queue): frames and counters must be identical.
usage: python tools/stress_frames.py [seconds] [seed]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import volumerenderer_amd as vr
from volumerenderer_amd import capi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
L = capi.lib()
vols = {"skull": vr.Data.loadUCD(os.path.join(ROOT, "tests", "data", "Skull.vol.gz")), "c60": vr.Data.loadUCD(os.path.join(ROOT, "tests", "data", "C60.vol"))}
rnd = {}
for name, data in vols.items():
    r = vr.Renderer(0); r.upload_volume(data); r.build_kdt("AlwaysCenter"); rnd[name] = r

def render(r, v, W, b, n, r0, r1, variant):
    r.set_kernel(variant)
    d_rgb = torch.full((n, r1 - r0, W, 3), 5, dtype=torch.uint8, device="cuda")
    d_stats = torch.zeros((n, 7), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    capi.check(L.vr_render_device(r.handle, C.byref(v), b, n, r0, r1, d_rgb.data_ptr(), d_stats.data_ptr(), None, None), "render")
    capi.check(L.vr_sync(r.handle), "sync")
    r.set_kernel(capi.KERNEL_DEFAULT)
    return d_rgb, d_stats.cpu().numpy()

t_end = time.time() + budget
n_views = n_bad = 0
while time.time() < t_end:
    name = ["skull", "c60"][int(rng.integers(2))]
    data, r = vols[name], rnd[name]
    res = np.array(data.resulution, dtype=np.float64)
    centre = res / 2.0
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    pos = centre + d * float(rng.uniform(0.7, 1.6)) * float(res.max())
    forw = centre + rng.normal(size=3) * 0.15 * res - pos; forw /= np.linalg.norm(forw)
    up = np.cross(forw, rng.normal(size=3)); up /= np.linalg.norm(up)
    W, H = int(rng.integers(17, 1300)), int(rng.integers(9, 1100))
    n = int(rng.integers(1, 8)); b = int(rng.integers(0, 257 - n))
    r0 = int(rng.integers(0, H // 3)); r1 = int(rng.integers(r0 + 1, H + 1)) if rng.random() < 0.5 else H
    vp = float(rng.uniform(0.8, 3.0))
    v = vr.raytrace._view(tuple(pos), tuple(forw), tuple(up), W, H, vp)
    ref = render(r, v, W, b, n, r0, r1, capi.KERNEL_RESUME)
    for i in range(2):  # the second call uses the ray list the first one left
        got = render(r, v, W, b, n, r0, r1, capi.KERNEL_DEFAULT)
        if not (got[1] == ref[1]).all() or not bool(torch.equal(got[0], ref[0])):
            n_bad += 1
            ds = np.argwhere(got[1] != ref[1])
            dp = (got[0] != ref[0]).any(dim=3).nonzero()
            print("DIFFERS (call %d): %s %dx%d rows %d..%d iso %d+%d; counters differ at %s: %s vs %s; %d pixels differ, first %s; camera %r %r %r vp %r" % (
                i, name, W, H, r0, r1, b, n, ds[:6].tolist(), got[1][ds[:6, 0], ds[:6, 1]].tolist() if len(ds) else [], ref[1][ds[:6, 0], ds[:6, 1]].tolist() if len(ds) else [],
                int(dp.shape[0]), dp[:3].tolist(), tuple(float(x) for x in pos), tuple(float(x) for x in forw), tuple(float(x) for x in up), vp), flush=True)
    n_views += 1
print("done: %d random views, %d differences" % (n_views, n_bad))
sys.exit(1 if n_bad else 0)
