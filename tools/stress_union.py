#!/usr/bin/env python3
"""GPU box (VR_STRESS_WIDE=1: a 320^3 volume on the wide node array; VR_STRESS_NOALLOC=1: the default selection through
vr_render_device_noalloc): random views of Skull, C60 and the 256^3 stand-in -- camera anywhere around the volume, image sizes up to 900 x 700,
iso ranges of 8 .. 256 values, row ranges -- rendered by the default kernel selection (the union kernel: first call, repeated calls with
the tiles longest first and two cuts, with and without counters) and by the kernel with one traversal per iso value; frames and all
counters must be identical.  usage: python tools/stress_union.py [seconds] [seed]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import volumerenderer_amd as vr
from volumerenderer_amd import capi, datasets

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
L = capi.lib()
vols = {"skull": vr.Data.loadUCD(os.path.join(ROOT, "tests", "data", "Skull.vol.gz")),
        "c60": vr.Data.loadUCD(os.path.join(ROOT, "tests", "data", "C60.vol")),
        "standin256": vr.Data(datasets.bonsai_standin())}
if os.environ.get("VR_STRESS_WIDE"):  # a volume beyond 256 voxels per axis: the wide node array and its kernels
    vols = {"gyroid320": vr.Data(datasets.gyroid_noise(320))}
print("volumes ready", flush=True)
rnd = {}
for name, data in vols.items():
    r = vr.Renderer(0); r.upload_volume(data); r.build_kdt("AlwaysCenter"); rnd[name] = r

def render(r, v, W, b, n, r0, r1, variant, with_stats):
    r.set_kernel(variant)
    d_rgb = torch.full((n, r1 - r0, W, 3), 5, dtype=torch.uint8, device="cuda")
    d_stats = torch.zeros((n, 7), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    fn = L.vr_render_device_noalloc if (variant == capi.KERNEL_DEFAULT and os.environ.get("VR_STRESS_NOALLOC")) else L.vr_render_device
    capi.check(fn(r.handle, C.byref(v), b, n, r0, r1, d_rgb.data_ptr(), d_stats.data_ptr() if with_stats else None, None, None), "render")
    capi.check(L.vr_sync(r.handle), "sync")
    r.set_kernel(capi.KERNEL_DEFAULT)
    return d_rgb, d_stats.cpu().numpy()

print("trees built", flush=True)
t_end = time.time() + budget
n_views = n_bad = n_err_views = 0
while time.time() < t_end:
    name = list(vols)[int(rng.integers(len(vols)))]
    data, r = vols[name], rnd[name]
    res = np.array(data.resulution, dtype=np.float64)
    centre = res / 2.0
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    pos = centre + d * float(rng.uniform(0.7, 2.5)) * float(res.max())
    forw = centre + rng.normal(size=3) * 0.15 * res - pos; forw /= np.linalg.norm(forw)
    up = np.cross(forw, rng.normal(size=3)); up /= np.linalg.norm(up)
    W, H = int(rng.integers(33, 900)), int(rng.integers(17, 700))
    n = int(rng.choice([8, 9, 16, 40, 100, 256])); b = int(rng.integers(0, 257 - n))
    r0 = int(rng.integers(0, H // 3)); r1 = int(rng.integers(r0 + 1, H + 1)) if rng.random() < 0.5 else H
    vp = float(rng.uniform(0.8, 3.0))
    v = vr.raytrace._view(tuple(pos), tuple(forw), tuple(up), W, H, vp)
    what = "%s %dx%d rows %d..%d iso %d+%d" % (name, W, H, r0, r1, b, n)
    if os.environ.get("VR_STRESS_VERBOSE"): print("view %d: %s at %.1f s" % (n_views, what, time.time() - t_end + budget), flush=True)
    ref = render(r, v, W, b, n, r0, r1, capi.KERNEL_PIPELINE, True)
    for i in range(3):
        got = render(r, v, W, b, n, r0, r1, capi.KERNEL_DEFAULT, True)
        if not (got[1] == ref[1]).all() or not bool(torch.equal(got[0], ref[0])):
            n_bad += 1; print("DIFFERS (call %d, with counters): %s" % (i, what), flush=True)
    got = render(r, v, W, b, n, r0, r1, capi.KERNEL_DEFAULT, False)
    if not bool(torch.equal(got[0], ref[0])):
        n_bad += 1; print("DIFFERS (without counters): %s" % what, flush=True)
    if int(ref[1][:, 6].sum()) != 0:
        n_err_views += 1
        ks = np.nonzero(ref[1][:, 6])[0]
        print("errors counted (a ray past the bound on its checks; all kernels agree): %s; iso %s: %s; camera %r %r %r viewport %r" % (
            what, (b + ks[:4]).tolist(), ref[1][ks[:4], 6].tolist(), tuple(pos), tuple(forw), tuple(up), vp), flush=True)
    n_views += 1
    if n_views % 20 == 0: print("%d views, %d differences" % (n_views, n_bad), flush=True)
print("done: %d random views, %d differences, %d views with rays past the bound on their checks" % (n_views, n_bad, n_err_views))
sys.exit(1 if n_bad else 0)
