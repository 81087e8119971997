"""One sweep as two concurrent launches (two contexts, two streams) against one launch: does a second launch that
moves into the CUs the first one frees shorten a single sweep?  Splits tried: iso halves, centre rows / border rows."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import volumerenderer_amd as vr  # noqa: E402
from volumerenderer_amd import capi  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = 1024
data = vr.Data.loadUCD(os.path.join(ROOT, "tests/data/Skull.vol.gz"))
pos, forw, up = vr.main_camera(data.resulution)
L = capi.lib()
rs = []
for _ in range(3):
    r = vr.Renderer(0)
    r.upload_volume(data)
    r.build_kdt("AlwaysCenter")
    rs.append(r)
view = vr.raytrace._view(pos, forw, up, W, W, 2.0)
streams = [torch.cuda.Stream() for _ in range(3)]


def launch(k, iso_b, iso_n, r0, r1, buf):
    capi.check(L.vr_render_device(rs[k].handle, C.byref(view), iso_b, iso_n, r0, r1, buf.data_ptr(), None, None,
                                  C.c_void_p(streams[k].cuda_stream)), "render")


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        torch.cuda.synchronize()  # one sweep at a time: latency of a single sweep
    return (time.perf_counter() - t0) / n * 1e3


full = torch.empty((256, W, W, 3), dtype=torch.uint8, device="cuda")
print("one launch: %.3f ms" % timed(lambda: launch(0, 0, 256, 0, W, full)))
a = torch.empty((128, W, W, 3), dtype=torch.uint8, device="cuda")
b = torch.empty((128, W, W, 3), dtype=torch.uint8, device="cuda")
print("iso halves, two streams: %.3f ms" % timed(lambda: (launch(0, 0, 128, 0, W, a), launch(1, 128, 128, 0, W, b))))
for lo, hi in ((256, 768), (320, 704), (192, 832)):
    c = torch.empty((256, hi - lo, W, 3), dtype=torch.uint8, device="cuda")
    t = torch.empty((256, lo, W, 3), dtype=torch.uint8, device="cuda")
    u = torch.empty((256, W - hi, W, 3), dtype=torch.uint8, device="cuda")
    print("rows %d..%d first, border rows on two more streams: %.3f ms" % (lo, hi, timed(
        lambda: (launch(0, 0, 256, lo, hi, c), launch(1, 0, 256, 0, lo, t), launch(2, 0, 256, hi, W, u)))))
for r in rs:
    r.close()
