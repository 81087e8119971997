"""What each rank of an N-GPU run would do, timed alone on ONE GPU: for N = 1, 2, 4, 8 and the three splits (blocks of
iso values, contiguous blocks of rows, rows in bands of 32 dealt out round-robin) the whole-call time of every shard
(pre-pass + memset + traversal kernel, frames resident on the device).  The slowest shard is the step time an N-GPU
run can reach with that split (the tree is replicated, nothing is exchanged before the optional gather).
usage: python tools/shard_times.py {skull|standin256|c60} [width] [iso_begin] [iso_count]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import volumerenderer_amd as vr  # noqa: E402
from volumerenderer_amd import capi, sharding  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1] if len(sys.argv) > 1 else "skull"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
I0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
IN = int(sys.argv[4]) if len(sys.argv) > 4 else 256
if which == "skull":
    data = vr.Data.loadUCD(os.path.join(ROOT, "tests/data/Skull.vol.gz"))
elif which == "c60":
    data = vr.Data.loadUCD(os.path.join(ROOT, "tests/data/C60.vol"))
else:
    data = vr.Data(vr.datasets.bonsai_standin())
pos, forw, up = vr.main_camera(data.resulution)
L = capi.lib()
H = W
stream = torch.cuda.Stream()


def timed(call, reps):
    with torch.cuda.stream(stream):
        for _ in range(16):  # untimed, one by one: the library settles its cut of the iso range for this shard
            call()
            stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        stream.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


with vr.Renderer(0) as r:
    r.upload_volume(data)
    r.build_kdt("AlwaysCenter")
    view = vr.raytrace._view(pos, forw, up, W, H, 2.0)
    d_rgb = torch.empty(IN * H * W * 3, dtype=torch.uint8, device="cuda")
    d_stats = torch.zeros((IN, 7), dtype=torch.int64, device="cuda")
    sp = C.c_void_p(stream.cuda_stream)
    reps = 10 if IN * W * W <= (1 << 28) else 4
    print("%s %dx%d, iso %d..%d: ms per call of every shard alone on one GPU (slowest shard = step time of the split)" % (
        which, W, H, I0, I0 + IN - 1))
    for world in (1, 2, 4, 8):
        rows = {}
        if IN >= world:
            ts = []
            for k in range(world):
                b, n = sharding.iso_shard(k, world, I0, IN)
                ts.append(timed(lambda: capi.check(L.vr_render_device(r.handle, C.byref(view), b, n, 0, H, d_rgb.data_ptr(),
                                                                      d_stats.data_ptr(), None, sp), "render"), reps))
            rows["iso blocks"] = ts
        ts = []
        for k in range(world):
            r0, r1 = sharding.row_shard(k, world, H)
            ts.append(timed(lambda: capi.check(L.vr_render_device(r.handle, C.byref(view), I0, IN, r0, r1, d_rgb.data_ptr(),
                                                                  d_stats.data_ptr(), None, sp), "render"), reps))
        rows["row blocks"] = ts
        ts = []
        for k in range(world):
            sh = sharding.band_shard(k, world, H)
            ts.append(timed(lambda: capi.check(L.vr_render_device_bands(r.handle, C.byref(view), I0, IN, sh["row_begin"], sh["band_rows"],
                                                                        sh["band_pitch"], sh["n_bands"], d_rgb.data_ptr(),
                                                                        d_stats.data_ptr(), None, sp), "bands"), reps))
        rows["row bands of 32"] = ts
        for name, ts in rows.items():
            print("N=%d %-16s max %7.3f  sum %7.3f  | %s" % (world, name, max(ts), sum(ts), " ".join("%.3f" % t for t in ts)))
