"""Times vr_render_sweep (frames delivered to host memory) into pageable and page-locked buffers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
import volumerenderer_amd as vr
data = vr.Data.loadUCD(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/data/Skull.vol.gz"))
pos, forw, up = vr.main_camera(data.resulution)
with vr.Renderer(0) as r:
    r.upload_volume(data); r.build_kdt("AlwaysCenter")
    pageable = np.zeros((256, 1024, 1024, 3), np.uint8)
    pinned = r.pinned_frames(256, 1024, 1024)
    for name, buf in (("pageable", pageable), ("page-locked", pinned)):
        for it in range(3):
            t0 = time.perf_counter()
            out = r.renderSweep(pos, forw, up, 1024, 1024, 2.0, 0, 256, out=buf)
            dt = time.perf_counter() - t0
        print("host-buffer sweep into %s memory: %.1f ms (%.1f GB/s of frames)" % (name, dt * 1e3, buf.nbytes / dt / 1e9))
    assert (pageable == pinned).all()
