"""Per-frame counters of the 1024^3 workload (run on the GPU box): moves, escape-help events and checks per ray."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (its HIP runtime must load first)
import volumerenderer_amd as vr
from volumerenderer_amd import datasets

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
vox = datasets.gyroid_noise(n)
pos, forw, up = vr.main_camera(vox.shape)
with vr.Renderer(0) as r:
    r.upload_volume(vr.Data(vox))
    r.build_kdt("AlwaysCenter")
    for iso in (20, 80, 140, 200):
        t = time.perf_counter()
        s = r.renderSweep(pos, forw, up, W, W, 2.0, iso, 1)["stats"][0]
        dt = time.perf_counter() - t
        act = s["hit_count"] + s["miss_count"]
        print("iso %3d: %.1f ms host-timed; rays in volume %d (hit %d); per such ray: checks %.1f moves %.2f escapes %.2f" % (
            iso, dt * 1e3, act, s["hit_count"], s["nodes_checked"] / act, s["moves"] / act, s["escape_help_events"] / act))
