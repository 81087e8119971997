"""Runs one sweep of the pipeline kernel on a chosen volume (for the diagnostic build's stderr report).
usage: python tools/diag_run.py {skull|c60|cube16|cube32|noise64} [width] [kernel id, default 0]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import volumerenderer_amd as vr
from volumerenderer_amd import capi
which = sys.argv[1]; W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
kernel = int(sys.argv[3]) if len(sys.argv) > 3 else 0
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(3)
if which == "skull": data = vr.Data.loadUCD(os.path.join(root, "tests/data/Skull.vol.gz"))
elif which == "c60": data = vr.Data.loadUCD(os.path.join(root, "tests/data/C60.vol"))
elif which.startswith("cube"):
    n = int(which[4:]); g = np.stack(np.meshgrid(*[np.linspace(-1, 1, n)] * 3, indexing="ij")); r = np.sqrt((g ** 2).sum(0))
    data = vr.Data(np.clip(255 * np.exp(-3 * r * r) + rng.normal(0, 6, (n, n, n)), 0, 255).astype(np.uint8))
else: data = vr.Data(rng.integers(0, 256, (64, 64, 64), dtype=np.uint8))
kdt = vr.SlimKDT.newHeursiticPartition(data, "AlwaysCenter")
pos, forw, up = vr.main_camera(data.resulution)
vp = 2.0 * max(data.resulution) / 256.0  # keep the volume's share of the image similar to Skull's
L = capi.lib(); r = vr.Renderer(0); r.upload_kdt(kdt); r.set_kernel(kernel)
view = vr.raytrace._view(pos, forw, up, W, W, vp)
d_rgb = torch.empty((256, W, W, 3), dtype=torch.uint8, device="cuda"); d_stats = torch.zeros((256, 7), dtype=torch.int64, device="cuda")
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    capi.check(L.vr_render_device(r.handle, C.byref(view), 0, 256, 0, W, d_rgb.data_ptr(), d_stats.data_ptr(), None, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "render")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = d_stats.cpu().numpy()
print("%s: %.2f ms/sweep, checks %d, hits %d, miss %d, oob %d, tree slots %d" % (which, dt * 1e3, st[:, 0].sum(), st[:, 1].sum(), st[:, 2].sum(), st[:, 3].sum(), len(kdt.nodes)))
