"""Where, along the iso axis, do the walks of a ray change?  Renders the Skull sweep with taps at 256 x 256 and counts,
per iso value, the rays whose result differs from the one at the iso value before (a new walk starts there)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import volumerenderer_amd as vr  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
data = vr.Data.loadUCD(os.path.join(ROOT, "tests/data/Skull.vol.gz"))
pos, forw, up = vr.main_camera(data.resulution)
W = 256
with vr.Renderer(0) as r:
    r.upload_volume(data)
    r.build_kdt("AlwaysCenter")
    out = r.renderSweep(pos, forw, up, W, W, 2.0, 0, 256, taps=True)
active = (out["cell_flag"][0] >> 24) != 2
key = out["checked"].astype(np.uint64) << np.uint64(32) | out["t"].view(np.uint32).astype(np.uint64)
new_walk = np.ones((256, W, W), bool)
new_walk[1:] = key[1:] != key[:-1]
starts = new_walk[:, active].sum(1)           # walks that start at iso i
cost = (out["checked"] * new_walk)[:, active].sum(1, dtype=np.int64)  # reference-defined checks of those walks
print("rays in the volume", int(active.sum()), "walks per ray %.1f" % (starts.sum() / active.sum()))
for b in range(11):
    lo, hi = b * 24, min(256, b * 24 + 24)
    print("iso %3d..%3d: walks %5.1f%%  checks %5.1f%%" % (lo, hi - 1, 100.0 * starts[lo:hi].sum() / starts.sum(),
                                                         100.0 * cost[lo:hi].sum() / cost.sum()))
