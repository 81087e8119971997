"""Oracle renders at the full sizes the GPU kernels are benchmarked on, as committed fixtures (run once, in the build
container, CPU only: about ten minutes on 8 cores).

The reference's driver loop (src/main.zig:111-138: iso = 0..255, one renderImage each, src/raytrace.zig:11-106) is run
by the CPU oracle (oracle/vr_oracle.c, pinned to the reference's own renders by tests/test_oracle_golden.py) on

  skull_1024     Skull.vol (tests/data/Skull.vol.gz), 1024 x 1024, iso 0..255, camera of src/main.zig:31-43
  standin_1024   the seeded 256^3 stand-in volume (volumerenderer_amd/datasets.py), 1024 x 1024, iso 0..255
  skull_4096     Skull.vol, 4096 x 4096, iso 80

and writes tests/golden/fullsize_oracle.json: per frame the SHA-256 of the RGB24 buffer and the seven counters of
vr_render_stats (nodes_checked, hit, miss, oob, escape helps, MOVEs, errors), plus the SHA-256 of each volume (the
stand-in is generated with float32 numpy arithmetic; a machine whose numpy rounds differently gets another volume and
must not be compared with these frames).  It also records, for two span-heuristic trees of Skull that take the oracle
minutes to build (SlimKDT.newHeursiticPartition, src/SlimKDT.zig:222-322 with FlexibleFirst5 :369-375 and
FlexibleCenterWeighted100 :568-598), the SHA-256 of the node array up to its last used slot and PartitionBuildData.

The GPU tests hash what the default kernel leaves in device memory and compare (tests/test_gpu_golden_fullsize.py).
"""
import gzip
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from volumerenderer_amd import datasets  # noqa: E402

THREADS = int(os.environ.get("GOLDEN_THREADS", "8"))
STAT_FIELDS = ("nodes_checked", "hit_count", "miss_count", "oob_count", "escape_help_events", "moves", "errors")


def skull():
    raw = gzip.open(os.path.join(ROOT, "tests", "data", "Skull.vol.gz")).read()
    return np.frombuffer(raw[28:], dtype=np.uint8).reshape(68, 256, 256).copy()


def sweep(pool, vox, W, isos, what):
    """One oracle render per iso value, image rows spread over the pool's threads (ctypes releases the GIL)."""
    nodes, rd, _ = O.build_kdt(vox)
    nodes = O.trim_nodes(nodes)
    pos, forw, up = O.main_camera(vox.shape)
    bands = [(W * i // (4 * THREADS), W * (i + 1) // (4 * THREADS)) for i in range(4 * THREADS)]
    frames = []
    t0 = time.time()
    for iso in isos:
        img = np.zeros((W, W, 3), np.uint8)
        parts = list(pool.map(lambda b: O.render(nodes, rd, pos, forw, up, W, W, 2.0, iso, rows=b, taps=False, out=img)["stats"], bands))
        st = [sum(int(getattr(p, f)) for p in parts) for f in STAT_FIELDS[:6]] + [0]  # errors: the oracle has none to count
        st[3] = W * W - st[1] - st[2]  # whatever a band reports for rows outside it: oob = the rays that neither hit nor missed
        frames.append({"iso": int(iso), "sha256": hashlib.sha256(img.tobytes()).hexdigest(), "stats": st})
        if iso % 32 == 0:
            print("%s iso %d (%.0f s)" % (what, iso, time.time() - t0), flush=True)
    return {"width": W, "height": W, "viewport_width": 2.0, "camera_pos": [-30.0, -300.0, -30.0],
            "volume_shape": list(vox.shape), "volume_sha256": hashlib.sha256(vox.tobytes()).hexdigest(),
            "stat_fields": list(STAT_FIELDS), "frames": frames}


def tree(vox, heuristic):
    nodes, rd, st = O.build_kdt(vox, heuristic, max_slots=1 << 32)
    used = int(np.flatnonzero(nodes)[-1]) + 1
    return {"heuristic": heuristic, "used_slots": used, "sha256": hashlib.sha256(nodes[:used].tobytes()).hexdigest(),
            "layer_count": int(st.layer_count), "number_of_nodes": int(st.number_of_nodes),
            "non_center_partition_count": int(st.non_center_partition_count)}


def main():
    out = {"made_by": "tests/golden/make_fullsize_golden.py (oracle/vr_oracle.c on the CPU)"}
    sk = skull()
    with ThreadPoolExecutor(THREADS) as pool:
        # the two slow tree builds run next to the renders
        trees = [pool.submit(tree, sk, h) for h in ("FlexibleFirst5", "FlexibleCenterWeighted100")]
        # check the band bookkeeping against a whole-frame render once
        nodes, rd, _ = O.build_kdt(sk)
        whole = O.render(O.trim_nodes(nodes), rd, *O.main_camera(sk.shape), 256, 256, 2.0, 80, taps=False)
        probe = sweep(pool, sk, 256, [80], "probe")["frames"][0]
        assert probe["sha256"] == hashlib.sha256(whole["rgb"].tobytes()).hexdigest()
        assert probe["stats"] == [int(getattr(whole["stats"], f)) for f in STAT_FIELDS[:6]] + [0], probe["stats"]
        out["skull_4096"] = sweep(pool, sk, 4096, [80], "skull 4096")
        out["skull_1024"] = sweep(pool, sk, 1024, range(256), "skull 1024")
        out["standin_1024"] = sweep(pool, datasets.bonsai_standin(), 1024, range(256), "stand-in 1024")
        out["skull_trees"] = [t.result() for t in trees]
    with open(os.path.join(HERE, "fullsize_oracle.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("written", os.path.getsize(os.path.join(HERE, "fullsize_oracle.json")), "bytes")


if __name__ == "__main__":
    main()
