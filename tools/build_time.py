"""Times the centre-split tree build: GPU (vr_build_kdt) against the host builder, on the Skull volume
and the 256^3 stand-in.  Prints one JSON line per volume."""
import gzip
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import volumerenderer_amd as vr  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    vols = {"Skull": vr.Data.loadUCD(os.path.join(ROOT, "tests/data/Skull.vol.gz")).data,
            "standin256": vr.datasets.bonsai_standin()}
    for name, vox in vols.items():
        d = vr.Data(vox)
        t0 = time.perf_counter()
        k = vr.SlimKDT.newHeursiticPartition(d, "AlwaysCenter")
        host_ms = (time.perf_counter() - t0) * 1e3
        with vr.Renderer(0) as r:
            r.upload_volume(d)
            r.build_kdt("AlwaysCenter")
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                st = r.build_kdt("AlwaysCenter")
                best = min(best, (time.perf_counter() - t0) * 1e3)
        print(json.dumps({"volume": name, "shape": list(vox.shape), "nodes": int(st.number_of_nodes),
                          "slots": len(k.nodes), "host_build_ms": round(host_ms, 2), "gpu_build_ms": round(best, 3),
                          "gpu_Mnodes_per_s": round(st.number_of_nodes / best / 1e3, 1)}))


if __name__ == "__main__":
    main()
