"""Times the tree build: GPU (vr_build_kdt) against the host builder, on the Skull volume and the 256^3
stand-in -- the centre split and, with `span` as the first argument, the span heuristics of the reference driver's
list that the reference can build for the volume.  Prints one JSON line per volume and heuristic."""
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
    heuristics = ["AlwaysCenter"]
    if len(sys.argv) > 1 and sys.argv[1] == "span":
        heuristics = ["FlexibleFirst5", "FlexibleAfter15", "FlexibleAfter20", "FlexibleCenterWeighted30",
                      "FlexibleCenterWeighted50", "FlexibleCenterWeighted75", "FlexibleCenterWeighted100"]
    vols = {"Skull": vr.Data.loadUCD(os.path.join(ROOT, "tests/data/Skull.vol.gz")).data,
            "standin256": vr.datasets.bonsai_standin()}
    for name, vox in vols.items():
      for heur in heuristics:
        d = vr.Data(vox)
        t0 = time.perf_counter()
        try:
            k = vr.SlimKDT.newHeursiticPartition(d, heur, 1 << 30)
        except vr.capi.VrError:
            print(json.dumps({"volume": name, "heuristic": heur, "host": "too deep", "host_ms": round((time.perf_counter() - t0) * 1e3, 1)}))
            continue
        host_ms = (time.perf_counter() - t0) * 1e3
        with vr.Renderer(0) as r:
            r.upload_volume(d)
            r.build_kdt(heur)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                st = r.build_kdt(heur)
                best = min(best, (time.perf_counter() - t0) * 1e3)
        print(json.dumps({"volume": name, "heuristic": heur, "shape": list(vox.shape), "nodes": int(st.number_of_nodes),
                          "layers": int(st.layer_count), "slots": len(k.nodes), "host_build_ms": round(host_ms, 2),
                          "gpu_build_ms": round(best, 3), "gpu_Mnodes_per_s": round(st.number_of_nodes / best / 1e3, 1)}), flush=True)


if __name__ == "__main__":
    main()
