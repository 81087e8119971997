#!/bin/bash
# Runs on the GPU box: bench.py with the union kernel's tiles in list order (VR_UNION_LPT=0) and longest first (default),
# alternating, for the volumes named on the command line (default: skull c60 standin256).
vols=("$@"); [ ${#vols[@]} -eq 0 ] && vols=(skull c60 standin256)
mkdir -p gpurun_out/lpt
for vol in "${vols[@]}"; do
  for rep in 1 2; do
    for mode in 0 1; do
      VR_UNION_LPT=$mode timeout -k 10 300 python -u bench.py --volume $vol --steps 20 --warmup 3 --no-cpu-baseline --no-host-delivered > gpurun_out/lpt/${vol}_$mode.json 2> gpurun_out/lpt/${vol}_$mode.err || { echo "$vol $mode FAILED"; tail -5 gpurun_out/lpt/${vol}_$mode.err; exit 1; }
      python - <<PY
import json; d=json.load(open("gpurun_out/lpt/${vol}_$mode.json")); r=d["roofline"]; print("%-10s lpt=%s kernel_ms %.3f step_ms %.3f errors %d ipl %s drain %.2f" % ("$vol", "$mode", r["kernel_ms"], d["ms_per_step"], d["errors"], r.get("items_per_lane"), r.get("drain_frac",0)))
PY
    done
  done
done
