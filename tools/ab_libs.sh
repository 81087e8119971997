#!/bin/bash
# GPU box: bench.py once per library build named on the command line ("base" = the product library), twice round-robin,
# printing kernel and step time.  usage: tools/ab_libs.sh [--args "bench args"] name...
extra=""
if [ "$1" = "--args" ]; then extra=$2; shift; shift; fi
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for name in "$@"; do
    lib=$PWD/volumerenderer_amd/tune/libvr_$name.so; [ "$name" = base ] && lib=$PWD/volumerenderer_amd/libvr_hip.so
    VR_LIB_PATH=$lib timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-delivered $extra > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err || { echo "$name FAILED"; tail -5 gpurun_out/ab/$name.err; exit 1; }
    python - <<PY
import json; d=json.load(open("gpurun_out/ab/$name.json")); r=d["roofline"]; f=d.get("first_call") or {}
print("%-12s kernel_ms %.3f step_ms %.3f first_call step %.3f kernel %.3f errors %d ipl %s drain %.2f" % ("$name", r["kernel_ms"], d["ms_per_step"], f.get("ms_per_step",0), f.get("kernel_ms",0), d["errors"], r.get("items_per_lane"), r.get("drain_frac",0)))
PY
  done
done
