#!/bin/bash
# Runs on the GPU box: bench.py once per tuning build named on the command line (volumerenderer_amd/tune/libvr_<name>.so),
# Skull sweep by default; extra bench arguments after "--".  Usage: tools/tune_variants.sh f0 f16 -- --volume c60
names=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done; [ "${1:-}" = "--" ] && shift
mkdir -p gpurun_out/tune
for v in "${names[@]}"; do
  VR_LIB_PATH=$PWD/volumerenderer_amd/tune/libvr_$v.so timeout -k 10 300 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-delivered "$@" > gpurun_out/tune/$v.json 2> gpurun_out/tune/$v.err || { echo "$v FAILED"; tail -5 gpurun_out/tune/$v.err; exit 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/tune/$v.json")); r=d["roofline"]; print("%-10s kernel_ms %.3f step_ms %.3f errors %d ipl %s drain %.2f" % ("$v", r["kernel_ms"], d["ms_per_step"], d["errors"], r.get("items_per_lane"), r.get("drain_frac",0)))
PY
done
