#!/bin/bash
# Runs on the GPU box: the 1024^3 gyroid at 4096^2, a block of iso values, with the union kernel (default) and the kernel
# with shared traversals.  Usage: tools/wide_union_time.sh [iso_begin] [iso_count]
set -u
b=${1:-124}; n=${2:-32}
mkdir -p gpurun_out/wide_union
for k in 0 5; do
  timeout -k 10 400 python -u bench.py --volume gyroid1024 --width 4096 --iso-begin $b --iso-count $n --steps 3 --warmup 1 \
      --no-cpu-baseline --no-host-delivered --kernel $k > gpurun_out/wide_union/k${k}_${b}_${n}.json 2> gpurun_out/wide_union/k${k}_${b}_${n}.err \
      || { echo "kernel $k failed"; tail -5 gpurun_out/wide_union/k${k}_${b}_${n}.err; exit 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/wide_union/k${k}_${b}_${n}.json")); r=d["roofline"]
print("kernel %d  iso %d+%d  %s  kernel_ms %.3f  step_ms %.3f  errors %d  value %.0f %s" % ($k, $b, $n, r.get("kernel"), r["kernel_ms"], d["ms_per_step"], d["errors"], d["value"], d["unit"]))
PY
done
