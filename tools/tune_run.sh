#!/bin/bash
# Runs bench.py once per tuning build / knob setting and prints kernel_ms per setting (GPU box).
mkdir -p gpurun_out/tune
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout -k 10 120 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/tune/$tag.json 2> gpurun_out/tune/$tag.err || { echo "$tag FAILED"; tail -3 gpurun_out/tune/$tag.err; return 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/tune/$tag.json")); print("%-14s kernel_ms %.3f  step_ms %.3f  errors %d" % ("$tag", d["roofline"]["kernel_ms"], d["ms_per_step"], d["errors"]))
PY
}
"$@"
