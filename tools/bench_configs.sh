#!/bin/bash
# Runs on the GPU box: the bench line of every BASELINE configuration that fits one GPU (no CPU legs), one JSON line each.
# Usage: tools/bench_configs.sh <tag> [extra bench args]
set -u
tag=$1; shift
out=gpurun_out/cfg_$tag
mkdir -p $out
run() { name=$1; shift; timeout -k 10 180 python3 bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; return 1; }
  python3 - $out/$name.json $name <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("%-26s %9.0f Mrays/s  step %7.3f ms  kernel %7.3f ms (%s)  frac %.3f  errors %d" % (sys.argv[2], d["value"], d["ms_per_step"], r["kernel_ms"], r["kernel"], r["frac"], d["errors"]))
PY
}
run skull1024_sweep --steps 20 "$@" &&
run skull1024_iso80 --iso-begin 80 --iso-count 1 --steps 50 "$@" &&
run skull4096_iso80 --width 4096 --iso-begin 80 --iso-count 1 --steps 20 "$@" &&
run c60_1024_iso100 --volume c60 --iso-begin 100 --iso-count 1 --steps 50 "$@" &&
run c60_1024_sweep --volume c60 --steps 20 "$@" &&
run standin256_sweep --volume standin256 --steps 10 "$@"
