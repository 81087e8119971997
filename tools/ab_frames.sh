#!/bin/bash
# single-frame workloads per library build
for rep in 1 2; do
for name in "$@"; do
  lib=$PWD/volumerenderer_amd/tune/libvr_$name.so; [ "$name" = base ] && lib=$PWD/volumerenderer_amd/libvr_hip.so
  out="$name"
  for a in "--iso-begin 80 --iso-count 1" "--volume c60 --iso-begin 100 --iso-count 1" "--width 4096 --iso-begin 80 --iso-count 1"; do
    r=$(VR_LIB_PATH=$lib timeout -k 10 200 python -u bench.py $a --steps 40 --no-latency-floor --no-cpu-baseline --no-host-delivered --no-first-call --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f/%.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))")
    out="$out  $r"
  done
  echo "$out"
done
done
