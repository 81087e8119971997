#!/bin/bash
# GPU box: the 256-iso sweep of Skull, C60 and the 256^3 stand-in per library build ("base" = the product library), twice:
# step / kernel ms.  usage: tools/ab_volumes.sh name...
for rep in 1 2; do
for name in "$@"; do
  lib=$PWD/volumerenderer_amd/tune/libvr_$name.so; [ "$name" = base ] && lib=$PWD/volumerenderer_amd/libvr_hip.so
  out=$(printf "%-10s" "$name")
  for v in skull c60 standin256; do
    r=$(VR_LIB_PATH=$lib timeout -k 10 200 python -u bench.py --volume $v --steps 20 --warmup 3 --no-latency-floor --no-cpu-baseline --no-host-delivered --no-first-call --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f/%.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']))")
    out="$out  $v $r"
  done
  echo "$out"
done
done
