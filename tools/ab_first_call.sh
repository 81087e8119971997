#!/bin/bash
# GPU box: first call and repeated call of the three 256-iso sweeps per library build.  usage: tools/ab_first_call.sh name...
for name in "$@"; do
  lib=$PWD/volumerenderer_amd/tune/libvr_$name.so; [ "$name" = base ] && lib=$PWD/volumerenderer_amd/libvr_hip.so
  for v in skull c60 standin256; do
    r=$(VR_LIB_PATH=$lib timeout -k 10 300 python -u bench.py --volume $v --steps 10 --warmup 2 --no-latency-floor --no-cpu-baseline --no-host-delivered --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['first_call']; print('first %.3f (kernel %.3f) repeated %.3f (kernel %.3f)' % (f['ms_per_step'], f['kernel_ms'], d['ms_per_step'], d['roofline']['kernel_ms']))")
    echo "$name $v $r"
  done
done
