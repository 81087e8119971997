#!/bin/bash
# GPU box: sweep-kernel time of each tuning library given (names under volumerenderer_amd/tune/, or "default") on Skull, C60 and the 256^3 stand-in.
mkdir -p gpurun_out/tune
for name in "$@"; do
  if [ "$name" = default ]; then unset VR_LIB_PATH; else export VR_LIB_PATH=$PWD/volumerenderer_amd/tune/libvr_$name.so; fi
  line="$name:"
  for cfg in "skull|--steps 30" "c60|--volume c60 --steps 30" "standin|--volume standin256 --steps 30"; do
    tag=${cfg%%|*}; args=${cfg#*|}
    timeout -k 10 120 python3 bench.py --no-cpu-baseline --warmup 2 $args > gpurun_out/tune/${name}_$tag.json 2> gpurun_out/tune/${name}_$tag.err || { echo "$name $tag FAILED"; tail -3 gpurun_out/tune/${name}_$tag.err; exit 1; }
    line="$line $(python3 -c "import json;d=json.load(open('gpurun_out/tune/${name}_$tag.json'));print('$tag k=%.3f s=%.3f e=%d |' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['errors']))")"
  done
  echo "$line"
done
