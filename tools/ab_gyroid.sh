for name in "$@"; do
  lib=$PWD/volumerenderer_amd/tune/libvr_$name.so; [ "$name" = base ] && lib=$PWD/volumerenderer_amd/libvr_hip.so
  r=$(VR_LIB_PATH=$lib timeout -k 10 300 python -u bench.py --volume gyroid1024 --width 4096 --iso-begin 128 --iso-count 32 --steps 10 --warmup 2 --no-latency-floor --no-cpu-baseline --no-host-delivered --no-first-call --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f/%.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']))")
  echo "$name gyroid $r"
done
