#!/bin/bash
# GPU box: the frames cleared by hipMemsetAsync (0) or by vr_k_clear_frames with that many blocks of 256 threads.  usage: tools/ab_clear.sh grid...
for rep in 1 2; do
for ck in "$@"; do
  out=$(printf "clear_kernel=%-6s" $ck)
  for a in "--volume skull" "--volume c60" "--volume standin256" "--width 4096 --iso-begin 80 --iso-count 1"; do
    r=$(VR_CLEAR_KERNEL=$ck timeout -k 10 200 python -u bench.py $a --steps 20 --warmup 3 --no-latency-floor --no-cpu-baseline --no-host-delivered --no-first-call --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f/%.4f e%d' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['errors']))")
    out="$out  $r"
  done
  echo "$out"
done
done
