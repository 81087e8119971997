#!/bin/bash
# GPU box: the three sweeps and the three frames with and without the kept ray list (VR_KEEP_LIST=0: every call makes its list again)
for rep in 1 2; do
for keep in 0 1; do
  out="keep_list=$keep"
  for a in "--volume skull" "--volume c60" "--volume standin256" "--iso-begin 80 --iso-count 1" "--volume c60 --iso-begin 100 --iso-count 1"; do
    r=$(VR_KEEP_LIST=$keep timeout -k 10 200 python -u bench.py $a --steps 20 --warmup 3 --no-latency-floor --no-cpu-baseline --no-host-delivered --no-first-call --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f/%.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))")
    out="$out  $r"
  done
  echo "$out"
done
done
