#!/bin/bash
# GPU box: first call and settled step of the three sweeps with the cut of the iso range forced.  usage: tools/ab_first_cut.sh ipl...
for ipl in "$@"; do
  for v in skull c60 standin256; do
    r=$(VR_SWEEP_ITEMS=$ipl timeout -k 10 300 python -u bench.py --volume $v --steps 10 --warmup 2 --no-latency-floor --no-cpu-baseline --no-host-delivered --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['first_call']; print('first %.3f (kernel %.3f) repeated %.3f' % (f['ms_per_step'], f['kernel_ms'], d['ms_per_step']))")
    echo "items per lane $ipl $v $r"
  done
done
