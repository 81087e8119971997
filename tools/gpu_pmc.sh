#!/bin/bash
# Usage: tools/gpu_pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ... ; env vars pass through to bench.py
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $R/$out/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $VR_PMC_BENCH_ARGS > $R/$out/p$i.json 2> $R/$out/p$i.err || { echo "pass $i failed"; tail -3 $R/$out/p$i.err; exit 1; }
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$R/$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'vr_k_trace' in r['Kernel_Name'] or 'vr_k_render' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg): print("%-40s %.4g" % (k, sum(agg[k])/len(agg[k])))
PY
