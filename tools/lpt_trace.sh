#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of the Skull sweep with the union kernel's tiles in list order and longest first
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
  export VR_UNION_LPT=$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lpt_trace$mode -- python3 $R/bench.py --steps 20 --warmup 3 --settle 0 --no-cpu-baseline --no-host-delivered "$@" > $R/gpurun_out/lpt_trace$mode.json 2> $R/gpurun_out/lpt_trace$mode.err || { tail -5 $R/gpurun_out/lpt_trace$mode.err; exit 1; }
  f=$(find $R/gpurun_out/lpt_trace$mode -name "*kernel_stats.csv" | head -1)
  echo "== VR_UNION_LPT=$mode"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %5s avg_us %9.1f total_ms %8.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
