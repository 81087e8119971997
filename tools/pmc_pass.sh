#!/bin/bash
# GPU box: one rocprofv3 counter pass (its own run, no tracing options) of bench.py on a named library build, per-kernel means printed.
# usage: tools/pmc_pass.sh <lib name | base> "<counters>" [bench args...]
name=$1; counters=$2; shift; shift
R=$GRAFT_REPO_ROOT
lib=$R/volumerenderer_amd/tune/libvr_$name.so; [ "$name" = base ] && lib=$R/volumerenderer_amd/libvr_hip.so
out=$R/gpurun_out/pmc_$name
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export VR_LIB_PATH=$lib VR_SWEEP_ITEMS=${VR_SWEEP_ITEMS:-3}
timeout -s KILL 200 rocprofv3 --pmc $counters --output-format csv -d $out/raw -- python3 $R/bench.py --steps 4 --warmup 1 --settle 0 --no-cpu-baseline --no-host-delivered --no-first-call "$@" > $out/bench.json 2> $out/err.txt || { echo "pass failed"; tail -3 $out/err.txt; exit 1; }
python3 - "$out" "$name" <<'PY'
import csv, glob, collections, re, sys
out, name = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(vr_k_\w+)', r['Kernel_Name'])
        if m: agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
for k in ("vr_k_union", "vr_k_sweep", "vr_k_trace"):
    if k in agg:
        print(name, k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(agg[k].items())))
PY
