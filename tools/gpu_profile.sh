#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the default bench, then the HBM-traffic PMC passes
# (FETCH_SIZE and WRITE_SIZE in separate runs, no tracing options mixed in).
# Usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $R/$out/bench_trace.json 2> $R/$out/trace.err || { tail -5 $R/$out/trace.err; exit 1; }
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $R/$out/pmc$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/$out/pmc$i.json 2> $R/$out/pmc$i.err || { echo "pass $i failed"; tail -3 $R/$out/pmc$i.err; exit 1; }
done
python3 - <<PY
import csv,glob,collections,json,re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/$out/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(re.search(r'(vr_k_\w+)', r['Kernel_Name']) or [None, r['Kernel_Name'][:40]])[1]][r['Counter_Name']].append(float(r['Counter_Value']))
summary={k:{c:sum(v)/len(v) for c,v in d.items()} for k,d in agg.items()}
json.dump(summary, open("$R/$out/pmc_summary.json","w"), indent=1)
for k,d in summary.items():
    if 'vr_k_trace' in k or 'vr_k_render' in k or 'vr_k_sweep' in k:
        for c in sorted(d): print("%-36s %.5g" % (c, d[c]))
PY
