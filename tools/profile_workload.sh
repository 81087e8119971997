#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of one bench workload, then its counter passes (HBM
# traffic in runs of their own, no tracing options mixed in), and writes the two files bench.py and the judge read:
#   gpurun_out/prof_<tag>/r03_stats_<workload>.csv   kernel stats of `bench.py --steps 30 --warmup 1 <args>` (the first launch of a sweep has no tile order yet)
#   gpurun_out/prof_<tag>/r03_pmc_<workload>.json    per-kernel counter means, the commit and the kernel-source fingerprint
# Usage: tools/profile_workload.sh <tag> <commit> [bench args...]
set -u
tag=$1; commit=$2; shift; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# the cut of the iso range that the library settles on for this workload (an un-profiled run with the usual settling), forced
# for the profiled runs below so that every traced launch runs with it
cut=$(timeout -k 10 300 python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-delivered --no-first-call --no-latency-floor --no-extras "$@" 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['roofline'].get('items_per_lane') or '')")
if [ -n "$cut" ]; then export VR_SWEEP_ITEMS=$cut; echo "settled cut: $cut items per lane"; fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py --steps 30 --warmup 1 --settle 0 --no-cpu-baseline --no-host-delivered --no-first-call --no-latency-floor --no-extras "$@" > $out/bench_trace.json 2> $out/trace.err || { tail -5 $out/trace.err; exit 1; }
if [ -n "${PROFILE_TRACE_ONLY:-}" ]; then # only the kernel stats again (the counter summary of this workload stays)
  f=$(find $out/trace -name "*kernel_stats.csv" | head -1); w=$(python3 -c "import json; print(json.load(open('$out/bench_trace.json'))['config']['workload'])")
  cp $f $out/r03_stats_$w.csv; echo "$w: kernel stats only"; exit 0
fi
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $out/pmc$i -- python3 $R/bench.py --steps 4 --warmup 1 --settle 0 --no-cpu-baseline --no-host-delivered --no-first-call --no-latency-floor --no-extras "$@" > $out/pmc$i.json 2> $out/pmc$i.err || { echo "pass $i failed"; tail -3 $out/pmc$i.err; exit 1; }
done
python3 - "$out" "$commit" "$R" <<'PY'
import csv, glob, collections, json, re, sys, hashlib, os, shutil
out, commit, root = sys.argv[1:4]
bench = json.load(open(out + "/bench_trace.json"))
workload = bench["config"]["workload"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(vr_k_\w+)', r['Kernel_Name'])
        if m:
            agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
h = hashlib.sha256()
for name in ["vr_kernels.hip", "vr_trace_core.h", "vr_trace_wide.h", "vr_trace_union.h", "vr_trace_union_wide.h"]:
    h.update(open(os.path.join(root, "volumerenderer_amd", "csrc", name), "rb").read())
summary = {"workload": workload, "commit": commit, "kernel_sources_sha": h.hexdigest()[:16],
           "command": "rocprofv3 --pmc <one pass per run> -- python3 bench.py --steps 4 --warmup 1 --settle 0 --no-cpu-baseline --no-host-delivered --no-first-call " + " ".join(sys.argv[4:]),
           "forced_items_per_lane": os.environ.get("VR_SWEEP_ITEMS"), "note": "per-dispatch means; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (gfx950: double FETCH_SIZE)",
           "bench_under_trace": {k: bench[k] for k in ("value", "ms_per_step")}, "kernel_ms_under_trace": bench["roofline"]["kernel_ms"],
           "kernels": {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}}
json.dump(summary, open("%s/r03_pmc_%s.json" % (out, workload), "w"), indent=1)
stats = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], "%s/r03_stats_%s.csv" % (out, workload))
print(workload, "kernel_ms %.3f" % bench["roofline"]["kernel_ms"], "step %.3f" % bench["ms_per_step"])
for k, d in summary["kernels"].items():
    if k in ("vr_k_union", "vr_k_sweep", "vr_k_trace", "vr_k_trace_wide"):
        for c in sorted(d):
            print("  %-14s %-34s %.5g" % (k, c, d[c]))
PY
