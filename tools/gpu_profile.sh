#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the default bench, then PMC passes.
# Usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $R/$out/bench_trace.json 2> $R/$out/trace.err || exit 1
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_LDS" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
            "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $R/$out/pmc_$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/$out/pmc_$name.json 2> $R/$out/pmc_$name.err || { echo "pass $name failed"; tail -3 $R/$out/pmc_$name.err; exit 1; }
done
echo done
