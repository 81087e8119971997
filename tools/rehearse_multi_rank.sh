#!/bin/bash
# Runs on the GPU box (via gpurun): the N = 2 and N = 4 code paths of bench.py on ONE GPU (VR_BENCH_REHEARSAL=1: all
# ranks share device 0, gloo instead of RCCL).  Checks control flow and the per-rank shards, not speed.
set -u
export VR_BENCH_REHEARSAL=1
out=gpurun_out/rehearsal
mkdir -p $out
i=0
for args in "--gpus 2" "--gpus 4 --scaling strong --shard rows --gather --no-extras" "--gpus 3 --scaling strong --shard bands" "--gpus 2 --scaling weak --no-extras" \
            "--gpus 4 --scaling strong --width 4096 --iso-begin 80 --iso-count 1 --gather --no-extras" \
            "--gpus 2 --scaling strong --shard rows --volume gyroid1024 --width 2048 --iso-begin 140 --iso-count 1 --no-extras"; do
  i=$((i+1))
  n=$(echo $args | cut -d' ' -f2)
  timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+i)) \
      bench.py $args --steps 3 --warmup 1 > $out/run$i.json 2> $out/run$i.err || { echo "FAILED: $args"; tail -5 $out/run$i.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$out/run$i.json").read().strip().splitlines()[-1])
print("$args ->", d["n_gpus"], d["scaling"], d["config"]["sharding"], "gather", d["config"]["gather"], "rays/step", d["config"]["rays_per_step"], "errors", d["errors"], "%.1f Mrays/s" % d["value"],
      {k: (round(d[k]["value"]), d[k]["rays_per_step"]) for k in ("weak_scaling", "strong_scaling", "strong_bands", "strong_iso", "strong_with_gather") if k in d})
PY
done
