source tools/tune_run.sh true
W="--volume gyroid1024 --width 4096 --iso-begin 140 --iso-count 1"
runw() { tag=$1; shift; env "$@" timeout -k 10 120 python -u bench.py --steps 3 --warmup 1 $W > gpurun_out/tune/$tag.json 2> gpurun_out/tune/$tag.err || { echo "$tag FAILED"; tail -3 gpurun_out/tune/$tag.err; return 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/tune/$tag.json")); print("%-14s kernel_ms %.3f  step_ms %.3f  errors %d" % ("$tag", d["roofline"]["kernel_ms"], d["ms_per_step"], d["errors"]))
PY
}
runw base X=1 && for t in 6_3 6_4 4_4 4_8 8_8 6_64; do runw p$t VR_LIB_PATH=$PWD/tune_p$t.so || exit 1; done
