source tools/tune_run.sh true
W="--volume gyroid1024 --width 4096 --iso-begin 140 --iso-count 1"
runw() { tag=$1; shift; env "$@" timeout -k 10 120 python -u bench.py --steps 3 --warmup 1 $W > gpurun_out/tune/$tag.json 2> gpurun_out/tune/$tag.err || { echo "$tag FAILED"; tail -3 gpurun_out/tune/$tag.err; return 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/tune/$tag.json")); print("%-14s kernel_ms %.3f  step_ms %.3f  errors %d" % ("$tag", d["roofline"]["kernel_ms"], d["ms_per_step"], d["errors"]))
PY
}
runw base X=1 && for t in 640_2 512_2 256_5 256_4 320_4; do runw w$t VR_LIB_PATH=$PWD/tune_w$t.so || exit 1; done
