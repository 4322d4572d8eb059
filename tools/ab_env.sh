#!/bin/bash
# A/B of one environment switch on the bench line: bash tools/ab_env.sh NAME=VALUE [bench args...]   (GPU box)
# runs the GPU tests first when AB_TESTS=1, then bench.py without / with the switch, twice each
sw=$1; shift
out=gpurun_out/ab; mkdir -p $out
if [ "${AB_TESTS:-0}" = "1" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?
  tail -3 $out/tests.log
  if [ $rc -ne 0 ]; then echo "tests rc=$rc"; grep -n "Error\|FAILED\|assert" $out/tests.log | head -20; exit $rc; fi
fi
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-host-out --no-single-frame $BENCH_ARGS > $out/bench_$name.json 2> $out/bench_$name.err || { echo "bench $name failed"; tail -5 $out/bench_$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$out/bench_$name.json").read().strip().splitlines()[-1])
print("$name:", round(d["value"]), round(d["ms_per_step"],4), {k: round(v,4) for k,v in d["roofline"]["all_kernels_ms_per_step"].items()})
PY
}
BENCH_ARGS="$*"
run base1 AB_NONE=1
run with1 $sw
run base2 AB_NONE=1
run with2 $sw
