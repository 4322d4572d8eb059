#!/bin/bash
# round 5, first GPU call: parity of the new fp forms, the whole GPU suite, then library-against-library A/B of the default arithmetic
# (round 4's library against this tree's) and the bench line under the contracted forms
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_round5.py -m gpu -x -q > gpurun_out/r05_gputest_fp.log 2>&1; rc=$?; tail -3 gpurun_out/r05_gputest_fp.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_round5.py > gpurun_out/r05_gputest_all.log 2>&1; rc=$?; tail -3 gpurun_out/r05_gputest_all.log
[ $rc -ne 0 ] && exit $rc
bash tools/ab_old_new.sh r4 r5b 2>&1 | tee gpurun_out/r05_ab_r4_r5b.txt
for c in 0 7 15 8; do
  python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-host-out --no-single-frame --contract $c > gpurun_out/r05_bench_contract$c.json 2> gpurun_out/r05_bench_contract$c.err || { echo "bench contract $c failed"; tail -3 gpurun_out/r05_bench_contract$c.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05_bench_contract$c.json").read().strip().splitlines()[-1])
print("contract $c:", round(d["value"]), round(d["ms_per_step"],4), {k: round(v,4) for k,v in d["roofline"]["all_kernels_ms_per_step"].items()}, round(d["roofline"]["frac"],4))
PY
done
