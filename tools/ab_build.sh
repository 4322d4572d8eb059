#!/bin/bash
# A/B of a compile-time experiment on the bench line (GPU box): bash tools/ab_build.sh "<extra hipcc flags>"
# default build -> bench; build with the flags -> GPU parity tests + bench; again both; the tree's library is rebuilt at the end.
out=gpurun_out/abb; mkdir -p $out
run() { name=$1; python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-host-out --no-single-frame > $out/$name.json 2> $out/$name.err || { echo "bench $name failed"; tail -3 $out/$name.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$out/$name.json").read().strip().splitlines()[-1])
print("$name:", round(d["value"]), round(d["ms_per_step"],4), {k: round(v,4) for k,v in d["roofline"]["all_kernels_ms_per_step"].items()})
PY
}
run base1
TINYORB_BUILD_EXTRA="$1" python -m tinyslam_amd.build --force > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?; tail -2 $out/tests.log
if [ $rc -ne 0 ]; then python -m tinyslam_amd.build --force > /dev/null 2>&1; exit $rc; fi
run with1
python -m tinyslam_amd.build --force > $out/build.log 2>&1
run base2
TINYORB_BUILD_EXTRA="$1" python -m tinyslam_amd.build --force > $out/build.log 2>&1
run with2
python -m tinyslam_amd.build --force > $out/build.log 2>&1
