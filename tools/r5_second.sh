#!/bin/bash
# round 5, second collection: whole GPU suite at the tree, the bench line (default + --force-collate + node host), matcher counters
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_b.log 2>&1; rc=$?; tail -3 gpurun_out/r05_gputest_b.log
[ $rc -ne 0 ] && exit $rc
python bench.py > gpurun_out/r05_bench_b.json 2> gpurun_out/r05_bench_b.err || { tail -5 gpurun_out/r05_bench_b.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_bench_b.json").read().strip().splitlines()[-1])
print("bench:", round(d["value"]), round(d["ms_per_step"],4), {k: round(v,4) for k,v in d["roofline"]["all_kernels_ms_per_step"].items()}, round(d["roofline"]["frac"],4))
print("single:", round(d["single_frame_us"],1), round(d["single_frame_us_blocking_wait"],1))
for k in ("single_frame_loop","single_frame_loop_blocking_wait"):
    print(k, {kk: {a: (round(b,1) if isinstance(b,float) else b) for a,b in vv.items()} if isinstance(vv,dict) else vv for kk,vv in d[k].items() if kk not in ("what","wait")})
PY
python bench.py --force-collate --cpu-sample 0 --no-single-frame > gpurun_out/r05_bench_force_collate.json 2> gpurun_out/r05_bench_force_collate.err || { tail -5 gpurun_out/r05_bench_force_collate.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_bench_force_collate.json").read().strip().splitlines()[-1])
c=d["collate"]
print("force-collate:", round(d["value"]), round(d["ms_per_step"],4), "sharded", round(d["value_sharded"]), round(d["ms_per_step_sharded"],4), "host ms", round(c["host_ms_per_batch"],4), "wait", round(c["host_wait_ms_per_batch"],4), "enqueue", round(c["host_enqueue_ms_per_batch"],4), c["root_check"], c["exact"])
PY
python bench.py --host node --cpu-sample 0 --no-single-frame > gpurun_out/r05_bench_node_n1.json 2> gpurun_out/r05_bench_node_n1.err || { tail -5 gpurun_out/r05_bench_node_n1.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_bench_node_n1.json").read().strip().splitlines()[-1])
print("node:", round(d["value"]), round(d["ms_per_step"],4), "sharded", round(d["value_sharded"]), round(d["ms_per_step_sharded"],4), {k: round(v,4) for k,v in d["collate"].items() if isinstance(v,float)})
PY
bash tools/pmc_match.sh gpurun_out/r05_match_pmc > gpurun_out/r05_match_pmc.txt 2>&1; tail -22 gpurun_out/r05_match_pmc.txt
