# usage (GPU box, after tools/publish_round5_profiles.sh has stamped profiles/traffic*.json): bash tools/rebench_round5.sh -- the bench lines that quote them
cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/r05g_bench.json 2> /dev/null
python3 bench.py --contract 7 > gpurun_out/r05g_bench_contract7.json 2> /dev/null
python3 bench.py --mode intended > gpurun_out/r05g_bench_intended.json 2> /dev/null
python3 bench.py --mode intended --angle-bins 1024 > gpurun_out/r05g_bench_intended_bins1024.json 2> /dev/null
python3 bench.py --input y8 --cpu-sample 0 --no-single-frame > gpurun_out/r05g_bench_y8.json 2> /dev/null
python3 - <<'PY'
import json
for n in ("bench", "bench_contract7", "bench_intended", "bench_intended_bins1024", "bench_y8"):
    d = json.loads(open("gpurun_out/r05g_%s.json" % n).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-26s %8.0f frames/s %.4f ms  %s %.4f ms frac %.4f traffic %s %s" % (n, d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["frac"], r["traffic"], r.get("traffic_note", "")))
PY
