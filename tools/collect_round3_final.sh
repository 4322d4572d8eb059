# usage (GPU box): bash tools/collect_round3_final.sh   -- everything profiles/r03_* is made from, at the final kernels of round 3
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/collect_profiles.sh r03f > gpurun_out/r03f_collect.log 2>&1
echo "literal done"
bash tools/collect_intended.sh r03f > gpurun_out/r03f_collect_i.log 2>&1
echo "intended done"
python3 bench.py --input y8 > gpurun_out/r03f_bench_y8.json 2> /dev/null
python3 bench.py --host node --cpu-sample 0 > gpurun_out/r03f_bench_node_n1.json 2> /dev/null
python3 bench.py --in-flight 3 --cpu-sample 0 --no-host-out --no-single-frame > gpurun_out/r03f_bench_inflight3.json 2> /dev/null
python3 bench.py --in-flight 2 --cpu-sample 0 --no-host-out --no-single-frame > gpurun_out/r03f_bench_inflight2.json 2> /dev/null
echo "bench lines done"
(for s in "1280 720" "640 480" "1920 1080"; do python3 tools/single_frame_latency.py $s 2>/dev/null; done; echo "# TINYORB_SINGLE_SYNC=1 (hipStreamSynchronize instead of the polled sequence number)"; TINYORB_SINGLE_SYNC=1 python3 tools/single_frame_latency.py 1280 720 2>/dev/null; echo "# TINYORB_SINGLE_SERIAL=1 (one k_front launch per level)"; TINYORB_SINGLE_SERIAL=1 python3 tools/single_frame_latency.py 1280 720 2>/dev/null; echo "# TINYORB_SINGLE_SPLIT=1 (round 2: six launches)"; TINYORB_SINGLE_SPLIT=1 python3 tools/single_frame_latency.py 1280 720 2>/dev/null) > gpurun_out/r03f_single_frame_latency.txt
timeout -k 10 120 tools/ubench/launch_floor > gpurun_out/r03f_launch_floor.txt 2>&1 || true
bash tools/ablate_intended.sh > gpurun_out/r03f_ablate_intended.txt 2>&1
bash tools/pmc_intended.sh gpurun_out/r03f_pmc_i > gpurun_out/r03f_pmc_i.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r03f_pmc_i gpurun_out/r03f_pmc_intended_summary.csv > /dev/null 2>&1
echo "all done"
