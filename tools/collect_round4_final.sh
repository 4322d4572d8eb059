# usage (GPU box): bash tools/collect_round4_final.sh   -- everything profiles/r04_* is made from, at the final kernels of round 4
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/collect_profiles.sh r04f > gpurun_out/r04f_collect.log 2>&1
echo "literal done"
bash tools/collect_intended.sh r04f > gpurun_out/r04f_collect_i.log 2>&1
bash tools/pmc_intended.sh gpurun_out/r04f_pmc_i > gpurun_out/r04f_pmc_i.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r04f_pmc_i gpurun_out/r04f_pmc_intended_summary.csv > /dev/null 2>&1
echo "intended done"
mkdir -p gpurun_out/r04f_y8
Y8="python3 bench.py --input y8 --steps 3 --warmup 1 --cpu-sample 0 --no-single-frame --preheat-ms 0 --no-host-out --repeats 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r04f_y8/fetch -- $Y8 > gpurun_out/r04f_y8/fetch.json 2> gpurun_out/r04f_y8/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r04f_y8/write -- $Y8 > gpurun_out/r04f_y8/write.json 2> gpurun_out/r04f_y8/write.err
echo "y8 counters done"
python3 bench.py --host node --cpu-sample 0 > gpurun_out/r04f_bench_node_n1.json 2> /dev/null
TINYORB_NODE_LOOPBACK=2 python3 bench.py --host node --cpu-sample 0 --no-single-frame > gpurun_out/r04f_bench_node_n1_rccl_self.json 2> /dev/null
python3 bench.py --gpus 1 --force-collate --cpu-sample 0 --no-single-frame > gpurun_out/r04f_bench_force_collate.json 2> /dev/null
echo "bench lines done"
bash tools/content_axis.sh gpurun_out/r04f_content_axis.txt > /dev/null 2>&1
(for s in "1280 720" "640 480" "1920 1080"; do python3 tools/single_frame_latency.py $s 2>/dev/null; done) > gpurun_out/r04f_single_frame_latency.txt
python3 tools/pinned_loop_probe.py > gpurun_out/r04f_pinned_loop.txt 2>&1
bash tools/pmc_masks.sh gpurun_out/r04f_masks > gpurun_out/r04f_pmc_masks.txt 2>&1 || true
echo "all done"
