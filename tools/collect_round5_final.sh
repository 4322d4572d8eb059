# usage (GPU box): bash tools/collect_round5_final.sh   -- everything profiles/r05_* is made from, at the final kernels of round 5
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/collect_profiles.sh r05f > gpurun_out/r05f_collect.log 2>&1
echo "literal done"
# the contracted arithmetic (OrbOptions::fp_contract = 7): bench line, kernel stats, its own stamped counter pass
mkdir -p gpurun_out/r05f_fp7
C7="python3 bench.py --contract 7 --steps 3 --warmup 1 --cpu-sample 0 --no-single-frame --preheat-ms 0 --no-host-out --repeats 1"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05f_fp7/kt -- python3 bench.py --contract 7 --steps 20 --warmup 3 --cpu-sample 0 --no-single-frame --no-host-out > gpurun_out/r05f_fp7/bench_under_rocprof.json 2> gpurun_out/r05f_fp7/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r05f_fp7/fetch -- $C7 > gpurun_out/r05f_fp7/fetch.json 2> gpurun_out/r05f_fp7/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r05f_fp7/write -- $C7 > gpurun_out/r05f_fp7/write.json 2> gpurun_out/r05f_fp7/write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r05f_fp7/sq -- $C7 > gpurun_out/r05f_fp7/sq.json 2> gpurun_out/r05f_fp7/sq.err
echo "contract 7 counters done"
bash tools/collect_intended.sh r05f > gpurun_out/r05f_collect_i.log 2>&1
bash tools/pmc_intended.sh gpurun_out/r05f_pmc_i > gpurun_out/r05f_pmc_i.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r05f_pmc_i gpurun_out/r05f_pmc_intended_summary.csv > /dev/null 2>&1
mkdir -p gpurun_out/r05f_pmc_i1024
I1024="python3 bench.py --mode intended --angle-bins 1024 --steps 3 --warmup 1 --cpu-sample 0 --no-single-frame --preheat-ms 0 --no-host-out --repeats 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r05f_pmc_i1024/fetch -- $I1024 > gpurun_out/r05f_pmc_i1024/fetch.json 2> gpurun_out/r05f_pmc_i1024/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r05f_pmc_i1024/write -- $I1024 > gpurun_out/r05f_pmc_i1024/write.json 2> gpurun_out/r05f_pmc_i1024/write.err
python3 tools/pmc_summary.py gpurun_out/r05f_pmc_i1024 gpurun_out/r05f_pmc_intended_bins1024_summary.csv > /dev/null 2>&1
echo "intended done"
mkdir -p gpurun_out/r05f_y8
Y8="python3 bench.py --input y8 --steps 3 --warmup 1 --cpu-sample 0 --no-single-frame --preheat-ms 0 --no-host-out --repeats 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r05f_y8/fetch -- $Y8 > gpurun_out/r05f_y8/fetch.json 2> gpurun_out/r05f_y8/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r05f_y8/write -- $Y8 > gpurun_out/r05f_y8/write.json 2> gpurun_out/r05f_y8/write.err
echo "y8 counters done"
python3 bench.py --contract 7 > gpurun_out/r05f_bench_contract7.json 2> /dev/null
python3 bench.py --contract 15 --cpu-sample 0 --no-single-frame > gpurun_out/r05f_bench_contract15.json 2> /dev/null
python3 bench.py --contract 8 --cpu-sample 0 --no-single-frame > gpurun_out/r05f_bench_contract8.json 2> /dev/null
python3 bench.py --input y8 --cpu-sample 0 --no-single-frame > gpurun_out/r05f_bench_y8.json 2> /dev/null
python3 bench.py --mode intended --angle-bins 1024 > gpurun_out/r05f_bench_intended_bins1024.json 2> /dev/null
python3 bench.py --host node --cpu-sample 0 > gpurun_out/r05f_bench_node_n1.json 2> /dev/null
TINYORB_NODE_LOOPBACK=2 python3 bench.py --host node --cpu-sample 0 --no-single-frame > gpurun_out/r05f_bench_node_n1_rccl_self.json 2> /dev/null
python3 bench.py --gpus 1 --force-collate --cpu-sample 0 --no-single-frame > gpurun_out/r05f_bench_force_collate.json 2> /dev/null
TINYORB_DIST_BACKEND=gloo python3 bench.py --gpus 2 --frames 64 --cpu-sample 0 --no-single-frame > gpurun_out/r05f_rehearsal_gloo2_weak.json 2> gpurun_out/r05f_rehearsal_gloo2_weak.err || true
echo "bench lines done"
bash tools/content_axis.sh gpurun_out/r05f_content_axis.txt > /dev/null 2>&1
(for s in "1280 720" "640 480" "1920 1080"; do python3 tools/single_frame_latency.py $s 2>/dev/null; done) > gpurun_out/r05f_single_frame_latency.txt
bash tools/pmc_masks.sh gpurun_out/r05f_masks > gpurun_out/r05f_pmc_masks.txt 2>&1 || true
bash tools/pmc_match.sh gpurun_out/r05f_match_pmc > gpurun_out/r05f_match_pmc.txt 2>&1 || true
echo "all done"
