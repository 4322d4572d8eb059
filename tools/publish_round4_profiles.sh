# usage (here, after tools/collect_round4_final.sh on the GPU box): bash tools/publish_round4_profiles.sh  -- gpurun_out/r04f* -> profiles/r04_*
set -e
cd "$(dirname "$0")/.."
python tools/make_profiles.py r04f r04 > /dev/null
python tools/make_y8_traffic.py gpurun_out/r04f_y8 r04 > /dev/null
python tools/make_intended_traffic.py gpurun_out/r04f_pmc_intended_summary.csv r04 > /dev/null
cp gpurun_out/r04f_bench_node_n1.json profiles/r04_bench_node_n1.json
cp gpurun_out/r04f_bench_node_n1_rccl_self.json profiles/r04_bench_node_n1_rccl_self.json
cp gpurun_out/r04f_bench_force_collate.json profiles/r04_bench_force_collate_nccl_n1.json
cp gpurun_out/r04f_content_axis.txt profiles/r04_content_axis.txt
cp gpurun_out/r04f_single_frame_latency.txt profiles/r04_single_frame_latency.txt
cp gpurun_out/r04f_pinned_loop.txt profiles/r04_pinned_loop_probe.txt
cp gpurun_out/r04f_pmc_masks.txt profiles/r04_pmc_masks.txt
cp gpurun_out/r04f_pmc_intended_summary.csv profiles/r04_pmc_intended_summary.csv
cp gpurun_out/r04f_intended/bench.json profiles/r04_bench_intended.json
cp "$(ls -t gpurun_out/r04f_intended/kt/*/*_kernel_stats.csv | head -1)" profiles/r04_intended_kernel_stats.csv
