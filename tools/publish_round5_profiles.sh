# usage (here, after tools/collect_round5_final.sh on the GPU box): bash tools/publish_round5_profiles.sh  -- gpurun_out/r05f* -> profiles/r05_*
set -e
cd "$(dirname "$0")/.."
python tools/make_profiles.py r05f r05 > /dev/null
python tools/make_l0_traffic.py gpurun_out/r05f_y8 r05 traffic_literal_y8.json y8 "--input y8" > /dev/null
python tools/make_l0_traffic.py gpurun_out/r05f_fp7 r05 traffic_literal_rgba_fp7.json rgba "--contract 7" > /dev/null
python tools/make_intended_traffic.py gpurun_out/r05f_pmc_intended_summary.csv r05 > /dev/null
python tools/make_intended_traffic.py gpurun_out/r05f_pmc_intended_bins1024_summary.csv r05 traffic_intended_rgba_bins1024.json "--angle-bins 1024" > /dev/null
cp "$(ls -t gpurun_out/r05f_fp7/kt/*/*_kernel_stats.csv | head -1)" profiles/r05_contract7_kernel_stats.csv
cp gpurun_out/r05f_fp7/bench_under_rocprof.json profiles/r05_bench_contract7_under_rocprof.json
for n in bench_contract7 bench_contract15 bench_contract8 bench_y8 bench_intended_bins1024 bench_node_n1 bench_node_n1_rccl_self rehearsal_gloo2_weak; do
  [ -s gpurun_out/r05f_$n.json ] && cp gpurun_out/r05f_$n.json profiles/r05_$n.json
done
cp gpurun_out/r05f_bench_force_collate.json profiles/r05_bench_force_collate_nccl_n1.json
cp gpurun_out/r05f_content_axis.txt profiles/r05_content_axis.txt
cp gpurun_out/r05f_single_frame_latency.txt profiles/r05_single_frame_latency.txt
cp gpurun_out/r05f_pmc_masks.txt profiles/r05_pmc_masks.txt
cp gpurun_out/r05f_match_pmc.txt profiles/r05_match_pmc.txt
cp gpurun_out/r05f_pmc_intended_summary.csv profiles/r05_pmc_intended_summary.csv
cp gpurun_out/r05f_pmc_intended_bins1024_summary.csv profiles/r05_pmc_intended_bins1024_summary.csv
cp gpurun_out/r05f_intended/bench.json profiles/r05_bench_intended.json
cp "$(ls -t gpurun_out/r05f_intended/kt/*/*_kernel_stats.csv | head -1)" profiles/r05_intended_kernel_stats.csv
# the bench lines that quote a traffic file are taken AGAIN once the stamped files exist (tools/rebench_round5.sh on the GPU box), then:
for n in bench bench_contract7 bench_intended bench_intended_bins1024 bench_y8; do
  [ -s gpurun_out/r05g_$n.json ] && cp gpurun_out/r05g_$n.json profiles/r05_$n.json
done
true
