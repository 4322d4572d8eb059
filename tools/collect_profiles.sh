# Runs on the GPU box (gpurun):  bash tools/collect_profiles.sh <tag>
# rocprofv3 kernel-trace stats of the bench command + separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit
# one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots") + the plain bench line.  Results under gpurun_out/<tag>/.
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-r01}; OUT=gpurun_out/$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-single-frame --no-host-out > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
PMC_ARGS="python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-single-frame --preheat-ms 0 --no-host-out --repeats 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $PMC_ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $PMC_ARGS > $OUT/write.json 2> $OUT/write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $PMC_ARGS > $OUT/sq.json 2> $OUT/sq.err
# stall / pipe counters (what bounds k_front: VERDICT r01 item 3)
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq2 -- $PMC_ARGS > $OUT/sq2.json 2> $OUT/sq2.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/sq3 -- $PMC_ARGS > $OUT/sq3.json 2> $OUT/sq3.err || true
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
