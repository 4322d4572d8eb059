# usage (GPU box): bash tools/pmc_brief.sh <outdir>  -- counters for the BRIEF kernels (k_brief_t, k_brief_nf)
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_brief}; mkdir -p $OUT
ARGS="python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-single-frame --preheat-ms 0 --no-host-out --repeats 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq1 -- $ARGS > $OUT/sq1.json 2> $OUT/sq1.err
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- $ARGS > $OUT/sq2.json 2> $OUT/sq2.err
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/tcp -- $ARGS > $OUT/tcp.json 2> $OUT/tcp.err || true
rocprofv3 --pmc TA_BUSY_avr TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum --output-format csv -d $OUT/ta -- $ARGS > $OUT/ta.json 2> $OUT/ta.err || true
ls $OUT
