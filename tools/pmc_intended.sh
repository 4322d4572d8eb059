# usage: bash tools/pmc_intended.sh <outdir>   -- rocprofv3 counter passes over a short run of the intended mode (separate passes)
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_i}; mkdir -p $OUT
ARGS="python3 bench.py --mode intended --steps 3 --warmup 1 --cpu-sample 0 --no-single-frame --preheat-ms 0 --no-host-out --repeats 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -- $ARGS > $OUT/sq1.json 2> $OUT/sq1.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq2 -- $ARGS > $OUT/sq2.json 2> $OUT/sq2.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $ARGS > $OUT/write.json 2> $OUT/write.err
ls $OUT
