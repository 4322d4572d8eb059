cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/pmc_masks; rm -rf $OUT; mkdir -p $OUT
for m in 1 3 15; do
TINYORB_PHASE_MASK=$m rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/m$m -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > $OUT/m$m.json 2> $OUT/m$m.err
done
ls $OUT
