# usage (GPU box): bash tools/pmc_masks.sh  -- SQ_INSTS_VALU / SALU / LDS of k_front per phase mask (0: A only, 1: +B1, 3: +B2, 7: +C0, 15: all)
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/pmcm; mkdir -p $OUT
for m in 0 1 3 7 15; do
  TINYORB_PHASE_MASK=$m rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/m$m -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-single-frame --preheat-ms 0 --no-host-out --repeats 1 > $OUT/m$m.json 2> $OUT/m$m.err
done
python3 - <<'PY'
import glob, pandas as pd
for m in (0,1,3,7,15):
    f=sorted(glob.glob(f"gpurun_out/pmcm/m{m}/*/*counter_collection.csv"))[-1]
    d=pd.read_csv(f); d=d[d.Kernel_Name.str.contains("k_front<true")]
    g=d.groupby("Counter_Name").Counter_Value.mean()
    print(m, {k:int(v) for k,v in g.items()})
PY
