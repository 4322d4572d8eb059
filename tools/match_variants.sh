# usage (GPU box): bash tools/match_variants.sh name1 name2 ...  -- the matcher (tools/match_rate.py) and its matrix-pipe / LDS counters for whole libraries
# tinyslam_amd/libtinyorb_<name>.so (built with different TINYORB_MATCH_* settings), one after the other on one box; the tree's library is put back
export TMPDIR=/tmp
out=gpurun_out/matchvar; mkdir -p $out
cp tinyslam_amd/libtinyorb.so $out/keep.so
for n in "$@"; do
  cp tinyslam_amd/libtinyorb_$n.so tinyslam_amd/libtinyorb.so
  timeout -k 10 120 python tools/match_rate.py 2>&1 | tail -1 | sed "s/^/$n: /"
  timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F6F4 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/$n.pmc -- python3 tools/match_rate.py > $out/$n.pmc.log 2>&1
  timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $out/$n.pmc2 -- python3 tools/match_rate.py > $out/$n.pmc2.log 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for d in ("pmc", "pmc2"):
    for f in glob.glob("$out/$n.%s/*/*_counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if "k_match_fp4" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
v = {k: a[0] / max(a[1], 1) for k, a in acc.items()}
if v:
    print("$n: mfma %.3e valu %.3e (%.2f per mfma)  pipe busy %.3f  wait_inst %.2f wait_any %.2f of wave cycles  lds conflict %.2f of idx_active, %.2f cycles per lds instr"
          % (v.get("SQ_INSTS_VALU_MFMA_F6F4", 0), v.get("SQ_INSTS_VALU", 0), (v.get("SQ_INSTS_VALU", 0) - v.get("SQ_INSTS_VALU_MFMA_F6F4", 0)) / max(v.get("SQ_INSTS_VALU_MFMA_F6F4", 1), 1),
             v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(v.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024, 1), v.get("SQ_WAIT_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1),
             v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1), v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1),
             v.get("SQ_LDS_IDX_ACTIVE", 0) / max(v.get("SQ_INSTS_LDS", 1), 1)))
PY
done
cp $out/keep.so tinyslam_amd/libtinyorb.so
