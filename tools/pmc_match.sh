# usage (GPU box): bash tools/pmc_match.sh <outdir>  -- kernel stats and matrix-pipe counters of the matcher (tools/match_rate.py: six launches of k_desc_expand4 + k_match_fp4)
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=${1:-gpurun_out/match_pmc}; mkdir -p $OUT
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 tools/match_rate.py > $OUT/kt.log 2>&1
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F6F4 SQ_INSTS_VALU_MFMA_MOPS_F6F4 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc -- python3 tools/match_rate.py > $OUT/pmc.log 2>&1
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc2 -- python3 tools/match_rate.py > $OUT/pmc2.log 2>&1
python3 - <<PY
import csv, glob, collections
out = open("$OUT/summary.txt", "w")
for f in glob.glob("$OUT/kt/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_match" in r["Name"] or "k_desc" in r["Name"]:
            print("%-20s calls %s avg %.4f ms" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e6), file=out)
for d in ("pmc", "pmc2"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if "k_match_fp4" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    # one row per (dispatch, counter): mean per launch
    for k, (v, n) in sorted(acc.items()):
        print("k_match_fp4 %-28s %.4e per launch (%d launches)" % (k, v / max(n, 1), n), file=out)
out.close()
print(open("$OUT/summary.txt").read())
PY
