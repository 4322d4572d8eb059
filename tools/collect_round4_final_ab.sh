bash tools/collect_round4_final.sh 2>&1 | tail -1
for i in 1 2; do  # the three level-constant precomputes (TINYORB_NO_COLTAB switches them off together)
  python bench.py --cpu-sample 0 --no-single-frame --no-host-out > gpurun_out/r04f_ab_coltab_on_$i.json 2>/dev/null
  TINYORB_NO_COLTAB=1 python bench.py --cpu-sample 0 --no-single-frame --no-host-out > gpurun_out/r04f_ab_coltab_off_$i.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("on_1","off_1","on_2","off_2"):
    d=json.loads(open("gpurun_out/r04f_ab_coltab_%s.json" % n).read().strip().splitlines()[-1])
    print(n, round(d["value"]), round(d["ms_per_step"],4), {k: round(v,4) for k,v in d["roofline"]["all_kernels_ms_per_step"].items()})
PY
