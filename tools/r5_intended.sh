#!/bin/bash
# round 5: intended mode with angle bins -- parity, then the bench line with 0 / 1024 / 30 bins and the HBM counters of k_brief_i
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py -m gpu -x -q -k "angle_bins" 2>&1 | tail -2
[ ${PIPESTATUS[0]} -ne 0 ] && exit 1
for b in 0 1024 30; do
  python bench.py --mode intended --angle-bins $b --cpu-sample 0 --no-single-frame --no-host-out > gpurun_out/r05_bench_intended_bins$b.json 2> gpurun_out/r05_bench_intended_bins$b.err || { tail -3 gpurun_out/r05_bench_intended_bins$b.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05_bench_intended_bins$b.json").read().strip().splitlines()[-1])
print("bins $b:", round(d["value"]), round(d["ms_per_step"],4), {k: round(v,4) for k,v in d["roofline"]["all_kernels_ms_per_step"].items()})
PY
done
for b in 0 1024; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/r05_pmc_intended_bins$b/$c -- python3 bench.py --mode intended --angle-bins $b --steps 3 --warmup 1 --repeats 1 --cpu-sample 0 --no-single-frame --no-host-out --preheat-ms 0 > gpurun_out/r05_pmc_intended_bins$b.$c.log 2>&1
  done
done
python - <<'PY'
import csv, glob, collections
for b in (0, 1024):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("gpurun_out/r05_pmc_intended_bins%d/%s/*/*_counter_collection.csv" % (b, c)):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void orb::", "")
                a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    tot = 0.0
    for k, v in sorted(acc.items()):
        if "synth" in k: continue
        fs = v["FETCH_SIZE"][0] / max(v["FETCH_SIZE"][1], 1); ws = v["WRITE_SIZE"][0] / max(v["WRITE_SIZE"][1], 1)
        gb = (2 * fs + ws) * 1024 / 1e9  # KB units; gfx950: FETCH_SIZE counts half (MI355X_MICROARCH.md)
        print("bins %4d  %-28s FETCH %.3e KB WRITE %.3e KB per launch -> %.3f GB" % (b, k[:28], fs, ws, gb))
        if k.startswith(("k_front_i", "k_brief_i", "k_select_i")): tot += gb * (1 if "k_front_i<false" not in k else 1)
    print("bins %4d  sum over the mode's kernels (one launch each; k_front_i<false> once per upper level): %.3f GB" % (b, tot))
PY
