#!/usr/bin/env python3
"""profiles/traffic_literal_y8.json from FETCH_SIZE / WRITE_SIZE passes over `bench.py --input y8` (tools/collect_round4_final.sh):
    python tools/make_y8_traffic.py gpurun_out/r04f_y8 r04
HBM bytes per launch of the level-0 kernel = (2 x FETCH_SIZE + WRITE_SIZE) KB (MI355X_MICROARCH.md, HBM section)."""
import glob, json, os, sys
import pandas as pd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinyslam_amd import build as orb_build
src, tag = sys.argv[1], sys.argv[2]
vals = {}
for sub in ("fetch", "write"):
    f = sorted(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1]
    d = pd.read_csv(f)
    d = d[d["Kernel_Name"].str.contains("k_front<true")]
    d = d[d["Grid_Size"] == d["Grid_Size"].max()]  # the batch launches
    vals[sub] = float(d.groupby("Counter_Name")["Counter_Value"].mean().iloc[0])
out = {"kernel": "k_front_l0", "mode": "literal", "input": "y8", "frames_per_launch": 256.0, "csrc_sha256": orb_build.source_hash(),
       "hbm_bytes_per_launch": (2.0 * vals["fetch"] + vals["write"]) * 1024.0, "fetch_size_kb": vals["fetch"], "write_size_kb": vals["write"],
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --input y8 --steps 3 --warmup 1` (%s); "
                 "bytes = (2*FETCH_SIZE + WRITE_SIZE) KB" % tag}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_literal_y8.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
