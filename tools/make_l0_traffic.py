#!/usr/bin/env python3
"""A stamped traffic file for the level-0 kernel of a literal-mode variant (profiles/traffic_literal_<input>[_fp<N>].json) from FETCH_SIZE /
WRITE_SIZE passes over `bench.py <flags> --steps 3 --warmup 1`:
    python tools/make_l0_traffic.py <src dir with fetch/ and write/> <tag> <output name> <input: rgba | y8> "<bench flags>"
HBM bytes per launch of the level-0 kernel = (2 x FETCH_SIZE + WRITE_SIZE) KB (MI355X_MICROARCH.md, HBM section)."""
import glob, json, os, sys
import pandas as pd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinyslam_amd import build as orb_build
src, tag, name, inp, flags = sys.argv[1:6]
vals = {}
for sub in ("fetch", "write"):
    f = sorted(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1]
    d = pd.read_csv(f)
    d = d[d["Kernel_Name"].str.contains("k_front<true")]
    d = d[d["Grid_Size"] == d["Grid_Size"].max()]  # the batch launches
    vals[sub] = float(d.groupby("Counter_Name")["Counter_Value"].mean().iloc[0])
out = {"kernel": "k_front_l0", "mode": "literal", "input": inp, "frames_per_launch": 256.0, "csrc_sha256": orb_build.source_hash(),
       "hbm_bytes_per_launch": (2.0 * vals["fetch"] + vals["write"]) * 1024.0, "fetch_size_kb": vals["fetch"], "write_size_kb": vals["write"],
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py %s --steps 3 --warmup 1` (%s); "
                 "bytes = (2*FETCH_SIZE + WRITE_SIZE) KB" % (flags, tag)}
json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
print(json.dumps(out, indent=1))
