#!/usr/bin/env python3
"""profiles/traffic_intended_rgba.json from the counter passes of tools/pmc_intended.sh:
    python tools/make_intended_traffic.py gpurun_out/r03f_pmc_intended_summary.csv r03
HBM bytes per launch of each kernel = (2 x FETCH_SIZE + WRITE_SIZE) KB (MI355X_MICROARCH.md, HBM section); the profile id
k_front_i of the bench line covers two launches (level 0, level 1): their mean."""
import json, os, sys
import pandas as pd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinyslam_amd import build as orb_build
src, tag = sys.argv[1], sys.argv[2]
d = pd.read_csv(src)
piv = d.pivot_table(index="kernel", columns="counter", values="mean_per_launch")
allk = {k: {"fetch_kb": float(piv.loc[k, "FETCH_SIZE"]), "write_kb": float(piv.loc[k, "WRITE_SIZE"])} for k in piv.index}
b = lambda k: (2.0 * allk[k]["fetch_kb"] + allk[k]["write_kb"]) * 1024.0
out = {"kernel": "k_front_i", "mode": "intended", "input": "rgba", "frames_per_launch": 128.0, "csrc_sha256": orb_build.source_hash(),
       "hbm_bytes_per_launch": 0.5 * (b("k_front_i<true>") + b("k_front_i<false>")),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --mode intended --steps 3 --warmup 1` "
                 "(profiles/%s_pmc_intended_summary.csv); bytes = (2*FETCH_SIZE + WRITE_SIZE) KB, mean of the two launches (level 0, level 1) "
                 "that the profile id k_front_i covers" % tag,
       "hbm_bytes_per_batch_all_kernels": sum(b(k) for k in allk),
       "all_kernels": allk}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_intended_rgba.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
