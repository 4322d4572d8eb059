#!/usr/bin/env python3
"""profiles/traffic_intended_rgba.json from the counter passes of tools/pmc_intended.sh:
    python tools/make_intended_traffic.py gpurun_out/r03f_pmc_intended_summary.csv r03
HBM bytes per launch of each kernel = (2 x FETCH_SIZE + WRITE_SIZE) KB (MI355X_MICROARCH.md, HBM section); the profile id
k_front_i_l0 of the bench line is the level-0 launch (the levels above have their own id, k_front_i_ln)."""
import json, os, sys
import pandas as pd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinyslam_amd import build as orb_build
src, tag = sys.argv[1], sys.argv[2]
name = sys.argv[3] if len(sys.argv) > 3 else "traffic_intended_rgba.json"   # e.g. traffic_intended_rgba_bins1024.json
flags = sys.argv[4] if len(sys.argv) > 4 else ""                          # the bench flags beside --mode intended
d = pd.read_csv(src)
piv = d.pivot_table(index="kernel", columns="counter", values="mean_per_launch")
allk = {k: {"fetch_kb": float(piv.loc[k, "FETCH_SIZE"]), "write_kb": float(piv.loc[k, "WRITE_SIZE"])} for k in piv.index}
b = lambda k: (2.0 * allk[k]["fetch_kb"] + allk[k]["write_kb"]) * 1024.0
out = {"kernel": "k_front_i_l0", "mode": "intended", "input": "rgba", "frames_per_launch": 256.0, "csrc_sha256": orb_build.source_hash(),
       "hbm_bytes_per_launch": b("k_front_i<true>"),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --mode intended " + flags + " --steps 3 --warmup 1` "
                 "(profiles/%s_pmc_intended*_summary.csv); bytes = (2*FETCH_SIZE + WRITE_SIZE) KB of the level-0 launch "
                 "(profile id k_front_i_l0; the launches of the levels above are k_front_i_ln)" % tag,
       "hbm_bytes_per_batch_all_kernels": sum(b(k) for k in allk),
       "all_kernels": allk}
json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
print(json.dumps(out, indent=1))
