#!/usr/bin/env python3
"""Turns gpurun_out/<tag>/ (tools/collect_profiles.sh) into the committed summaries under profiles/:
<tag>_kernel_stats.csv, <tag>_pmc_summary.csv, <tag>_bench.json and traffic.json (HBM bytes per launch of each
kernel = 2 x FETCH_SIZE + WRITE_SIZE, in KB -> bytes; the factor 2 is the gfx950 correction for wide coalesced
reads, MI355X_MICROARCH.md section HBM)."""
import glob
import json
import os
import shutil
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinyslam_amd import build as orb_build  # noqa: E402  (source_hash: no GPU, no compile)
src_tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
tag = sys.argv[2] if len(sys.argv) > 2 else src_tag  # name of the committed files (one set per round)
src = os.path.join(ROOT, "gpurun_out", src_tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)



def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1] if files else None


shutil.copy(newest(os.path.join(src, "kt", "*", "*_kernel_stats.csv")), os.path.join(dst, tag + "_kernel_stats.csv"))
for name in ("bench.json", "bench_under_rocprof.json"):
    shutil.copy(os.path.join(src, name), os.path.join(dst, tag + "_" + name))


def short(name):
    if "k_front<true" in name:   # k_front<true, false> / <true, true> (Y8): level 0
        return "k_front<true>"
    if "k_front<false" in name:
        return "k_front<false>"
    for k in ("k_front<true>", "k_front<false>", "k_brief_rows", "k_brief_t", "k_brief_nf", "k_slot_prefix", "k_compact",
              "k_synth", "k_grayscale", "k_mip", "k_blur_rows", "k_fast", "k_brief"):
        if k in name:
            return k
    return name[:40]


rows = []
for sub in ("fetch", "write", "sq", "sq2", "sq3"):
    f = newest(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    if not f:
        continue
    df = pd.read_csv(f)
    df["kernel"] = df["Kernel_Name"].map(short)
    g = df.groupby(["kernel", "Counter_Name"])["Counter_Value"].agg(["mean", "count"]).reset_index()
    rows.append(g)
pmc = pd.concat(rows)
pmc.columns = ["kernel", "counter", "mean_per_launch", "launches"]
pmc.to_csv(os.path.join(dst, tag + "_pmc_summary.csv"), index=False)

piv = pmc.pivot(index="kernel", columns="counter", values="mean_per_launch")
bench = json.load(open(os.path.join(src, "bench.json")))
dom = {"k_front_l0": "k_front<true>", "k_front_ln": "k_front<false>"}.get(
    bench["roofline"]["kernel"], bench["roofline"]["kernel"])
fetch_kb, write_kb = float(piv.loc[dom, "FETCH_SIZE"]), float(piv.loc[dom, "WRITE_SIZE"])
traffic = {
    "kernel": bench["roofline"]["kernel"], "frames_per_launch": bench["roofline"]["frames_per_launch"],
    "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
    "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
    "csrc_sha256": orb_build.source_hash(),  # the kernels these counters were taken on (bench.py checks it)
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1` "
              "(profiles/%s_pmc_summary.csv); bytes = (2*FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE reads half of a wide "
              "coalesced stream on gfx950" % tag,
    "all_kernels": {k: {"fetch_kb": float(piv.loc[k, "FETCH_SIZE"]), "write_kb": float(piv.loc[k, "WRITE_SIZE"])}
                    for k in piv.index if k.startswith("k_") and k != "k_synth"},
}
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
