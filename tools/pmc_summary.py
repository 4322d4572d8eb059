#!/usr/bin/env python3
"""Summarises the counter passes of tools/pmc.sh (gpurun_out/<dir>/{sq1,sq2,fetch,write,tcc}) per kernel:
    python tools/pmc_summary.py gpurun_out/r02a_pmc [profiles/r02_pmc_summary.csv]"""
import glob
import os
import sys

import pandas as pd

src = sys.argv[1]
KERNELS = ("k_front<true>", "k_front<false>", "k_brief_rows", "k_slot_prefix", "k_front_y", "k_compact", "k_brief_t", "k_brief_nfb", "k_brief_nf",
           "k_front_i<true>", "k_front_i<false>", "k_brief_i", "k_select_i", "k_gauss")


def short(name):
    if "k_front_i<true" in name or "k_front_i<(bool)1" in name:
        return "k_front_i<true>"
    if "k_front_i<" in name:
        return "k_front_i<false>"
    if "k_front<true" in name:
        return "k_front<true>"
    if "k_front<false" in name:
        return "k_front<false>"
    for k in KERNELS:
        if k in name:
            return k
    return None


rows = []
for sub in sorted(os.listdir(src)):
    fs = sorted(glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")), key=os.path.getmtime)
    if not fs:
        continue
    d = pd.read_csv(fs[-1])
    d["kernel"] = d["Kernel_Name"].map(short)
    d = d.dropna(subset=["kernel"])
    d = d[d["Grid_Size"] == d.groupby("kernel")["Grid_Size"].transform("max")]  # the batch launches, not single-frame ones
    g = d.groupby(["kernel", "Counter_Name"])["Counter_Value"].agg(["mean", "count"]).reset_index()
    g["pass"] = sub
    rows.append(g)
out = pd.concat(rows)
out.columns = ["kernel", "counter", "mean_per_launch", "launches", "pass"]
pd.set_option("display.width", 200)
pd.set_option("display.max_rows", 500)
print(out.to_string(index=False, float_format=lambda v: "%.0f" % v))
if len(sys.argv) > 2:
    out.to_csv(sys.argv[2], index=False)
