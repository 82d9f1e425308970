#!/usr/bin/env python3
"""Per-kernel means of a rocprofv3 --pmc counter CSV for the convolution kernels:
    python tools/pmc_mfma_summary.py counter_collection.csv out.csv"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "conv_mfma" in k:
        agg[k[k.find("<"):k.find(">") + 1]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    f.write("kernel,launches,counter,mean_per_launch\n")
    for k, d in agg.items():
        for c, v in d.items():
            f.write(f"\"conv_mfma_f32_kernel{k}\",{len(v)},{c},{sum(v) / len(v):.1f}\n")
