#!/usr/bin/env python3
"""Turn rocprofv3 counter CSVs (one pass with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, both
over the same command) into a per-kernel traffic summary.

FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte read
requests of a streaming kernel at 64 bytes each, so fetched bytes = 2 * FETCH_SIZE * 1024;
WRITE_SIZE is exact.  (Check on this kernel: with the XCD-aware tile order a 64->64 3x3 launch at
1088x1920 reports 2 * 312.1 MiB = 1.195 x its input, exactly the (18*34)/(16*32) halo factor.)

  python tools/pmc_summary.py FETCH.csv WRITE.csv OUT.json "command that was profiled"
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


def short(name):
    m = re.search(r"conv_mfma_f32_kernel<(\d+), (\d+), (\d+), (\d+)(?:, \w+)*>", name)
    if m:
        return "conv_mfma_f32_kernel<%s,%s,%s,%s>" % m.groups()
    m = re.search(r"(\w+_kernel)\b", name)
    return m.group(1) if m else name[:60]


def main():
    fetch_csv, write_csv, out, cmd = sys.argv[1:5]
    fe, wr = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for name, (fsum, fn) in fe.items():
        if name not in wr:
            continue
        wsum, wn = wr[name]
        fetch = 2.0 * fsum / fn * 1024.0
        write = wsum / wn * 1024.0
        kernels[short(name)] = {"launches": fn, "fetch_size_kib_raw": fsum / fn, "write_size_kib_raw": wsum / wn,
                                "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                                "traffic_bytes_per_launch": fetch + write}
    doc = {"command": cmd,
           "correction": "fetch = 2 x FETCH_SIZE (gfx950 counts 128-B read requests as 64 B); WRITE_SIZE as reported; KiB -> bytes",
           "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"]))}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for k, v in list(doc["kernels"].items())[:8]:
        print(f"{k:40s} n={v['launches']:5d} fetch {v['fetch_bytes_per_launch']/1e6:9.1f} MB  write {v['write_bytes_per_launch']/1e6:9.1f} MB")


if __name__ == "__main__":
    main()
