#!/usr/bin/env python3
"""Mean counter values per dispatch of the kernels whose name contains <substr>, from one or more rocprofv3 --pmc output dirs.
usage: pmc_one.py <substr> <dir> [<dir> ...]"""
import collections
import csv
import glob
import os
import sys

sub = sys.argv[1]
for d in sys.argv[2:]:
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    # the first dispatch is the warm-up call of tools/one_gemm.py: drop it when there are more
    print(os.path.basename(d.rstrip("/")), "  ".join(f"{k}={sum(v[1:] or v) / len(v[1:] or v):.4g}" for k, v in sorted(agg.items())))
