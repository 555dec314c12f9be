#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per (kernel, counter).
   python tools/pmc_summary.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            name = ("k_project" if "k_project" in k else "k_pairwise_mfma" if "k_pairwise_mfma" in k else None)
            if name:
                agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print("%s %s n=%d mean=%.6g" % (k[0], k[1], len(v), sum(v) / len(v)))
