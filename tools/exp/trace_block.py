#!/usr/bin/env python3
"""Kernels between two consecutive occurrences of a marker kernel in a rocprofv3 --kernel-trace CSV (one row block of the
streamed pipeline): start offset, duration, gap before, name.   python tools/exp/trace_block.py trace.csv marker [which]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -8
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
idx = [i for i, e in enumerate(ev) if marker in e[2]]
lo, hi = idx[which], idx[which + 1]
t0 = ev[lo][0]
prev_end = t0
for s, e, n in ev[lo:hi]:
    name = n.replace("void ", "").replace("mvs::(anonymous namespace)::", "").split("(")[0][:58]
    print("%8.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name))
    prev_end = max(prev_end, e)
print("block span %.1f us" % ((ev[hi][0] - t0) / 1e3))
