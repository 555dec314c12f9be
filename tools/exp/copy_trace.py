#!/usr/bin/env python3
"""Summary of a rocprofv3 --memory-copy-trace CSV: copies >= 1 MiB, rate per copy over time (last large copies)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
big = []
for r in rows:
    size = int(r.get("Size", r.get("Bytes", 0)) or 0)
    if size >= (1 << 20):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        big.append((s, e, size, r.get("Direction", ""), r.get("Source_Agent_Id", ""), r.get("Destination_Agent_Id", "")))
big.sort()
print("columns:", list(rows[0].keys()) if rows else None)
print("large copies:", len(big))
for s, e, size, d, a, b in big[-50:]:
    print("%10.3f ms  dur %7.3f ms  %6.1f MB  %5.1f GB/s  %s %s->%s" % ((s - big[0][0]) / 1e6, (e - s) / 1e6, size / 1e6, size / (e - s), d, a, b))
