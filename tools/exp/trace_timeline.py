#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace CSV: for the LAST occurrence of a marker kernel onwards (one steady-state
run), per-kernel totals, the union of busy intervals and the idle gaps.   python tools/exp/trace_timeline.py trace.csv [marker]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_filter_meta"
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ev.sort()
starts = [i for i, e in enumerate(ev) if marker in e[2]]
# one run = from a marker to the next marker that is more than 5 ms later
runs = []
for i in starts:
    if not runs or ev[i][0] - ev[runs[-1]][0] > 20_000_000:
        runs.append(i)
lo = runs[-1]
hi = len(ev)
sel = ev[lo:hi]
# cut trailing kernels that belong to the link benchmark / teardown: stop at a gap > 20 ms
cut = len(sel)
for i in range(1, len(sel)):
    if sel[i][0] - max(e[1] for e in sel[:i]) > 20_000_000:
        cut = i
        break
sel = sel[:cut]
t0 = sel[0][0]
tot = defaultdict(lambda: [0, 0])
for s, e, n in sel:
    key = n.split("(")[0].replace("void ", "").replace("mvs::(anonymous namespace)::", "")[:60]
    tot[key][0] += e - s
    tot[key][1] += 1
busy = 0
cur_s, cur_e = sel[0][0], sel[0][1]
gaps = []
for s, e, n in sel[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t0, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(e for _, e, _ in sel) - t0
print("span %.2f ms, busy (union) %.2f ms, %d kernels" % (span / 1e6, busy / 1e6, len(sel)))
for k, (ns, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:16]:
    print("  %8.3f ms %5d  %s" % (ns / 1e6, c, k))
gaps.sort(reverse=True)
print("largest idle gaps (ms @ offset ms, next kernel):")
for g, at, n in gaps[:10]:
    print("  %.3f @ %.2f  %s" % (g / 1e6, at / 1e6, n[:50]))
print("sum of gaps %.2f ms in %d gaps" % (sum(g for g, _, _ in gaps) / 1e6, len(gaps)))
